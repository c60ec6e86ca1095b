/* TEST INFRASTRUCTURE: stands in for DirectXTexP.h (which drags in the Windows SDK) when compiling the reference's block
 * codecs alone: SAL annotations vanish, a few CRT names are mapped. */
#pragma once
#include <assert.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <memory>
#include "directxmath.h"
#include "directxpackedvector.h"

#define _In_
#define _Out_
#define _Inout_
#define _In_opt_
#define _Out_opt_
#define _In_z_
#define _In_reads_(x)
#define _In_reads_opt_(x)
#define _In_reads_bytes_(x)
#define _Out_writes_(x)
#define _Out_writes_all_(x)
#define _Out_writes_bytes_(x)
#define _Inout_updates_all_(x)
#define _Inout_updates_(x)
#define _In_range_(a, b)
#define _Use_decl_annotations_
#define _Analysis_assume_(x)
#define __analysis_assume(x)
#define _Success_(x)
#define _countof(a) (sizeof(a) / sizeof((a)[0]))
#define UNREFERENCED_PARAMETER(x) (void)(x)
#define __cdecl
typedef long HRESULT;
typedef unsigned long DWORD;
typedef int BOOL;
#ifndef TRUE
#define TRUE 1
#define FALSE 0
#endif
#include <math.h>
#define _isnan(x) isnan(x)
#define ARRAYSIZE(a) (sizeof(a) / sizeof((a)[0]))
