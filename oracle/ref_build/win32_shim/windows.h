/*
 * oracle/ref_build/win32_shim/windows.h -- TEST INFRASTRUCTURE.
 *
 * Just enough of the Win32 API, on pthreads, for the reference's dispatch layer
 * (/root/reference/3rdParty/Intel/Source/win32Threads.cpp, .h) to compile UNMODIFIED on Linux, so that the
 * reference's own CompressImageMT / CompressImageST / 17 CompressImage* trampolines can be run as the *caller* of
 * libispc_texcomp.so ("the plugin's dispatch code calls the ABI unchanged") and as the pin of this project's
 * portable dispatch layer (csrc/dispatch.hip).  Semantics implemented in win32_shim.cpp: manual/auto-reset events,
 * wait-all over events or thread handles, SignalObjectAndWait, CreateThread.  Everything else the file touches is
 * error reporting (FormatMessage / MessageBox), mapped to stderr.
 */
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <wchar.h>

typedef void*          HANDLE;
typedef void*          LPVOID;
typedef uint32_t       DWORD;
typedef DWORD*         LPDWORD;
typedef int            BOOL;
typedef uint8_t        BYTE;
typedef char           TCHAR;
typedef char*          LPTSTR;
typedef const char*    LPCTSTR;
typedef const char*    LPCSTR;
typedef void*          HWND;
typedef void*          HLOCAL;
typedef unsigned int   UINT;
typedef size_t         SIZE_T;

#define WINAPI
#define TRUE  1
#define FALSE 0
#define INFINITE              0xFFFFFFFFu
#define WAIT_OBJECT_0         0u
#define WAIT_TIMEOUT          258u
#define WAIT_FAILED           0xFFFFFFFFu
#define MAXIMUM_WAIT_OBJECTS  64
#define MB_OK                 0u
#define LMEM_ZEROINIT         0x40u
#define FORMAT_MESSAGE_ALLOCATE_BUFFER 0x100u
#define FORMAT_MESSAGE_IGNORE_INSERTS  0x200u
#define FORMAT_MESSAGE_FROM_SYSTEM     0x1000u
#define LANG_NEUTRAL    0
#define SUBLANG_DEFAULT 1
#define MAKELANGID(p, s) ((((unsigned)(s)) << 10) | (unsigned)(p))
#define TEXT(x) x

struct SYSTEM_INFO { DWORD dwNumberOfProcessors; };
struct SECURITY_ATTRIBUTES;
typedef DWORD (*LPTHREAD_START_ROUTINE)(LPVOID);

DWORD  GetLastError();
void   SetLastError(DWORD e);
void   GetSystemInfo(SYSTEM_INFO* si);                 /* honours ITW_REF_THREADS (test knob: forces the core count) */
HANDLE CreateEvent(SECURITY_ATTRIBUTES*, BOOL manual_reset, BOOL initial_state, LPCTSTR name);
BOOL   SetEvent(HANDLE h);
BOOL   ResetEvent(HANDLE h);
BOOL   CloseHandle(HANDLE h);
HANDLE CreateThread(SECURITY_ATTRIBUTES*, SIZE_T stack, LPTHREAD_START_ROUTINE fn, LPVOID arg, DWORD flags, LPDWORD tid);
DWORD  WaitForSingleObject(HANDLE h, DWORD ms);
DWORD  WaitForMultipleObjects(DWORD n, const HANDLE* hs, BOOL wait_all, DWORD ms);
DWORD  SignalObjectAndWait(HANDLE to_signal, HANDLE to_wait, DWORD ms, BOOL alertable);

DWORD  FormatMessage(DWORD flags, const void* src, DWORD id, DWORD lang, LPTSTR buf, DWORD n, va_list* args);
HLOCAL LocalAlloc(UINT flags, SIZE_T bytes);
SIZE_T LocalSize(HLOCAL p);
HLOCAL LocalFree(HLOCAL p);
int    lstrlen(LPCTSTR s);
int    MessageBox(HWND, LPCTSTR text, LPCTSTR caption, UINT type);
void   OutputDebugString(LPCTSTR s);

template <size_t N, typename... A>
inline int swprintf_s(wchar_t (&buf)[N], const wchar_t* fmt, A... a) { return swprintf(buf, N, fmt, a...); }
