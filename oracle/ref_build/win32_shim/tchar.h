/* TEST INFRASTRUCTURE: MBCS flavour of <tchar.h> for win32Threads.cpp (see windows.h in this directory). */
#pragma once
#define _T(x) x
