/* TEST INFRASTRUCTURE: the one <strsafe.h> function win32Threads.cpp uses (see windows.h in this directory). */
#pragma once
#include <stdio.h>
template <typename... A>
inline long StringCchPrintf(char* dst, size_t n, const char* fmt, A... a) { snprintf(dst, n, fmt, a...); return 0; }
