/*
 * oracle/ref_build/win32_shim.cpp -- TEST INFRASTRUCTURE: pthread implementation of win32_shim/windows.h.
 *
 * One process-wide mutex + condition variable serve every handle: SetEvent / thread exit broadcast, waiters re-check.
 * Wait-all consumes auto-reset events only once every handle is signalled (Win32's rule).  Two Win32 guarantees the
 * event protocol of win32Threads.cpp:211-274 relies on are kept: SetEvent on a manual-reset event satisfies every
 * thread waiting at that moment even if the event is reset right afterwards (a per-event pulse count the waiter
 * samples when it starts to wait), and SignalObjectAndWait signals and starts waiting atomically (one critical
 * section).
 */
#include "win32_shim/windows.h"
#include <pthread.h>
#include <stdlib.h>
#include <unistd.h>

namespace {
pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
pthread_cond_t  g_cv = PTHREAD_COND_INITIALIZER;
thread_local DWORD t_last_error = 0;

struct Obj {
    enum Kind { Event, Thread } kind;
    bool manual = false, signalled = false;
    uint64_t pulses = 0;                      /* SetEvent count (manual-reset events) */
    pthread_t tid{};
    LPTHREAD_START_ROUTINE fn = nullptr;
    LPVOID arg = nullptr;
};

void* thread_main(void* p)
{
    Obj* o = static_cast<Obj*>(p);
    o->fn(o->arg);
    pthread_mutex_lock(&g_mu);
    o->signalled = true;                       /* a thread handle becomes signalled when the thread ends */
    pthread_cond_broadcast(&g_cv);
    pthread_mutex_unlock(&g_mu);
    return nullptr;
}

void consume(Obj* o) { if (o->kind == Obj::Event && !o->manual) o->signalled = false; }
}

DWORD GetLastError() { return t_last_error; }
void  SetLastError(DWORD e) { t_last_error = e; }

void GetSystemInfo(SYSTEM_INFO* si)
{
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    if (const char* e = getenv("ITW_REF_THREADS")) n = atol(e);
    si->dwNumberOfProcessors = (DWORD)(n < 1 ? 1 : n);
}

HANDLE CreateEvent(SECURITY_ATTRIBUTES*, BOOL manual_reset, BOOL initial_state, LPCTSTR)
{
    Obj* o = new Obj;
    o->kind = Obj::Event;
    o->manual = manual_reset != 0;
    o->signalled = initial_state != 0;
    return o;
}

static void signal_locked(Obj* o)
{
    o->signalled = true;
    o->pulses++;
    pthread_cond_broadcast(&g_cv);
}

BOOL SetEvent(HANDLE h)
{
    pthread_mutex_lock(&g_mu);
    signal_locked(static_cast<Obj*>(h));
    pthread_mutex_unlock(&g_mu);
    return TRUE;
}

BOOL ResetEvent(HANDLE h)
{
    pthread_mutex_lock(&g_mu);
    static_cast<Obj*>(h)->signalled = false;
    pthread_mutex_unlock(&g_mu);
    return TRUE;
}

BOOL CloseHandle(HANDLE h)
{
    Obj* o = static_cast<Obj*>(h);
    if (!o) return FALSE;
    if (o->kind == Obj::Thread) pthread_join(o->tid, nullptr);
    delete o;
    return TRUE;
}

HANDLE CreateThread(SECURITY_ATTRIBUTES*, SIZE_T, LPTHREAD_START_ROUTINE fn, LPVOID arg, DWORD, LPDWORD tid)
{
    Obj* o = new Obj;
    o->kind = Obj::Thread;
    o->fn = fn;
    o->arg = arg;
    if (pthread_create(&o->tid, nullptr, thread_main, o) != 0) { delete o; return nullptr; }
    if (tid) *tid = (DWORD)(uintptr_t)o;
    return o;
}

static DWORD wait_locked(DWORD n, const HANDLE* hs, BOOL wait_all)
{
    uint64_t seen[MAXIMUM_WAIT_OBJECTS];
    for (DWORD i = 0; i < n; i++) seen[i] = static_cast<Obj*>(hs[i])->pulses;
    DWORD ret = WAIT_OBJECT_0;
    for (;;) {
        DWORD ready = 0, first = n;
        for (DWORD i = 0; i < n; i++) {
            Obj* o = static_cast<Obj*>(hs[i]);
            if (o->signalled || (o->kind == Obj::Event && o->manual && o->pulses != seen[i])) { ready++; if (first == n) first = i; }
        }
        if (wait_all ? ready == n : ready > 0) {
            if (wait_all) for (DWORD i = 0; i < n; i++) consume(static_cast<Obj*>(hs[i]));
            else { consume(static_cast<Obj*>(hs[first])); ret = WAIT_OBJECT_0 + first; }
            break;
        }
        pthread_cond_wait(&g_cv, &g_mu);
    }
    return ret;
}

DWORD WaitForMultipleObjects(DWORD n, const HANDLE* hs, BOOL wait_all, DWORD)
{
    if (n > MAXIMUM_WAIT_OBJECTS) return WAIT_FAILED;
    pthread_mutex_lock(&g_mu);
    DWORD ret = wait_locked(n, hs, wait_all);
    pthread_mutex_unlock(&g_mu);
    return ret;
}

DWORD WaitForSingleObject(HANDLE h, DWORD ms) { return WaitForMultipleObjects(1, &h, TRUE, ms); }

DWORD SignalObjectAndWait(HANDLE to_signal, HANDLE to_wait, DWORD, BOOL)
{
    pthread_mutex_lock(&g_mu);
    signal_locked(static_cast<Obj*>(to_signal));
    DWORD ret = wait_locked(1, &to_wait, TRUE);
    pthread_mutex_unlock(&g_mu);
    return ret;
}

DWORD FormatMessage(DWORD, const void*, DWORD id, DWORD, LPTSTR buf, DWORD, va_list*)
{
    /* only used with FORMAT_MESSAGE_ALLOCATE_BUFFER: buf is really a char** */
    char* m = static_cast<char*>(LocalAlloc(LMEM_ZEROINIT, 64));
    snprintf(m, 64, "error %u", id);
    *reinterpret_cast<char**>(buf) = m;
    return (DWORD)strlen(m);
}

HLOCAL LocalAlloc(UINT, SIZE_T bytes)
{
    size_t* p = static_cast<size_t*>(calloc(1, bytes + sizeof(size_t) * 2));
    p[0] = bytes;
    return p + 2;
}
SIZE_T LocalSize(HLOCAL p) { return p ? static_cast<size_t*>(p)[-2] : 0; }
HLOCAL LocalFree(HLOCAL p) { if (p) free(static_cast<size_t*>(p) - 2); return nullptr; }
int    lstrlen(LPCTSTR s) { return s ? (int)strlen(s) : 0; }
int    MessageBox(HWND, LPCTSTR text, LPCTSTR caption, UINT) { fprintf(stderr, "[%s] %s\n", caption, text); return 1; }
void   OutputDebugString(LPCTSTR s) { fputs(s, stderr); }
