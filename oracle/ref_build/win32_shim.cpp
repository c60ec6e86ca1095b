/*
 * oracle/ref_build/win32_shim.cpp -- TEST INFRASTRUCTURE: pthread implementation of win32_shim/windows.h.
 *
 * Every handle has its own mutex + condition variable (a process-wide pair made each of the 64 workers' signals wake all
 * the others: thousands of futile wake-ups per CompressImageMT).  Two Win32 guarantees the event protocol of
 * win32Threads.cpp:211-274 relies on are kept: SetEvent on a manual-reset event satisfies every thread waiting at that
 * moment even if the event is reset right afterwards (a per-event pulse count the waiter samples when it starts to wait),
 * and SignalObjectAndWait signals and starts waiting atomically with respect to the waited object.
 */
#include "win32_shim/windows.h"
#include <pthread.h>
#include <stdlib.h>
#include <unistd.h>

namespace {
thread_local DWORD t_last_error = 0;

struct Obj {
    enum Kind { Event, Thread } kind;
    bool manual = false, signalled = false;
    uint64_t pulses = 0;                      /* SetEvent count (manual-reset events) */
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    pthread_cond_t cv = PTHREAD_COND_INITIALIZER;
    pthread_t tid{};
    LPTHREAD_START_ROUTINE fn = nullptr;
    LPVOID arg = nullptr;
};

void signal_locked(Obj* o)
{
    o->signalled = true;
    o->pulses++;
    pthread_cond_broadcast(&o->cv);
}

/* o->mu held; `seen` = pulse count sampled when the wait began */
void wait_locked(Obj* o, uint64_t seen)
{
    while (!(o->signalled || (o->kind == Obj::Event && o->manual && o->pulses != seen))) pthread_cond_wait(&o->cv, &o->mu);
    if (o->kind == Obj::Event && !o->manual) o->signalled = false;          /* auto-reset: consumed by the waiter */
}

void* thread_main(void* p)
{
    Obj* o = static_cast<Obj*>(p);
    o->fn(o->arg);
    pthread_mutex_lock(&o->mu);
    signal_locked(o);                          /* a thread handle becomes signalled when the thread ends */
    pthread_mutex_unlock(&o->mu);
    return nullptr;
}
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

BOOL SetEvent(HANDLE h)
{
    Obj* o = static_cast<Obj*>(h);
    pthread_mutex_lock(&o->mu);
    signal_locked(o);
    pthread_mutex_unlock(&o->mu);
    return TRUE;
}

BOOL ResetEvent(HANDLE h)
{
    Obj* o = static_cast<Obj*>(h);
    pthread_mutex_lock(&o->mu);
    o->signalled = false;
    pthread_mutex_unlock(&o->mu);
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

DWORD WaitForSingleObject(HANDLE h, DWORD)
{
    Obj* o = static_cast<Obj*>(h);
    pthread_mutex_lock(&o->mu);
    wait_locked(o, o->pulses);
    pthread_mutex_unlock(&o->mu);
    return WAIT_OBJECT_0;
}

/* wait-all only (the one form win32Threads.cpp uses): the objects are waited for one after the other.  Win32 would
 * consume the auto-reset events atomically once all are set; with a single waiter (the submitting thread) the outcome
 * is the same. */
DWORD WaitForMultipleObjects(DWORD n, const HANDLE* hs, BOOL wait_all, DWORD ms)
{
    if (!wait_all || n > MAXIMUM_WAIT_OBJECTS) return WAIT_FAILED;
    for (DWORD i = 0; i < n; i++) WaitForSingleObject(hs[i], ms);
    return WAIT_OBJECT_0;
}

/* atomic with respect to the waited object: its pulse count is sampled under its lock BEFORE the other object is
 * signalled, so a SetEvent + ResetEvent pair issued by whoever the signal wakes cannot be missed */
DWORD SignalObjectAndWait(HANDLE to_signal, HANDLE to_wait, DWORD, BOOL)
{
    Obj* w = static_cast<Obj*>(to_wait);
    pthread_mutex_lock(&w->mu);
    const uint64_t seen = w->pulses;
    SetEvent(to_signal);
    wait_locked(w, seen);
    pthread_mutex_unlock(&w->mu);
    return WAIT_OBJECT_0;
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
