// Error plumbing + version for the C-ABI library (libpasero_hip.so).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" void pk_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* pk_last_error(void) { return g_err; }

extern "C" int pk_version(void) { return 100; }
