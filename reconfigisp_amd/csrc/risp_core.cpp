// Version + thread-local error text of libreconfigisp_hip (no device code here).
#include <stdarg.h>
#include <stdio.h>
#include "risp.h"

static thread_local char g_err[512] = "";

void risp_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {
int risp_version(void) { return RISP_VERSION; }
const char *risp_last_error(void) { return g_err; }
}
