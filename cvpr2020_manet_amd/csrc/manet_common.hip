// version / error plumbing of libmanet_hip.so
#include "manet_common.h"

static thread_local char g_manet_err[512] = "";

int manet_set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_manet_err, sizeof(g_manet_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" {
const char *manet_version(void) { return "manet_hip 0.1 (gfx950)"; }
const char *manet_last_error_string(void) { return g_manet_err; }
}
