// dcl_capi.cpp -- error plumbing and host-only helpers of the C ABI (include/dcl_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/dcl_hip.h"

static thread_local char g_err[512] = "";

void dcl_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *dcl_last_error(void) { return g_err; }

// Which kernel symbol did the last convolution / weight-gradient entry point of this thread launch?  (The tile and variant
// are chosen inside the library; bench.py's in-step kernel timer names its rows with this.)  Off unless switched on.
static thread_local char g_kernel[160] = "";
static int g_trace_kernels = 0;

void dcl_note_kernel(const char *fmt, ...)
{
    if (!g_trace_kernels)
        return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
    va_end(ap);
}

extern "C" int dcl_trace_kernels(int on)
{
    g_trace_kernels = on ? 1 : 0;
    g_kernel[0] = 0;
    return 0;
}

extern "C" const char *dcl_last_kernel(void) { return g_kernel; }
extern "C" int dcl_version(void) { return 1; }

// Column splits for the sweep kernels: row blocks x splits workgroups should fill a whole
// number of 256-CU rounds as evenly as possible without making the splits tiny.
extern "C" int dcl_suggest_nsplit(int N1, int N2)
{
    if (N1 <= 0 || N2 <= 0)
        return 1;
    int rb = (N1 + DCL_ROW_TILE - 1) / DCL_ROW_TILE;
    int nchunk = (N2 + 31) / 32;
    int best = 1;
    double best_eff = 0.0;
    for (int s = 1; s <= 32 && s <= nchunk; ++s) {
        int wgs = rb * s;
        int rounds = (wgs + 255) / 256;
        double eff = (double)wgs / (rounds * 256.0);
        // each split should keep >= 4 chunks of work to amortise the A-panel load
        if (nchunk / s < 4 && s > 1)
            break;
        if (eff > best_eff + 0.02) {
            best_eff = eff;
            best = s;
        }
    }
    return best;
}
