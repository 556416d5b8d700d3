// liblade_hip.so: error channel, version and the kernel-timing helper used by bench.py.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "common.hpp"

namespace lade {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return LADE_E_LAUNCH;
    }
    return LADE_OK;
}

}  // namespace lade

namespace lade {
int debug_int(const char* name) {
    const char* e = getenv("LADE_DEBUG");
    const size_t n = strlen(name);
    while (e && *e) {
        while (*e == ',' || *e == ' ') ++e;
        const char* end = strchr(e, ',');
        const size_t len = end ? (size_t)(end - e) : strlen(e);
        if (len >= n && strncmp(e, name, n) == 0 && (len == n || e[n] == '=')) return len == n ? 1 : atoi(e + n + 1);
        e += len;
    }
    return 0;
}
}  // namespace lade

extern "C" int lade_version(void) { return LADE_ABI_VERSION; }
extern "C" int lade_build_flags(void) {
#ifdef LADE_EXPERIMENTAL
    return 1;
#else
    return 0;
#endif
}
extern "C" const char* lade_last_error_string(void) { return lade::g_err; }

// Mean duration (us) of `reps` back-to-back launches of the attention kernel pair on `stream`,
// measured with hipEvents recorded on that same stream.
extern "C" int lade_time_attn(const lade_attn_args* a, int32_t reps, float* mean_us, void* stream) {
    LADE_REQUIRE(a && mean_us && reps > 0, LADE_E_ARG, "lade_time_attn: bad args");
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        lade::set_error("lade_time_attn: hipEventCreate failed");
        return LADE_E_LAUNCH;
    }
    int rc = lade_attn_fwd(a, stream);            // warm-up, also validates
    if (rc == 0 && a->n_splits > 1) rc = lade_attn_combine(a, stream);
    if (rc == 0) {
        (void)hipEventRecord(e0, st);
        for (int i = 0; i < reps && rc == 0; ++i) {
            rc = lade_attn_fwd(a, stream);
            if (rc == 0 && a->n_splits > 1) rc = lade_attn_combine(a, stream);
        }
        (void)hipEventRecord(e1, st);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        *mean_us = ms * 1000.f / (float)reps;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

// Same measurement over `n` argument sets used round-robin (one launch pair per repetition): with n K/V caches whose
// total size exceeds the 256 MB Infinity Cache every launch streams its keys/values from HBM, as a decode step does
// (consecutive layers own different caches); with n = 1 the cache stays resident in the Infinity Cache.
extern "C" int lade_time_attn_rot(const lade_attn_args* a, int32_t n, int32_t reps, float* mean_us, void* stream) {
    LADE_REQUIRE(a && mean_us && reps > 0 && n > 0, LADE_E_ARG, "lade_time_attn_rot: bad args");
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        lade::set_error("lade_time_attn_rot: hipEventCreate failed");
        return LADE_E_LAUNCH;
    }
    int rc = 0;
    for (int i = 0; i < n && rc == 0; ++i) {          // one untimed pass: validation, code objects, LDS attributes
        rc = lade_attn_fwd(a + i, stream);
        if (rc == 0 && a[i].n_splits > 1) rc = lade_attn_combine(a + i, stream);
    }
    if (rc == 0) {
        (void)hipEventRecord(e0, st);
        for (int i = 0; i < reps && rc == 0; ++i) {
            const lade_attn_args* x = a + (i % n);
            rc = lade_attn_fwd(x, stream);
            if (rc == 0 && x->n_splits > 1) rc = lade_attn_combine(x, stream);
        }
        (void)hipEventRecord(e1, st);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        *mean_us = ms * 1000.f / (float)reps;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}
