// XCD placement census + same-XCD hand-off probe for the MI355X box
// (standalone; build: hipcc --offload-arch=gfx950 -O3 tools/xcd_probe.hip -o tools/xcd_probe).
//
// The split-KV attention merges its key splits; a merge INSIDE the launch is only cheaper than a second launch when the splits of one
// head hand their partials over through ONE XCD's L2 (plain stores, s_waitcnt vmcnt(0), an L2-scope atomic, L1-bypassing loads) -
// a cross-XCD hand-off costs what the launch boundary costs.  That needs two facts about this chip/runtime:
//   1. census : does block b of a 1-D grid run on XCD b % 8 (HW_REG_XCC_ID), for the attention kernel's resource shape
//               (512 threads, ~128 KB of LDS), idle and next to another kernel on a second stream?
//   2. handoff: groups of NS work-groups on one XCD publish a payload each; the last arriver (L2-scope counter) re-reads all NS
//               payloads with sc1 loads and checks EVERY word.  Uneven load (per-block delay), consumer L1 warmed with the stale
//               payload first, thousands of launches with changing patterns.  Reported: stale words, and s_memtime spans of the hand-off.
//   3. the same hand-off with device-scope operations (sc1 write-through stores, agent-scope atomic) for the price comparison.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

__global__ __launch_bounds__(512) void census_kernel(uint32_t* xcc, uint32_t* cu, int spin) {
    extern __shared__ unsigned char smem[];
    if (threadIdx.x == 0) {
        xcc[blockIdx.x] = xcc_id();
        uint32_t hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        cu[blockIdx.x] = hw;
        smem[0] = 1;
    }
    // stay resident for a while so that the whole grid is in flight at once (like the attention launch)
    unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) {}
}

__global__ void noise_kernel(float* p, int n, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    float a = p[i % n];
    for (int k = 0; k < iters; ++k) a = a * 1.0001f + 0.5f;
    p[i % n] = a;
}

// ---- hand-off -------------------------------------------------------------------------------------------------
// payload of group g, split s: PW 16-byte words per thread x 512 threads; word value = f(seq, g, s, index)
constexpr int PAY_PER_THREAD = 2;                 // 2 x 16 B x 512 threads = 16 KB per work-group (the attention partial is 15 KB)
__device__ __forceinline__ uint32_t pat(uint32_t seq, uint32_t g, uint32_t s, uint32_t i) { return (seq * 2654435761u) ^ (g * 40503u) ^ (s << 24) ^ (i * 97u + 12345u); }

struct HandArgs {
    u32x4* payload;        // [groups][ns][PAY_PER_THREAD*512]
    uint32_t* counter;     // [groups]  (one 128-byte line each)
    uint32_t* errors;      // [0] stale words, [1] merges done, [2] groups whose members sat on different XCDs (by recorded id)
    uint32_t* xccs;        // [groups][ns]
    unsigned long long* stamps;   // [groups][4]
    int ns, groups, seq, mode, delay;   // mode 0: same-XCD (L2 scope), 1: device scope
};

__global__ __launch_bounds__(512) void handoff_kernel(HandArgs a) {
    const int b = blockIdx.x;
    const int xcd = b & 7, j = b >> 3;
    const int g = xcd + 8 * (j / a.ns), s = j % a.ns;
    if (g >= a.groups) return;
    const int tid = threadIdx.x;
    u32x4* mine = a.payload + ((size_t)g * a.ns + s) * (PAY_PER_THREAD * 512);
    // warm this CU's L1 with the STALE payload of every split of the group (what a previous layer's merge left behind)
    uint32_t warm = 0;
    for (int ss = 0; ss < a.ns; ++ss) {
        const u32x4* p = a.payload + ((size_t)g * a.ns + ss) * (PAY_PER_THREAD * 512);
        for (int k = 0; k < PAY_PER_THREAD; ++k) warm ^= p[k * 512 + tid][0];
    }
    // uneven load: blocks are delayed by different amounts
    const unsigned long long d = (unsigned long long)a.delay * ((b * 2654435761u >> 28) & 15);
    unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < d) {}
    const unsigned long long t_pub = __builtin_readcyclecounter();
    for (int k = 0; k < PAY_PER_THREAD; ++k) {
        const uint32_t i = k * 512 + tid;
        u32x4 v = {pat(a.seq, g, s, 4 * i), pat(a.seq, g, s, 4 * i + 1), pat(a.seq, g, s, 4 * i + 2), pat(a.seq, g, s, 4 * i + 3) ^ (warm & 0u)};
        if (a.mode == 0) mine[i] = v;
    }
    if (a.mode == 1) {
        // device-scope variant: write-through stores (sc0 sc1)
        for (int k = 0; k < PAY_PER_THREAD; ++k) {
            const uint32_t i = k * 512 + tid;
            u32x4 v = {pat(a.seq, g, s, 4 * i), pat(a.seq, g, s, 4 * i + 1), pat(a.seq, g, s, 4 * i + 2), pat(a.seq, g, s, 4 * i + 3)};
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(mine + i), "v"(v) : "memory");
        }
    }
    if (tid == 0) a.xccs[g * a.ns + s] = xcc_id();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ uint32_t old_s;
    if (tid == 0) {
        uint32_t old;
        if (a.mode == 0) old = __hip_atomic_fetch_add(a.counter + g * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else old = __hip_atomic_fetch_add(a.counter + g * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old_s = old;
    }
    __syncthreads();
    if (old_s != (uint32_t)(a.ns - 1)) return;
    const unsigned long long t_last = __builtin_readcyclecounter();
    // last arriver of the group: re-read every split's payload, bypassing this CU's L1
    uint32_t bad = 0;
    for (int ss = 0; ss < a.ns; ++ss) {
        const u32x4* p = a.payload + ((size_t)g * a.ns + ss) * (PAY_PER_THREAD * 512);
        u32x4 v[PAY_PER_THREAD];
        for (int k = 0; k < PAY_PER_THREAD; ++k) {
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[k]) : "v"(p + k * 512 + tid) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int k = 0; k < PAY_PER_THREAD; ++k) {
            asm volatile("" : "+v"(v[k]));
            const uint32_t i = k * 512 + tid;
            for (int e = 0; e < 4; ++e) bad += v[k][e] != pat(a.seq, g, ss, 4 * i + e);
        }
    }
    const unsigned long long t_done = __builtin_readcyclecounter();
    if (bad) atomicAdd(a.errors, bad);
    if (tid == 0) {
        atomicAdd(a.errors + 1, 1u);
        uint32_t x0 = xcc_id(), spread = 0;
        for (int ss = 0; ss < a.ns; ++ss) spread |= (a.xccs[g * a.ns + ss] != x0);      // plain load: informational only
        if (spread) atomicAdd(a.errors + 2, 1u);
        // reset through the same path the arrivals take (an RMW): a plain / sc1 store could sit in a different cache level than the
        // one that executes the atomics
        (void)__hip_atomic_fetch_add(a.counter + g * 32, (uint32_t)(0 - a.ns), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.stamps[g * 4 + 0] = t_pub; a.stamps[g * 4 + 1] = t_last; a.stamps[g * 4 + 2] = t_done;
    }
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    hipStream_t s0, s1;
    CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
    // ---- 1. census ----
    const int lds = 128 * 1024;
    CK(hipFuncSetAttribute((const void*)census_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    float* noise; CK(hipMalloc(&noise, 1 << 20));
    for (int with_noise = 0; with_noise < 2; ++with_noise)
        for (int n : {32, 192, 256, 264, 1024}) {
            uint32_t *xcc, *cu;
            CK(hipMalloc(&xcc, n * 4)); CK(hipMalloc(&cu, n * 4));
            int viol = 0, launches = 200;
            std::vector<uint32_t> h(n), hc(n);
            std::vector<int> per_xcd(8, 0);
            for (int l = 0; l < launches; ++l) {
                if (with_noise) hipLaunchKernelGGL(noise_kernel, dim3(64 + (l % 5) * 37), dim3(256), 0, s1, noise, 1 << 18, 2000);
                hipLaunchKernelGGL(census_kernel, dim3(n), dim3(512), lds, s0, xcc, cu, 3000 + (l % 3) * 2000);
                CK(hipStreamSynchronize(s0));
                CK(hipMemcpy(h.data(), xcc, n * 4, hipMemcpyDeviceToHost));
                for (int b = 0; b < n; ++b) { viol += (int)h[b] != (b & 7); if (l == 0) per_xcd[h[b] & 7]++; }
            }
            CK(hipStreamSynchronize(s1));
            printf("census grid=%4d noise=%d: %d launches, blocks with XCC_ID != b%%8: %d   (blocks per XCD, first launch: %d %d %d %d %d %d %d %d)\n", n, with_noise, launches, viol,
                   per_xcd[0], per_xcd[1], per_xcd[2], per_xcd[3], per_xcd[4], per_xcd[5], per_xcd[6], per_xcd[7]);
            CK(hipFree(xcc)); CK(hipFree(cu));
        }
    // ---- 2./3. hand-off ----
    for (int mode = 0; mode < 2; ++mode)
        for (int ns : {6, 8}) {
            const int groups = 32;
            HandArgs a;
            const size_t pay_words = (size_t)groups * ns * PAY_PER_THREAD * 512;
            CK(hipMalloc(&a.payload, pay_words * 16)); CK(hipMemset(a.payload, 0, pay_words * 16));
            CK(hipMalloc(&a.counter, groups * 128)); CK(hipMemset(a.counter, 0, groups * 128));
            CK(hipMalloc(&a.errors, 16)); CK(hipMemset(a.errors, 0, 16));
            CK(hipMalloc(&a.xccs, groups * ns * 4)); CK(hipMemset(a.xccs, 0, groups * ns * 4));
            CK(hipMalloc(&a.stamps, groups * 4 * 8)); CK(hipMemset(a.stamps, 0, groups * 4 * 8));
            a.ns = ns; a.groups = groups; a.mode = mode;
            const int grid = 8 * ((groups + 7) / 8) * ns;
            const int launches = 3000;
            double sum_wait = 0, sum_read = 0; int n_st = 0;
            std::vector<unsigned long long> st(groups * 4);
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            float ms_total = 0;
            for (int l = 0; l < launches; ++l) {
                a.seq = l + 1; a.delay = (l % 4 == 0) ? 0 : 300 * (l % 7);
                if (l % 3 == 1) hipLaunchKernelGGL(noise_kernel, dim3(40 + (l % 11) * 20), dim3(256), 0, s1, noise, 1 << 18, 1500);
                if (l >= launches - 200) { a.delay = 0; CK(hipEventRecord(e0, s0)); }
                hipLaunchKernelGGL(handoff_kernel, dim3(grid), dim3(512), 0, s0, a);
                if (l >= launches - 200) {
                    CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms_total += ms;
                    CK(hipMemcpy(st.data(), a.stamps, groups * 4 * 8, hipMemcpyDeviceToHost));
                    for (int g = 0; g < groups; ++g) { sum_wait += (double)(st[g * 4 + 1] - st[g * 4 + 0]); sum_read += (double)(st[g * 4 + 2] - st[g * 4 + 1]); ++n_st; }
                }
            }
            CK(hipDeviceSynchronize());
            uint32_t err[4];
            CK(hipMemcpy(err, a.errors, 16, hipMemcpyDeviceToHost));
            printf("handoff mode=%s ns=%d: %d launches x %d groups: merges %u (expected %d), STALE WORDS %u, groups spread over XCDs %u;  last arriver: publish->arrival %.0f cycles, "
                   "re-read of %d x 16 KB %.0f cycles;  kernel %.2f us (no delay, last 200 launches)\n",
                   mode == 0 ? "same-XCD(L2 scope, plain stores)" : "device scope (sc1 stores, agent atomic)", ns, launches, groups, err[1], launches * groups, err[0], err[2],
                   sum_wait / n_st, ns, sum_read / n_st, ms_total / 200 * 1e3);
        }
    return 0;
}
