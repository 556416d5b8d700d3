// Feasibility probe for the fragment-packed weight-streaming GEMM (standalone; hipcc --offload-arch=gfx950 -O3).
//   C[M][N] = A[M][K] . W[N][K]^T,  M = 32*MT rows, bf16 in, fp32 out.
// W is packed offline in MFMA-fragment order: Wp[panel = n/32][kstep = k/16][lane][8]  with lane = (k%16/8)*32 + n%32, so one
// wave-wide global_load_dwordx4 (1 KiB, contiguous) IS the A operand of v_mfma_f32_32x32x16_bf16 for (panel, kstep).
// A is packed the same way: Ap[mb = m/32][kstep][lane][8].  No LDS, no barriers in the main loop: a wave streams a contiguous
// K range of one panel with PD k-steps in flight; the 4 K-groups of a work-group are reduced through LDS once at the end.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <cmath>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <bool NT>
__device__ __forceinline__ u32x4 ld16(const u32x4* p) {
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}

// grid (N/64, S).  WG = 8 waves: wave = (kg = w>>1 in 0..3, pn = w&1).  k-steps of split s: [s*ks_per, (s+1)*ks_per), kg takes a quarter.
template <int MT, int PD, bool NT>
__global__ __launch_bounds__(512) void gemm_v2(const u32x4* __restrict__ Wp, const u32x4* __restrict__ Ap, float* __restrict__ out,
                                               int N, int K, int S) {
    extern __shared__ float4 red[];          // [kg 1..3][pn][mb][g4][lane]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave >> 1, pn = wave & 1;
    const int panel = blockIdx.x * 2 + pn;
    const int ksteps = K / 16;
    const int ks_per_split = ksteps / S, ks_per = ks_per_split / 4;
    const int ks0 = blockIdx.y * ks_per_split + kg * ks_per;
    const u32x4* wp = Wp + ((size_t)panel * ksteps + ks0) * 64 + lane;
    const u32x4* ap = Ap + (size_t)ks0 * 64 + lane;
    const size_t a_mb = (size_t)ksteps * 64;

    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    u32x4 wf[PD], af[PD][MT];
#pragma unroll
    for (int j = 0; j < PD; ++j) {
        if (j < ks_per) {
            wf[j] = ld16<NT>(wp + (size_t)j * 64);
#pragma unroll
            for (int mb = 0; mb < MT; ++mb) af[j][mb] = ap[(size_t)mb * a_mb + (size_t)j * 64];
        }
    }
    for (int ks = 0; ks < ks_per; ks += PD) {
#pragma unroll
        for (int j = 0; j < PD; ++j) {
            if (ks + j < ks_per) {
#pragma unroll
                for (int mb = 0; mb < MT; ++mb)
                    acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[j]), __builtin_bit_cast(bf16x8, af[j][mb]), acc[mb], 0, 0, 0);
                const int nk = ks + j + PD;
                if (nk < ks_per) {
                    wf[j] = ld16<NT>(wp + (size_t)nk * 64);
#pragma unroll
                    for (int mb = 0; mb < MT; ++mb) af[j][mb] = ap[(size_t)mb * a_mb + (size_t)nk * 64];
                }
            }
        }
    }
    // reduce the 4 K-groups through LDS: kg 1..3 write, then every thread sums one (pn, mb, g4, lane) slot ... here simply kg 0 sums
    if (kg > 0) {
#pragma unroll
        for (int mb = 0; mb < MT; ++mb)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                red[((((kg - 1) * 2 + pn) * MT + mb) * 4 + g4) * 64 + lane] = float4{acc[mb][4 * g4], acc[mb][4 * g4 + 1], acc[mb][4 * g4 + 2], acc[mb][4 * g4 + 3]};
    }
    __syncthreads();
    if (kg == 0) {
        float* o = out + (size_t)blockIdx.y * (32 * MT) * N;
#pragma unroll
        for (int mb = 0; mb < MT; ++mb)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float4 v = float4{acc[mb][4 * g4], acc[mb][4 * g4 + 1], acc[mb][4 * g4 + 2], acc[mb][4 * g4 + 3]};
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const float4 r = red[(((q * 2 + pn) * MT + mb) * 4 + g4) * 64 + lane];
                    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                }
                const int m = mb * 32 + (lane & 31), n = panel * 32 + 8 * g4 + 4 * (lane >> 5);
                *reinterpret_cast<float4*>(o + (size_t)m * N + n) = v;
            }
    }
}

__global__ void ref_gemm(const uint16_t* A, const uint16_t* W, float* out, int M, int N, int K) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc += __uint_as_float((uint32_t)A[(size_t)m * K + k] << 16) * __uint_as_float((uint32_t)W[(size_t)n * K + k] << 16);
    out[(size_t)m * N + n] = acc;
}

// row-major [R][K] -> fragment order [R/32][K/16][64 lanes][8]
__global__ void pack_frag(const uint16_t* src, uint16_t* dst, int R, int K) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // one 16-byte chunk
    const size_t total = (size_t)R * K / 8;
    if (idx >= total) return;
    const int lane = idx & 63;
    const size_t t = idx >> 6;
    const int ks = t % (K / 16);
    const int rb = t / (K / 16);
    const int r = rb * 32 + (lane & 31), k = ks * 16 + (lane >> 5) * 8;
    *reinterpret_cast<uint4*>(dst + idx * 8) = *reinterpret_cast<const uint4*>(src + (size_t)r * K + k);
}

template <int MT, int PD, bool NT>
static float run(const uint16_t* const* Wp, int nW, const uint16_t* Ap, float* out, int N, int K, int S, int reps) {
    const size_t lds = (size_t)3 * 2 * MT * 4 * 64 * 16;
    CK(hipFuncSetAttribute((const void*)gemm_v2<MT, PD, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid(N / 64, S);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_v2<MT, PD, NT>), grid, dim3(512), lds, 0, (const u32x4*)Wp[i % nW], (const u32x4*)Ap, out, N, K, S);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gemm_v2<MT, PD, NT>), grid, dim3(512), lds, 0, (const u32x4*)Wp[i % nW], (const u32x4*)Ap, out, N, K, S);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

int main() {
    struct Shape { const char* name; int N, K; } shapes[] = {{"qkv", 12288, 4096}, {"o", 4096, 4096}, {"gate_up", 22016, 4096}, {"down", 4096, 11008}};
    const int M = 64;
    for (auto& sh : shapes) {
        const int N = sh.N, K = sh.K;
        const size_t wbytes = (size_t)N * K * 2;
        const int nW = (int)std::max<size_t>(2, (600u << 20) / wbytes + 1);
        std::vector<uint16_t*> Wp(nW);
        uint16_t *Wrow, *Arow, *Ap;
        float *out, *ref;
        CK(hipMalloc(&Wrow, wbytes)); CK(hipMalloc(&Arow, (size_t)2 * M * K * 2)); CK(hipMalloc(&Ap, (size_t)2 * M * K * 2));
        CK(hipMalloc(&out, (size_t)4 * 2 * M * N * 4)); CK(hipMalloc(&ref, (size_t)M * N * 4));
        std::vector<uint16_t> h((size_t)N * K);
        uint32_t s = 12345;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 + ((s >> 16) & 0x1ff) + ((s >> 30) << 15)); }     // ~ +-[0.008, 0.03]
        CK(hipMemcpy(Wrow, h.data(), wbytes, hipMemcpyHostToDevice));
        std::vector<uint16_t> ha((size_t)2 * M * K);
        for (auto& v : ha) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x3f00 + ((s >> 16) & 0xff) + ((s >> 30) << 15)); }
        CK(hipMemcpy(Arow, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
        for (int i = 0; i < nW; ++i) {
            CK(hipMalloc(&Wp[i], wbytes));
            hipLaunchKernelGGL(pack_frag, dim3((unsigned)((size_t)N * K / 8 / 256)), dim3(256), 0, 0, Wrow, Wp[i], N, K);
        }
        hipLaunchKernelGGL(pack_frag, dim3((unsigned)((size_t)2 * M * K / 8 / 256)), dim3(256), 0, 0, Arow, Ap, 2 * M, K);
        hipLaunchKernelGGL(ref_gemm, dim3(N / 256, M), dim3(256), 0, 0, Arow, Wrow, ref, M, N, K);
        CK(hipDeviceSynchronize());
        // correctness (S = 1)
        run<2, 8, false>(Wp.data(), nW, Ap, out, N, K, 1, 1);
        std::vector<float> ho((size_t)M * N), hr((size_t)M * N);
        CK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hr.data(), ref, hr.size() * 4, hipMemcpyDeviceToHost));
        double maxerr = 0, maxref = 0;
        for (size_t i = 0; i < ho.size(); ++i) { maxerr = std::max(maxerr, (double)std::fabs(ho[i] - hr[i])); maxref = std::max(maxref, (double)std::fabs(hr[i])); }
        printf("%-8s N=%5d K=%5d  W=%.1f MB  max|err|=%.3g (max|ref|=%.3g)\n", sh.name, N, K, wbytes / 1e6, maxerr, maxref);
        const int Ss[] = {1, 2, 4};
        for (int S : Ss) {
            if ((K / 16) % (4 * S) != 0) continue;
            float t;
            t = run<2, 4, false>(Wp.data(), nW, Ap, out, N, K, S, 60);  printf("   S=%d PD=4         %7.2f us  %6.2f TB/s\n", S, t, wbytes / t / 1e6);
            t = run<2, 8, false>(Wp.data(), nW, Ap, out, N, K, S, 60);  printf("   S=%d PD=8         %7.2f us  %6.2f TB/s\n", S, t, wbytes / t / 1e6);
            t = run<2, 8, true>(Wp.data(), nW, Ap, out, N, K, S, 60);   printf("   S=%d PD=8  nt     %7.2f us  %6.2f TB/s\n", S, t, wbytes / t / 1e6);
            t = run<2, 16, false>(Wp.data(), nW, Ap, out, N, K, S, 60); printf("   S=%d PD=16        %7.2f us  %6.2f TB/s\n", S, t, wbytes / t / 1e6);
            t = run<2, 16, true>(Wp.data(), nW, Ap, out, N, K, S, 60);  printf("   S=%d PD=16 nt     %7.2f us  %6.2f TB/s\n", S, t, wbytes / t / 1e6);
            t = run<4, 8, true>(Wp.data(), nW, Ap, out, N, K, S, 60);   printf("   S=%d PD=8  nt MT=4 (A rows re-read; timing only) %7.2f us  %6.2f TB/s\n", S, t, wbytes / t / 1e6);
        }
        for (int i = 0; i < nW; ++i) CK(hipFree(Wp[i]));
        CK(hipFree(Wrow)); CK(hipFree(Arow)); CK(hipFree(Ap)); CK(hipFree(out)); CK(hipFree(ref));
    }
    return 0;
}
