// 16-ROW-GRANULAR form of the skinny weight-streaming GEMM: v_mfma_f32_16x16x32 tiles, so that a step of 65..80 / 97..112 /
// 129..144 / 161..176 rows runs an 80 / 112 / 144 / 176-row activation tile instead of padding to the next multiple of 32.
//
// One wave holds ALL MT 16-row activation blocks (like the 160-row class of gemm_kernel.hpp: the waves lie along N) against its NT 16-row weight
// blocks; C^T = W . A^T again, so a lane owns one activation row (column lane % 16 of the MFMA result) and 4 consecutive weight rows per tile.  Same
// LDS-DMA ring, same stage layout [weight tile | activation tile], same XOR swizzle (g_off) as gemm_skinny_kernel.  A 64-deep K tile is two MFMA k-steps
// of 32: per output element the sum order differs from the 32x32x16 kernel's (four k-steps of 16) - results equal up to fp32 summation order, NOT bit
// for bit; a given row count always runs the same kernel.
// PROBE STATE (round 6): split-K fp32 partials only (stored straight from the accumulators).  MEASURED (profiles/r6_gemm16_probe.txt): correct (1e-6 of an
// fp64 product), but 10 % fewer padded rows buy 0-4 % at the 13B widths and nothing at the 7B widths against the 32-row classes - a 16 x 16 tile feeds each
// MFMA with twice the fragment reads per flop.  Not integrated; built only with make EXPERIMENTAL=1.
#pragma once
#include "gemm_kernel.hpp"

namespace lade {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct GMfma16;
template <> struct GMfma16<BF16> {
    __device__ static __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct GMfma16<F16> {
    __device__ static __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

// MT 16-row activation blocks per work-group (all in every wave), NG waves along N with NT 16-row weight blocks each: BN = 16 NT NG
template <typename T, int MT, int NG, int NT>
__global__ __launch_bounds__(G_THREADS) void gemm_skinny16_kernel(GemmK g) {
    constexpr int NW = G_THREADS / 64;
    constexpr int BN = 16 * NT * NG;
    constexpr int BM = 16 * MT;
    constexpr int W_BYTES = BN * 128, A_BYTES = BM * 128, STAGE = W_BYTES + A_BYTES;
    constexpr int W_PIECES = W_BYTES / 1024, A_PIECES = A_BYTES / 1024;
    constexpr int TOTAL_PIECES = W_PIECES + A_PIECES;
    constexpr int PIECES = (TOTAL_PIECES + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ng = wave;
    const bool computes = ng < NG;
    const int l16 = lane & 15, q4 = lane >> 4;
    const int n0 = blockIdx.x * BN, split = blockIdx.y, m0 = blockIdx.z * BM;

    const int k_tiles = (g.K + G_BK - 1) / G_BK;
    const int tps = (k_tiles + g.n_split - 1) / g.n_split;
    const int t0 = split * tps;
    const int nt = max(0, min(t0 + tps, k_tiles) - t0);

    const int64_t w_rs = g.w_ts ? G_BK : g.ldw, w_ts = g.w_ts ? g.w_ts : G_BK;
    const bool moves = wave * PIECES < TOTAL_PIECES;
    const uint16_t* p_src[PIECES];
    int p_dst[PIECES];
    bool p_w[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int piece = min(wave * PIECES + i, TOTAL_PIECES - 1);
        const bool isw = piece < W_PIECES;
        const int p = isw ? piece : piece - W_PIECES;
        const int row = p * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        p_src[i] = (isw ? g.W + (size_t)min(n0 + row, g.N - 1) * w_rs + (size_t)t0 * w_ts
                        : g.A + (size_t)min(m0 + row, g.M - 1) * g.lda + (size_t)t0 * G_BK) + c * 8;
        p_dst[i] = (isw ? 0 : W_BYTES) + p * 1024;
        p_w[i] = isw;
    }
    auto issue = [&](int j, int stage) {
        if (!moves) return;
        unsigned char* sbase = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const uint16_t* src = p_src[i] + (int64_t)j * (p_w[i] ? w_ts : (int64_t)G_BK);
            unsigned char* dst = sbase + p_dst[i];
            if (p_w[i])
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 2);
            else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[a][i][e] = 0.f;

    auto main_loop = [&](auto ns_c) __attribute__((always_inline)) {
        constexpr int NS = decltype(ns_c)::value;
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (s < nt) issue(s, s);
        for (int i = 0; i < nt; ++i) {
            const int stage = i % NS;
            const int younger = min(nt, i + NS) - (i + 1);
            g_wait_younger<PIECES, NS - 1>(younger);
            g_barrier();
            const unsigned char* ws = smem + stage * STAGE;
            const unsigned char* as = ws + W_BYTES;
            if (computes)
#pragma unroll
            for (int ks = 0; ks < G_BK / 32; ++ks) {
                u32x4 af[MT], wf[NT];
#pragma unroll
                for (int a = 0; a < MT; ++a) af[a] = *reinterpret_cast<const u32x4*>(as + g_off(a * 16 + l16, ks * 4 + q4));
#pragma unroll
                for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const u32x4*>(ws + g_off((ng * NT + j) * 16 + l16, ks * 4 + q4));
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[a][j] = GMfma16<T>::run(wf[j], af[a], acc[a][j]);
            }
            if (i + NS < nt) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                g_barrier();
                issue(i + NS, stage);
            }
        }
    };
    switch (g.n_stage) {
        case 2: main_loop(std::integral_constant<int, 2>{}); break;
        case 3: main_loop(std::integral_constant<int, 3>{}); break;
        default: main_loop(std::integral_constant<int, 4>{}); break;
    }

    // fp32 partials [split][M][N] straight from the accumulators: lane = activation row a*16 + l16, four consecutive weight rows per tile
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    float* outp = g.Cpart + (size_t)split * g.M * g.N;
    if (computes)
#pragma unroll
    for (int a = 0; a < MT; ++a) {
        const int m = m0 + a * 16 + l16;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n0 + (ng * NT + j) * 16 + q4 * 4;
            if (m < g.M && n < g.N)
                *reinterpret_cast<float4*>(outp + (size_t)m * g.N + n) = float4{acc[a][j][0], acc[a][j][1], acc[a][j][2], acc[a][j][3]};
        }
    }
}

template <typename T, int MT, int NG, int NT>
static int launch_gemm16(const GemmK& g0, hipStream_t st) {
    constexpr int BN = 16 * NT * NG, BM = 16 * MT;
    constexpr size_t STAGE = (size_t)(BN + BM) * 128;
    GemmK g = g0;
    if (g.n_stage == 0) { const int fit = (int)(G_LDS_MAX / STAGE); g.n_stage = fit >= 4 ? 4 : (fit >= 3 ? 3 : 2); }
    LADE_REQUIRE((g.n_stage == 2 || g.n_stage == 3 || g.n_stage == 4) && g.n_stage * STAGE <= (size_t)G_LDS_MAX, LADE_E_ARG,
                 "lade_gemm_skinny (16-row tiles): a ring of %d stages (2 | 3 | 4) of %d + %d rows does not fit the LDS", g.n_stage, BN, BM);
    LADE_REQUIRE(g.n_split > 1 && g.Cpart && g.epi == 0, LADE_E_ARG, "lade_gemm_skinny (16-row tiles): split-K partials only in this build");
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_skinny16_kernel<T, MT, NG, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS_MAX);
        attr = true;
    }
    dim3 grid(cdiv(g.N, BN), g.n_split, cdiv(g.M, BM));
    hipLaunchKernelGGL((gemm_skinny16_kernel<T, MT, NG, NT>), grid, dim3(G_THREADS), g.n_stage * STAGE, st, g);
    return check_launch("lade_gemm_skinny (16-row tiles)");
}

#define S16(TT, MTv, NGv, NTv) if (mt16 == MTv && ng == NGv && nt16 == NTv) return launch_gemm16<TT, MTv, NGv, NTv>(g, st);
template <typename T>
static int gemm16_dispatch(const GemmK& g, hipStream_t st, int mt16, int ng, int nt16) {
    /* 80 rows */  S16(T,5,8,1) S16(T,5,8,2) S16(T,5,4,2) S16(T,5,8,3) S16(T,5,4,4)
    /* 112 rows */ S16(T,7,8,1) S16(T,7,8,2) S16(T,7,4,2) S16(T,7,4,4)
    /* 144 rows */ S16(T,9,8,1) S16(T,9,8,2) S16(T,9,4,2) S16(T,9,4,4)
    /* 176 rows */ S16(T,11,8,1) S16(T,11,8,2) S16(T,11,4,2)
    return -1;
}
#undef S16

}  // namespace lade
