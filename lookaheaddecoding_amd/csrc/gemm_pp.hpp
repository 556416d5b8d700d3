// PING-PONG form of the skinny weight-streaming GEMM: the eight waves of a work-group are two groups of four (one wave of each group on
// every SIMD) that ALTERNATE roles from K tile to K tile - while group p % 2 multiplies tile p (fragment reads + MFMAs), the other group
// requests tile p - 1 + NS into the ring slot it has just left and waits for the tile it multiplies next.  One s_barrier per K tile.
//
// Why: in gemm_skinny_kernel all eight waves walk through [wait - barrier - fragment reads - MFMAs - barrier - DMA requests] in lock-step, so
// a K tile costs the SUM of those phases (13B gate/up at 120 rows: ~1650 cycles per 16 KB of weights, profiles/r6_clock_probe.txt) and a CU is
// bound by its own instruction stream, not by HBM; at 120 rows the chip also runs into its power cap (1.34 kW, 2.06 GHz) and every cycle of that
// chain gets longer.  Here the MFMA phase of one group hides the request phase of the other: a tile costs max(multiply, request) instead.
//
// A group's four waves form an MW x NG grid (MW * NG == 4) of MT x NT MFMA tiles each; every tile of the K slice is requested by the group
// that multiplies it (NS is even), 32 / 4 pieces per wave.  Each group accumulates the tiles of its parity; at the end group 1 hands its
// accumulators to group 0 through the LDS (same lane, same register: conflict-free 16-byte rows) and the shared epilogue runs on the sum.
// Per output element: (sum over even tiles, in order) + (sum over odd tiles, in order) - deterministic, NOT the bits of gemm_skinny_kernel's
// single chain (tests compare against fp32 references and pin determinism / K-tile-major == row-major).
#pragma once
#include "gemm_kernel.hpp"

namespace lade {

constexpr int g_pp_stages(int bn, int bm) {
    const int fit = G_LDS_MAX / ((bn + bm) * 128);
    return fit >= 6 ? 6 : (fit >= 4 ? 4 : 2);
}

template <typename T, int MW, int MT, int NG, int NT>
__global__ __launch_bounds__(G_THREADS) void gemm_pp_kernel(GemmK g) {
    static_assert(MW * NG == 4, "a group is four waves, one per SIMD");
    constexpr int GW = 4;                       // waves per group
    constexpr int MB = MW * MT;
    constexpr int BN = 32 * NT * NG;
    constexpr int BM = 32 * MB;
    constexpr int W_BYTES = BN * 128, A_BYTES = BM * 128, STAGE = W_BYTES + A_BYTES;
    constexpr int W_PIECES = W_BYTES / 1024, A_PIECES = A_BYTES / 1024;
    constexpr int TOTAL_PIECES = W_PIECES + A_PIECES;
    constexpr int PIECES = (TOTAL_PIECES + GW - 1) / GW;                     // per wave of the requesting group and tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3;                                // waves w and w + 4 share a SIMD: one of each group per SIMD
    const int mw = wq % MW, ng = wq / MW;
    const int ql = lane & 31, hi = lane >> 5;
    const int n0 = blockIdx.x * BN, split = blockIdx.y, m0 = blockIdx.z * BM;

    const int k_tiles = (g.K + G_BK - 1) / G_BK;
    const int tps = (k_tiles + g.n_split - 1) / g.n_split;
    const int t0 = split * tps;
    const int nt = max(0, min(t0 + tps, k_tiles) - t0);

    // this wave's DMA pieces of a tile, resolved once (as in gemm_skinny_kernel): source pointer at the split's first K tile + offset inside a stage
    const int64_t w_rs = g.w_ts ? G_BK : g.ldw, w_ts = g.w_ts ? g.w_ts : G_BK;
    const bool moves = wq * PIECES < TOTAL_PIECES;
    const uint16_t* p_src[PIECES];
    int p_dst[PIECES];
    bool p_w[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int piece = min(wq * PIECES + i, TOTAL_PIECES - 1);
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
            if (p_w[i])          // non-temporal weight stream (every byte read once per step by one work-group)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 2);
            else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][i][e] = 0.f;

    auto main_loop = [&](auto ns_c) __attribute__((always_inline)) {
        constexpr int NS = decltype(ns_c)::value;
        static_assert(NS % 2 == 0, "every tile is requested by the group that multiplies it");
        constexpr int MAXY = (NS - 2) / 2;                                   // own tiles that may stay in flight behind the one waited for
        // prologue: tiles 0 .. NS - 2, each by its own group; group 0 waits for tile 0
#pragma unroll
        for (int s = 0; s < NS - 1; ++s)
            if (s < nt && (s & 1) == grp) issue(s, s);
        if (grp == 0) {
            int younger = 0;                                                 // tiles 2, 4, .. <= NS - 2 requested behind tile 0
#pragma unroll
            for (int s = 2; s < NS - 1; s += 2) younger += s < nt ? 1 : 0;
            g_wait_younger<PIECES, MAXY>(younger);
        }
        g_barrier();
        for (int p = 0; p < nt; ++p) {
            if ((p & 1) == grp) {
                // ---- multiply tile p ----
                const unsigned char* ws = smem + (p % NS) * STAGE;
                const unsigned char* as = ws + W_BYTES;
                if (!(g.dbg & 4))
#pragma unroll
                for (int kk = 0; kk < G_BK / 16; ++kk) {
                    u32x4 af[MT], wf[NT];
#pragma unroll
                    for (int a = 0; a < MT; ++a) af[a] = *reinterpret_cast<const u32x4*>(as + g_off((mw * MT + a) * 32 + ql, kk * 2 + hi));
#pragma unroll
                    for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const u32x4*>(ws + g_off((ng * NT + j) * 32 + ql, kk * 2 + hi));
#pragma unroll
                    for (int a = 0; a < MT; ++a)
#pragma unroll
                        for (int j = 0; j < NT; ++j) acc[a][j] = GMfma<T>::run(wf[j], af[a], acc[a][j]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the slot is this group's to refill after the barrier
            } else {
                // ---- request tile p - 1 + NS into the slot of tile p - 1 (multiplied by this group in the previous phase), then wait for tile p + 1 ----
                const int j = p - 1 + NS;
                if (j < nt) issue(j, j % NS);
                if (p + 1 < nt) {
                    // own tiles behind tile p + 1: p + 3, p + 5, .. up to min(nt - 1, p - 1 + NS)
                    const int last = min(nt - 1, p - 1 + NS);
                    const int younger = last >= p + 3 ? (last - (p + 3)) / 2 + 1 : 0;
                    g_wait_younger<PIECES, MAXY>(younger);
                }
            }
            g_barrier();
        }
    };
    switch (g.n_stage) {
        case 2: main_loop(std::integral_constant<int, 2>{}); break;
        case 6: main_loop(std::integral_constant<int, 6>{}); break;
        default: main_loop(std::integral_constant<int, 4>{}); break;
    }

    // ---- group 1 hands its accumulators to group 0: [wave][tile][e / 4][lane] rows of 16 bytes ----
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    g_barrier();
    {
        float4* xb = reinterpret_cast<float4*>(smem) + (size_t)wq * (MT * NT * 4 * 64) + lane;
        if (grp == 1) {
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4)
                        xb[((a * NT + j) * 4 + e4) * 64] = float4{acc[a][j][4 * e4 + 0], acc[a][j][4 * e4 + 1], acc[a][j][4 * e4 + 2], acc[a][j][4 * e4 + 3]};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        g_barrier();
        if (grp == 0) {
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        const float4 v = xb[((a * NT + j) * 4 + e4) * 64];
                        acc[a][j][4 * e4 + 0] += v.x; acc[a][j][4 * e4 + 1] += v.y; acc[a][j][4 * e4 + 2] += v.z; acc[a][j][4 * e4 + 3] += v.w;
                    }
        }
    }
    // (gemm_epilogue opens with s_waitcnt lgkmcnt(0) + a barrier: group 0's reads are done before the staging writes)
    gemm_epilogue<T, MW, MT, NG, NT>(g, acc, smem, mw, ng, grp == 0, n0, m0, split);
}

template <typename T, int MW, int MT, int NG, int NT>
static int launch_gemm_pp(const GemmK& g0, hipStream_t st) {
    constexpr int BN = 32 * NT * NG, BM = 32 * MW * MT;
    constexpr size_t STAGE = (size_t)(BN + BM) * 128;
    static_assert((size_t)BM * BN * 4 <= (size_t)G_LDS_MAX, "the hand-over of group 1's accumulators must fit the LDS");
    GemmK g = g0;
    if (g.n_stage == 0) g.n_stage = g_pp_stages(BN, BM);
    LADE_REQUIRE((g.n_stage == 2 || g.n_stage == 4 || g.n_stage == 6) && g.n_stage * STAGE <= (size_t)G_LDS_MAX, LADE_E_ARG,
                 "lade_gemm_skinny (ping-pong): a ring of %d stages (2 | 4 | 6) of %d + %d rows does not fit the %d KB of LDS", g.n_stage, BN, BM, G_LDS_MAX / 1024);
    size_t stg = g.n_split == 1 ? (g.epi == 1 ? 0 : (g.epi == 2 ? (size_t)NG * BM * 8 : (size_t)BM * (BN * 2 + 16))) : (size_t)BM * (BN * 4 + 16);
    if (g.n_split > 1 && stg > (size_t)G_LDS_MAX) { g.dbg |= 32; stg = 0; }
    size_t lds = g.n_stage * STAGE;
    if (stg > lds) lds = stg;
    if ((size_t)BM * BN * 4 > lds) lds = (size_t)BM * BN * 4;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_pp_kernel<T, MW, MT, NG, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS_MAX);
        attr = true;
    }
    dim3 grid(cdiv(g.N, BN), g.n_split, cdiv(g.M, BM));
    hipLaunchKernelGGL((gemm_pp_kernel<T, MW, MT, NG, NT>), grid, dim3(G_THREADS), lds, st, g);
    return check_launch("lade_gemm_skinny (ping-pong)");
}

// the group wave grids that are built: (MW x NG == 4 waves, MT x NT MFMA tiles per wave)
#define PP_SHAPE(TT, MWv, MTv, NGv, NTv) if (mw == MWv && mt == MTv && ng == NGv && nt == NTv) return launch_gemm_pp<TT, MWv, MTv, NGv, NTv>(g, st);
#define PP_GO(TT)                                                                                                       \
    /* 32 rows */  PP_SHAPE(TT,1,1,4,1) PP_SHAPE(TT,1,1,4,2)                                                            \
    /* 64 rows */  PP_SHAPE(TT,2,1,2,1) PP_SHAPE(TT,2,1,2,2) PP_SHAPE(TT,2,1,2,3) PP_SHAPE(TT,2,1,2,4) PP_SHAPE(TT,1,2,4,1) PP_SHAPE(TT,1,2,4,2) \
    /* 96 rows */  PP_SHAPE(TT,1,3,4,1) PP_SHAPE(TT,1,3,4,2)                                                            \
    /* 128 rows */ PP_SHAPE(TT,2,2,2,1) PP_SHAPE(TT,2,2,2,2) PP_SHAPE(TT,2,2,2,3) PP_SHAPE(TT,1,4,4,1) PP_SHAPE(TT,4,1,1,4) \
    /* 160 rows */ PP_SHAPE(TT,1,5,4,1) PP_SHAPE(TT,1,5,4,2)                                                            \
    /* 192 rows */ PP_SHAPE(TT,2,3,2,1) PP_SHAPE(TT,2,3,2,2)                                                            \
    /* 256 rows */ PP_SHAPE(TT,2,4,2,1) PP_SHAPE(TT,2,4,2,2)

template <typename T>
static int gemm_pp_dispatch(const GemmK& g, hipStream_t st, int mw, int mt, int ng, int nt) {
    PP_GO(T)
    return -1;
}
#undef PP_GO
#undef PP_SHAPE

}  // namespace lade
