// Kernel template of the skinny weight-streaming GEMM and its shape table.  Included by gemm_bf16.hip / gemm_f16.hip (one translation
// unit per dtype, so that the instantiations compile in parallel); gemm.hip holds the C entry points.
#pragma once
#include "common.hpp"
#include <type_traits>

#include "gemm_decl.hpp"

namespace lade {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// LDS ring depth: n_stage - 1 tiles stay in flight while one is multiplied.  It is a LAUNCH parameter (GemmK::n_stage, 2..G_NSTAGE_CAP):
// a deeper ring keeps more bytes in flight per work-group but costs LDS, i.e. co-resident work-groups - which of the two a projection
// needs depends on its split count, so the autotune picks the depth per (projection, row class).  Default (n_stage = 0 at the C ABI):
// 4 stages where they fit, else as many as fit (round 4, same box in alternation: 3 -> 4 stages took the 7B step from 3.91 / 3.86 to
// 3.78 / 3.77 ms; round 3 had measured -1.4 % on the isolated projections).
#ifndef LADE_G_NSTAGE_DEFAULT
#define LADE_G_NSTAGE_DEFAULT 4
#endif
// cache-policy bits of the weight stream (1 = sc0, 2 = nt, 16 = sc1): a COMPILE-time constant - variants are separate builds (tools/build_variant.sh NAME
// "-DLADE_W_AUX=18"): a run-time case in the DMA issue loop costs the step several per cent by itself
#ifndef LADE_W_AUX
#define LADE_W_AUX 2
#endif
#ifndef LADE_A_AUX
#define LADE_A_AUX 0          // the activation tile (and the weight stream when LADE_DEBUG=gemm_dbg=16 turns nt off)
#endif
constexpr int G_NSTAGE_CAP = 8;
constexpr int G_LDS_MAX = 160 * 1024;
constexpr int g_stages(int bn, int bm) {
    const int fit = G_LDS_MAX / ((bn + bm) * 128);
    // the compute-shaped 256 x 256 tile (64 KB per stage) runs as a double buffer; every other shape holds >= 3 stages
    return fit < LADE_G_NSTAGE_DEFAULT ? (fit < 2 ? 2 : fit) : LADE_G_NSTAGE_DEFAULT;
}
// counted wait: at most `younger` whole tiles (of PIECES pieces per wave) may stay in flight (vmcnt holds 6 bits: beyond 63 pieces the
// wait is stricter than needed, never weaker)
template <int PIECES, int MAXY>
__device__ __forceinline__ void g_wait_younger(int younger) {
    if constexpr (MAXY <= 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        if (younger >= MAXY) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAXY * PIECES < 63 ? MAXY * PIECES : 63) : "memory");
        else g_wait_younger<PIECES, MAXY - 1>(younger);
    }
}
constexpr int G_THREADS = 512;

__device__ __forceinline__ void g_barrier() { asm volatile("s_barrier" ::: "memory"); }
template <int N> __device__ __forceinline__ void g_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// 128-byte rows: two rows per 256-byte bank row, XOR the 16-byte chunk index with (row/2)&7
__device__ __forceinline__ int g_off(int row, int c16) { return row * 128 + ((c16 ^ ((row >> 1) & 7)) << 4); }

template <typename T> struct GMfma;
template <> struct GMfma<BF16> {
    __device__ static __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct GMfma<F16> {
    __device__ static __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};


// value after one rounding to the model dtype (the GEMM output, then every elementwise op, rounds like torch does)
template <typename T> __device__ __forceinline__ float g_rnd(float f) { return to_f32<T>(from_f32<T>(f)); }

// Epilogue shared by the kernels of this file: the C^T tile a work-group holds in its accumulators (wave (mw, ng) of an MW x NG grid, MT x NT MFMA
// tiles each; lane = activation row ql of an m-tile, 16 weight rows per tile) -> row-major C / SwiGLU / row argmax / fp32 split-K partials.
// Every wave of the work-group calls it (barriers inside); `computes` = this wave holds accumulators.
template <typename T, int MW, int MT, int NG, int NT>
__device__ __forceinline__ void gemm_epilogue(const GemmK& g, f32x16 (&acc)[MT][NT], unsigned char* smem, int mw, int ng, bool computes, int n0, int m0, int split) {
    constexpr int BN = 32 * NT * NG;
    constexpr int BM = 32 * MW * MT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int ql = lane & 31, hi = lane >> 5;
    // ---- epilogue: C^T tile (lane = activation row ql of block mb, 16 weight rows per MFMA tile) -> row-major C ----
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    g_barrier();
    if ((g.dbg & 1) && acc[0][0][0] != 12345.678f) return;
    if (g.n_split == 1 && g.epi == 1) {
        // SwiGLU in the epilogue (LlamaMLP, lade/models/modeling_llama.py:360-380: act_fn(gate_proj(x)) * up_proj(x)).  The fused
        // gate/up weight is interleaved in groups of 16 rows ([16 gate rows | their 16 up rows] per 32-row MFMA tile), so a lane's
        // accumulators e and e+8 are gate and up of the same output column: the product is lane-local.  Rounded as the separate ops
        // round: GEMM outputs to the dtype, silu to the dtype, product to the dtype.  out[m][16*tile + (e&3) + 8*(e>>2) + 4*hi].
        if (computes)
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            const int m = m0 + (mw * MT + a) * 32 + ql;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int nrow = n0 + (ng * NT + j) * 32;              // first fused row of this tile
                if (m < g.M && nrow < g.N) {
                    uint16_t* dst = g.C + (size_t)m * g.ldc + nrow / 2 + 4 * hi;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        float o[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float gt = g_rnd<T>(acc[a][j][4 * h + e]), up = g_rnd<T>(acc[a][j][8 + 4 * h + e]);
                            o[e] = g_rnd<T>(gt / (1.f + __expf(-gt))) * up;
                        }
                        u32x2 w;
                        w[0] = pack2<T>(o[0], o[1]);
                        w[1] = pack2<T>(o[2], o[3]);
                        *reinterpret_cast<u32x2*>(dst + 8 * h) = w;
                    }
                }
            }
        }
    } else if (g.n_split == 1 && g.epi == 2) {
        // Row argmax in the epilogue (greedy steps: `torch.argmax(logits)` of lade/decoding.py:1021 on the rows of
        // modeling_llama.py:1541-1544): the [rows, V] logits never reach HBM.  Per activation row this work-group's best (value, column)
        // over its BN columns, the value rounded to the model dtype first (the logits the reference compares are 16-bit), the LOWEST
        // column among equal values (torch.argmax; lade_argmax_rows) - written as a pair to Cpart[(m * gridDim.x + blockIdx.x) * 2],
        // merged across the column blocks by lade_argmax_pairs.  A lane holds the columns 8 g4 + 4 hi + e of tile j in ascending order.
        float* sv = reinterpret_cast<float*>(smem);
        int* si = reinterpret_cast<int*>(smem + NG * BM * 4);
        if (computes)
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            float best = -INFINITY;
            int bi = 0x7fffffff;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int n = n0 + (ng * NT + j) * 32 + 8 * g4 + 4 * hi + e;
                        const float v = g_rnd<T>(acc[a][j][4 * g4 + e]);
                        if (n < g.N && argmax_better(v, n, best, bi)) { best = v; bi = n; }
                    }
            const float ob = __shfl_xor(best, 32);
            const int oi = __shfl_xor(bi, 32);
            if (oi != 0x7fffffff && argmax_better(ob, oi, best, bi)) { best = ob; bi = oi; }
            if (hi == 0) {
                sv[ng * BM + (mw * MT + a) * 32 + ql] = best;
                si[ng * BM + (mw * MT + a) * 32 + ql] = bi;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        g_barrier();
        if (tid < BM && m0 + tid < g.M) {
            float best = sv[tid];
            int bi = si[tid];
#pragma unroll
            for (int q = 1; q < NG; ++q) {
                const float ob = sv[q * BM + tid];
                const int oi = si[q * BM + tid];
                if (oi != 0x7fffffff && argmax_better(ob, oi, best, bi)) { best = ob; bi = oi; }
            }
            float* dst = g.Cpart + ((size_t)(m0 + tid) * gridDim.x + blockIdx.x) * 2;
            *reinterpret_cast<float2*>(dst) = float2{best, __int_as_float(bi)};
        }
    } else if (g.n_split == 1) {
        // stage [BM][BN] in the model dtype, then whole-row 16-byte stores
        constexpr int RS = BN * 2 + 16;
        unsigned char* stg = smem;
        if (computes)
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                u32x2 w;
                w[0] = pack2<T>(acc[a][j][4 * g4 + 0], acc[a][j][4 * g4 + 1]);
                w[1] = pack2<T>(acc[a][j][4 * g4 + 2], acc[a][j][4 * g4 + 3]);
                *reinterpret_cast<u32x2*>(stg + ((mw * MT + a) * 32 + ql) * RS + ((ng * NT + j) * 32 + 8 * g4 + 4 * hi) * 2) = w;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        g_barrier();
        constexpr int CPR = BN * 2 / 16;
        for (int idx = tid; idx < BM * CPR; idx += G_THREADS) {
            const int row = idx / CPR, c = idx % CPR;
            if (m0 + row < g.M && n0 + c * 8 < g.N)
                *reinterpret_cast<u32x4*>(g.C + (size_t)(m0 + row) * g.ldc + n0 + c * 8) = *reinterpret_cast<const u32x4*>(stg + row * RS + c * 16);
        }
    } else if (g.dbg & 32) {
        // fp32 partials straight from the accumulators: shapes whose fp32 tile does not fit the LDS (the 256 x 256 tile: the launcher sets
        // the bit) and the experiment LADE_DEBUG=gemm_dbg=32 - lane (m = ql, hi) of tile (a, j) holds the four
        // consecutive weight rows n = 8*g4 + 4*hi .. +3 of activation row m, one 16-byte store each, no LDS staging and no barrier in
        // the tail.  Measured the same or slower than the staged whole-row stores below (7B, 60 rows: 92.9 vs 90.4-92.2 us per layer;
        // gate/up 40.2 vs 37.5-39.2): what the partial stores cost (16 us per layer, tools/gemm_flags.py with LADE_DEBUG=gemm_dbg=1) is
        // their 48.6 MB, not the staging round trip.
        float* outp = g.Cpart + (size_t)split * g.M * g.N;
        if (computes)
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            const int m = m0 + (mw * MT + a) * 32 + ql;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int nb = n0 + (ng * NT + j) * 32 + 4 * hi;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
                    if (m < g.M && nb + 8 * g4 < g.N)
                        *reinterpret_cast<float4*>(outp + (size_t)m * g.N + nb + 8 * g4) =
                            float4{acc[a][j][4 * g4 + 0], acc[a][j][4 * g4 + 1], acc[a][j][4 * g4 + 2], acc[a][j][4 * g4 + 3]};
            }
        }
    } else {
        // fp32 partials [split][M][N]; stage through LDS in two halves of BN to stay within the ring
        constexpr int RSF = BN * 4 + 16;
        float* outp = g.Cpart + (size_t)split * g.M * g.N;
        unsigned char* stg = smem;
        if (computes)
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                *reinterpret_cast<float4*>(stg + ((mw * MT + a) * 32 + ql) * RSF + ((ng * NT + j) * 32 + 8 * g4 + 4 * hi) * 4) =
                    float4{acc[a][j][4 * g4 + 0], acc[a][j][4 * g4 + 1], acc[a][j][4 * g4 + 2], acc[a][j][4 * g4 + 3]};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        g_barrier();
        constexpr int CPR = BN * 4 / 16;
        for (int idx = tid; idx < BM * CPR; idx += G_THREADS) {
            const int row = idx / CPR, c = idx % CPR;
            if (m0 + row < g.M && n0 + c * 4 < g.N) {
                const float4 v = *reinterpret_cast<const float4*>(stg + row * RSF + c * 16);
                float* dstp = outp + (size_t)(m0 + row) * g.N + n0 + c * 4;
                // plain stores.  Round 6 measured the alternatives IN THE STEP, alternating, two boxes (profiles/r6_partial_store_policy_ab*.txt): write-through
                // (sc1; LADE_DEBUG=gemm_dbg=8) c2 cold -0.8 % / -0.1 %, mid and hot regimes -0.6 ... -1.3 %, c4 0 ... +1 %: inside the noise taken together;
                // non-temporal (gemm_dbg=64) equal in the step, plain decoding +2 %.  (The switch sits in the epilogue, outside the K loop.)
                if (g.dbg & 8) {
                    const u32x4 vv = u32x4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dstp), "v"(vv) : "memory");
                }
                else if (g.dbg & 64)
                    __builtin_nontemporal_store(u32x4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}, reinterpret_cast<u32x4*>(dstp));
                else
                    *reinterpret_cast<float4*>(dstp) = v;
            }
        }
    }
}

// Work-group tile = BM activation rows x BN weight rows, BM = 32*MW*MT, BN = 32*NG*NT: the waves form an MW x NG grid
// (MW*NG <= 8 waves compute, all 8 issue DMA; m-group = w % MW, n-group = w / MW) and one wave owns MT x NT MFMA tiles
// of 32 x 32.  Per 16-deep K step a wave reads MT activation + NT weight fragments from LDS for MT*NT MFMAs, so the LDS
// bytes read per weight byte are 8 waves * (MT + NT) / (NG*NT) ... = (MT + NT) / NT * MW (+ the DMA write): with MT = 1 a
// 128-row step moves ~7 LDS bytes per weight byte and the 128 B/clk LDS port caps a CU at ~38 GB/s of weights; 2 x 2
// wave tiles bring that to ~5.5.
template <typename T, int MW, int MT, int NG, int NT>
__global__ __launch_bounds__(G_THREADS) void gemm_skinny_kernel(GemmK g) {
    static_assert(MW * NG <= 8, "at most 8 computing waves");
    constexpr int NW = G_THREADS / 64;          // all 8 waves move DMA pieces, the first MW * NG of them compute
    constexpr int MB = MW * MT;
    constexpr int BN = 32 * NT * NG;
    constexpr int BM = 32 * MB;
    // a stage = one 64-deep K tile: [weight tile | activation tile]
    constexpr int W_BYTES = BN * 128, A_BYTES = BM * 128, STAGE = W_BYTES + A_BYTES;
    constexpr int W_PIECES = W_BYTES / 1024, A_PIECES = A_BYTES / 1024;      // 1-KiB DMA pieces per tile
    constexpr int TOTAL_PIECES = W_PIECES + A_PIECES;
    constexpr int PIECES = (TOTAL_PIECES + NW - 1) / NW;                     // per wave and stage (the tail repeats the last piece)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mw = wave % MW, ng = wave / MW;
    const bool computes = ng < NG;
    const int ql = lane & 31, hi = lane >> 5;
    const int n0 = blockIdx.x * BN, split = blockIdx.y, m0 = blockIdx.z * BM;

    const int k_tiles = (g.K + G_BK - 1) / G_BK;
    const int tps = (k_tiles + g.n_split - 1) / g.n_split;
    const int t0 = split * tps;
    const int nt = max(0, min(t0 + tps, k_tiles) - t0);

    // ---- this wave's DMA pieces, resolved ONCE: per piece a lane's source pointer at the split's first K tile and the piece's offset
    // inside a stage.  Inside the K loop a piece then costs one 64-bit add, the M0 write
    // and the global_load_lds.  (Left inside the loop, the address arithmetic - two clamps, a 64-bit multiply, and the kernel
    // arguments re-read through the scalar cache after every asm barrier - cost a wave ~120 ns per piece: more than the transfer.)
    // weight addressing: row-major W[N][ldw] (a K tile of a row = 128 bytes, rows ldw apart) or K-tile-major Wkt[K/64][N][64] (g.w_ts =
    // 64 N: the 128-byte segments of ALL rows of one K tile are contiguous, so a 1-KiB DMA piece is one contiguous KiB, a work-group's
    // tile one contiguous BN x 128 bytes, and the work-groups of a split sweep memory linearly as they walk along K)
    const int64_t w_rs = g.w_ts ? G_BK : g.ldw, w_ts = g.w_ts ? g.w_ts : G_BK;
    const bool moves = wave * PIECES < TOTAL_PIECES;     // a wave moves PIECES pieces or none (vmcnt is per wave: nothing requested, nothing to wait for)
    const uint16_t* p_src[PIECES];
    int p_dst[PIECES];
    bool p_w[PIECES];
    const bool nt_weights = !(g.dbg & 16);
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int piece = min(wave * PIECES + i, TOTAL_PIECES - 1);
        const bool isw = piece < W_PIECES;
        const int p = isw ? piece : piece - W_PIECES;
        const int row = p * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        p_src[i] = (isw ? g.W + (size_t)min(n0 + row, g.N - 1) * w_rs + (size_t)t0 * w_ts
                        : g.A + (size_t)min(m0 + row, g.M - 1) * g.lda + (size_t)t0 * G_BK) + c * 8;
        if (!isw && (g.dbg & 128)) p_src[i] = g.A + (lane & 7) * 8;        // ablation (tools/gemm_ingest_probe.py): every activation piece re-reads one cached 128-byte line
        p_dst[i] = (isw ? 0 : W_BYTES) + p * 1024;
        p_w[i] = isw;
    }
    // tile j of this split into ring slot `stage`
    auto issue = [&](int j, int stage) {
        if (!moves) return;
        unsigned char* sbase = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const uint16_t* src = p_src[i] + (int64_t)j * (p_w[i] ? w_ts : ((g.dbg & 128) ? (int64_t)0 : (int64_t)G_BK));
            unsigned char* dst = sbase + p_dst[i];
            // cache policy of the weight stream (bits: 1 = sc0, 2 = nt, 16 = sc1): NON-TEMPORAL (round 2) - every weight byte is read by exactly one
            // work-group, once per step, so keeping it in L2 / the Infinity Cache only evicts what is re-read (activation tiles, partials): 97.2 -> 92.0 us
            // per 7B layer at 60 rows, decode step 4.65 -> 4.51 ms (LADE_DEBUG=gemm_dbg=16 turns it off).  nt + sc1 (round 6, -DLADE_W_AUX=18) wins 7-17 % on
            // ISOLATED back-to-back launches and is neutral to +1 % in the step (c2 3.90 / 3.90 vs 3.84 / 3.89 ms, c4 7.95 / 7.95 vs 7.95 / 7.84 with one
            // library per arm: profiles/r6_cache_policy_variants_ab.txt).  A first A/B through a RUN-TIME switch had read +7 %: the third case in this loop cost
            // the step ~5 % by itself and fell on one arm - the policy is a compile-time constant now.
            if (p_w[i] && nt_weights)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, LADE_W_AUX);
            else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, LADE_A_AUX);
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][i][e] = 0.f;

    // The ring depth is a launch parameter, but the loop is compiled once per depth (the counted waits need immediates, and a run-time
    // chain of compare-and-branch per K tile in front of every barrier cost the step 6 %: 4.29 / 4.33 vs 4.05 / 4.10 ms against the
    // compile-time ring on one box); the kernel switches to its copy once.
    auto main_loop = [&](auto ns_c) __attribute__((always_inline)) {
        constexpr int NS = decltype(ns_c)::value;
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (s < nt) issue(s, s);
        for (int i = 0; i < nt; ++i) {
            const int stage = i % NS;
            const int younger = min(nt, i + NS) - (i + 1);        // tiles requested after tile i that may stay in flight
            g_wait_younger<PIECES, NS - 1>(younger);
            g_barrier();
            const unsigned char* ws = smem + stage * STAGE;
            const unsigned char* as = ws + W_BYTES;
            if (computes && !(g.dbg & 4))
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
        case 5: main_loop(std::integral_constant<int, 5>{}); break;
        case 6: main_loop(std::integral_constant<int, 6>{}); break;
        case 8: main_loop(std::integral_constant<int, 8>{}); break;
        default: main_loop(std::integral_constant<int, 4>{}); break;
    }

    gemm_epilogue<T, MW, MT, NG, NT>(g, acc, smem, mw, ng, computes, n0, m0, split);
}

// sums the n_split fp32 partials in split order and writes the model dtype:  C[m][n] = sum_s part[s][m][n]
template <typename T, int MW, int MT, int NG, int NT>
static int launch_gemm(const GemmK& g0, hipStream_t st) {
    constexpr int BN = 32 * NT * NG, BM = 32 * MW * MT;
    constexpr size_t STAGE = (size_t)(BN + BM) * 128;
    GemmK g = g0;
    if (g.n_stage == 0) g.n_stage = g_stages(BN, BM);
    LADE_REQUIRE(g.n_stage >= 2 && g.n_stage <= G_NSTAGE_CAP && g.n_stage != 7 && g.n_stage * STAGE <= (size_t)G_LDS_MAX, LADE_E_ARG,
                 "lade_gemm_skinny: a ring of %d stages of %d + %d rows does not fit the %d KB of LDS", g.n_stage, BN, BM, G_LDS_MAX / 1024);
    // the epilogue stages the [BM][BN] tile through the LDS (model dtype without split-K, fp32 partials with it); an fp32 tile larger
    // than the LDS (the 256 x 256 shape) is stored straight from the accumulators instead
    size_t stg = g.n_split == 1 ? (g.epi == 1 ? 0 : (g.epi == 2 ? (size_t)NG * BM * 8 : (size_t)BM * (BN * 2 + 16))) : (size_t)BM * (BN * 4 + 16);
    if (g.n_split > 1 && stg > (size_t)G_LDS_MAX) { g.dbg |= 32; stg = 0; }
    const size_t ring = g.n_stage * STAGE, lds = ring > stg ? ring : stg;
    static_assert((size_t)BM * (BN * 2 + 16) <= (size_t)G_LDS_MAX, "model-dtype staging must fit the LDS");
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel<T, MW, MT, NG, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS_MAX);
        attr = true;
    }
    dim3 grid(cdiv(g.N, BN), g.n_split, cdiv(g.M, BM));
    hipLaunchKernelGGL((gemm_skinny_kernel<T, MW, MT, NG, NT>), grid, dim3(G_THREADS), lds, st, g);
    return check_launch("lade_gemm_skinny");
}

// the shapes that are built: (wave grid MW x NG, MT x NT MFMA tiles per wave)
#define SHAPE(TT, MWv, MTv, NGv, NTv) if (mw == MWv && mt == MTv && ng == NGv && nt == NTv) return launch_gemm<TT, MWv, MTv, NGv, NTv>(g, st);
// four groups: each is instantiated in a translation unit of its own per dtype (gemm_part.hip compiled with -DLADE_GEMM_PART=0..3), so that the
// ~70 shapes x 6 ring depths compile in parallel (one unit per dtype took 160 s and was the whole build)
#define GO_0(TT)                                                                                                             \
    /* 32 rows */  SHAPE(TT,1,1,1,1) SHAPE(TT,1,1,2,1) SHAPE(TT,1,1,4,1) SHAPE(TT,1,1,8,1) SHAPE(TT,1,1,2,2) SHAPE(TT,1,1,4,2) SHAPE(TT,1,1,3,1)  \
    /* 64 rows */  SHAPE(TT,2,1,1,1) SHAPE(TT,2,1,2,1) SHAPE(TT,2,1,4,1) SHAPE(TT,2,1,3,2) SHAPE(TT,2,1,4,2) SHAPE(TT,2,1,3,1) SHAPE(TT,1,2,3,1)  \
                   SHAPE(TT,1,2,2,1) SHAPE(TT,1,2,4,1) SHAPE(TT,1,2,6,1) SHAPE(TT,1,2,8,1) SHAPE(TT,1,2,2,2) SHAPE(TT,1,2,3,2)  \
                   SHAPE(TT,1,2,4,2) SHAPE(TT,1,2,2,3) SHAPE(TT,1,2,2,4)
#define GO_1(TT)                                                                                                             \
    /* 96 rows */  SHAPE(TT,3,1,1,1) SHAPE(TT,3,1,2,1) SHAPE(TT,3,1,2,2) SHAPE(TT,3,1,2,3) SHAPE(TT,3,1,2,4) SHAPE(TT,1,3,3,1)  \
                   SHAPE(TT,1,3,4,1) SHAPE(TT,1,3,6,1) SHAPE(TT,1,3,8,1) SHAPE(TT,1,3,3,2) SHAPE(TT,1,3,4,2)                    \
    /* 224-row weight blocks (N = 57344 = 256 x 224: Llama-2-70B gate/up on 256 CUs in one wave of work-groups) */        \
                   SHAPE(TT,1,1,7,1) SHAPE(TT,1,2,7,1) SHAPE(TT,1,3,7,1) SHAPE(TT,1,4,7,1)
#define GO_2(TT)                                                                                                             \
    /* 128 rows */ SHAPE(TT,4,1,1,1) SHAPE(TT,4,1,2,1) SHAPE(TT,4,1,2,2) SHAPE(TT,4,1,2,3) SHAPE(TT,4,1,2,4) SHAPE(TT,1,4,3,1) SHAPE(TT,2,2,3,1)  \
                   SHAPE(TT,2,2,2,1) SHAPE(TT,2,2,4,1) SHAPE(TT,2,2,3,2) SHAPE(TT,2,2,4,2) SHAPE(TT,2,2,2,2)                    \
                   SHAPE(TT,1,4,4,1) SHAPE(TT,1,4,6,1) SHAPE(TT,1,4,8,1) SHAPE(TT,1,4,3,2) SHAPE(TT,1,4,4,2) SHAPE(TT,1,4,2,2)
#define GO_3(TT)                                                                                                             \
    /* 160 rows (round 6: a 129..160-row step no longer pads to 192): ONE m-group of five m-blocks per wave, the waves along N */            \
                   SHAPE(TT,1,5,8,1) SHAPE(TT,1,5,4,2) SHAPE(TT,1,5,4,1) SHAPE(TT,1,5,2,2) SHAPE(TT,1,5,3,1) SHAPE(TT,1,5,6,1) SHAPE(TT,1,5,2,1)                    \
    /* 192 rows */ SHAPE(TT,2,3,2,1) SHAPE(TT,2,3,4,1) SHAPE(TT,2,3,2,2) SHAPE(TT,3,2,2,1) SHAPE(TT,3,2,2,2)                                     \
    /* 256 rows */ SHAPE(TT,2,4,2,1) SHAPE(TT,2,4,4,1) SHAPE(TT,2,4,2,2) SHAPE(TT,4,2,2,1) SHAPE(TT,4,2,2,2)                                     \
    /* 256 x 256 compute-shaped tile (steps and prefill chunks wider than 256 rows run as several row blocks): 4 x 2 / 2 x 4 MFMA tiles  \
       per wave, 0.75 LDS fragment reads per MFMA, double-buffered 64 KB stages */                                        \
                   SHAPE(TT,2,4,4,2) SHAPE(TT,4,2,2,4)

// -1: no kernel of this group for the wave grid
template <typename T, int PART>
static int gemm_dispatch_part(const GemmK& g, hipStream_t st, int mw, int mt, int ng, int nt) {
    if constexpr (PART == 0) { GO_0(T) }
    if constexpr (PART == 1) { GO_1(T) }
    if constexpr (PART == 2) { GO_2(T) }
    if constexpr (PART == 3) { GO_3(T) }
    return -1;
}
#undef GO_0
#undef GO_1
#undef GO_2
#undef GO_3
#undef SHAPE

}  // namespace lade
