// Split-K weight-streaming GEMM with REGISTER-RESIDENT activations ("RA" kernel) for steps of <= 128 rows.
//
// The LDS-ring kernel of gemm_kernel.hpp brings an activation tile into the CU beside every weight tile: at 120 rows half of what a
// CU ingests (and half of its LDS ring) is activations, re-fetched by every work-group for every 64-deep K tile (DESIGN 4.6:
// the 128-row class).  Here a work-group is PERSISTENT over the weight rows of its column group: it owns one K slice of at most KT
// tiles and keeps the activation fragments of that slice in REGISTERS for its whole life - 4 waves, one per SIMD, up to 512 VGPRs
// + AGPRs each: wave w holds the MFMA operand of activation rows 32 (w % MW) .. + 31 for every 16-deep k-step of the slice
// (4 registers per k-step: 320 at KT = 20) - so that
//   * the activations pass through the CU ONCE per work-group (while the first chunk of weight rows streams, in the ring like the
//     old kernel's tile) instead of once per 128 weight rows,
//   * all 160 KB of LDS are a ring of 4-KB weight units (one 32-row strip x one K tile) from the second chunk on, and
//   * a weight fragment read from LDS meets its activation operand in a register (half the ds_reads of a 2 x 2 wave tile).
// The split-K partials leave straight from the accumulators at the end of every chunk of CS strips, while the next chunk's weights
// are already in flight (the old kernel stores them in a tail after its K loop, all work-groups of the launch at once).
//
// Arithmetic: the same MFMA sequence per (row, column, split) as gemm_skinny_kernel on the same K slices - bit-identical partials
// (tests/test_gpu_ktile.py).  Weights K-tile-major only (lade_weight_to_ktile).
#pragma once
#include "gemm_kernel.hpp"

namespace lade {

constexpr int RA_THREADS = 256;
constexpr int RA_UNIT = 4096;                     // 32 weight rows x 64 k x 2 bytes
constexpr int RA_RING = G_LDS_MAX / RA_UNIT;      // 40 units: all of the LDS

template <int N> __device__ __forceinline__ void ra_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// s_waitcnt vmcnt(<= y) with an immediate (vmcnt holds 6 bits).  FAST = the value of a steady iteration: one compare on that path.
template <int FAST>
__device__ __forceinline__ void ra_wait_le(int y) {
    if (y >= FAST) { ra_wait<FAST>(); return; }
    if (y >= 16) {
        if (y >= 24) { if (y >= 28) ra_wait<28>(); else ra_wait<24>(); } else { if (y >= 20) ra_wait<20>(); else ra_wait<16>(); }
    } else if (y >= 8) {
        if (y >= 12) ra_wait<12>(); else ra_wait<8>();
    } else if (y >= 4) {
        ra_wait<4>();
    } else if (y >= 2) {
        ra_wait<2>();
    } else {
        ra_wait<0>();
    }
}

// Work-group = MW m-tiles (32 activation rows each) x NG strip groups, MW * NG <= 4 computing waves (all 4 waves move DMA pieces).
// A chunk = CS weight strips (32 rows each) over the whole K slice; wave (mw, ng) multiplies the strips ng * NTW .. + NTW - 1 of it.
// The N / (32 CS) chunks of a split are dealt out to the n_groups work-groups of the split as evenly as whole chunks allow.
template <typename T, int MW, int NG, int CS, int KT>
__global__ __launch_bounds__(RA_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_ra_kernel(GemmRA g) {
    static_assert(MW * NG <= 4 && CS % NG == 0 && CS <= 4, "wave grid");
    constexpr int NTW = CS / NG;                 // strips per wave and chunk
    constexpr int R = RA_RING;
    constexpr int ASLOTS = 4;                    // ring slots a first-chunk tile reserves for the activation units (one per wave)
    constexpr int TS0 = CS + ASLOTS, TS1 = CS;   // slots per tile: first chunk / later chunks
    constexpr int CHUNK_BYTES = CS * 32 * G_BK * 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mw = wave % MW, ng = wave / MW;
    const bool computes = ng < NG;
    const int ql = lane & 31, hi = lane >> 5;
    const int split = blockIdx.x % g.n_split, grp = blockIdx.x / g.n_split;
    const int k_tiles = g.K / G_BK;
    const int t0 = split * g.tps;
    const int nt = min(g.tps, k_tiles - t0);                                  // K tiles of this slice (host: >= 1)
    const int nc_all = g.N / (32 * CS);
    const int c_lo = (int)((int64_t)grp * nc_all / g.n_groups), c_hi = (int)((int64_t)(grp + 1) * nc_all / g.n_groups);
    const int n_chunks = c_hi - c_lo;
    if (n_chunks <= 0) return;

    // ---- per-lane constants of the DMA pieces (a piece = 8 rows x 128 bytes = 1 KiB, lane -> 16 bytes; XOR chunk swizzle on the source) ----
    const int prow = wave * 8 + (lane >> 3);                                  // row of this lane inside a 32-row unit, W pieces (piece = wave)
    const uint32_t w_lane = (uint32_t)(prow * G_BK + (((lane & 7) ^ ((prow >> 1) & 7)) << 3)) * 2u;     // byte offset inside the unit's 4 KB source
    // activation unit of this wave: its own 32 rows x one K tile, 4 pieces; rows past M repeat row M - 1
    uint32_t a_lane[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = p * 8 + (lane >> 3);
        const int m = min(mw * 32 + r, g.M - 1);
        a_lane[p] = (uint32_t)(m * (int)g.lda + (((lane & 7) ^ ((r >> 1) & 7)) << 3)) * 2u;
    }
    const size_t w_tile_stride = (size_t)g.N * G_BK * 2;                      // bytes between two K tiles of a strip
    // fragment read offsets inside a unit: row ql, 16-byte chunk (2 kk + hi) ^ swizzle
    int f_off[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) f_off[kk] = g_off(ql, kk * 2 + hi);

    // ---- issue side (wave-uniform): the next tile to request is K tile ii of chunk ic ----
    int ic = 0, ii = 0;
    int i_head = 0;                                // ring slot of its first unit
    int used = 0;                                  // ring slots held by tiles that are requested and not yet left by every wave
    int iss_p = 0;                                 // VMEM pieces this wave has requested so far
    const unsigned char* a_src = reinterpret_cast<const unsigned char*>(g.A) + (size_t)t0 * G_BK * 2;      // activation K tile ii (first chunk)
    const unsigned char* w_chunk = reinterpret_cast<const unsigned char*>(g.W) + ((size_t)t0 * g.N + (size_t)c_lo * CS * 32) * G_BK * 2;
    const unsigned char* w_src = w_chunk;          // weight K tile ii of chunk ic
    auto issue_tile = [&]() {
        int slot = i_head;
        if (ic == 0) {
            if (computes) {
                int as = slot + wave;
                if (as >= R) as -= R;
                unsigned char* dst = smem + as * RA_UNIT;
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src + a_lane[p]),
                                                     (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
                iss_p += 4;
            }
            a_src += G_BK * 2;
            slot += ASLOTS;
            if (slot >= R) slot -= R;
            used += ASLOTS;
        }
        const unsigned char* src = w_src + w_lane;
#pragma unroll
        for (int j = 0; j < CS; ++j) {
            // the weight stream is non-temporal (aux = 2): every byte is read by one work-group, once per step
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)j * 32 * G_BK * 2),
                                             (__attribute__((address_space(3))) void*)(smem + slot * RA_UNIT + wave * 1024), 16, 0, 2);
            ++slot;
            if (slot >= R) slot -= R;
        }
        iss_p += CS;
        used += CS;
        i_head = slot;
        w_src += w_tile_stride;
        if (++ii == nt) { ii = 0; ++ic; w_chunk += CHUNK_BYTES; w_src = w_chunk; }
    };
    auto fits = [&]() { return ic < n_chunks && used + (ic == 0 ? TS0 : TS1) <= R; };

    u32x4 af[KT * 4];
    f32x16 acc[NTW];

    // prologue: fill the ring
    while (fits()) issue_tile();

    int c_head = 0;                    // ring slot of the current tile's first unit
    int prev_slots = 0;                // slots of the tile consumed in the previous iteration (free after this iteration's barrier)
    int need = 0;                      // pieces this wave has requested through the current tile
    // the partial stores of the previous chunk sit in the same in-order queue: st_n of them, issued when iss_p stood at st_mark
    int st_mark = -1, st_n = 0;
    float* outp = g.Cpart + (size_t)split * g.M * g.N + (size_t)min(mw * 32 + ql, g.M - 1) * g.N + (size_t)(c_lo * CS + ng * NTW) * 32 + 4 * hi;
    for (int c = 0; c < n_chunks; ++c) {
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        const bool first = c == 0;
        const int tp = CS + ((first && computes) ? 4 : 0), ts = first ? TS0 : TS1;
#pragma unroll
        for (int i = 0; i < KT; ++i) {
            if (i < nt) {
                need += tp;
                // what this wave requested after the current tile may stay in flight: later tiles' pieces, and the previous chunk's partial
                // stores where they were issued after this tile's pieces (the queue retires in order)
                int y = iss_p - need;
                if (st_n > 0) {
                    if (need <= st_mark) y += st_n; else st_n = 0;
                }
                ra_wait_le<R - 2 * CS>(y);
                g_barrier();
                used -= prev_slots;
                prev_slots = ts;
                int slot = c_head;
                if (first) {
                    if (computes) {
                        int as = slot + wave;
                        if (as >= R) as -= R;
                        const unsigned char* ab = smem + as * RA_UNIT;
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) af[i * 4 + kk] = *reinterpret_cast<const u32x4*>(ab + f_off[kk]);
                    }
                    slot += ASLOTS;
                    if (slot >= R) slot -= R;
                }
                // fragment reads of this wave's strips, then the requests that refill the slots the previous tile left (the DMA instructions
                // issue while the reads return), then the MFMAs
                u32x4 wf[NTW][4];
                const bool mul = computes && !(g.dbg & 4);
                if (mul) {
#pragma unroll
                    for (int jj = 0; jj < NTW; ++jj) {
                        int sl = slot + ng * NTW + jj;
                        if (sl >= R) sl -= R;
                        const unsigned char* wb = smem + sl * RA_UNIT;
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) wf[jj][kk] = *reinterpret_cast<const u32x4*>(wb + f_off[kk]);
                    }
                }
                if (fits()) issue_tile();
                if (fits()) issue_tile();          // (behind the first chunk the tiles are smaller: the ring deepens by one tile per iteration)
                if (mul) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int jj = 0; jj < NTW; ++jj) acc[jj] = GMfma<T>::run(wf[jj][kk], af[i * 4 + kk], acc[jj]);
                }
                c_head += ts;
                if (c_head >= R) c_head -= R;
            }
        }
        // ---- partials of this chunk, straight from the accumulators: lane (ql, hi) of a strip holds the weight rows 8 g4 + 4 hi .. + 3 of
        // activation row m (rows past M were computed from row M - 1's operands: the same bits go to the same address) ----
        if (computes && !(g.dbg & 1)) {
#pragma unroll
            for (int jj = 0; jj < NTW; ++jj)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
                    *reinterpret_cast<float4*>(outp + jj * 32 + 8 * g4) = float4{acc[jj][4 * g4 + 0], acc[jj][4 * g4 + 1], acc[jj][4 * g4 + 2], acc[jj][4 * g4 + 3]};
            // in the queue behind every piece requested so far
            if (st_n > 0 && need <= st_mark) st_n += 4 * NTW;        // an older batch is still younger than the current tile: merge (never over-allows)
            else { st_mark = iss_p; st_n = 4 * NTW; }
        }
        outp += CS * 32;
    }
}

template <typename T, int MW, int NG, int CS, int KT>
static int launch_gemm_ra(const GemmRA& g, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_ra_kernel<T, MW, NG, CS, KT>, hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS_MAX);
        attr = true;
    }
    hipLaunchKernelGGL((gemm_ra_kernel<T, MW, NG, CS, KT>), dim3(g.n_groups * g.n_split), dim3(RA_THREADS), (size_t)RA_RING * RA_UNIT, st, g);
    return check_launch("lade_gemm_ra");
}

// -1: no kernel for this shape
template <typename T>
static int gemm_ra_dispatch(const GemmRA& g, hipStream_t st, int mw, int cs) {
#define RA_SHAPE(MWv, NGv, CSv) if (mw == MWv && cs == CSv) return launch_gemm_ra<T, MWv, NGv, CSv, RA_KT>(g, st);
    RA_SHAPE(4, 1, 4) RA_SHAPE(4, 1, 2) RA_SHAPE(3, 1, 4) RA_SHAPE(3, 1, 2) RA_SHAPE(2, 2, 4) RA_SHAPE(2, 2, 2) RA_SHAPE(1, 4, 4)
#undef RA_SHAPE
    return -1;
}

}  // namespace lade
