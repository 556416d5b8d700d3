// Shared between the C entry points (gemm.hip) and the per-dtype translation units of the skinny GEMM (gemm_bf16.hip, gemm_f16.hip).
#pragma once
#include "common.hpp"

namespace lade {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int G_BK = 64;          // K depth of one LDS tile (128-byte rows)

struct GemmK {
    const uint16_t* A;     // [M][lda]
    const uint16_t* W;     // [N][ldw], or [K/64][N][64] (w_ts != 0)
    uint16_t* C;           // [M][ldc] model dtype (n_split == 1)
    float* Cpart;          // [n_split][M][N] fp32 (n_split > 1)
    int64_t lda, ldw, ldc;
    int64_t w_ts;          // 0: W is row-major [N][ldw];  64 N: W is K-tile-major [K/64][N][64] (elements between the K tiles of a row)
    int M, N, K, n_split;
    int n_stage;           // LDS ring depth of this launch (0: the shape's default)
    int dbg;               // ablation switches for tools/gemm_ablate.py (LADE_DEBUG=gemm_dbg=<bits>): 1 = no output stores, 4 = no LDS reads / MFMA
    int epi;               // n_split == 1 only: 0 = C = A.W^T;  1 = SwiGLU over interleaved gate / up rows, C is [M][N/2];
                           // 2 = row argmax: no C, Cpart holds one (value, column) pair per row and column block
};

// register-resident-activation kernel (gemm_ra.hpp)
struct GemmRA {
    const uint16_t* A;     // [M][lda]
    const uint16_t* W;     // K-tile-major [K/64][N][64]
    float* Cpart;          // [n_split][M][N] fp32
    int64_t lda;
    int M, N, K;
    int n_split, tps;      // K tiles per split (<= KT; the last split may hold fewer)
    int n_groups;          // column groups per split; grid = n_groups * n_split work-groups
    int dbg;               // 1 = no stores, 4 = no fragment reads / MFMA (tools)
};

constexpr int RA_KT = 20;          // K tiles (64 deep) a work-group of the RA kernel holds in registers: K slices of <= 1280
int gemm_ra_dispatch_bf16(const GemmRA& g, hipStream_t st, int mw, int cs);
int gemm_ra_dispatch_f16(const GemmRA& g, hipStream_t st, int mw, int cs);

// launch the kernel built for this wave grid (mw x ng waves compute, mt x nt MFMA tiles each); -1 when it is not in the shape table
int gemm_dispatch_bf16(const GemmK& g, hipStream_t st, int mw, int mt, int ng, int nt);
int gemm_dispatch_f16(const GemmK& g, hipStream_t st, int mw, int mt, int ng, int nt);
// the 16-row-granular form (gemm16.hpp): mt16 16-row activation blocks in every wave, ng waves along N with nt16 16-row weight blocks each
int gemm16_dispatch_bf16(const GemmK& g, hipStream_t st, int mt16, int ng, int nt16);
int gemm16_dispatch_f16(const GemmK& g, hipStream_t st, int mt16, int ng, int nt16);
// the ping-pong form (gemm_pp.hpp): mw x ng = the grid of ONE group of four waves
int gemm_pp_dispatch_bf16(const GemmK& g, hipStream_t st, int mw, int mt, int ng, int nt);
int gemm_pp_dispatch_f16(const GemmK& g, hipStream_t st, int mw, int mt, int ng, int nt);

}  // namespace lade
