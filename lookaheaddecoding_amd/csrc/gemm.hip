// Skinny weight-streaming GEMM for the decode step:  C[M,N] = A[M,K] . W[N,K]^T,  M <= 128 per row block.
//
// Reference: the nn.Linear projections of the step (q/k/v/o, gate/up/down, lm_head;
// lade/models/modeling_llama.py:360-380, 492-494, 558, 1541) - SURVEY.md 8(f) rank 2.  With M = T <= 240 rows
// the GEMM is a stream of the weight matrix: bound by HBM (and, per CU, by the ~55-68 GB/s one CU can ingest from HBM), so
// the weight bytes must be spread over all 256 CUs.  Work-group = (BN weight rows) x (one K slice) x (one row
// block); the weight tile and the activation tile of each 64-deep K step arrive by LDS-DMA into a 3-stage ring,
// eight waves hold the C^T tile in MFMA accumulators (lane = one activation row, like the attention kernel),
// split-K partials are fp32 and are summed in a fixed order (deterministic) by lade_splitk_reduce or by the
// consumer kernel.
#include "common.hpp"
#include <cstdlib>

namespace lade {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int G_BK = 64;          // K depth of one LDS tile (128-byte rows)
constexpr int G_NSTAGE = 3;
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

struct GemmK {
    const uint16_t* A;     // [M][lda]
    const uint16_t* W;     // [N][ldw]
    uint16_t* C;           // [M][ldc] model dtype (n_split == 1)
    float* Cpart;          // [n_split][M][N] fp32 (n_split > 1)
    int64_t lda, ldw, ldc;
    int M, N, K, n_split;
    int dbg;               // ablation switches for tools/gemm_ablate.py (LADE_GEMM_DBG): 1 = no output stores, 4 = no LDS reads / MFMA
};

// MB = 32-row activation blocks per work-group (2 or 4); NG = weight-row groups (MB*NG <= 8 waves compute, all 8
// waves issue DMA); NT = 32-row weight tiles per wave.  waves: m-block = w % MB, n-group = w / MB;
// BN = 32 * NT * NG weight rows per work-group.
template <typename T, int MB, int NG, int NT>
__global__ __launch_bounds__(G_THREADS) void gemm_skinny_kernel(GemmK g) {
    static_assert(MB * NG <= 8, "at most 8 computing waves");
    constexpr int BN = 32 * NT * NG;
    constexpr int BM = 32 * MB;
    constexpr int W_BYTES = BN * 128, A_BYTES = BM * 128, STAGE = W_BYTES + A_BYTES;
    constexpr int W_PIECES = W_BYTES / 1024, A_PIECES = A_BYTES / 1024;      // per work-group
    constexpr int TOTAL_PIECES = W_PIECES + A_PIECES;
    constexpr int PIECES = (TOTAL_PIECES + 7) / 8;                           // per wave (the tail repeats the last piece)

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mb = wave % MB, ng = wave / MB;
    const bool computes = ng < NG;
    const int ql = lane & 31, hi = lane >> 5;
    const int n0 = blockIdx.x * BN, split = blockIdx.y, m0 = blockIdx.z * BM;

    const int k_tiles = (g.K + G_BK - 1) / G_BK;
    const int tps = (k_tiles + g.n_split - 1) / g.n_split;
    const int t0 = split * tps;
    const int nt = max(0, min(t0 + tps, k_tiles) - t0);

    auto issue = [&](int tile, int stage) {
        const int k0 = tile * G_BK;
        unsigned char* ws = smem + stage * STAGE;
        unsigned char* as = ws + W_BYTES;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int piece = min(wave * PIECES + i, TOTAL_PIECES - 1);
            const bool isw = piece < W_PIECES;
            const int p = isw ? piece : piece - W_PIECES;
            const int row = p * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            const uint16_t* src;
            if (isw) src = g.W + (size_t)min(n0 + row, g.N - 1) * g.ldw + k0 + c * 8;
            else src = g.A + (size_t)min(m0 + row, g.M - 1) * g.lda + k0 + c * 8;
            unsigned char* dst = (isw ? ws : as) + p * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
#pragma unroll
    for (int s = 0; s < G_NSTAGE; ++s)
        if (s < nt) issue(t0 + s, s);

    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    for (int i = 0; i < nt; ++i) {
        const int stage = i % G_NSTAGE;
        const unsigned char* ws = smem + stage * STAGE;
        const unsigned char* as = ws + W_BYTES;
        const int younger = min(nt, i + G_NSTAGE) - (i + 1);
        if (younger >= 2) g_wait_vm<2 * PIECES>();
        else if (younger == 1) g_wait_vm<PIECES>();
        else g_wait_vm<0>();
        g_barrier();
        if (computes && !(g.dbg & 4))
#pragma unroll
        for (int kk = 0; kk < G_BK / 16; ++kk) {
            const u32x4 af = *reinterpret_cast<const u32x4*>(as + g_off(mb * 32 + ql, kk * 2 + hi));
            u32x4 wf[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const u32x4*>(ws + g_off((ng * NT + j) * 32 + ql, kk * 2 + hi));
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j] = GMfma<T>::run(wf[j], af, acc[j]);
        }
        if (i + G_NSTAGE < nt) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            g_barrier();
            issue(t0 + i + G_NSTAGE, stage);
        }
    }

    // ---- epilogue: C^T tile (lane = activation row ql of block mb, 16 weight rows per MFMA tile) -> row-major C ----
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    g_barrier();
    const int m = m0 + mb * 32 + ql;
    if ((g.dbg & 1) && acc[0][0] != 12345.678f) return;
    if (g.n_split == 1) {
        // stage [BM][BN] in the model dtype, then whole-row 16-byte stores
        constexpr int RS = BN * 2 + 16;
        unsigned char* stg = smem;
        if (computes)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                u32x2 w;
                w[0] = pack2<T>(acc[j][4 * g4 + 0], acc[j][4 * g4 + 1]);
                w[1] = pack2<T>(acc[j][4 * g4 + 2], acc[j][4 * g4 + 3]);
                *reinterpret_cast<u32x2*>(stg + (mb * 32 + ql) * RS + ((ng * NT + j) * 32 + 8 * g4 + 4 * hi) * 2) = w;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        g_barrier();
        constexpr int CPR = BN * 2 / 16;
        for (int idx = tid; idx < BM * CPR; idx += G_THREADS) {
            const int row = idx / CPR, c = idx % CPR;
            if (m0 + row < g.M && n0 + c * 8 < g.N)
                *reinterpret_cast<u32x4*>(g.C + (size_t)(m0 + row) * g.ldc + n0 + c * 8) = *reinterpret_cast<const u32x4*>(stg + row * RS + c * 16);
        }
    } else {
        // fp32 partials [split][M][N]; stage through LDS in two halves of BN to stay within the ring
        constexpr int RSF = BN * 4 + 16;
        float* outp = g.Cpart + (size_t)split * g.M * g.N;
        unsigned char* stg = smem;
        static_assert((size_t)BM * RSF <= (size_t)G_NSTAGE * STAGE, "fp32 staging must fit in the ring");
        if (computes)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                *reinterpret_cast<float4*>(stg + (mb * 32 + ql) * RSF + ((ng * NT + j) * 32 + 8 * g4 + 4 * hi) * 4) =
                    float4{acc[j][4 * g4 + 0], acc[j][4 * g4 + 1], acc[j][4 * g4 + 2], acc[j][4 * g4 + 3]};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        g_barrier();
        constexpr int CPR = BN * 4 / 16;
        for (int idx = tid; idx < BM * CPR; idx += G_THREADS) {
            const int row = idx / CPR, c = idx % CPR;
            if (m0 + row < g.M && n0 + c * 4 < g.N)
                *reinterpret_cast<float4*>(outp + (size_t)(m0 + row) * g.N + n0 + c * 4) = *reinterpret_cast<const float4*>(stg + row * RSF + c * 16);
        }
    }
    (void)m;
}

// sums the n_split fp32 partials in split order and writes the model dtype:  C[m][n] = sum_s part[s][m][n]
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* part, uint16_t* C, int64_t ldc, int M, int N, int n_split) {
    const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int m = blockIdx.y;
    if (i >= N) return;
    float4 a = *reinterpret_cast<const float4*>(part + (size_t)m * N + i);
    for (int s = 1; s < n_split; ++s) {
        const float4 b = *reinterpret_cast<const float4*>(part + ((size_t)s * M + m) * N + i);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    u32x2 w;
    w[0] = pack2<T>(a.x, a.y);
    w[1] = pack2<T>(a.z, a.w);
    *reinterpret_cast<u32x2*>(C + (size_t)m * ldc + i) = w;
}

template <typename T, int MB, int NG, int NT>
static int launch_gemm(const GemmK& g, hipStream_t st) {
    constexpr int BN = 32 * NT * NG, BM = 32 * MB;
    constexpr size_t lds = (size_t)G_NSTAGE * (BN + BM) * 128;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel<T, MB, NG, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    dim3 grid(cdiv(g.N, BN), g.n_split, cdiv(g.M, BM));
    hipLaunchKernelGGL((gemm_skinny_kernel<T, MB, NG, NT>), grid, dim3(G_THREADS), lds, st, g);
    return check_launch("lade_gemm_skinny");
}



}  // namespace lade

using namespace lade;

// bn: weight rows per work-group (64/128/192/256 with 96- or 128-row blocks; 32..256 with 64-row blocks); mb: 32-row
// activation blocks per work-group (1 | 2 | 3 | 4; 0 = by M)
extern "C" int lade_gemm_skinny(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, float* Cpart,
                                int32_t M, int32_t N, int32_t K, int32_t n_split, int32_t bn, int32_t mb, int32_t dtype, void* stream) {
    LADE_REQUIRE(A && W && M > 0 && N > 0 && K > 0 && n_split >= 1, LADE_E_ARG, "lade_gemm_skinny: M=%d N=%d K=%d split=%d", M, N, K, n_split);
    LADE_REQUIRE(K % G_BK == 0 && lda % 8 == 0 && ldw % 8 == 0 && N % 8 == 0, LADE_E_ARG,
                 "lade_gemm_skinny: K=%d must be a multiple of %d, strides / N multiples of 8", K, G_BK);
    LADE_REQUIRE(n_split == 1 ? (C != nullptr && ldc % 8 == 0) : (Cpart != nullptr), LADE_E_ARG, "lade_gemm_skinny: missing output buffer");
    LADE_REQUIRE(dtype == LADE_BF16 || dtype == LADE_F16, LADE_E_DTYPE, "lade_gemm_skinny: dtype=%d", dtype);
    GemmK g;
    g.A = (const uint16_t*)A; g.W = (const uint16_t*)W; g.C = (uint16_t*)C; g.Cpart = Cpart;
    g.lda = lda; g.ldw = ldw; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.n_split = n_split;
    { static int dbg = -1; if (dbg < 0) { const char* e = getenv("LADE_GEMM_DBG"); dbg = e ? atoi(e) : 0; } g.dbg = dbg; }
    hipStream_t st = (hipStream_t)stream;
    const bool tiny_m = mb == 1 || (mb == 0 && M <= 32);
    const bool small_m = mb == 2 || (mb == 0 && M <= 64);
    const bool mid_m = mb == 3 || (mb == 0 && M > 64 && M <= 96);
#define GO(TT)                                                                        \
    if (tiny_m) {                          /* one 32-row block: lookahead-parallel ranks, config 1 (T <= 16) */ \
        if (bn <= 64) return launch_gemm<TT, 1, 2, 1>(g, st);                         \
        if (bn <= 128) return launch_gemm<TT, 1, 4, 1>(g, st);                        \
        return launch_gemm<TT, 1, 8, 1>(g, st);                                       \
    }                                                                                 \
    if (mid_m) {                           /* 96-row blocks: 6 computing waves */     \
        if (bn <= 64) return launch_gemm<TT, 3, 2, 1>(g, st);                         \
        if (bn <= 128) return launch_gemm<TT, 3, 2, 2>(g, st);                        \
        if (bn <= 192) return launch_gemm<TT, 3, 2, 3>(g, st);                        \
        return launch_gemm<TT, 3, 2, 4>(g, st);                                       \
    }                                                                                 \
    if (small_m) {                                                                    \
        if (bn <= 32) return launch_gemm<TT, 2, 1, 1>(g, st);                         \
        if (bn <= 64) return launch_gemm<TT, 2, 2, 1>(g, st);                         \
        if (bn <= 128) return launch_gemm<TT, 2, 4, 1>(g, st);                        \
        return launch_gemm<TT, 2, 4, 2>(g, st);                                       \
    } else {                                                                          \
        if (bn <= 32) return launch_gemm<TT, 4, 1, 1>(g, st);                         \
        if (bn <= 64) return launch_gemm<TT, 4, 2, 1>(g, st);                         \
        if (bn <= 128) return launch_gemm<TT, 4, 2, 2>(g, st);                        \
        if (bn <= 192) return launch_gemm<TT, 4, 2, 3>(g, st);                        \
        return launch_gemm<TT, 4, 2, 4>(g, st);                                       \
    }
    if (dtype == LADE_BF16) { GO(BF16) } else { GO(F16) }
#undef GO
}


extern "C" int lade_splitk_reduce(const float* part, void* C, int64_t ldc, int32_t M, int32_t N, int32_t n_split, int32_t dtype, void* stream) {
    LADE_REQUIRE(part && C && M > 0 && N > 0 && N % 4 == 0 && n_split >= 1, LADE_E_ARG, "lade_splitk_reduce: bad args");
    dim3 grid(cdiv(N / 4, 256), M);
    if (dtype == LADE_BF16) hipLaunchKernelGGL(splitk_reduce_kernel<BF16>, grid, dim3(256), 0, (hipStream_t)stream, part, (uint16_t*)C, ldc, M, N, n_split);
    else if (dtype == LADE_F16) hipLaunchKernelGGL(splitk_reduce_kernel<F16>, grid, dim3(256), 0, (hipStream_t)stream, part, (uint16_t*)C, ldc, M, N, n_split);
    else LADE_REQUIRE(false, LADE_E_DTYPE, "lade_splitk_reduce: dtype=%d", dtype);
    return check_launch("lade_splitk_reduce");
}
