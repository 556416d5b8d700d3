// Skinny weight-streaming GEMM for the decode step:  C[M,N] = A[M,K] . W[N,K]^T,  M <= 256 per row block (a work-group holds every
// activation row of the step, so the weights are streamed once whatever M is).
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

#include "gemm_decl.hpp"

namespace lade {

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

// output chunk c (16 bytes) of the K-tile-major weight: c = (kt * N + n) * 8 + j  <-  W[n][64 kt + 8 j ..]
__global__ __launch_bounds__(256) void weight_to_ktile_kernel(const uint16_t* W, int64_t ldw, uint16_t* Wkt, int N, int K) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= (int64_t)N * (K / 8)) return;
    const int j = (int)(c & 7);
    const int64_t row = c >> 3;                     // kt * N + n
    const int kt = (int)(row / N), n = (int)(row - (int64_t)kt * N);
    *reinterpret_cast<u32x4*>(Wkt + c * 8) = *reinterpret_cast<const u32x4*>(W + (size_t)n * ldw + (size_t)kt * G_BK + j * 8);
}

// the inverse: row-major chunk c = n * (K/8) + kc  <-  Wkt[(kc/8) * N + n][8 (kc % 8) ..]
__global__ __launch_bounds__(256) void weight_from_ktile_kernel(const uint16_t* Wkt, uint16_t* W, int64_t ldw, int N, int K) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int cpr = K / 8;
    if (c >= (int64_t)N * cpr) return;
    const int n = (int)(c / cpr), kc = (int)(c - (int64_t)n * cpr);
    *reinterpret_cast<u32x4*>(W + (size_t)n * ldw + (size_t)kc * 8) =
        *reinterpret_cast<const u32x4*>(Wkt + ((size_t)(kc >> 3) * N + n) * G_BK + (kc & 7) * 8);
}

}  // namespace lade

using namespace lade;

// bn: weight rows per work-group (32..256); mb: 32-row activation blocks per work-group (1..4; 0 = by M); mt: m-blocks per
// wave (1 | 2 | 3 | 4, divides mb; 0 = 1); nt: 32-row weight tiles per wave (1..4; 0 = spread the tiles over as many n-groups
// as there are waves).  The waves form an (mb/mt) x (bn/32/nt) grid.  Only the shapes in the table of gemm_kernel.hpp are built.
static int gemm_skinny(const void* A, int64_t lda, const void* W, int64_t ldw, bool ktile, void* C, int64_t ldc, float* Cpart,
                       int32_t M, int32_t N, int32_t K, int32_t n_split, int32_t bn, int32_t mb, int32_t mt, int32_t nt, int32_t ring,
                       int32_t epilogue, int32_t dtype, void* stream) {
    // ring 10 / 12 / 14 / 16: the PING-PONG form of the K loop (gemm_pp.hpp: two groups of four waves alternate between multiplying a tile and
    // requesting one) on a ring of its default depth / 2 / 4 / 6 stages
    const bool pp = ring >= 10;
    LADE_REQUIRE(ring == 0 || (ring >= 2 && ring <= 8) || ring == 10 || ring == 12 || ring == 14 || ring == 16, LADE_E_ARG,
                 "lade_gemm_skinny: ring=%d (0 = default, 2..8 stages; 10 / 12 / 14 / 16 = ping-pong K loop on its default / 2 / 4 / 6 stages)", ring);
    LADE_REQUIRE(ring != 7, LADE_E_ARG, "lade_gemm_skinny: no loop is compiled for a ring of 7 stages (2, 3, 4, 5, 6, 8 are)");
    LADE_REQUIRE(epilogue == 0 || (epilogue == 1 && n_split == 1 && N % 32 == 0 && C != nullptr) || (epilogue == 2 && n_split == 1 && Cpart != nullptr), LADE_E_ARG,
                 "lade_gemm_skinny: epilogue=%d needs n_split == 1 and N %% 32 == 0 + an output matrix (1) or the pair buffer in Cpart (2)", epilogue);
    LADE_REQUIRE(A && W && M > 0 && N > 0 && K > 0 && n_split >= 1, LADE_E_ARG, "lade_gemm_skinny: M=%d N=%d K=%d split=%d", M, N, K, n_split);
    LADE_REQUIRE(K % G_BK == 0 && lda % 8 == 0 && ldw % 8 == 0 && N % 8 == 0, LADE_E_ARG,
                 "lade_gemm_skinny: K=%d must be a multiple of %d, strides / N multiples of 8", K, G_BK);
    LADE_REQUIRE(epilogue == 2 || (n_split == 1 ? (C != nullptr && ldc % (epilogue ? 4 : 8) == 0) : (Cpart != nullptr)), LADE_E_ARG, "lade_gemm_skinny: missing output buffer");
    LADE_REQUIRE(dtype == LADE_BF16 || dtype == LADE_F16, LADE_E_DTYPE, "lade_gemm_skinny: dtype=%d", dtype);
    if (mt == 16) {
        // 16-ROW-GRANULAR tiles (gemm16.hpp; probe state: split-K partials only): mb = 16-row activation blocks of the step (5 / 7 / 9 / 11), nt = 16-row weight
        // blocks per wave (1..4), bn = weight rows per work-group: the waves lie along N
        LADE_REQUIRE(nt >= 1 && nt <= 4 && bn % (16 * nt) == 0 && bn / (16 * nt) <= 8 && M <= 16 * mb && N % 4 == 0, LADE_E_ARG, "lade_gemm_skinny (16-row tiles): mb=%d bn=%d nt=%d M=%d", mb, bn, nt, M);
        GemmK g;
        g.A = (const uint16_t*)A; g.W = (const uint16_t*)W; g.C = (uint16_t*)C; g.Cpart = Cpart;
        g.lda = lda; g.ldw = ldw; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.n_split = n_split;
        g.w_ts = ktile ? (int64_t)N * G_BK : 0; g.dbg = 0; g.epi = epilogue; g.n_stage = ring;
#ifdef LADE_EXPERIMENTAL
        const int rc16 = dtype == LADE_BF16 ? gemm16_dispatch_bf16(g, (hipStream_t)stream, mb, bn / (16 * nt), nt) : gemm16_dispatch_f16(g, (hipStream_t)stream, mb, bn / (16 * nt), nt);
        if (rc16 >= 0) return rc16;
        LADE_REQUIRE(false, LADE_E_ARG, "lade_gemm_skinny (16-row tiles): no kernel for %d blocks x %d waves x %d tiles", mb, bn / (16 * nt), nt);
#else
        // 10 % fewer padded rows bought 0-4 % (13B) or nothing (7B) against the 32-row classes: the 16x16x32 tiles need more fragment reads per flop
        // (profiles/r6_gemm16_probe.txt): built only with make EXPERIMENTAL=1
        (void)g;
        LADE_REQUIRE(false, LADE_E_ARG, "lade_gemm_skinny: the 16-row-granular tiles (mt=16) are not in this build (make EXPERIMENTAL=1; lade_build_flags() bit 0)");
#endif
    }
    if (mb == 0) mb = M <= 32 ? 1 : (M <= 64 ? 2 : (M <= 96 ? 3 : (M <= 128 ? 4 : (M <= 160 ? 5 : (M <= 192 ? 6 : 8)))));
    if (mt == 0) mt = mb <= 4 ? 1 : (mb == 5 ? 5 : mb / 2);
    if (mb > 4 && nt == 0) nt = 1;
    LADE_REQUIRE(mb >= 1 && mb <= 8 && mt >= 1 && mt <= 5 && mb % mt == 0, LADE_E_ARG, "lade_gemm_skinny: mb=%d mt=%d", mb, mt);
    // 192 / 256-row work-groups: the activation tile leaves room for <= 128 weight rows per stage of a 3-stage ring; the wider
    // shapes are the double-buffered 256 x 256 tile (mb = 8, bn = 256, nt = 2 | 4) and the 160-row class (mb = 5: one wave holds all five
    // m-blocks, the eight waves - or four with two n-tiles each - lie along N: 256 weight rows + 160 activation rows = 52 KB per stage, three stages)
    if (mb > 5 && bn > 128 && !(mb == 8 && bn == 256 && (nt == 2 || nt == 4))) {
        // (the argmax epilogue's pair buffer is indexed by the CALLER's ceil(N / bn): a silently narrower block would shift every row's pairs)
        LADE_REQUIRE(epilogue != 2, LADE_E_ARG, "lade_gemm_skinny: epilogue 2 with mb=%d needs bn <= 128 (got %d): the pair buffer's stride is ceil(N / bn)", mb, bn);
        bn = 128;
    }
    const int mw = mb / mt;
    const int tiles = bn <= 32 ? 1 : (bn <= 64 ? 2 : (bn <= 96 ? 3 : (bn <= 128 ? 4 : (bn <= 192 ? 6 : (bn <= 224 ? 7 : 8)))));      // 32-row weight tiles per work-group
    if (nt == 0) {                                     // default: as many n-groups as waves allow (a ping-pong group is four waves)
        const int ng_max = (pp ? 4 : 8) / mw > 0 ? (pp ? 4 : 8) / mw : 1;
        nt = 1;
        while (tiles / nt > ng_max || tiles % nt != 0) ++nt;
    }
    LADE_REQUIRE(nt >= 1 && nt <= 4 && tiles % nt == 0, LADE_E_ARG, "lade_gemm_skinny: bn=%d (%d tiles) nt=%d", bn, tiles, nt);
    const int ng = tiles / nt;
    GemmK g;
    g.A = (const uint16_t*)A; g.W = (const uint16_t*)W; g.C = (uint16_t*)C; g.Cpart = Cpart;
    g.lda = lda; g.ldw = ldw; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.n_split = n_split;
    g.w_ts = ktile ? (int64_t)N * G_BK : 0;
    { static int dbg = -1; if (dbg < 0) dbg = debug_int("gemm_dbg"); g.dbg = dbg; }
    g.epi = epilogue;
    g.n_stage = pp ? ring - 10 : ring;
    hipStream_t st = (hipStream_t)stream;
#ifdef LADE_EXPERIMENTAL
    const int rc = pp ? (dtype == LADE_BF16 ? gemm_pp_dispatch_bf16(g, st, mw, mt, ng, nt) : gemm_pp_dispatch_f16(g, st, mw, mt, ng, nt))
                      : (dtype == LADE_BF16 ? gemm_dispatch_bf16(g, st, mw, mt, ng, nt) : gemm_dispatch_f16(g, st, mw, mt, ng, nt));
#else
    // the ping-pong K loop tied or lost against the lock-step loop at every BASELINE width (profiles/r6_gemm_160_pingpong_probe.txt): built only with make EXPERIMENTAL=1
    LADE_REQUIRE(!pp, LADE_E_ARG, "lade_gemm_skinny: the ping-pong K loop (ring=%d) is not in this build (make EXPERIMENTAL=1; lade_build_flags() bit 0)", ring);
    const int rc = dtype == LADE_BF16 ? gemm_dispatch_bf16(g, st, mw, mt, ng, nt) : gemm_dispatch_f16(g, st, mw, mt, ng, nt);
#endif
    if (rc >= 0) return rc;
    LADE_REQUIRE(!pp, LADE_E_ARG, "lade_gemm_skinny: no ping-pong kernel for mb=%d mt=%d bn=%d nt=%d (a group is a %d x %d grid of four waves)", mb, mt, bn, nt, mw, ng);
    LADE_REQUIRE(false, LADE_E_ARG, "lade_gemm_skinny: no kernel for mb=%d mt=%d bn=%d nt=%d (wave grid %d x %d)", mb, mt, bn, nt, mw, ng);
}

extern "C" int lade_gemm_skinny(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, float* Cpart,
                                int32_t M, int32_t N, int32_t K, int32_t n_split, int32_t bn, int32_t mb, int32_t mt, int32_t nt,
                                int32_t ring, int32_t epilogue, int32_t dtype, void* stream) {
    return gemm_skinny(A, lda, W, ldw, false, C, ldc, Cpart, M, N, K, n_split, bn, mb, mt, nt, ring, epilogue, dtype, stream);
}

// The same GEMM on a weight stored K-tile-major: Wkt[K/64][N][64] (lade_weight_to_ktile).  Same arithmetic in the same order, so the
// results are bit-identical to lade_gemm_skinny on the row-major weight; only the addresses the weight DMA reads differ.
extern "C" int lade_gemm_skinny_kt(const void* A, int64_t lda, const void* Wkt, void* C, int64_t ldc, float* Cpart,
                                   int32_t M, int32_t N, int32_t K, int32_t n_split, int32_t bn, int32_t mb, int32_t mt, int32_t nt,
                                   int32_t ring, int32_t epilogue, int32_t dtype, void* stream) {
    return gemm_skinny(A, lda, Wkt, 8, true, C, ldc, Cpart, M, N, K, n_split, bn, mb, mt, nt, ring, epilogue, dtype, stream);
}

// Split-K GEMM with register-resident activations (gemm_ra.hpp): M <= 128 rows, K-tile-major weights, fp32 partials for a `*_parts`
// consumer.  The K slices are those of lade_gemm_skinny_kt with the same n_split, and so are the partials - bit for bit.
extern "C" int lade_gemm_ra_kt(const void* A, int64_t lda, const void* Wkt, float* Cpart, int32_t M, int32_t N, int32_t K, int32_t n_split,
                               int32_t cs, int32_t n_groups, int32_t dtype, void* stream) {
    LADE_REQUIRE(A && Wkt && Cpart && M > 0 && M <= 128 && N > 0 && K > 0 && n_split >= 1, LADE_E_ARG, "lade_gemm_ra_kt: M=%d (1..128) N=%d K=%d split=%d", M, N, K, n_split);
    LADE_REQUIRE(K % G_BK == 0 && lda % 8 == 0 && (cs == 2 || cs == 4) && N % (32 * cs) == 0, LADE_E_ARG,
                 "lade_gemm_ra_kt: K=%d must be a multiple of %d, lda of 8, cs=%d one of 2 | 4, N=%d a multiple of 32 cs", K, G_BK, cs, N);
    LADE_REQUIRE(dtype == LADE_BF16 || dtype == LADE_F16, LADE_E_DTYPE, "lade_gemm_ra_kt: dtype=%d", dtype);
    const int k_tiles = K / G_BK, tps = (k_tiles + n_split - 1) / n_split;
    LADE_REQUIRE(tps <= RA_KT && (n_split - 1) * tps < k_tiles, LADE_E_ARG,
                 "lade_gemm_ra_kt: K=%d in %d splits = %d K tiles per work-group (a work-group holds <= %d in registers; no split may be empty)", K, n_split, tps, RA_KT);
    LADE_REQUIRE((int64_t)lda * 128 * 2 < ((int64_t)1 << 31), LADE_E_LIMIT, "lade_gemm_ra_kt: lda=%lld too large", (long long)lda);
    if (n_groups <= 0) {
        static int n_cu = 0;
        if (n_cu == 0) {
            int dev = 0;
            hipDeviceProp_t prop;
            n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        }
        n_groups = n_cu / n_split > 0 ? n_cu / n_split : 1;
    }
    const int nc = N / (32 * cs);
    if (n_groups > nc) n_groups = nc;
    GemmRA g;
    g.A = (const uint16_t*)A; g.W = (const uint16_t*)Wkt; g.Cpart = Cpart; g.lda = lda;
    g.M = M; g.N = N; g.K = K; g.n_split = n_split; g.tps = tps; g.n_groups = n_groups;
    { static int dbg = -1; if (dbg < 0) dbg = debug_int("gemm_dbg"); g.dbg = dbg; }
    const int mw = (M + 31) / 32;
    hipStream_t st = (hipStream_t)stream;
#ifdef LADE_EXPERIMENTAL
    const int rc = dtype == LADE_BF16 ? gemm_ra_dispatch_bf16(g, st, mw, cs) : gemm_ra_dispatch_f16(g, st, mw, cs);
    if (rc >= 0) return rc;
    LADE_REQUIRE(false, LADE_E_ARG, "lade_gemm_ra_kt: no kernel for %d row blocks x %d-strip chunks", mw, cs);
#else
    // bit-identical to the LDS-ring kernel, 10-35 % slower at every BASELINE width (profiles/r6_gemm_ra_probe.txt): built only with make EXPERIMENTAL=1
    (void)mw; (void)st; (void)g;
    LADE_REQUIRE(false, LADE_E_ARG, "lade_gemm_ra_kt: the register-resident-activation kernel is not in this build (make EXPERIMENTAL=1; lade_build_flags() bit 0)");
#endif
}

// Wkt[kt][n][0..63] = W[n][64 kt .. 64 kt + 63]: one 16-byte chunk per thread, a wave writes 1 KiB contiguous
extern "C" int lade_weight_to_ktile(const void* W, int64_t ldw, void* Wkt, int32_t N, int32_t K, int32_t dtype, void* stream) {
    LADE_REQUIRE(W && Wkt && W != Wkt && N > 0 && K > 0 && K % G_BK == 0 && ldw >= K && ldw % 8 == 0, LADE_E_ARG,
                 "lade_weight_to_ktile: N=%d K=%d ldw=%lld (K must be a multiple of %d, out of place)", N, K, (long long)ldw, G_BK);
    LADE_REQUIRE(dtype == LADE_BF16 || dtype == LADE_F16, LADE_E_DTYPE, "lade_weight_to_ktile: dtype=%d", dtype);
    const int64_t chunks = (int64_t)N * (K / 8);
    LADE_REQUIRE(chunks / 256 + 1 < (int64_t)1 << 31, LADE_E_LIMIT, "lade_weight_to_ktile: N=%d K=%d too large", N, K);
    hipLaunchKernelGGL(weight_to_ktile_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)W, ldw, (uint16_t*)Wkt, N, K);
    return check_launch("lade_weight_to_ktile");
}

// W[n][64 kt + j] = Wkt[kt][n][j]: the row-major matrix a library GEMM needs (prefill chunks of a model whose weights are held
// K-tile-major only), written into a caller-provided scratch
extern "C" int lade_weight_from_ktile(const void* Wkt, void* W, int64_t ldw, int32_t N, int32_t K, int32_t dtype, void* stream) {
    LADE_REQUIRE(W && Wkt && W != Wkt && N > 0 && K > 0 && K % G_BK == 0 && ldw >= K && ldw % 8 == 0, LADE_E_ARG,
                 "lade_weight_from_ktile: N=%d K=%d ldw=%lld (K must be a multiple of %d, out of place)", N, K, (long long)ldw, G_BK);
    LADE_REQUIRE(dtype == LADE_BF16 || dtype == LADE_F16, LADE_E_DTYPE, "lade_weight_from_ktile: dtype=%d", dtype);
    const int64_t chunks = (int64_t)N * (K / 8);
    LADE_REQUIRE(chunks / 256 + 1 < (int64_t)1 << 31, LADE_E_LIMIT, "lade_weight_from_ktile: N=%d K=%d too large", N, K);
    hipLaunchKernelGGL(weight_from_ktile_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)Wkt, (uint16_t*)W, ldw, N, K);
    return check_launch("lade_weight_from_ktile");
}

extern "C" int lade_splitk_reduce(const float* part, void* C, int64_t ldc, int32_t M, int32_t N, int32_t n_split, int32_t dtype, void* stream) {
    LADE_REQUIRE(part && C && M > 0 && N > 0 && N % 4 == 0 && n_split >= 1, LADE_E_ARG, "lade_splitk_reduce: bad args");
    dim3 grid(cdiv(N / 4, 256), M);
    if (dtype == LADE_BF16) hipLaunchKernelGGL(splitk_reduce_kernel<BF16>, grid, dim3(256), 0, (hipStream_t)stream, part, (uint16_t*)C, ldc, M, N, n_split);
    else if (dtype == LADE_F16) hipLaunchKernelGGL(splitk_reduce_kernel<F16>, grid, dim3(256), 0, (hipStream_t)stream, part, (uint16_t*)C, ldc, M, N, n_split);
    else LADE_REQUIRE(false, LADE_E_DTYPE, "lade_splitk_reduce: dtype=%d", dtype);
    return check_launch("lade_splitk_reduce");
}
