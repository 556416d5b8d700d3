// Elementwise / row kernels around the GEMMs of one model step: RMSNorm (+ residual add), SwiGLU
// gate, row gather (embedding lookup, logits-row selection) and the fp32 row softmax used by the
// sampling verify.  Reference: LlamaRMSNorm / LlamaMLP (lade/models/modeling_llama.py:222-227,
// :360-380), lade/decoding.py:484-489.  All HBM-streaming: 16-byte accesses, one pass.
#include "common.hpp"

namespace lade {

template <typename T> struct St;
template <> struct St<BF16> { typedef uint16_t S; };
template <> struct St<F16> { typedef uint16_t S; };
template <> struct St<F32> { typedef float S; };
template <typename T> __device__ __forceinline__ float ldf(const typename St<T>::S* p, size_t i);
template <> __device__ __forceinline__ float ldf<BF16>(const uint16_t* p, size_t i) { return to_f32<BF16>(p[i]); }
template <> __device__ __forceinline__ float ldf<F16>(const uint16_t* p, size_t i) { return to_f32<F16>(p[i]); }
template <> __device__ __forceinline__ float ldf<F32>(const float* p, size_t i) { return p[i]; }
template <typename T> __device__ __forceinline__ typename St<T>::S stf(float f);
template <> __device__ __forceinline__ uint16_t stf<BF16>(float f) { return from_f32<BF16>(f); }
template <> __device__ __forceinline__ uint16_t stf<F16>(float f) { return from_f32<F16>(f); }
template <> __device__ __forceinline__ float stf<F32>(float f) { return f; }

__device__ __forceinline__ float block_sum(float v, float* sm) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += sm[i];
    __syncthreads();
    return t;
}
__device__ __forceinline__ float block_max(float v, float* sm) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    float t = -INFINITY;
    for (int i = 0; i < nw; ++i) t = fmaxf(t, sm[i]);
    __syncthreads();
    return t;
}

// ---- 16-byte vector access of the storage types --------------------------------------------------
template <typename T> struct Vec;             // VEC elements per 16 bytes
template <> struct Vec<BF16> { static constexpr int N = 8; };
template <> struct Vec<F16> { static constexpr int N = 8; };
template <> struct Vec<F32> { static constexpr int N = 4; };
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <typename T> __device__ __forceinline__ void unpack(const u32x4& v, float* f);
template <> __device__ __forceinline__ void unpack<BF16>(const u32x4& v, float* f) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[2 * e] = __uint_as_float(v[e] << 16); f[2 * e + 1] = __uint_as_float(v[e] & 0xffff0000u); }
}
template <> __device__ __forceinline__ void unpack<F16>(const u32x4& v, float* f) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[2 * e] = to_f32<F16>((uint16_t)(v[e] & 0xffffu)); f[2 * e + 1] = to_f32<F16>((uint16_t)(v[e] >> 16)); }
}
template <> __device__ __forceinline__ void unpack<F32>(const u32x4& v, float* f) {
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = __uint_as_float(v[e]);
}
template <typename T> __device__ __forceinline__ u32x4 repack(const float* f);
template <> __device__ __forceinline__ u32x4 repack<BF16>(const float* f) {
    u32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (uint32_t)from_f32<BF16>(f[2 * e]) | ((uint32_t)from_f32<BF16>(f[2 * e + 1]) << 16);
    return v;
}
template <> __device__ __forceinline__ u32x4 repack<F16>(const float* f) {
    u32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (uint32_t)from_f32<F16>(f[2 * e]) | ((uint32_t)from_f32<F16>(f[2 * e + 1]) << 16);
    return v;
}
template <> __device__ __forceinline__ u32x4 repack<F32>(const float* f) {
    return u32x4{__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])};
}
// Sum of the n_parts fp32 split-K partials of a GEMM output (fixed order: deterministic), rounded once to the
// storage type like a library GEMM's output would be.  part_stride = elements between consecutive partials.
template <typename T, int N, int R>
__device__ __forceinline__ void load_parts_r(const float* parts, int n_parts, size_t part_stride, size_t idx, float* out) {
#pragma unroll
    for (int e = 0; e < N; ++e) out[e] = 0.f;
    // R partials per round: all their loads are in flight together (these kernels are latency bound); the summation order stays
    // 0,1,2,...  The first round is straight-line code (no loop header in front of its loads: the compiler puts a wait for the
    // caller's earlier loads in front of a loop, which serialises two memory latencies); further rounds (n_parts > R) loop.
    {
        float4 v[R][N / 4];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const float* p = parts + (size_t)min(j, n_parts - 1) * part_stride + idx;
#pragma unroll
            for (int e = 0; e < N / 4; ++e) v[j][e] = *reinterpret_cast<const float4*>(p + 4 * e);
        }
#pragma unroll
        for (int j = 0; j < R; ++j)
            if (j < n_parts) {                                     // a partial past the end was the last one again: not added
#pragma unroll
                for (int e = 0; e < N / 4; ++e) {
                    out[4 * e] += v[j][e].x; out[4 * e + 1] += v[j][e].y; out[4 * e + 2] += v[j][e].z; out[4 * e + 3] += v[j][e].w;
                }
            }
    }
    for (int s0 = R; s0 < n_parts; s0 += R) {
        float4 v[R][N / 4];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const float* p = parts + (size_t)min(s0 + j, n_parts - 1) * part_stride + idx;
#pragma unroll
            for (int e = 0; e < N / 4; ++e) v[j][e] = *reinterpret_cast<const float4*>(p + 4 * e);
        }
#pragma unroll
        for (int j = 0; j < R; ++j)
            if (s0 + j < n_parts) {
#pragma unroll
                for (int e = 0; e < N / 4; ++e) {
                    out[4 * e] += v[j][e].x; out[4 * e + 1] += v[j][e].y; out[4 * e + 2] += v[j][e].z; out[4 * e + 3] += v[j][e].w;
                }
            }
    }
#pragma unroll
    for (int e = 0; e < N; ++e) { const typename St<T>::S sv = stf<T>(out[e]); out[e] = ldf<T>(&sv, 0); }
}

template <typename T, int N>
__device__ __forceinline__ void load_parts(const float* parts, int n_parts, size_t part_stride, size_t idx, float* out) {
    if (n_parts <= 4) load_parts_r<T, N, 4>(parts, n_parts, part_stride, idx, out);      // wave-uniform
    else load_parts_r<T, N, 8>(parts, n_parts, part_stride, idx, out);
}

// value after one rounding to the storage type
template <typename T> __device__ __forceinline__ float rnd_st(float f) { typename St<T>::S s = stf<T>(f); return ldf<T>(&s, 0); }

// y = w * cast(x_f32 * rsqrt(mean(x^2)+eps));  ADD: x <- cast(x + r) first (residual), norm of the sum.
// One block of BT threads per row, the row lives in registers (<= CH 16-byte chunks per thread): one HBM pass.
// Latency bound (rows <= 240): every load of the row - x, the residual or all its split-K partials, the norm
// weight - is issued before the first use.
template <typename T, bool ADD, int CH, int BT>
__global__ __launch_bounds__(BT) void rmsnorm_kernel(typename St<T>::S* x, const typename St<T>::S* r, const typename St<T>::S* w,
                                                     typename St<T>::S* y, int hidden, float eps, const float* parts, int n_parts,
                                                     size_t part_stride, const int32_t* sel = nullptr, int src_rows = 0) {
    constexpr int N = Vec<T>::N;
    __shared__ float sm[BT / 64];
    // sel: the block's INPUT row is sel[blockIdx.x] (row-pruned tail of the step: only the rows whose logits are read are normed),
    // its output row is blockIdx.x, and x is not written back
    size_t base = (size_t)blockIdx.x * hidden;
    const size_t obase = base;
    if (sel) {
        int rsel = sel[blockIdx.x];
        rsel = rsel < 0 ? 0 : (rsel >= src_rows ? src_rows - 1 : rsel);
        base = (size_t)rsel * hidden;
    }
    const int nvec = hidden / N;
    float v[CH][N];
    u32x4 wv[CH];
    float ss = 0.f;
    // without ADD, `r` may name a row table (the embedding matrix, lade_embed_rmsnorm): the block's input row is r[sel[blockIdx.x]] and
    // is also copied to x[blockIdx.x] - the step's embedding gather and its first norm in one launch (modeling_llama.py:1413, :857)
    const typename St<T>::S* xin = (!ADD && r) ? r : x;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int i = threadIdx.x + c * BT;
        if (i < nvec) {
            wv[c] = *reinterpret_cast<const u32x4*>(w + (size_t)i * N);
            const u32x4 xr = *reinterpret_cast<const u32x4*>(xin + base + (size_t)i * N);      // used only after the partial loads are in flight
            if (!ADD && r) *reinterpret_cast<u32x4*>(x + obase + (size_t)i * N) = xr;
            float rr[N];
            if (ADD) {
                if (parts) load_parts<T, N>(parts, n_parts, part_stride, base + (size_t)i * N, rr);
                else unpack<T>(*reinterpret_cast<const u32x4*>(r + base + (size_t)i * N), rr);
            }
            unpack<T>(xr, v[c]);
            if (ADD) {
#pragma unroll
                for (int e = 0; e < N; ++e) v[c][e] = rnd_st<T>(v[c][e] + rr[e]);
                if (!sel) *reinterpret_cast<u32x4*>(x + base + (size_t)i * N) = repack<T>(v[c]);
            }
#pragma unroll
            for (int e = 0; e < N; ++e) ss += v[c][e] * v[c][e];
        }
    }
    const float var = block_sum(ss, sm) / (float)hidden;
    const float inv = rsqrtf(var + eps);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int i = threadIdx.x + c * BT;
        if (i < nvec) {
            float ww[N], o[N];
            unpack<T>(wv[c], ww);
#pragma unroll
            for (int e = 0; e < N; ++e) o[e] = ww[e] * rnd_st<T>(v[c][e] * inv);     // weight * hidden.to(dtype)
            *reinterpret_cast<u32x4*>(y + obase + (size_t)i * N) = repack<T>(o);
        }
    }
}

// out[r][i] = silu(gu[r][i]) * gu[r][inter+i], rounded like torch: act(gate) -> dtype, then * up -> dtype
// layout 0: a row is [gate (inter) | up (inter)] (two projections concatenated); layout 1: groups of 16 - [16 gate | their 16 up] - the
// row order of the engine's fused gate/up weight (see the SwiGLU epilogue of gemm.hip)
template <typename T>
__global__ __launch_bounds__(256) void silu_mul_kernel(const typename St<T>::S* gu, typename St<T>::S* out, int inter, const float* parts,
                                                       int n_parts, size_t part_stride, int layout) {
    constexpr int N = Vec<T>::N;
    const size_t rb = (size_t)blockIdx.y * 2 * inter;
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) * N;
    if (i >= inter) return;
    const size_t gi = layout ? (size_t)(i >> 4) * 32 + (i & 15) : (size_t)i;
    const size_t ui = layout ? gi + 16 : (size_t)inter + i;
    float g[N], u[N], o[N];
    if (parts) {
        load_parts<T, N>(parts, n_parts, part_stride, rb + gi, g);
        load_parts<T, N>(parts, n_parts, part_stride, rb + ui, u);
    } else {
        unpack<T>(*reinterpret_cast<const u32x4*>(gu + rb + gi), g);
        unpack<T>(*reinterpret_cast<const u32x4*>(gu + rb + ui), u);
    }
#pragma unroll
    for (int e = 0; e < N; ++e) o[e] = rnd_st<T>(g[e] / (1.f + __expf(-g[e]))) * u[e];
    *reinterpret_cast<u32x4*>(out + (size_t)blockIdx.y * inter + i) = repack<T>(o);
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const unsigned char* src, const int32_t* idx, unsigned char* dst, int row_bytes, int src_rows) {
    int r = idx[blockIdx.x];
    r = r < 0 ? 0 : (r >= src_rows ? src_rows - 1 : r);
    const unsigned char* s = src + (size_t)r * row_bytes;
    unsigned char* d = dst + (size_t)blockIdx.x * row_bytes;
    if ((row_bytes & 15) == 0) {
        for (int i = threadIdx.x * 16; i < row_bytes; i += blockDim.x * 16) *reinterpret_cast<uint4*>(d + i) = *reinterpret_cast<const uint4*>(s + i);
    } else {
        for (int i = threadIdx.x; i < row_bytes; i += blockDim.x) d[i] = s[i];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const typename St<T>::S* logits, int64_t ld, int V, float inv_temp, float* probs) {
    __shared__ float sm[8];
    const size_t base = (size_t)blockIdx.x * ld;
    float* out = probs + (size_t)blockIdx.x * V;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < V; i += blockDim.x) mx = fmaxf(mx, ldf<T>(logits, base + i) * inv_temp);
    mx = block_max(mx, sm);
    float s = 0.f;
    for (int i = threadIdx.x; i < V; i += blockDim.x) {
        const float e = expf(ldf<T>(logits, base + i) * inv_temp - mx);
        out[i] = e;
        s += e;
    }
    s = block_sum(s, sm);
    const float inv = 1.f / s;
    for (int i = threadIdx.x; i < V; i += blockDim.x) out[i] *= inv;
}

// The same row softmax with the row held in registers: 1024 threads, 16-byte loads, CH chunks per thread (V <= 1024 * CH * N), one read of
// the logits and one write of the probabilities.  The sampling loop calls this once per step for the ONE row a token is drawn from; the
// three dependent scalar sweeps above cost 97 us for a 32000-entry row (a quarter-occupied CU walking 125 elements per thread three
// times), this form ~6 us.  Used when the row is 16-byte aligned and V a multiple of the vector width.
template <typename T, int CH>
__global__ __launch_bounds__(1024) void softmax_rows_vec_kernel(const typename St<T>::S* logits, int64_t ld, int V, float inv_temp, float* probs) {
    constexpr int N = Vec<T>::N;
    __shared__ float sm[16];
    const size_t base = (size_t)blockIdx.x * ld;
    float* out = probs + (size_t)blockIdx.x * V;
    const int nvec = V / N;
    float v[CH][N];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int i = threadIdx.x + c * 1024;
        if (i < nvec) {
            unpack<T>(*reinterpret_cast<const u32x4*>(logits + base + (size_t)i * N), v[c]);
#pragma unroll
            for (int e = 0; e < N; ++e) { v[c][e] *= inv_temp; mx = fmaxf(mx, v[c][e]); }
        }
    }
    mx = block_max(mx, sm);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int i = threadIdx.x + c * 1024;
        if (i < nvec) {
#pragma unroll
            for (int e = 0; e < N; ++e) { v[c][e] = expf(v[c][e] - mx); s += v[c][e]; }
        }
    }
    s = block_sum(s, sm);
    const float inv = 1.f / s;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int i = threadIdx.x + c * 1024;
        if (i < nvec) {
#pragma unroll
            for (int q = 0; q < N / 4; ++q)
                *reinterpret_cast<float4*>(out + (size_t)i * N + 4 * q) = float4{v[c][4 * q] * inv, v[c][4 * q + 1] * inv, v[c][4 * q + 2] * inv, v[c][4 * q + 3] * inv};
        }
    }
}

// Sampling verify, device side (lade/decoding.py:484-540): the host-side acceptance loop only ever looks at the probability of a
// DRAFT token under the distribution that follows an accepted prefix.  Row 0 is the distribution after the input token,
// row 1 + c*gs + j the one after position j of candidate c.  For row 0 the drafts in question are the candidates' first
// tokens, for row 1 + c*gs + j their tokens at position j+1.  One block per row: softmax statistics (max, sum of exponentials
// of logits/temperature) in one sweep over the V logits, then the g probabilities are gathered; the [rows, V] probability
// matrix the reference materialises (guess_probs, 15 MB per step at config 3) is never written.
template <typename T>
__global__ __launch_bounds__(1024) void softmax_gather_kernel(const typename St<T>::S* logits, int64_t ld, int V, float inv_temp, int skip,
                                                              const int32_t* guess, const int32_t* g_dev, int g_host, int gs, int g_cap,
                                                              float* scal, float* stats) {
    __shared__ float sm[16];
    const int row = blockIdx.x;
    const int g = g_dev ? min(*g_dev, g_cap) : g_host;
    int pos = 0;                                    // position inside the n-gram whose drafts this row judges
    if (row > 0) {
        pos = (row - 1) % gs + 1;
        if (pos >= gs || (row - 1) / gs >= g) { pos = -1; }      // last row of a candidate / padded candidate slot: never consulted
    }
    const size_t base = (size_t)(row == 0 ? 0 : row + skip) * ld;
    // online max / sum: one pass over the row
    float mx = -INFINITY, sum = 0.f;
    for (int i = threadIdx.x; i < V; i += blockDim.x) {
        const float x = ldf<T>(logits, base + i) * inv_temp;
        if (x > mx) { sum = sum * expf(mx - x) + 1.f; mx = x; }
        else if (x > -INFINITY) sum += expf(x - mx);
    }
    const float bm = block_max(mx, sm);
    sum = (mx == -INFINITY) ? 0.f : sum * expf(mx - bm);
    const float bs = block_sum(sum, sm);
    if (threadIdx.x == 0) { stats[2 * row] = bm; stats[2 * row + 1] = bs; }
    if (pos >= 0)
        for (int c = threadIdx.x; c < g; c += blockDim.x) {
            int tok = guess[c * gs + pos];
            tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
            scal[(size_t)row * g_cap + c] = expf(ldf<T>(logits, base + tok) * inv_temp - bm) / bs;
        }
}

}  // namespace lade

using namespace lade;


#define DISPATCH_DTYPE(dtype, CALL)                                   \
    switch (dtype) {                                                  \
        case LADE_BF16: { typedef BF16 TT; CALL; } break;             \
        case LADE_F16: { typedef F16 TT; CALL; } break;               \
        case LADE_F32: { typedef F32 TT; CALL; } break;               \
        default: LADE_REQUIRE(false, LADE_E_DTYPE, "unsupported dtype %d", dtype); \
    }

template <bool ADD>
static int launch_rmsnorm(void* x, const void* r, const void* weight, void* y, int rows, int hidden, float eps, int dtype, hipStream_t st,
                          const float* parts = nullptr, int n_parts = 0, size_t part_stride = 0, const int32_t* sel = nullptr, int src_rows = 0) {
    const int nvec_bytes = dtype == LADE_F32 ? 4 : 8;
    LADE_REQUIRE(hidden % nvec_bytes == 0, LADE_E_ARG, "lade_rmsnorm: hidden=%d must be a multiple of %d", hidden, nvec_bytes);
    const int vecs = hidden / nvec_bytes;              // 16-byte chunks per row
    LADE_REQUIRE(vecs <= 4 * 1024, LADE_E_LIMIT, "lade_rmsnorm: hidden=%d too large for the register-resident row", hidden);
#define RMS_LAUNCH(CH, BT) DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((rmsnorm_kernel<TT, ADD, CH, BT>), dim3(rows), dim3(BT), 0, st, (St<TT>::S*)x, \
                                          (const St<TT>::S*)r, (const St<TT>::S*)weight, (St<TT>::S*)y, hidden, eps, parts, n_parts, part_stride, sel, src_rows))
    // one 16-byte chunk per thread up to 1024 threads - every load of the row (x, all its split-K partials, the weight) is in flight at
    // once; with two chunks per thread the store of chunk 0 into x sits between the loads of chunk 0 and chunk 1 (x may alias them), so the
    // row pays its memory latency twice: 8.7-12 us instead of 5.5-6.5 at hidden 5120 / 8192 (profiles/r5_bench_c4_kernel_medians.txt)
    if (vecs <= 256) { RMS_LAUNCH(1, 256); } else if (vecs <= 512) { RMS_LAUNCH(1, 512); } else if (vecs <= 640) { RMS_LAUNCH(1, 640); }
    else if (vecs <= 768) { RMS_LAUNCH(1, 768); } else if (vecs <= 1024) { RMS_LAUNCH(1, 1024); } else if (vecs <= 2048) { RMS_LAUNCH(2, 1024); }
    else { RMS_LAUNCH(4, 1024); }
#undef RMS_LAUNCH
    return check_launch(ADD ? "lade_add_rmsnorm" : "lade_rmsnorm");
}

extern "C" int lade_rmsnorm(const void* x, const void* weight, void* y, int32_t rows, int32_t hidden, float eps, int32_t dtype, void* stream) {
    LADE_REQUIRE(x && weight && y && rows >= 0 && hidden > 0, LADE_E_ARG, "lade_rmsnorm: rows=%d hidden=%d", rows, hidden);
    if (rows == 0) return LADE_OK;
    hipStream_t st = (hipStream_t)stream;
    return launch_rmsnorm<false>((void*)x, nullptr, weight, y, rows, hidden, eps, dtype, st);
}

extern "C" int lade_embed_rmsnorm(const void* table, int32_t table_rows, const int32_t* ids, void* x, const void* weight, void* y, int32_t rows,
                                  int32_t hidden, float eps, int32_t dtype, void* stream) {
    LADE_REQUIRE(table && ids && x && weight && y && rows >= 0 && hidden > 0 && table_rows > 0 && x != table, LADE_E_ARG,
                 "lade_embed_rmsnorm: rows=%d hidden=%d table_rows=%d", rows, hidden, table_rows);
    if (rows == 0) return LADE_OK;
    return launch_rmsnorm<false>(x, table, weight, y, rows, hidden, eps, dtype, (hipStream_t)stream, nullptr, 0, 0, ids, table_rows);
}

extern "C" int lade_add_rmsnorm(void* x, const void* r, const void* weight, void* y, int32_t rows, int32_t hidden, float eps, int32_t dtype, void* stream) {
    LADE_REQUIRE(x && r && weight && y && rows >= 0 && hidden > 0, LADE_E_ARG, "lade_add_rmsnorm: rows=%d hidden=%d", rows, hidden);
    if (rows == 0) return LADE_OK;
    hipStream_t st = (hipStream_t)stream;
    return launch_rmsnorm<true>(x, r, weight, y, rows, hidden, eps, dtype, st);
}

extern "C" int lade_add_rmsnorm_rows(const void* x, const void* r, const float* parts, int32_t n_parts, int64_t part_stride, const int32_t* sel,
                                     const void* weight, void* y, int32_t n_sel, int32_t src_rows, int32_t hidden, float eps, int32_t dtype, void* stream) {
    LADE_REQUIRE(x && sel && weight && y && n_sel >= 0 && src_rows > 0 && hidden > 0 && ((r != nullptr) != (parts != nullptr && n_parts > 0)), LADE_E_ARG,
                 "lade_add_rmsnorm_rows: n_sel=%d src_rows=%d hidden=%d n_parts=%d (exactly one of r / parts)", n_sel, src_rows, hidden, n_parts);
    LADE_REQUIRE(!parts || dtype != LADE_F32, LADE_E_DTYPE, "lade_add_rmsnorm_rows: split-K partials come from the 16-bit GEMM only");
    if (n_sel == 0) return LADE_OK;
    return launch_rmsnorm<true>((void*)x, r, weight, y, n_sel, hidden, eps, dtype, (hipStream_t)stream, parts, n_parts, (size_t)part_stride, sel, src_rows);
}

extern "C" int lade_silu_mul(const void* gu, void* out, int32_t rows, int32_t inter, int32_t layout, int32_t dtype, void* stream) {
    LADE_REQUIRE(gu && out && rows >= 0 && inter > 0 && (layout == 0 || (layout == 1 && inter % 16 == 0)), LADE_E_ARG, "lade_silu_mul: rows=%d inter=%d layout=%d", rows, inter, layout);
    if (rows == 0) return LADE_OK;
    hipStream_t st = (hipStream_t)stream;
    LADE_REQUIRE(inter % 8 == 0, LADE_E_ARG, "lade_silu_mul: inter=%d must be a multiple of 8", inter);
    const int per_thr = dtype == LADE_F32 ? 4 : 8;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(silu_mul_kernel<TT>, dim3(cdiv(inter / per_thr, 256), rows), dim3(256), 0, st, (const St<TT>::S*)gu, (St<TT>::S*)out, inter, (const float*)nullptr, 0, (size_t)0, layout));
    return check_launch("lade_silu_mul");
}

extern "C" int lade_gather_rows(const void* src, const int32_t* idx, void* dst, int32_t rows, int32_t width, int32_t elem_bytes, int32_t src_rows, void* stream) {
    LADE_REQUIRE(src && idx && dst && rows >= 0 && width > 0 && elem_bytes > 0 && src_rows > 0, LADE_E_ARG, "lade_gather_rows: rows=%d width=%d", rows, width);
    if (rows == 0) return LADE_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)src, idx, (unsigned char*)dst, width * elem_bytes, src_rows);
    return check_launch("lade_gather_rows");
}

extern "C" int lade_softmax_rows(const void* logits, int64_t ld, int32_t rows, int32_t V, int32_t dtype, float temperature, float* probs, void* stream) {
    LADE_REQUIRE(logits && probs && rows >= 0 && V > 0 && ld >= V && temperature > 0.f, LADE_E_ARG, "lade_softmax_rows: rows=%d V=%d T=%f", rows, V, temperature);
    if (rows == 0) return LADE_OK;
    hipStream_t st = (hipStream_t)stream;
    const int esz = dtype == LADE_F32 ? 4 : 2, nvw = 16 / esz;
    const bool vec_ok = V % nvw == 0 && V % 4 == 0 && ((uintptr_t)logits % 16) == 0 && ((size_t)ld * esz) % 16 == 0 && ((uintptr_t)probs % 16) == 0;
    const int chunks = (V / nvw + 1023) / 1024;             // 16-byte chunks per thread
    if (vec_ok && chunks <= 8) {
        if (chunks <= 2) { DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((softmax_rows_vec_kernel<TT, 2>), dim3(rows), dim3(1024), 0, st, (const St<TT>::S*)logits, ld, V, 1.f / temperature, probs)); }
        else if (chunks <= 4) { DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((softmax_rows_vec_kernel<TT, 4>), dim3(rows), dim3(1024), 0, st, (const St<TT>::S*)logits, ld, V, 1.f / temperature, probs)); }
        else { DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((softmax_rows_vec_kernel<TT, 8>), dim3(rows), dim3(1024), 0, st, (const St<TT>::S*)logits, ld, V, 1.f / temperature, probs)); }
    } else {
        DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(softmax_rows_kernel<TT>, dim3(rows), dim3(256), 0, st, (const St<TT>::S*)logits, ld, V, 1.f / temperature, probs));
    }
    return check_launch("lade_softmax_rows");
}

extern "C" int lade_softmax_gather(const void* logits, int64_t ld, int32_t rows, int32_t V, int32_t dtype, float temperature, int32_t skip,
                                   const int32_t* guess, const int32_t* g_dev, int32_t g, int32_t gs, int32_t g_cap, float* scal,
                                   float* stats, void* stream) {
    LADE_REQUIRE(logits && scal && stats && rows >= 1 && V > 0 && ld >= V && temperature > 0.f && skip >= 0 && gs > 0 && g_cap >= 0 && g >= 0 && g <= g_cap &&
                     (g_cap == 0 || guess), LADE_E_ARG, "lade_softmax_gather: rows=%d V=%d T=%f g=%d gs=%d g_cap=%d", rows, V, temperature, g, gs, g_cap);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(softmax_gather_kernel<TT>, dim3(rows), dim3(1024), 0, st, (const St<TT>::S*)logits, ld, V, 1.f / temperature, skip,
                                             guess, g_dev, g, gs, g_cap, scal, stats));
    return check_launch("lade_softmax_gather");
}

// split-K aware variants: the GEMM output arrives as n_parts fp32 partials [n_parts][rows][width]
extern "C" int lade_add_rmsnorm_parts(void* x, const float* parts, int32_t n_parts, int64_t part_stride, const void* weight, void* y,
                                      int32_t rows, int32_t hidden, float eps, int32_t dtype, void* stream) {
    LADE_REQUIRE(x && parts && weight && y && rows >= 0 && hidden > 0 && n_parts >= 1, LADE_E_ARG, "lade_add_rmsnorm_parts: bad args");
    LADE_REQUIRE(dtype != LADE_F32, LADE_E_DTYPE, "lade_add_rmsnorm_parts: 16-bit dtypes only");
    if (rows == 0) return LADE_OK;
    return launch_rmsnorm<true>(x, nullptr, weight, y, rows, hidden, eps, dtype, (hipStream_t)stream, parts, n_parts, (size_t)part_stride);
}

extern "C" int lade_silu_mul_parts(const float* parts, int32_t n_parts, int64_t part_stride, void* out, int32_t rows, int32_t inter, int32_t layout,
                                   int32_t dtype, void* stream) {
    LADE_REQUIRE(parts && out && rows >= 0 && inter > 0 && inter % 8 == 0 && n_parts >= 1 && (layout == 0 || (layout == 1 && inter % 16 == 0)), LADE_E_ARG,
                 "lade_silu_mul_parts: bad args");
    LADE_REQUIRE(dtype != LADE_F32, LADE_E_DTYPE, "lade_silu_mul_parts: 16-bit dtypes only");
    if (rows == 0) return LADE_OK;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(silu_mul_kernel<TT>, dim3(cdiv(inter / 8, 256), rows), dim3(256), 0, st, (const St<TT>::S*)nullptr, (St<TT>::S*)out, inter,
                                             parts, n_parts, (size_t)part_stride, layout));
    return check_launch("lade_silu_mul_parts");
}
