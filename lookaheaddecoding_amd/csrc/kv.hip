// RoPE with gathered positions fused with the in-place KV append, and the post-accept KV commit.
//
// Reference: apply_rotary_pos_emb + torch.cat (lade/models/modeling_llama.py:321-346, :510-516) and
// the commit loop of lade/decoding.py:1154-1163.  The reference re-copies the whole cache with
// torch.cat in every layer of every step; here the cache is preallocated ([Hkv][S_max][d] keys,
// [Hkv][d][S_max] transposed values) and only the T new rows are written.
#include <type_traits>

#include "common.hpp"
#include <algorithm>

namespace lade {

// One launch does both halves of the append.
//   blocks [0, T*bpt)       : token t (bpt blocks per token).  q rotated in place, rotated k -> K cache row P+t.
//                             A thread owns VEC consecutive pairs (i, i+d/2) of one head: 16-byte loads / stores.
//   then n_vblk blocks      : V transpose.  Block (kv head, 64-token slab, 32-wide d chunk) stages the
//                             slab through LDS and writes V^T[d][P+t] with the token index fastest.
template <typename T>
__global__ __launch_bounds__(256) void rope_kv_append_kernel(typename Elem<T>::S* qkv, const int32_t* positions,
                                                             const typename Elem<T>::S* cos_tab,
                                                             const typename Elem<T>::S* sin_tab,
                                                             typename Elem<T>::S* k_cache, typename Elem<T>::S* vt_cache,
                                                             int T_, int P, const int32_t* dyn_P, int H, int Hkv, int d,
                                                             int S_max, int max_pos, const float* parts, int n_parts,
                                                             size_t part_stride, typename Elem<T>::S* q_out, int bpt) {
    typedef typename Elem<T>::S S;
    constexpr int VEC = 16 / sizeof(S);
    __shared__ S sm[64][32 + 2];
    if (dyn_P) {
        P = *dyn_P;
        if (P + T_ > S_max) return;     // device-side cache length (hipGraph steps): never write K/V rows past the cache
    }
    const int row_w = (H + 2 * Hkv) * d;
    const int tok_blocks = T_ * bpt;                       // bpt blocks share one token row
    if ((int)blockIdx.x < tok_blocks) {
        const int t = blockIdx.x / bpt, sub = blockIdx.x - t * bpt;
        int pos = positions[t];
        pos = pos < 0 ? 0 : (pos >= max_pos ? max_pos - 1 : pos);
        const S* c = cos_tab + (size_t)pos * d;
        const S* s = sin_tab + (size_t)pos * d;
        const int half = d >> 1;
        S* row = qkv + (size_t)t * row_w;
        if (half % VEC == 0) {
            const int vph = half / VEC;                               // vectors per half head
            for (int idx = sub * blockDim.x + threadIdx.x; idx < (H + Hkv) * vph; idx += bpt * blockDim.x) {
                const int h = idx / vph, i = (idx - h * vph) * VEC;
                S* x = row + (size_t)h * d;
                S x1[VEC], x2[VEC], c1[VEC], c2[VEC], s1[VEC], s2[VEC], o1[VEC], o2[VEC];
                // the cos / sin rows depend on the position only: requested before the partials are waited for (one memory latency
                // instead of two)
                *reinterpret_cast<uint4*>(c1) = *reinterpret_cast<const uint4*>(c + i);
                *reinterpret_cast<uint4*>(c2) = *reinterpret_cast<const uint4*>(c + i + half);
                *reinterpret_cast<uint4*>(s1) = *reinterpret_cast<const uint4*>(s + i);
                *reinterpret_cast<uint4*>(s2) = *reinterpret_cast<const uint4*>(s + i + half);
                if (parts) {                       // qkv arrives as split-K fp32 partials: sum, round once to the dtype
                    const size_t e0 = (size_t)t * row_w + (size_t)h * d + i;
                    float a[VEC], b[VEC];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) { a[e] = 0.f; b[e] = 0.f; }
                    // rounds of R partials: every load of a round is in flight before the first add (order 0,1,2,..).  R follows the split
                    // count (wave uniform): with 2 partials a round of 8 would request the last one seven times over
                    auto sum_parts = [&](auto r_c) {
                        constexpr int R = decltype(r_c)::value;
#pragma unroll 1
                        for (int s0 = 0; s0 < n_parts; s0 += R) {
                            float4 va[R][VEC / 4], vb[R][VEC / 4];
#pragma unroll
                            for (int j = 0; j < R; ++j) {
                                const float* pa = parts + (size_t)min(s0 + j, n_parts - 1) * part_stride + e0;
#pragma unroll
                                for (int e = 0; e < VEC / 4; ++e) {
                                    va[j][e] = *reinterpret_cast<const float4*>(pa + 4 * e);
                                    vb[j][e] = *reinterpret_cast<const float4*>(pa + half + 4 * e);
                                }
                            }
#pragma unroll
                            for (int j = 0; j < R; ++j)
                                if (s0 + j < n_parts) {
#pragma unroll
                                    for (int e = 0; e < VEC / 4; ++e) {
                                        a[4 * e] += va[j][e].x; a[4 * e + 1] += va[j][e].y; a[4 * e + 2] += va[j][e].z; a[4 * e + 3] += va[j][e].w;
                                        b[4 * e] += vb[j][e].x; b[4 * e + 1] += vb[j][e].y; b[4 * e + 2] += vb[j][e].z; b[4 * e + 3] += vb[j][e].w;
                                    }
                                }
                        }
                    };
                    if (n_parts <= 2) sum_parts(std::integral_constant<int, 2>{});
                    else if (n_parts <= 4) sum_parts(std::integral_constant<int, 4>{});
                    else sum_parts(std::integral_constant<int, 8>{});
#pragma unroll
                    for (int e = 0; e < VEC; ++e) { x1[e] = Elem<T>::st(a[e]); x2[e] = Elem<T>::st(b[e]); }
                } else {
                    *reinterpret_cast<uint4*>(x1) = *reinterpret_cast<const uint4*>(x + i);
                    *reinterpret_cast<uint4*>(x2) = *reinterpret_cast<const uint4*>(x + i + half);
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float a1 = Elem<T>::ld(x1[e]), a2 = Elem<T>::ld(x2[e]);
                    // q*cos + rotate_half(q)*sin, rotate_half = cat(-x2, x1); one rounding per torch op
                    o1[e] = Elem<T>::st(__fadd_rn(rnd<T>(__fmul_rn(a1, Elem<T>::ld(c1[e]))), rnd<T>(__fmul_rn(-a2, Elem<T>::ld(s1[e])))));
                    o2[e] = Elem<T>::st(__fadd_rn(rnd<T>(__fmul_rn(a2, Elem<T>::ld(c2[e]))), rnd<T>(__fmul_rn(a1, Elem<T>::ld(s2[e])))));
                }
                S* dst = h < H ? (q_out ? q_out + ((size_t)t * H + h) * d : x) : k_cache + ((size_t)(h - H) * S_max + P + t) * d;
                *reinterpret_cast<uint4*>(dst + i) = *reinterpret_cast<const uint4*>(o1);
                *reinterpret_cast<uint4*>(dst + i + half) = *reinterpret_cast<const uint4*>(o2);
            }
        } else {
            for (int idx = sub * blockDim.x + threadIdx.x; idx < (H + Hkv) * half; idx += bpt * blockDim.x) {
                const int h = idx / half, i = idx - h * half;
                S* x = row + (size_t)h * d;
                const float a1 = Elem<T>::ld(x[i]), a2 = Elem<T>::ld(x[i + half]);
                const float o1 = rnd<T>(__fadd_rn(rnd<T>(__fmul_rn(a1, Elem<T>::ld(c[i]))), rnd<T>(__fmul_rn(-a2, Elem<T>::ld(s[i])))));
                const float o2 = rnd<T>(__fadd_rn(rnd<T>(__fmul_rn(a2, Elem<T>::ld(c[i + half]))), rnd<T>(__fmul_rn(a1, Elem<T>::ld(s[i + half])))));
                S* dst = h < H ? x : k_cache + ((size_t)(h - H) * S_max + P + t) * d;
                dst[i] = Elem<T>::st(o1);
                dst[i + half] = Elem<T>::st(o2);
            }
        }
        return;
    }
    // ---- V transpose ----
    const int dch = (d + 31) / 32;
    int b = blockIdx.x - tok_blocks;
    const int dc = b % dch; b /= dch;
    const int kvh = b % Hkv;
    const int t0 = (b / Hkv) * 64;
    const int nt = min(64, T_ - t0);
    const int d0 = dc * 32, nd = min(32, d - d0);
    if (parts && nd == 32) {
        for (int idx = threadIdx.x; idx < 64 * 8; idx += blockDim.x) {       // 8 float4 per 32-wide row
            const int tt = idx >> 3, d4 = (idx & 7) * 4;
            if (tt < nt) {
                const size_t e0 = (size_t)(t0 + tt) * row_w + (size_t)(H + Hkv + kvh) * d + d0 + d4;
                float4 a = float4{0.f, 0.f, 0.f, 0.f};
                auto sum_v = [&](auto r_c) {
                    constexpr int R = decltype(r_c)::value;
                    for (int s0 = 0; s0 < n_parts; s0 += R) {
                        float4 pv[R];
#pragma unroll
                        for (int j = 0; j < R; ++j) pv[j] = *reinterpret_cast<const float4*>(parts + (size_t)min(s0 + j, n_parts - 1) * part_stride + e0);
#pragma unroll
                        for (int j = 0; j < R; ++j)
                            if (s0 + j < n_parts) { a.x += pv[j].x; a.y += pv[j].y; a.z += pv[j].z; a.w += pv[j].w; }
                    }
                };
                if (n_parts <= 2) sum_v(std::integral_constant<int, 2>{});
                else if (n_parts <= 4) sum_v(std::integral_constant<int, 4>{});
                else sum_v(std::integral_constant<int, 8>{});
                sm[tt][d4] = Elem<T>::st(a.x); sm[tt][d4 + 1] = Elem<T>::st(a.y); sm[tt][d4 + 2] = Elem<T>::st(a.z); sm[tt][d4 + 3] = Elem<T>::st(a.w);
            }
        }
    } else {
        for (int idx = threadIdx.x; idx < 64 * 32; idx += blockDim.x) {
            const int tt = idx >> 5, dd = idx & 31;
            if (tt < nt && dd < nd) {
                const size_t e0 = (size_t)(t0 + tt) * row_w + (size_t)(H + Hkv + kvh) * d + d0 + dd;
                if (parts) {
                    float a = 0.f;
                    for (int sp = 0; sp < n_parts; ++sp) a += parts[sp * part_stride + e0];
                    sm[tt][dd] = Elem<T>::st(a);
                } else {
                    sm[tt][dd] = qkv[e0];
                }
            }
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 32 * 64; idx += blockDim.x) {
        const int dd = idx >> 6, tt = idx & 63;
        if (tt < nt && dd < nd) vt_cache[((size_t)kvh * d + d0 + dd) * S_max + P + t0 + tt] = sm[tt][dd];
    }
}

// grid (L, Hkv): copy cnt rows src.. -> dst.. of K ([S_max][d]) and V^T ([d][S_max])
template <typename S>
__global__ __launch_bounds__(256) void kv_commit_kernel(S* cache, int64_t layer_stride, int64_t v_offset, int Hkv, int d,
                                                        int S_max, int src, int dst, int cnt, const int32_t* ctl) {
    if (ctl) { src = ctl[LADE_CTL_KV_SRC]; dst = ctl[LADE_CTL_KV_DST]; cnt = ctl[LADE_CTL_KV_CNT]; }
    if (cnt <= 0) return;
    S* kc = cache + (size_t)blockIdx.x * layer_stride + (size_t)blockIdx.y * S_max * d;
    S* vc = cache + (size_t)blockIdx.x * layer_stride + v_offset + (size_t)blockIdx.y * d * S_max;
    for (int idx = threadIdx.x; idx < cnt * d; idx += blockDim.x) {
        const int j = idx / d, dd = idx - j * d;
        kc[(size_t)(dst + j) * d + dd] = kc[(size_t)(src + j) * d + dd];
    }
    for (int idx = threadIdx.x; idx < cnt * d; idx += blockDim.x) {
        const int dd = idx / cnt, j = idx - dd * cnt;
        vc[(size_t)dd * S_max + dst + j] = vc[(size_t)dd * S_max + src + j];
    }
}

// K / V as the reference hands them to its flash kernel - [S][Hkv][d], token-major (modeling_llama.py:705-713 after the
// transposes of :636-638) - re-laid into this library's cache layout: K [Hkv][S_max][d], V^T [Hkv][d][S_max].
// grid (ceil(S / 64), Hkv): the block copies its 64 K rows as 16-byte vectors and transposes its 64 x d slab of V through LDS
// so that both sides of the transpose move whole 128-byte runs.
template <typename S>
__global__ __launch_bounds__(256) void kv_pack_bshd_kernel(const S* k, const S* v, S* k_cache, S* vt_cache, int S_tot, int Hkv, int d, int S_max,
                                                           int64_t tok_stride, int64_t head_stride) {
    constexpr int VEC = 16 / sizeof(S);
    extern __shared__ __attribute__((aligned(16))) unsigned char pack_smem[];
    S* sm = reinterpret_cast<S*>(pack_smem);                 // [64][d + VEC] (padded rows)
    const int s0 = blockIdx.x * 64, h = blockIdx.y;
    const int ns = min(64, S_tot - s0);
    const int cpr = d / VEC, ldr = d + VEC;
    for (int idx = threadIdx.x; idx < ns * cpr; idx += blockDim.x) {
        const int r = idx / cpr, c = (idx - r * cpr) * VEC;
        const size_t src = (size_t)(s0 + r) * tok_stride + (size_t)h * head_stride + c;
        *reinterpret_cast<uint4*>(k_cache + ((size_t)h * S_max + s0 + r) * d + c) = *reinterpret_cast<const uint4*>(k + src);
        *reinterpret_cast<uint4*>(sm + r * ldr + c) = *reinterpret_cast<const uint4*>(v + src);
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < d * 64; idx += blockDim.x) {
        const int dd = idx >> 6, r = idx & 63;
        if (r < ns) vt_cache[((size_t)h * d + dd) * S_max + s0 + r] = sm[r * ldr + dd];
    }
}

template <typename T>
static int launch_rope(void* qkv, const int32_t* positions, const void* cos_tab, const void* sin_tab, void* k_cache,
                       void* vt_cache, int T_, int P, const int32_t* dyn_P, int H, int Hkv, int d, int S_max, int max_pos,
                       hipStream_t st, const float* parts = nullptr, int n_parts = 0, size_t part_stride = 0, void* q_out = nullptr) {
    typedef typename Elem<T>::S S;
    const int n_vblk = Hkv * cdiv(T_, 64) * cdiv(d, 32);
    // one 16-byte pair item per thread: blocks per token = items / 256 (latency bound: more, smaller work-groups)
    constexpr int VEC = 16 / sizeof(S);
    const int items = (d / 2) % VEC == 0 ? (H + Hkv) * (d / 2 / VEC) : (H + Hkv) * (d / 2);
    const int bpt = std::max(1, std::min(8, cdiv(items, 256)));
    hipLaunchKernelGGL(rope_kv_append_kernel<T>, dim3(T_ * bpt + n_vblk), dim3(256), 0, st, (S*)qkv, positions, (const S*)cos_tab,
                       (const S*)sin_tab, (S*)k_cache, (S*)vt_cache, T_, P, dyn_P, H, Hkv, d, S_max, max_pos, parts, n_parts, part_stride, (S*)q_out, bpt);
    return check_launch("lade_rope_kv_append");
}

// LlamaDynamicNTKScalingRotaryEmbedding (lade/models/modeling_llama.py:292-318) for one step: the reference rebuilds its cos / sin tables -
// with a base that grows with the sequence - whenever a step's kv_seq_len = P + T (:502-510) exceeds the longest length it has seen
// (`max_seq_len_cached`, starting at max_position_embeddings), and leaves cached K rows with the rotation they were written with.  A step
// only ever gathers the rows of its own T positions, so this kernel writes exactly those T rows (cos_rows / sin_rows [T][d], then indexed
// by the token's row instead of its position) from the device-resident state: state[0] = longest length seen.  inv_tab[i][d/2] = the
// inverse frequencies of a table rebuilt at length mp + i (row 0: the original base), computed on the host by torch exactly as the
// reference computes them (`base ** (arange / dim)` in fp32), so that only the final cos / sin come from this device.
template <typename T>
__global__ __launch_bounds__(128) void rope_rows_dynamic_kernel(const int32_t* positions, int T_, int P, const int32_t* dyn_P, int len_hint,
                                                                int32_t* state, int mp, const float* inv_tab, int n_len, int d,
                                                                typename Elem<T>::S* cos_rows, typename Elem<T>::S* sin_rows,
                                                                const int32_t* g_dev, int gcap, int gs) {
    const int t = blockIdx.x, j = threadIdx.x;
    if (dyn_P) P = *dyn_P;
    int T_real = T_;
    if (g_dev) T_real -= (gcap - min(max(*g_dev, 0), gcap)) * gs;      // a hipGraph step padded to gcap candidates: only g of them exist
    const int seen = *state;
    int eff = max(seen, max(P + T_real, len_hint));          // every block computes the same value: the write-back below is idempotent
    const int idx = min(max(eff - mp, 0), n_len - 1);
    if (j < d / 2) {
        const float ang = (float)positions[t] * inv_tab[(size_t)idx * (d / 2) + j];
        const typename Elem<T>::S c = Elem<T>::st(cosf(ang)), sn = Elem<T>::st(sinf(ang));
        cos_rows[(size_t)t * d + j] = c;
        cos_rows[(size_t)t * d + j + d / 2] = c;
        sin_rows[(size_t)t * d + j] = sn;
        sin_rows[(size_t)t * d + j + d / 2] = sn;
    }
    if (t == 0 && j == 0 && eff != seen) *state = eff;
}

}  // namespace lade

using namespace lade;

extern "C" int lade_rope_rows_dynamic(const int32_t* positions, int32_t T, int32_t P, const int32_t* dyn_P, int32_t len_hint, int32_t* state,
                                      int32_t max_position_embeddings, const float* inv_tab, int32_t n_len, int32_t d, void* cos_rows,
                                      void* sin_rows, int32_t dtype, const int32_t* g_dev, int32_t gcap, int32_t gs, void* stream) {
    LADE_REQUIRE(positions && state && inv_tab && cos_rows && sin_rows && T > 0 && P >= 0 && n_len > 0 && max_position_embeddings > 0 && d > 0 && d % 2 == 0 &&
                     d <= 256, LADE_E_ARG, "lade_rope_rows_dynamic: T=%d P=%d n_len=%d d=%d", T, P, n_len, d);
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case LADE_BF16: hipLaunchKernelGGL(rope_rows_dynamic_kernel<BF16>, dim3(T), dim3(128), 0, st, positions, T, P, dyn_P, len_hint, state, max_position_embeddings, inv_tab, n_len, d, (uint16_t*)cos_rows, (uint16_t*)sin_rows, g_dev, gcap, gs); break;
        case LADE_F16: hipLaunchKernelGGL(rope_rows_dynamic_kernel<F16>, dim3(T), dim3(128), 0, st, positions, T, P, dyn_P, len_hint, state, max_position_embeddings, inv_tab, n_len, d, (uint16_t*)cos_rows, (uint16_t*)sin_rows, g_dev, gcap, gs); break;
        case LADE_F32: hipLaunchKernelGGL(rope_rows_dynamic_kernel<F32>, dim3(T), dim3(128), 0, st, positions, T, P, dyn_P, len_hint, state, max_position_embeddings, inv_tab, n_len, d, (float*)cos_rows, (float*)sin_rows, g_dev, gcap, gs); break;
        default: LADE_REQUIRE(false, LADE_E_DTYPE, "lade_rope_rows_dynamic: dtype=%d", dtype);
    }
    return check_launch("lade_rope_rows_dynamic");
}

extern "C" int lade_rope_kv_append(void* qkv, const int32_t* positions, const void* cos_tab, const void* sin_tab,
                                   void* k_cache, void* vt_cache, int32_t T, int32_t P, const int32_t* dyn_P, int32_t H,
                                   int32_t Hkv, int32_t d, int32_t S_max, int32_t max_pos, int32_t dtype, void* stream) {
    LADE_REQUIRE(qkv && positions && cos_tab && sin_tab && k_cache && vt_cache, LADE_E_ARG, "lade_rope_kv_append: null pointer");
    LADE_REQUIRE(T > 0 && P >= 0 && P + T <= S_max && H > 0 && Hkv > 0 && d > 0 && d % 2 == 0 && d <= 256 && max_pos > 0,
                 LADE_E_ARG, "lade_rope_kv_append: T=%d P=%d S_max=%d H=%d Hkv=%d d=%d", T, P, S_max, H, Hkv, d);
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case LADE_BF16: return launch_rope<BF16>(qkv, positions, cos_tab, sin_tab, k_cache, vt_cache, T, P, dyn_P, H, Hkv, d, S_max, max_pos, st);
        case LADE_F16: return launch_rope<F16>(qkv, positions, cos_tab, sin_tab, k_cache, vt_cache, T, P, dyn_P, H, Hkv, d, S_max, max_pos, st);
        case LADE_F32: return launch_rope<F32>(qkv, positions, cos_tab, sin_tab, k_cache, vt_cache, T, P, dyn_P, H, Hkv, d, S_max, max_pos, st);
    }
    LADE_REQUIRE(false, LADE_E_DTYPE, "lade_rope_kv_append: dtype=%d", dtype);
}

extern "C" int lade_kv_commit(void* cache, int64_t layer_stride, int64_t v_offset, int32_t L, int32_t Hkv, int32_t d,
                              int32_t S_max, int32_t src, int32_t dst, int32_t cnt, const int32_t* ctl, int32_t max_cnt,
                              int32_t elem_bytes, void* stream) {
    LADE_REQUIRE(cache && L > 0 && Hkv > 0 && d > 0 && S_max > 0, LADE_E_ARG, "lade_kv_commit: bad args");
    LADE_REQUIRE(elem_bytes == 2 || elem_bytes == 4, LADE_E_DTYPE, "lade_kv_commit: elem_bytes=%d", elem_bytes);
    if (!ctl) {
        LADE_REQUIRE(cnt >= 0 && src >= 0 && dst >= 0 && src + cnt <= S_max && dst + cnt <= S_max && (cnt == 0 || src >= dst + cnt || dst >= src + cnt),
                     LADE_E_ARG, "lade_kv_commit: src=%d dst=%d cnt=%d S_max=%d", src, dst, cnt, S_max);
        if (cnt == 0) return LADE_OK;
    }
    (void)max_cnt;
    hipStream_t st = (hipStream_t)stream;
    if (elem_bytes == 2)
        hipLaunchKernelGGL(kv_commit_kernel<uint16_t>, dim3(L, Hkv), dim3(256), 0, st, (uint16_t*)cache, layer_stride, v_offset, Hkv, d, S_max, src, dst, cnt, ctl);
    else
        hipLaunchKernelGGL(kv_commit_kernel<uint32_t>, dim3(L, Hkv), dim3(256), 0, st, (uint32_t*)cache, layer_stride, v_offset, Hkv, d, S_max, src, dst, cnt, ctl);
    return check_launch("lade_kv_commit");
}

extern "C" int lade_kv_pack_bshd(const void* k, const void* v, int64_t tok_stride, int64_t head_stride, void* k_cache, void* vt_cache, int32_t S,
                                 int32_t Hkv, int32_t d, int32_t S_max, int32_t elem_bytes, void* stream) {
    LADE_REQUIRE(k && v && k_cache && vt_cache, LADE_E_ARG, "lade_kv_pack_bshd: null pointer");
    LADE_REQUIRE(S > 0 && Hkv > 0 && d > 0 && d <= 256 && S <= S_max, LADE_E_ARG, "lade_kv_pack_bshd: S=%d Hkv=%d d=%d S_max=%d", S, Hkv, d, S_max);
    LADE_REQUIRE(elem_bytes == 2 || elem_bytes == 4, LADE_E_DTYPE, "lade_kv_pack_bshd: elem_bytes=%d", elem_bytes);
    LADE_REQUIRE((d * elem_bytes) % 16 == 0 && (tok_stride * elem_bytes) % 16 == 0 && (head_stride * elem_bytes) % 16 == 0 && tok_stride > 0 && head_stride > 0,
                 LADE_E_ARG, "lade_kv_pack_bshd: head rows and strides must be multiples of 16 bytes (d=%d strides %lld %lld)", d, (long long)tok_stride, (long long)head_stride);
    // the kernel moves 16-byte vectors: a view with an odd storage offset (flash_attn_func hands over whatever the caller sliced) must not reach it
    LADE_REQUIRE(((uintptr_t)k | (uintptr_t)v | (uintptr_t)k_cache | (uintptr_t)vt_cache) % 16 == 0 && ((int64_t)S_max * elem_bytes) % 16 == 0, LADE_E_ARG,
                 "lade_kv_pack_bshd: k, v, k_cache, vt_cache must be 16-byte aligned (%p %p %p %p) and cache rows a multiple of 16 bytes (S_max=%d)", k, v, k_cache, vt_cache, S_max);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)64 * (d + 16 / elem_bytes) * elem_bytes;
    const dim3 grid(cdiv(S, 64), Hkv);
    if (elem_bytes == 2) {
        hipLaunchKernelGGL(kv_pack_bshd_kernel<uint16_t>, grid, dim3(256), lds, st, (const uint16_t*)k, (const uint16_t*)v, (uint16_t*)k_cache, (uint16_t*)vt_cache, S, Hkv, d, S_max, tok_stride, head_stride);
    } else {
        if (lds > 64 * 1024) {
            const hipError_t ae = hipFuncSetAttribute((const void*)kv_pack_bshd_kernel<uint32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            LADE_REQUIRE(ae == hipSuccess, LADE_E_LIMIT, "lade_kv_pack_bshd: %zu bytes of LDS (fp32, d=%d) are more than this device grants a work-group: %s", lds, d, hipGetErrorString(ae));
        }
        hipLaunchKernelGGL(kv_pack_bshd_kernel<uint32_t>, grid, dim3(256), lds, st, (const uint32_t*)k, (const uint32_t*)v, (uint32_t*)k_cache, (uint32_t*)vt_cache, S, Hkv, d, S_max, tok_stride, head_stride);
    }
    return check_launch("lade_kv_pack_bshd");
}

// qkv arrives as n_parts fp32 split-K partials [n_parts][T][(H+2Hkv)*d]; the rotated q goes to q_out [T][H*d]
extern "C" int lade_rope_kv_append_parts(const float* parts, int32_t n_parts, int64_t part_stride, void* q_out, const int32_t* positions,
                                         const void* cos_tab, const void* sin_tab, void* k_cache, void* vt_cache, int32_t T, int32_t P,
                                         const int32_t* dyn_P, int32_t H, int32_t Hkv, int32_t d, int32_t S_max, int32_t max_pos, int32_t dtype,
                                         void* stream) {
    LADE_REQUIRE(parts && q_out && positions && cos_tab && sin_tab && k_cache && vt_cache && n_parts >= 1, LADE_E_ARG, "lade_rope_kv_append_parts: null pointer");
    LADE_REQUIRE(T > 0 && P >= 0 && P + T <= S_max && H > 0 && Hkv > 0 && d > 0 && (d / 2) % 8 == 0 && d <= 256 && max_pos > 0,
                 LADE_E_ARG, "lade_rope_kv_append_parts: T=%d P=%d S_max=%d H=%d Hkv=%d d=%d", T, P, S_max, H, Hkv, d);
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case LADE_BF16: return launch_rope<BF16>(nullptr, positions, cos_tab, sin_tab, k_cache, vt_cache, T, P, dyn_P, H, Hkv, d, S_max, max_pos, st, parts, n_parts, (size_t)part_stride, q_out);
        case LADE_F16: return launch_rope<F16>(nullptr, positions, cos_tab, sin_tab, k_cache, vt_cache, T, P, dyn_P, H, Hkv, d, S_max, max_pos, st, parts, n_parts, (size_t)part_stride, q_out);
    }
    LADE_REQUIRE(false, LADE_E_DTYPE, "lade_rope_kv_append_parts: dtype=%d", dtype);
}
