// Fused lookahead-branch + verification-branch attention for gfx950 (MI355X).
//
// Replaces the reference's eager op (lade/models/modeling_llama.py:520-541 under the dense mask of
// :115-207) and its out-of-tree flash_attn_lade kernel (:705-713).  One launch covers the P cached
// keys and the T new tokens of the step; the 2-D (W x N + G) mask is never materialised: every lane
// owns one query row and derives a 64-bit visibility word per 64-key tile from the closed form.
//
// Shape of the problem: T <= ~256 query tokens against P+T keys, per KV head.  It is HBM-bound
// (arithmetic intensity ~ T*H/Hkv flop/B, below the gfx950 ridge for MHA), so the kernel is a
// split-KV streaming kernel:  grid = (row blocks of 128, KV heads, KV splits), 4 waves per block,
// each wave owns 32 query rows, all four share the K / V^T tiles staged in LDS.
//
// Everything is computed "transposed" so that a lane owns ONE query row end to end:
//     S^T[key][q] = K[key][:] . Q[q][:]           A = K tile (LDS),   B = Q^T (registers)
//     O^T[d][q]   = V^T[d][key] . P^T[key][q]     A = V^T tile (LDS), B = P^T (registers, from S^T)
// With v_mfma_f32_32x32x16 the C layout is col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5): the
// softmax row reduction is 31 in-register max/adds plus one exchange with lane^32, the rescale of
// O^T is lane-local, and the P^T B-operand is taken straight from the S^T accumulators (the k-index
// permutation this implies is applied to the V^T A-operand addresses instead of shuffling data).
// V is kept TRANSPOSED in HBM ([Hkv][d][S_max]) so its fragments are contiguous along keys.
#include "common.hpp"

namespace lade {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int KT = 64;        // keys per tile
constexpr int ROWS_PER_WG = 128;
constexpr float NEG_BIG = -1.0e30f;

struct AttnK {
    const uint16_t* q;
    const uint16_t* k;
    const uint16_t* vt;
    uint16_t* out;
    float* part_o;
    float* part_ml;
    const int32_t* dyn_P;
    int64_t q_row_stride, out_row_stride;
    int H, Hkv, S_max, n_splits;
    float scale_log2;   // scale * log2(e)
    lade_mask_params m;
};

// ---- mask predicate --------------------------------------------------------------------
// Row descriptor derived once per lane from the closed form.
struct RowDesc {
    int kind;   // 0 causal (c <= t), 1 level row, 2 candidate row, 3 no row
    int t;      // row index in the new-token block
    int a, i, ll, cbase, pos;
};

__device__ __forceinline__ RowDesc make_row(int t, bool valid, const lade_mask_params& m) {
    RowDesc r;
    r.t = t; r.a = 0; r.i = 0; r.ll = 0; r.cbase = 0; r.pos = 0;
    const int A = m.level_offset + m.dist_offset;
    if (!valid) { r.kind = 3; return r; }
    if (m.is_prefill || t < A) { r.kind = 0; return r; }
    if (t >= m.T - m.lguess) {
        const int k = t - (m.T - m.lguess);
        const int cand = k / m.gs;
        r.kind = 2; r.pos = k - cand * m.gs; r.cbase = m.T - m.lguess + cand * m.gs;
        return r;
    }
    r.kind = 1; r.a = A;
    r.ll = (t - A) / m.s;
    r.i = (t - A) - r.ll * m.s;
    return r;
}

// bits b (0..63) with lo <= c0+b <= hi
__device__ __forceinline__ uint64_t range_bits(int c0, int lo, int hi) {
    int b0 = lo - c0, b1 = hi - c0;
    if (b0 < 0) b0 = 0;
    if (b1 > 63) b1 = 63;
    if (b1 < b0) return 0ull;
    const uint64_t upto = (b1 == 63) ? ~0ull : ((1ull << (b1 + 1)) - 1ull);
    return upto & ~((1ull << b0) - 1ull);
}

// visibility of keys c0 .. c0+63 (c = key - P, negative = cached key) for one row
__device__ __forceinline__ uint64_t vis_bits(int c0, const RowDesc& r, const lade_mask_params& m) {
    constexpr int NEG = -(1 << 30);
    uint64_t v;
    if (r.kind == 0) {
        v = range_bits(c0, NEG, r.t);
    } else if (r.kind == 1) {
        v = range_bits(c0, NEG, r.a + r.i);                       // cols < A, block-0 prefix j <= i
        for (int rr = 1; rr <= r.ll; ++rr) {                       // own column of blocks 1..ll
            const int c = r.a + r.i + rr * m.s;
            v |= range_bits(c0, c, c);
        }
    } else if (r.kind == 2) {
        v = range_bits(c0, NEG, m.level_offset) | range_bits(c0, r.cbase, r.cbase + r.pos);
    } else {
        v = 0ull;
    }
    return v & range_bits(c0, NEG, m.T - 1);                       // keys beyond P+T do not exist
}

__global__ void mask_render_kernel(lade_mask_params m, uint8_t* out) {
    const int t = blockIdx.x;
    const RowDesc r = make_row(t, true, m);
    const int S = m.P + m.T;
    for (int k0 = 0; k0 < S; k0 += 64) {
        const uint64_t v = vis_bits(k0 - m.P, r, m);
        for (int b = threadIdx.x; b < 64 && k0 + b < S; b += blockDim.x) out[(size_t)t * S + k0 + b] = (v >> b) & 1;
    }
}

// ---- LDS addressing (XOR swizzles keep ds_read_b128 / ds_read_b64 conflict-free) ---------
// K tile [KT keys][D]: 16-byte chunk c16 of row `row`.
template <int D>
__device__ __forceinline__ int k_lds_off(int row, int c16) {
    constexpr int CPR = D / 8;                       // 16-B chunks per row
    constexpr int ROWS_PER_BANKROW = 256 / (2 * D) > 0 ? 256 / (2 * D) : 1;
    const int swz = (row / ROWS_PER_BANKROW) & (CPR - 1);
    return row * (2 * D) + ((c16 ^ swz) << 4);
}
// V^T tile [D][KT keys]: 8-byte chunk c8 (4 keys) of row `row`; rows are 128 B.
__device__ __forceinline__ int vt_lds_off(int row, int c8) {
    const int swz = (row >> 1) & 15;
    return row * (2 * KT) + ((c8 ^ swz) << 3);
}

template <typename T> struct Mfma;
template <> struct Mfma<BF16> {
    __device__ static __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mfma<F16> {
    __device__ static __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

template <typename T, int D>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnK a) {
    constexpr int KCH = D / 8;            // 16-B chunks per K row
    constexpr int K_CHUNKS = KT * KCH;    // per tile
    constexpr int V_CHUNKS = D * (KT / 8);
    constexpr int K_PER_THR = K_CHUNKS / 256;
    constexpr int V_PER_THR = V_CHUNKS / 256;
    constexpr int KSTEPS = D / 16;        // MFMA k-steps of S^T
    constexpr int DBLK = D / 32;          // 32-row blocks of O^T

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* k_lds = smem;
    unsigned char* vt_lds = smem + KT * D * 2;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int ql = lane & 31, hi = lane >> 5;
    const int kvh = blockIdx.y, sp = blockIdx.z;
    const int n_rep = a.H / a.Hkv;

    lade_mask_params m = a.m;
    if (a.dyn_P) m.P = *a.dyn_P;
    const int S_tot = m.P + m.T;
    const int n_tiles = (S_tot + KT - 1) / KT;
    const int tps = (n_tiles + a.n_splits - 1) / a.n_splits;
    const int tile0 = sp * tps;
    const int tile1 = min(tile0 + tps, n_tiles);

    // this lane's query row
    const int r = blockIdx.x * ROWS_PER_WG + wave * 32 + ql;
    const bool valid = r < n_rep * m.T;
    const int hg = valid ? r / m.T : 0;
    const int t = valid ? r - hg * m.T : 0;
    const int qh = kvh * n_rep + hg;
    const RowDesc rd = make_row(t, valid, m);

    // Q^T B-operand fragments: Q[t][qh][kk*16 + hi*8 .. +8]
    u32x4 qf[KSTEPS];
    {
        const uint16_t* qp = a.q + (size_t)t * a.q_row_stride + (size_t)qh * D + hi * 8;
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            if (valid) qf[kk] = *reinterpret_cast<const u32x4*>(qp + kk * 16);
            else qf[kk] = u32x4{0, 0, 0, 0};
        }
    }

    f32x16 oacc[DBLK];
#pragma unroll
    for (int i = 0; i < DBLK; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) oacc[i][j] = 0.f;
    float m_run = NEG_BIG, l_run = 0.f;

    const uint16_t* kbase = a.k + (size_t)kvh * a.S_max * D;
    const uint16_t* vbase = a.vt + (size_t)kvh * D * a.S_max;

    u32x4 kreg[K_PER_THR], vreg[V_PER_THR];
    auto load_tile = [&](int tile) {
        const int k0 = tile * KT;
#pragma unroll
        for (int i = 0; i < K_PER_THR; ++i) {
            const int ch = tid + i * 256;
            const int row = ch / KCH, c16 = ch % KCH;
            kreg[i] = *reinterpret_cast<const u32x4*>(kbase + (size_t)(k0 + row) * D + c16 * 8);
        }
#pragma unroll
        for (int i = 0; i < V_PER_THR; ++i) {
            const int ch = tid + i * 256;
            const int row = ch / (KT / 8), c = ch % (KT / 8);
            u32x4 v = *reinterpret_cast<const u32x4*>(vbase + (size_t)row * a.S_max + k0 + c * 8);
            // keys >= P+T hold stale bytes: zero them so 0 * garbage can never make a NaN
            const int kfirst = k0 + c * 8;
            if (kfirst + 8 > S_tot) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    uint32_t w = v[e];
                    if (kfirst + 2 * e >= S_tot) w &= 0xffff0000u;
                    if (kfirst + 2 * e + 1 >= S_tot) w &= 0x0000ffffu;
                    v[e] = w;
                }
            }
            vreg[i] = v;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < K_PER_THR; ++i) {
            const int ch = tid + i * 256;
            const int row = ch / KCH, c16 = ch % KCH;
            *reinterpret_cast<u32x4*>(k_lds + k_lds_off<D>(row, c16)) = kreg[i];
        }
#pragma unroll
        for (int i = 0; i < V_PER_THR; ++i) {
            const int ch = tid + i * 256;
            const int row = ch / (KT / 8), c = ch % (KT / 8);
            *reinterpret_cast<u32x2*>(vt_lds + vt_lds_off(row, 2 * c)) = u32x2{vreg[i][0], vreg[i][1]};
            *reinterpret_cast<u32x2*>(vt_lds + vt_lds_off(row, 2 * c + 1)) = u32x2{vreg[i][2], vreg[i][3]};
        }
    };

    if (tile0 < tile1) load_tile(tile0);
    for (int tile = tile0; tile < tile1; ++tile) {
        __syncthreads();          // previous tile's LDS reads are done
        store_tile();
        __syncthreads();
        if (tile + 1 < tile1) load_tile(tile + 1);   // in flight while this tile is computed

        const int k0 = tile * KT;
        const bool full = (k0 + KT <= m.P);           // whole tile in the cache: every key visible
        uint64_t vis = ~0ull;
        if (!full) vis = vis_bits(k0 - m.P, rd, m);
        else if (!valid) vis = 0ull;
        if (__builtin_amdgcn_ballot_w64(vis != 0ull) == 0ull) continue;   // nothing for this wave here

        // ---- S^T = K Q^T : two 32-key sub-tiles ----
        f32x16 sacc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int e = 0; e < 16; ++e) sacc[j][e] = 0.f;
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                const u32x4 kf = *reinterpret_cast<const u32x4*>(k_lds + k_lds_off<D>(32 * j + ql, kk * 2 + hi));
                sacc[j] = Mfma<T>::run(kf, qf[kk], sacc[j]);
            }
        }
        // ---- online softmax (log2 domain), masked entries contribute exactly 0 ----
        float tmax = NEG_BIG;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int kb = 32 * j + (e & 3) + 8 * (e >> 2) + 4 * hi;
                const float s = ((vis >> kb) & 1ull) ? sacc[j][e] * a.scale_log2 : NEG_BIG;
                sacc[j][e] = s;
                tmax = fmaxf(tmax, s);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float s = sacc[j][e];
                const float p = (s > 0.5f * NEG_BIG) ? exp2f(s - m_new) : 0.f;
                sacc[j][e] = p;
                psum += p;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int i = 0; i < DBLK; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[i][e] *= alpha;

        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                u32x4 pf;
#pragma unroll
                for (int e = 0; e < 4; ++e) pf[e] = pack2<T>(sacc[j][8 * jj + 2 * e], sacc[j][8 * jj + 2 * e + 1]);
                const int c8 = 8 * j + 4 * jj + hi;   // keys 32j+16jj+4hi+{0..3}, and +8
#pragma unroll
                for (int db = 0; db < DBLK; ++db) {
                    const int row = db * 32 + ql;
                    const u32x2 v0 = *reinterpret_cast<const u32x2*>(vt_lds + vt_lds_off(row, c8));
                    const u32x2 v1 = *reinterpret_cast<const u32x2*>(vt_lds + vt_lds_off(row, c8 + 2));
                    oacc[db] = Mfma<T>::run(u32x4{v0[0], v0[1], v1[0], v1[1]}, pf, oacc[db]);
                }
            }
    }

    // ---- epilogue ----
    l_run += __shfl_xor(l_run, 32);
    if (!valid) return;
    if (a.n_splits == 1) {
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        uint16_t* op = a.out + (size_t)t * a.out_row_stride + (size_t)qh * D;
#pragma unroll
        for (int db = 0; db < DBLK; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d0 = db * 32 + 8 * g4 + 4 * hi;
                u32x2 w;
                w[0] = pack2<T>(oacc[db][4 * g4 + 0] * inv, oacc[db][4 * g4 + 1] * inv);
                w[1] = pack2<T>(oacc[db][4 * g4 + 2] * inv, oacc[db][4 * g4 + 3] * inv);
                *reinterpret_cast<u32x2*>(op + d0) = w;
            }
    } else {
        const size_t prow = ((size_t)sp * a.H + qh) * m.T + t;
        float* po = a.part_o + prow * D;
#pragma unroll
        for (int db = 0; db < DBLK; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d0 = db * 32 + 8 * g4 + 4 * hi;
                *reinterpret_cast<float4*>(po + d0) =
                    float4{oacc[db][4 * g4 + 0], oacc[db][4 * g4 + 1], oacc[db][4 * g4 + 2], oacc[db][4 * g4 + 3]};
            }
        if (hi == 0) {
            a.part_ml[prow * 2 + 0] = m_run;
            a.part_ml[prow * 2 + 1] = l_run;
        }
    }
}

// merges split-KV partials: out = sum_s 2^(m_s-m) o_s / sum_s 2^(m_s-m) l_s
template <typename T>
__global__ void attn_combine_kernel(AttnK a, int D) {
    const int t = blockIdx.x, qh = blockIdx.y;
    const int Tn = a.m.T;
    float mx = NEG_BIG;
    for (int s = 0; s < a.n_splits; ++s) mx = fmaxf(mx, a.part_ml[(((size_t)s * a.H + qh) * Tn + t) * 2]);
    for (int d0 = threadIdx.x * 4; d0 < D; d0 += blockDim.x * 4) {
        float4 acc = float4{0.f, 0.f, 0.f, 0.f};
        float l = 0.f;
        for (int s = 0; s < a.n_splits; ++s) {
            const size_t prow = ((size_t)s * a.H + qh) * Tn + t;
            const float w = exp2f(a.part_ml[prow * 2] - mx);
            const float ls = a.part_ml[prow * 2 + 1];
            if (ls > 0.f) {
                const float4 o = *reinterpret_cast<const float4*>(a.part_o + prow * D + d0);
                acc.x += w * o.x; acc.y += w * o.y; acc.z += w * o.z; acc.w += w * o.w;
                l += w * ls;
            }
        }
        const float inv = l > 0.f ? 1.f / l : 0.f;
        u32x2 wv;
        wv[0] = pack2<T>(acc.x * inv, acc.y * inv);
        wv[1] = pack2<T>(acc.z * inv, acc.w * inv);
        *reinterpret_cast<u32x2*>(a.out + (size_t)t * a.out_row_stride + (size_t)qh * D + d0) = wv;
    }
}

// ---- fp32 path: plain VALU kernel (parity / tiny models; not a BASELINE dtype) -------------
// one wave per (query row, head); lanes stride over keys, then over d.
struct AttnF32 {
    const float* q; const float* k; const float* vt; float* out;
    const int32_t* dyn_P;
    int64_t q_row_stride, out_row_stride;
    int H, Hkv, d, S_max;
    float scale;
    lade_mask_params m;
};

__global__ __launch_bounds__(64) void attn_fwd_f32_kernel(AttnF32 a) {
    extern __shared__ float sm[];           // [S_tot] probabilities
    const int t = blockIdx.x, qh = blockIdx.y, lane = threadIdx.x;
    lade_mask_params m = a.m;
    if (a.dyn_P) m.P = *a.dyn_P;
    const int S_tot = m.P + m.T;
    const int kvh = qh / (a.H / a.Hkv);
    const RowDesc rd = make_row(t, true, m);
    const float* qp = a.q + (size_t)t * a.q_row_stride + (size_t)qh * a.d;
    const float* kb = a.k + (size_t)kvh * a.S_max * a.d;
    const float* vb = a.vt + (size_t)kvh * a.d * a.S_max;
    float mx = NEG_BIG;
    for (int k0 = 0; k0 < S_tot; k0 += 64) {
        const uint64_t vis = vis_bits(k0 - m.P, rd, m);
        const int key = k0 + lane;
        float s = NEG_BIG;
        if (key < S_tot && ((vis >> lane) & 1ull)) {
            float acc = 0.f;
            for (int dd = 0; dd < a.d; ++dd) acc = fmaf(qp[dd], kb[(size_t)key * a.d + dd], acc);
            s = acc * a.scale;
        }
        if (key < S_tot) sm[key] = s;
        mx = fmaxf(mx, s);
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    __syncthreads();
    float l = 0.f;
    for (int key = lane; key < S_tot; key += 64) {
        const float s = sm[key];
        const float p = (s > 0.5f * NEG_BIG) ? expf(s - mx) : 0.f;
        sm[key] = p;
        l += p;
    }
    for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o);
    __syncthreads();
    const float inv = l > 0.f ? 1.f / l : 0.f;
    for (int dd = lane; dd < a.d; dd += 64) {
        float acc = 0.f;
        const float* vr = vb + (size_t)dd * a.S_max;
        for (int key = 0; key < S_tot; ++key) acc = fmaf(sm[key], vr[key], acc);
        a.out[(size_t)t * a.out_row_stride + (size_t)qh * a.d + dd] = acc * inv;
    }
}

static int validate(const lade_attn_args* a) {
    LADE_REQUIRE(a != nullptr, LADE_E_ARG, "lade_attn: null args");
    LADE_REQUIRE(a->q && a->k_cache && a->vt_cache && a->out, LADE_E_ARG, "lade_attn: null tensor pointer");
    LADE_REQUIRE(a->H > 0 && a->Hkv > 0 && a->H % a->Hkv == 0, LADE_E_ARG, "lade_attn: H=%d Hkv=%d", a->H, a->Hkv);
    LADE_REQUIRE(a->mask.T > 0 && a->mask.P >= 0, LADE_E_ARG, "lade_attn: T=%d P=%d", a->mask.T, a->mask.P);
    LADE_REQUIRE(a->S_max % 64 == 0 && a->mask.P + a->mask.T <= a->S_max, LADE_E_ARG,
                 "lade_attn: S_max=%d must be a multiple of 64 and >= P+T=%d", a->S_max, a->mask.P + a->mask.T);
    LADE_REQUIRE(a->n_splits >= 1, LADE_E_ARG, "lade_attn: n_splits=%d", a->n_splits);
    LADE_REQUIRE(a->n_splits == 1 || (a->part_o && a->part_ml), LADE_E_ARG, "lade_attn: split-KV needs partial buffers");
    if (!a->mask.is_prefill) {
        const lade_mask_params& m = a->mask;
        LADE_REQUIRE(m.s > 0 && m.gs > 0 && m.lguess >= 0 && m.lguess % m.gs == 0 && m.level_offset >= 0 && m.dist_offset >= 0,
                     LADE_E_ARG, "lade_attn: bad mask params s=%d gs=%d lguess=%d lo=%d do=%d", m.s, m.gs, m.lguess,
                     m.level_offset, m.dist_offset);
        const int body = m.T - m.lguess - (m.level_offset + m.dist_offset);
        LADE_REQUIRE(body >= 0 && body % m.s == 0, LADE_E_ARG,
                     "lade_attn: T=%d is not offsets(%d)+k*s(%d)+lguess(%d)", m.T, m.level_offset + m.dist_offset, m.s, m.lguess);
    }
    return LADE_OK;
}

static AttnK make_k(const lade_attn_args* a) {
    AttnK k;
    k.q = (const uint16_t*)a->q; k.k = (const uint16_t*)a->k_cache; k.vt = (const uint16_t*)a->vt_cache;
    k.out = (uint16_t*)a->out; k.part_o = a->part_o; k.part_ml = a->part_ml; k.dyn_P = a->dyn_P;
    k.q_row_stride = a->q_row_stride; k.out_row_stride = a->out_row_stride;
    k.H = a->H; k.Hkv = a->Hkv; k.S_max = a->S_max; k.n_splits = a->n_splits;
    k.scale_log2 = a->scale * 1.4426950408889634f;
    k.m = a->mask;
    return k;
}

template <typename T, int D>
static int launch_fwd(const lade_attn_args* a, hipStream_t st) {
    const AttnK k = make_k(a);
    const int n_rep = a->H / a->Hkv;
    dim3 grid(cdiv(n_rep * a->mask.T, ROWS_PER_WG), a->Hkv, a->n_splits);
    const size_t lds = (size_t)KT * D * 2 * 2;
    hipLaunchKernelGGL((attn_fwd_kernel<T, D>), grid, dim3(256), lds, st, k);
    return check_launch("lade_attn_fwd");
}

}  // namespace lade

using namespace lade;

extern "C" int lade_attn_fwd(const lade_attn_args* a, void* stream) {
    int rc = validate(a);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (a->dtype == LADE_F32) {
        LADE_REQUIRE(a->d > 0 && a->d <= 256, LADE_E_DTYPE, "lade_attn_fwd(f32): d=%d", a->d);
        LADE_REQUIRE((size_t)a->S_max * 4 <= 160 * 1024, LADE_E_LIMIT, "lade_attn_fwd(f32): S_max=%d exceeds the LDS row", a->S_max);
        AttnF32 k;
        k.q = (const float*)a->q; k.k = (const float*)a->k_cache; k.vt = (const float*)a->vt_cache; k.out = (float*)a->out;
        k.dyn_P = a->dyn_P; k.q_row_stride = a->q_row_stride; k.out_row_stride = a->out_row_stride;
        k.H = a->H; k.Hkv = a->Hkv; k.d = a->d; k.S_max = a->S_max; k.scale = a->scale; k.m = a->mask;
        const size_t lds = (size_t)(a->dyn_P ? a->S_max : a->mask.P + a->mask.T) * 4;
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void*)attn_fwd_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(attn_fwd_f32_kernel, dim3(a->mask.T, a->H), dim3(64), lds, st, k);
        return check_launch("lade_attn_fwd(f32)");
    }
    LADE_REQUIRE(a->dtype == LADE_BF16 || a->dtype == LADE_F16, LADE_E_DTYPE, "lade_attn_fwd: dtype=%d", a->dtype);
    LADE_REQUIRE(a->d == 128 || a->d == 64, LADE_E_DTYPE, "lade_attn_fwd: head_dim %d (MFMA kernel supports 64 and 128)", a->d);
    LADE_REQUIRE(a->q_row_stride % 8 == 0 && a->out_row_stride % 4 == 0, LADE_E_ARG, "lade_attn_fwd: row strides must keep 16-B alignment");
    if (a->dtype == LADE_BF16) return a->d == 128 ? launch_fwd<BF16, 128>(a, st) : launch_fwd<BF16, 64>(a, st);
    return a->d == 128 ? launch_fwd<F16, 128>(a, st) : launch_fwd<F16, 64>(a, st);
}

extern "C" int lade_attn_combine(const lade_attn_args* a, void* stream) {
    int rc = validate(a);
    if (rc) return rc;
    LADE_REQUIRE(a->n_splits > 1, LADE_E_ARG, "lade_attn_combine: n_splits=%d", a->n_splits);
    LADE_REQUIRE(a->dtype == LADE_BF16 || a->dtype == LADE_F16, LADE_E_DTYPE, "lade_attn_combine: dtype=%d", a->dtype);
    const AttnK k = make_k(a);
    dim3 grid(a->mask.T, a->H);
    const int thr = a->d / 4 < 64 ? 64 : a->d / 4;
    if (a->dtype == LADE_BF16) hipLaunchKernelGGL(attn_combine_kernel<BF16>, grid, dim3(thr), 0, (hipStream_t)stream, k, a->d);
    else hipLaunchKernelGGL(attn_combine_kernel<F16>, grid, dim3(thr), 0, (hipStream_t)stream, k, a->d);
    return check_launch("lade_attn_combine");
}

extern "C" int lade_mask_render(const lade_mask_params* m, uint8_t* out, void* stream) {
    LADE_REQUIRE(m && out && m->T > 0, LADE_E_ARG, "lade_mask_render: bad args");
    hipLaunchKernelGGL(mask_render_kernel, dim3(m->T), dim3(64), 0, (hipStream_t)stream, *m, out);
    return check_launch("lade_mask_render");
}
