// Fused lookahead-branch + verification-branch attention for gfx950 (MI355X).
//
// Replaces the reference's eager op (lade/models/modeling_llama.py:520-541 under the dense mask of
// :115-207) and its out-of-tree flash_attn_lade kernel (:705-713).  One launch covers the P cached
// keys and the T new tokens of the step; the 2-D (W x N + G) mask is never materialised: every lane
// owns one query row and derives a 64-bit visibility word per 64-key tile from the closed form.
//
// Shape of the problem: T <= ~256 query tokens against P+T keys, per KV head.  It is HBM-bound
// (arithmetic intensity ~ T*H/Hkv flop/B, below the gfx950 ridge for MHA), so the kernel is a
// split-KV streaming kernel:  grid = (row blocks, KV heads, KV splits); a work-group is RG x KQ waves (default 4 x 2 = 8 waves,
// 128 query rows): wave (rg, kq) owns 32 query rows and a 32-key part of every tile, all share the K / V^T tiles staged in LDS.
//
// Everything is computed "transposed" so that a lane owns ONE query row end to end:
//     S^T[key][q] = K[key][:] . Q[q][:]           A = K tile (LDS),   B = Q^T (registers)
//     O^T[d][q]   = V^T[d][key] . P^T[key][q]     A = V^T tile (LDS), B = P^T (registers, from S^T)
// With v_mfma_f32_32x32x16 the C layout is col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5): the
// softmax row reduction is 31 in-register max/adds plus one exchange with lane^32, the rescale of
// O^T is lane-local, and the P^T B-operand is taken straight from the S^T accumulators (the k-index
// permutation this implies is applied to the V^T A-operand addresses instead of shuffling data).
// V is kept TRANSPOSED in HBM ([Hkv][d][S_max]) so its fragments are contiguous along keys.
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"

// Cache policy of the K / V stream (bits 1 = sc0, 2 = nt, 16 = sc1), a COMPILE-time constant: NON-TEMPORAL + sc1 since round 6.  The cache of a 2 k-token sequence
// is 1 GB per 7B model - far beyond L2 and the Infinity Cache - and every step reads all of it once: at the default policy it only evicts what IS re-read
// (activation tiles, split-K partials, the attention partials the merge reads next).  One library per arm, alternating on one box, the shipped decision table
// in every arm (profiles/r6_cache_policy_variants_ab.txt): attention pair 16.3-16.5 vs 17.4-17.9 us at config 2 (0.283-0.286 vs 0.261-0.268 of 8 TB/s), 22.1-22.7
// vs 23.3-23.9 at config 4; plain decoding 3.27 vs 3.35 ms (7B), 5.41 vs 5.48 (13B); config 2 mid / hot regimes -1.5 %; the cold 60 / 120-row steps equal within
// the noise.  Variants: tools/build_variant.sh NAME "-DLADE_KV_AUX=0" (a run-time switch in the issue path would cost more than the policy gives).
#ifndef LADE_KV_AUX
#define LADE_KV_AUX 18
#endif
// the partial (or final) output rows stored write-through (sc1) instead of plain: a compile-time variant (tools/build_variant.sh NAME "-DLADE_PO_WT=1")
#ifndef LADE_PO_WT
#define LADE_PO_WT 0
#endif

namespace lade {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int KT = 64;        // keys per tile
constexpr float NEG_BIG = -1.0e30f;     // initial running max
constexpr float MASKED = -3.0e30f;      // masked score: 2^(MASKED - m) == 0 even while m is still NEG_BIG

struct AttnK {
    const uint16_t* q;
    const uint16_t* k;
    const uint16_t* vt;
    uint16_t* out;
    uint16_t* part_o;     // [n_splits][T][H][D] normalised partial outputs, model dtype
    float* part_ml;       // [n_splits][H][T][2] (running max in log2 units, running sum)
    const int32_t* dyn_P;
    int64_t q_row_stride, out_row_stride;
    int H, Hkv, S_max, n_splits, n_rep;
    float scale_log2;   // scale * log2(e)
    int dbg;
    // work-group -> (row block, KV head, split) decode of the 1-D grid (see block_decode): magic multipliers for / n_splits and / Hkv
    uint32_t ns_magic, hkv_magic, t_magic;          // ... and / T (row -> head-in-group)
    float inv_T;        // 1 / T, rounded once on the host
    int n_groups;       // row blocks x KV heads
    lade_mask_params m;
    // fused RoPE + KV append (kernels with NPC > 0): q and the step's new K / V rows come from the qkv projection's fp32 split-K partials
    const float* parts;          // [n_parts][T][row_w], part_stride elements apart
    size_t part_stride;
    const int32_t* positions;    // [T] or null (= table row t)
    const uint16_t* cos_tab;     // [max_pos][D]
    const uint16_t* sin_tab;
    uint16_t* k_w;               // the caches again, writable
    uint16_t* vt_w;
    int n_parts, max_pos, row_w;
    // producer mode of the fused form (sync_flags given): the first n_prod work-groups of the grid do lade_rope_kv_append_parts' work ONCE per KV head
    // (32 tokens each) and raise flags[kvh]; the attention work-groups of that head request the cache tiles that hold no new row, then wait for the flag
    int32_t* flags;
    int n_prod, prod_chunks;
};

// ---- 1-D grid, XCD aware ---------------------------------------------------------------------------------------------------------
// Consecutive work-group ids are dealt round-robin to the chip's 8 XCDs (each with its own L2).  Block b = x + 8 (q n_splits + sp) is
// split sp of group g = x + 8 q, group = (row block rb, KV head kvh) with g = rb Hkv + kvh: every split of a group - and, when Hkv is a
// multiple of 8 (Llama-2-70B: 8), every row block of a KV head - has the same b mod 8, i.e. one XCD (tools/xcd_probe: equal b mod 8
// -> equal XCC_ID in every one of 96 000 groups observed): the row blocks of one GQA head then share one L2 copy of its K / V stream.
// (Round 3 also merged the splits of a group INSIDE the launch through that L2 - last arriver, and last-ticket variants; bit-identical
// to the two-launch form and correct under stress, but 5-6 us SLOWER: a device-scope RMW costs ~2 us here and the merger's sc1 re-reads
// ~2 us per round while other work-groups still stream, DESIGN 4.1.  The merge is therefore a second launch, lade_attn_combine.)
__device__ __forceinline__ uint32_t div_magic(uint32_t x, uint32_t magic) { return magic ? __umulhi(x, magic) : x; }     // magic 0: divisor 1

struct BlockId { int rb, kvh, sp, group; bool live; };
__device__ __forceinline__ BlockId block_decode(const AttnK& a) {
    const uint32_t b = blockIdx.x - (uint32_t)a.n_prod;          // (n_prod is a multiple of 8: b mod 8 = the XCD of the block stays what it was)
    const uint32_t x = b & 7u, j = b >> 3;
    const uint32_t q = div_magic(j, a.ns_magic);
    BlockId r;
    r.sp = (int)(j - q * (uint32_t)a.n_splits);
    r.group = (int)(x + 8u * q);
    r.rb = (int)div_magic((uint32_t)r.group, a.hkv_magic);
    r.kvh = r.group - r.rb * a.Hkv;
    r.live = r.group < a.n_groups;
    return r;
}


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
    if (m.layout == 1 && t >= A + m.s) {
        // flash order (modeling_llama.py:1471-1485): after block 0, the levels >= 1 are interleaved column-major
        const int nl = (m.T - m.lguess - A - m.s) / m.s;      // levels >= 1
        const int q = t - (A + m.s);
        r.i = q / nl;
        r.ll = 1 + (q - r.i * nl);
        r.cbase = A + m.s + r.i * nl;                           // this column's keys of levels 1..: contiguous
        return r;
    }
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
        if (m.layout == 1) {
            if (r.ll >= 1) v |= range_bits(c0, r.cbase, r.cbase + r.ll - 1);   // own column, levels 1..ll (adjacent keys)
        } else {
            for (int rr = 1; rr <= r.ll; ++rr) {                   // own column of blocks 1..ll
                const int c = r.a + r.i + rr * m.s;
                v |= range_bits(c0, c, c);
            }
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

// ---- LDS tiles ------------------------------------------------------------------------------
// K tile [KT keys][D] and V^T tile [D][KT keys] are filled by LDS-DMA (global_load_lds, 16 B per
// lane, no VGPR staging).  The DMA writes LDS linearly (wave-uniform base + lane*16), so the XOR
// swizzle that keeps the ds_read_b128 fragment reads conflict-free is applied to the per-lane
// SOURCE address and to the read address (never to the destination).
template <int ROW_BYTES>
__device__ __forceinline__ int swz16(int row) {          // 16-B chunk swizzle of a row-major tile
    constexpr int CPR = ROW_BYTES / 16;                  // chunks per row
    constexpr int RPB = 256 / ROW_BYTES > 0 ? 256 / ROW_BYTES : 1;   // rows per 256-B bank row
    return (row / RPB) & (CPR - 1);
}
template <int ROW_BYTES>
__device__ __forceinline__ int tile_off(int row, int c16) { return row * ROW_BYTES + ((c16 ^ swz16<ROW_BYTES>(row)) << 4); }

// MFMA row index rho (0..31) of an S^T sub-tile <-> key kappa within the sub-tile.  With
// kappa = (rho&3) | ((rho>>3)&3)<<2 | ((rho>>2)&1)<<4 the 16 S^T values a lane holds are the 16
// CONTIGUOUS keys 16*hi .. 16*hi+15, so each PV k-step needs 8 contiguous keys of V^T = one 16-B read.
__device__ __forceinline__ int kappa(int rho) { return (rho & 3) | (((rho >> 3) & 3) << 2) | (((rho >> 2) & 1) << 4); }

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

// ---- merge of split-KV partials (shared by the in-launch merge and lade_attn_combine: identical arithmetic, identical results) ----
// out = sum_s w_s o_s / sum_s w_s with w_s = l_s 2^(m_s - m); o_s = normalised per-split outputs in the model dtype.  One call merges
// 16 bytes (8 values) of one (token, head); splits are taken 8 at a time and ALL loads of a group are requested before the first use.
// LD: loader with  float2 ml(int split)  and  u32x4 po(int split).
template <typename T, typename LD>
__device__ __forceinline__ u32x4 merge_splits(int ns, LD ld) {
    float mx = NEG_BIG, wsum = 0.f;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < ns; s0 += 8) {
        float2 v[8];
        u32x4 o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int sidx = min(s0 + j, ns - 1);
            v[j] = ld.ml(sidx);
            o[j] = ld.po(sidx);
        }
        // pin the (m, l) loads here: left alone, the compiler sinks each of them into the `s0 + j < ns` branch that first uses it,
        // one dependent memory latency after the other (five in a row at 8 splits)
        __builtin_amdgcn_sched_barrier(0);               // ... and keep all sixteen loads in front of the first wait
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(v[j].x), "+v"(v[j].y));
        float gm = mx;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (s0 + j < ns && v[j].y > 0.f) gm = fmaxf(gm, v[j].x);
        const float resc = __builtin_amdgcn_exp2f(mx - gm);      // 2^(NEG_BIG - gm) = 0 on the first group
        mx = gm;
        wsum *= resc;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] *= resc;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float ws = (s0 + j < ns && v[j].y > 0.f) ? v[j].y * __builtin_amdgcn_exp2f(v[j].x - mx) : 0.f;
            wsum += ws;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[2 * e] += ws * to_f32<T>((uint16_t)(o[j][e] & 0xffffu));
                acc[2 * e + 1] += ws * to_f32<T>((uint16_t)(o[j][e] >> 16));
            }
        }
    }
    const float inv = wsum > 0.f ? 1.f / wsum : 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] *= inv;
    u32x4 wv;
#pragma unroll
    for (int e = 0; e < 4; ++e) wv[e] = pack2<T>(acc[2 * e], acc[2 * e + 1]);
    return wv;
}

// plain loads (a separate launch: the partials were written by an earlier kernel)
struct PlainPartials {
    const float* ml_p; size_t ml_stride;           // floats
    const uint16_t* po_p; size_t po_stride;        // elements
    __device__ __forceinline__ float2 ml(int s) const { return *reinterpret_cast<const float2*>(ml_p + s * ml_stride); }
    __device__ __forceinline__ u32x4 po(int s) const { return *reinterpret_cast<const u32x4*>(po_p + s * po_stride); }
};

constexpr float RESCALE_THR = 8.0f;    // log2 units: the running max is only raised when it grows by more

// optional in-kernel timeline (build with -DLADE_ATTN_TIMELINE, run with LADE_DEBUG=attn_dbg=16): thread 0 of every
// work-group stamps s_memtime at 7 points into the words that follow part_ml
#ifdef LADE_ATTN_TIMELINE
__device__ __forceinline__ void dbg_stamp(const AttnK& a, int slot) {
    if ((a.dbg & 16) && threadIdx.x == 0) {
        const size_t base = (size_t)a.n_splits * a.H * a.m.T * 2;
        const int wg = blockIdx.x;
        unsigned long long tnow = __builtin_readcyclecounter();
        reinterpret_cast<unsigned long long*>(a.part_ml + base)[(size_t)wg * 8 + slot] = tnow;
    }
}
#else
__device__ __forceinline__ void dbg_stamp(const AttnK&, int) {}
#endif

// Work-group barrier that the compiler cannot move memory operations across.  The raw
// __builtin_amdgcn_s_barrier() is IntrNoMem: LDS reads that follow it in the source may be scheduled
// BEFORE it (observed: Q fragments read ahead of the barrier that publishes other waves' LDS-DMA).
__device__ __forceinline__ void wg_barrier() { asm volatile("s_barrier" ::: "memory"); }

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---- fused RoPE + KV append ------------------------------------------------------------------------------------------------------
// The qkv projection of a decode step is a split-K GEMM that leaves fp32 partials; round 1-4 summed them, rotated q / k and appended the
// new K / V rows in a launch of their own (lade_rope_kv_append_parts, 6.4-8.3 us per layer inside a step: latency, not bytes).  With
// NPC > 0 the attention work-groups do it themselves - the seam is per head, not all-to-all: a work-group of KV head h needs the q rows
// of its own heads (every split rebuilds them: 30 KB of fp32 per partial at T = 60) and, if its key range reaches into the new rows
// P .. P+T, exactly those rows of K and V^T - which it writes to the cache (for the steps to come) and then streams back like any other
// tile.  No work-group ever waits for another one: row blocks that share a KV head write the same bytes.  The arithmetic is
// lade_rope_kv_append_parts' own, operation for operation (sum in split order from 0, round once, one rounding per torch op of
// apply_rotary_pos_emb, lade/models/modeling_llama.py:321-346): q, the cache rows and therefore the attention output are bit-identical
// to the two-launch form (tests/test_gpu_fused_rope.py).
// One item = 8 + 8 values of one head row: columns i .. i+7 and their rotation partners i + D/2 ..
// (the cos / sin tables of a rotary embedding are cat(freqs, freqs) - lade/models/modeling_llama.py:252 - so ONE 16-byte load per table
// serves columns i .. i+7 and i + D/2 ..: the fused form requires such tables, include/lade_hip.h)
template <int NPC> struct RopeItem { float4 va[NPC][2], vb[NPC][2]; u32x4 c1, s1; };

template <int NPC, int D>
__device__ __forceinline__ void rope_load(RopeItem<NPC>& it, const AttnK& a, size_t e0, int trow, int i) {
#pragma unroll
    for (int j = 0; j < NPC; ++j) {                                       // a partial beyond n_parts re-reads the last one (never added)
        const float* pa = a.parts + (size_t)min(j, a.n_parts - 1) * a.part_stride + e0;
        it.va[j][0] = *reinterpret_cast<const float4*>(pa);
        it.va[j][1] = *reinterpret_cast<const float4*>(pa + 4);
        it.vb[j][0] = *reinterpret_cast<const float4*>(pa + D / 2);
        it.vb[j][1] = *reinterpret_cast<const float4*>(pa + D / 2 + 4);
    }
    const uint16_t* c = a.cos_tab + (size_t)trow * D + i;
    const uint16_t* sn = a.sin_tab + (size_t)trow * D + i;
    it.c1 = *reinterpret_cast<const u32x4*>(c);
    it.s1 = *reinterpret_cast<const u32x4*>(sn);
}

__device__ __forceinline__ uint16_t half_of(const u32x4& v, int e) { return (uint16_t)(v[e >> 1] >> ((e & 1) * 16)); }

template <typename T, int NPC>
__device__ __forceinline__ void rope_apply(const RopeItem<NPC>& it, int n_parts, u32x4& o1, u32x4& o2) {
    float a[8], b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = 0.f; b[e] = 0.f; }
#pragma unroll
    for (int j = 0; j < NPC; ++j)
        if (j < n_parts) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                a[4 * h2] += it.va[j][h2].x; a[4 * h2 + 1] += it.va[j][h2].y; a[4 * h2 + 2] += it.va[j][h2].z; a[4 * h2 + 3] += it.va[j][h2].w;
                b[4 * h2] += it.vb[j][h2].x; b[4 * h2 + 1] += it.vb[j][h2].y; b[4 * h2 + 2] += it.vb[j][h2].z; b[4 * h2 + 3] += it.vb[j][h2].w;
            }
        }
    uint16_t r1[8], r2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float a1 = Elem<T>::ld(Elem<T>::st(a[e])), a2 = Elem<T>::ld(Elem<T>::st(b[e]));
        // q*cos + rotate_half(q)*sin, rotate_half = cat(-x2, x1); one rounding per torch op
        r1[e] = Elem<T>::st(__fadd_rn(rnd<T>(__fmul_rn(a1, Elem<T>::ld(half_of(it.c1, e)))), rnd<T>(__fmul_rn(-a2, Elem<T>::ld(half_of(it.s1, e))))));
        r2[e] = Elem<T>::st(__fadd_rn(rnd<T>(__fmul_rn(a2, Elem<T>::ld(half_of(it.c1, e)))), rnd<T>(__fmul_rn(a1, Elem<T>::ld(half_of(it.s1, e))))));
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        o1[e] = (uint32_t)r1[2 * e] | ((uint32_t)r1[2 * e + 1] << 16);
        o2[e] = (uint32_t)r2[2 * e] | ((uint32_t)r2[2 * e + 1] << 16);
    }
}

// ---- producer mode of the fused form ---------------------------------------------------------------------------------------------------
// The first form above makes every KV split of a head rebuild the head's q rows from fp32 partials (+15.6 MB into the start-of-launch
// burst: measured +3.7 us per layer).  Here lade_rope_kv_append_parts' work is done ONCE - by dedicated work-groups at the head of the
// grid, one per (KV head, 32 tokens) - and handed to the attention work-groups of that head INSIDE the launch: the producer writes the
// rotated q rows (to the caller's q buffer), the K rows and the V^T columns with WRITE-THROUGH stores (sc0 sc1: the bytes leave the XCD's
// L2 as they are issued), drains them (s_waitcnt vmcnt(0) - inline asm, so that the compiler cannot drop it), and bumps flags[kvh]
// (relaxed, device scope).  A consumer polls the flag from ONE lane (relaxed sc1 loads, s_sleep between them, bounded), passes a
// barrier and only then requests q and the tiles that hold new rows - with sc1 loads, the consumer side of the guide's
// {write-through stores / sc1 loads} form (MI355X_MICROARCH.md, inter-workgroup visibility); everything it requested before
// (the tiles of the cache proper) does not depend on the producers.  Dispatch order is not relied upon for correctness: a consumer whose
// flag never comes (the producers queued behind a full chip) gives up after ~0.1 s and the launch ends - wrong results instead of
// a hung GPU; the producers sit at the head of the grid so that in practice they are resident first.  flags[] is reset by
// lade_attn_combine (this form needs n_splits > 1), so the buffer only has to be zero before the first launch.
__device__ __forceinline__ void store_wt_b128(void* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void store_wt_b16(void* p, uint32_t v) { asm volatile("global_store_short %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }

constexpr int PROD_TOK = 32;           // tokens per producer work-group

template <typename T, int D, int NPC, int NTHR>
__device__ __forceinline__ void rope_producer(const AttnK& a, unsigned char* smem) {
    constexpr int VPH = D / 16, LDV = D + 2;
    const int tid = threadIdx.x;
    const int pb = blockIdx.x;
    const int kvh = pb / a.prod_chunks, ch = pb - kvh * a.prod_chunks;
    if (kvh >= a.Hkv) return;                                   // padding of the producer range to a multiple of 8
    int P = a.m.P;
    if (a.dyn_P) P = *a.dyn_P;
    const int t0 = ch * PROD_TOK, nt_ = min(PROD_TOK, a.m.T - t0);
    const int n_rep = a.n_rep;
    const bool writes = P + a.m.T <= a.S_max;                   // (a device-side cache length past the cache: nothing is written, as in lade_rope_kv_append)
    uint16_t* q_out = const_cast<uint16_t*>(a.q);
    {
        // q rows of the group's heads + the K row: one item = 8 + 8 values (columns i.., i + D/2..) of one head row.  The q rows are rotated
        // whatever the cache length (the two-launch form does the same); only the cache stores are skipped past the cache.
        const int per_tok = (n_rep + 1) * VPH;
        for (int idx = tid; idx < nt_ * per_tok; idx += NTHR) {
            const int tt = idx / per_tok, rem = idx - tt * per_tok, hs = rem / VPH, i = (rem - hs * VPH) * 8;
            const int t = t0 + tt;
            int trow = t;
            if (a.positions) { trow = a.positions[t]; trow = trow < 0 ? 0 : (trow >= a.max_pos ? a.max_pos - 1 : trow); }
            if (hs >= n_rep && !writes) continue;
            const size_t col = hs < n_rep ? (size_t)(kvh * n_rep + hs) * D : (size_t)(a.H + kvh) * D;
            RopeItem<NPC> it;
            rope_load<NPC, D>(it, a, (size_t)t * a.row_w + col + i, trow, i);
            u32x4 o1, o2;
            rope_apply<T, NPC>(it, a.n_parts, o1, o2);
            uint16_t* dst = hs < n_rep ? q_out + (size_t)t * a.q_row_stride + (size_t)(kvh * n_rep + hs) * D + i
                                       : a.k_w + ((size_t)kvh * a.S_max + P + t) * D + i;
            store_wt_b128(dst, o1);
            store_wt_b128(dst + D / 2, o2);
        }
    }
    if (writes) {
        // V rows: partials summed in split order, rounded once, transposed through LDS ([token][d], padded rows)
        uint16_t* stage_v = reinterpret_cast<uint16_t*>(smem);
        const size_t v_col = (size_t)(a.H + a.Hkv + kvh) * D;
        for (int idx = tid; idx < nt_ * (D / 4); idx += NTHR) {
            const int tt = idx / (D / 4), d4 = (idx - tt * (D / 4)) * 4;
            const size_t e0 = (size_t)(t0 + tt) * a.row_w + v_col + d4;
            float4 vp[NPC];
#pragma unroll
            for (int j = 0; j < NPC; ++j) vp[j] = *reinterpret_cast<const float4*>(a.parts + (size_t)min(j, a.n_parts - 1) * a.part_stride + e0);
            float4 acc = float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NPC; ++j)
                if (j < a.n_parts) { acc.x += vp[j].x; acc.y += vp[j].y; acc.z += vp[j].z; acc.w += vp[j].w; }
            uint32_t* dst = reinterpret_cast<uint32_t*>(stage_v + tt * LDV + d4);
            dst[0] = (uint32_t)from_f32<T>(acc.x) | ((uint32_t)from_f32<T>(acc.y) << 16);
            dst[1] = (uint32_t)from_f32<T>(acc.z) | ((uint32_t)from_f32<T>(acc.w) << 16);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wg_barrier();
        for (int idx = tid; idx < D * PROD_TOK; idx += NTHR) {
            const int dd = idx / PROD_TOK, tt = idx - dd * PROD_TOK;
            if (tt < nt_) store_wt_b16(a.vt_w + ((size_t)kvh * D + dd) * a.S_max + P + t0 + tt, stage_v[tt * LDV + dd]);
        }
    }
    // every store of this wave has left the chip's caches; then the barrier; then ONE arrival per work-group
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();
    if (tid == 0) __hip_atomic_fetch_add(a.flags + kvh, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Work split inside a work-group of RG x KQ waves: wave (rg, kq) owns query rows [32*rg, 32*rg+32) of the block and the
// 32-key part kq of every stage; a stage is KQ/2 tiles of 64 keys.  Three shapes are built:
//   RG=4, KQ=2 (8 waves, 128 rows, stage = 1 tile,  3-stage ring)   steps of more than 64 (head-in-group, token) rows
//   RG=2, KQ=4 (8 waves,  64 rows, stage = 2 tiles, 2-stage ring)   <= 64 rows: every wave computes (a steady 7B step has 60)
//   RG=1, KQ=4 (4 waves,  32 rows, stage = 2 tiles, 2-stage ring)   <= 32 rows (lookahead-parallel ranks, TinyLlama steps)
// rg / kq from the wave index: waves w and w+4 share a SIMD (dispatch order 0,2,1,3,0,2,1,3); with RG=4 and T <= 64 rows the
// four busy waves sit on four different SIMDs, and with more rows every SIMD interleaves two waves - one in its MFMA phase
// while the other waits on LDS or runs the softmax VALU work.
// KV split sp of n_splits takes a CONTIGUOUS range of ceil(tiles / n_splits) 64-key tiles (the interleaved assignment sp, sp+n, ...
// is kept behind LADE_DEBUG=attn_dbg=64: it balances perfectly but loses DRAM locality, +1.6 us at the 7B shape).  The KQ key parts keep
// separate online-softmax states and are merged through LDS at the end.
// NPC = 0: q is read from memory and the new K / V rows are already in the cache; NPC = 2 | 4: fused RoPE + KV append from up to NPC partials
template <typename T, int D, int RG, int KQ, int NPC>
__global__ __launch_bounds__(64 * RG * KQ) void attn_fwd_kernel(AttnK a) {
    constexpr int NW = RG * KQ;                    // waves
    constexpr int NTHR = 64 * NW;
    constexpr int TPS = KQ / 2;                    // 64-key tiles per stage
    constexpr int NSTG = TPS == 1 ? 3 : 2;         // ring depth (a work-group's first NSTG stages are requested at once)
    constexpr int ROWS = 32 * RG;
    constexpr int KSTEPS = D / 16;                 // MFMA k-steps of S^T
    constexpr int DBLK = D / 32;                   // 32-row blocks of O^T
    constexpr int K_BYTES = KT * D * 2, V_BYTES = D * KT * 2, TILE_BYTES = K_BYTES + V_BYTES, STAGE_BYTES = TPS * TILE_BYTES;
    constexpr int Q_BYTES = ROWS * D * 2;
    constexpr int KPW = K_BYTES / 1024 / NW;       // 1-KiB DMA pieces per wave per tile
    constexpr int VPW = V_BYTES / 1024 / NW;
    constexpr int QPW = Q_BYTES / 1024 / NW;
    constexpr int PIECES = TPS * (KPW + VPW);      // per wave per stage
    constexpr int K_CPR = D / 8;                   // 16-B chunks per K (and Q) row
    static_assert(KPW >= 1 && VPW >= 1 && QPW >= 1, "every wave moves at least one piece of every tile");

    // LDS: [ring of NSTG stages of TPS x (K tile | V^T tile)] [Q tile]; the ring is reused for the merge + store staging
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* q_lds = smem + NSTG * STAGE_BYTES;
    if constexpr (NPC > 0) {
        if ((int)blockIdx.x < a.n_prod) {               // producer mode: the head of the grid does the RoPE + append work once per KV head
            rope_producer<T, D, NPC, NTHR>(a, smem);
            return;
        }
    }
    const bool pmode = NPC > 0 && a.n_prod > 0;         // ... and this work-group waits for its head's flag before it touches q or a new row
    dbg_stamp(a, 0);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int rg, kq;
    if (RG == 4) { rg = ((wave >> 2) << 1) | (wave & 1); kq = (wave >> 1) & 1; }
    else if (RG == 2) { rg = wave & 1; kq = wave >> 1; }
    else { rg = 0; kq = wave; }
    const int ts_mine = kq >> 1, kh = kq & 1;      // this wave's tile within the stage, and its 32-key half of that tile
    const int ql = lane & 31, hi = lane >> 5;
    const BlockId bid = block_decode(a);
    if (!bid.live) return;                          // padding of the grid to a multiple of 8 groups
    const int kvh = bid.kvh, sp = bid.sp, rbk = bid.rb;
    const int n_rep = a.n_rep;
    const int ns = a.n_splits;

    const uint16_t* kbase = a.k + (size_t)kvh * a.S_max * D;
    const uint16_t* vbase = a.vt + (size_t)kvh * D * a.S_max;

    const int n_rows = n_rep * a.m.T;
    const float invT = a.inv_T;
    // row r of the (head-in-group, token) row space -> (hg, t); exact for r < 4096, T <= 512
    auto split_row = [&](int r, int& hg, int& t) {
        if (n_rep == 1) { hg = 0; t = r; }
        else { hg = (int)(((float)r + 0.5f) * invT); t = r - hg * a.m.T; }
    };
    // the Q tile is requested FIRST: its addresses need nothing but the block id, so its pieces are on their way while the cache length
    // (a dependent scalar load in hipGraph steps) is still in flight; the first counted wait covers the Q tile + stage 0 (Q is the oldest)
    constexpr int VPH = D / 16;                    // fused: 16-byte vector pairs (i, i + D/2) of one head row
    constexpr int NPI = NPC ? NPC : 1;
    constexpr int QI = (ROWS * VPH + NTHR - 1) / NTHR;       // fused: rope items of the Q tile per thread
    RopeItem<NPI> qit[QI];
    auto q_item = [&](int it, int& row, int& i, int& hg, int& t) -> int {      // 0: nothing, 1: a real row, 2: a padding row of a live 32-row group (zeros)
        const int idx = tid + it * NTHR;
        row = idx / VPH;
        i = (idx - row * VPH) * 8;
        hg = 0; t = 0;
        if (idx >= ROWS * VPH) return 0;
        const int r = rbk * ROWS + row;
        if (r < n_rows) { split_row(r, hg, t); return 1; }
        return rbk * ROWS + (row & ~31) < n_rows ? 2 : 0;
    };
    auto issue_q = [&](auto sc1_c) {
        constexpr int AUX = decltype(sc1_c)::value ? 16 : 0;      // sc1: served by L2 / memory, never by this CU's L1 (producer mode)
#pragma unroll
        for (int i = 0; i < QPW; ++i) {
            const int piece = wave * QPW + i;
            // a 32-row group without any row (a steady 7B step has 60 rows: groups 2 and 3 of the 128-row block) is not requested at all:
            // its waves never read it (wave_rows below).  Pieces are wave uniform, the counted waits below do not depend on how many a wave issued.
            if (rbk * ROWS + (piece * (64 / K_CPR) / 32) * 32 >= n_rows) continue;
            const int row = piece * (64 / K_CPR) + lane / K_CPR;
            const int c = (lane % K_CPR) ^ swz16<2 * D>(row);
            int r = rbk * ROWS + row, hg, t;
            if (r >= n_rows) r = 0;                       // rows past the end read row 0; never stored
            split_row(r, hg, t);
            const uint16_t* src = a.q + (size_t)t * a.q_row_stride + (size_t)(kvh * n_rep + hg) * D + c * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(q_lds + piece * 1024), 16, 0, AUX);
        }
    };
    if constexpr (NPC == 0) {
        issue_q(std::false_type{});
    } else if (!pmode) {
        // fused: the q rows of this block as fp32 partials of the qkv projection + their cos / sin rows, requested before anything that
        // depends on the cache length; rotated and written to the Q tile further down, behind the K / V requests
#pragma unroll
        for (int it = 0; it < QI; ++it) {
            int row, i, hg, t;
            if (q_item(it, row, i, hg, t) == 1) {
                int trow = t;
                if (a.positions) { trow = a.positions[t]; trow = trow < 0 ? 0 : (trow >= a.max_pos ? a.max_pos - 1 : trow); }
                rope_load<NPI, D>(qit[it], a, (size_t)t * a.row_w + (size_t)(kvh * n_rep + hg) * D + i, trow, i);
            }
        }
    }
    // ---- split geometry: split sp covers the 64-key tiles base, base+stride, ... (my_tiles of them)
    lade_mask_params m = a.m;
    if (a.dyn_P) m.P = *a.dyn_P;
    const int S_tot = m.P + m.T;
    int n_tiles = (S_tot + KT - 1) / KT;
    if (m.is_prefill) {
        // plain causal rows (prefill chunks, modeling_llama.py:124-130): no row of this block sees a key beyond its last token, so
        // the tiles behind it are neither requested nor computed - the splits of the block partition only what it can see
        const int r0 = rbk * ROWS, r1 = min(r0 + ROWS, n_rep * m.T) - 1;
        const int hg0 = (int)div_magic((uint32_t)r0, a.t_magic), hg1 = (int)div_magic((uint32_t)r1, a.t_magic);
        const int tmax = hg0 != hg1 ? m.T - 1 : r1 - hg1 * m.T;
        n_tiles = min(n_tiles, (m.P + tmax + 1 + KT - 1) / KT);
    }
    // contiguous key ranges (an interleaved assignment sp, sp + n, .. balances perfectly but loses DRAM locality: +1.6 us at the 7B shape, round 2)
    const int tps = (int)div_magic((uint32_t)(n_tiles + ns - 1), a.ns_magic);       // no integer division on the way to the first DMA
    const int base = sp * tps;
    constexpr int stride = 1;
    const int my_tiles = max(0, min(base + tps, n_tiles) - base);
    // ---- LDS-DMA (no VGPR staging): the split's first NSTG stages are requested before the row bookkeeping; a tile that lies
    // beyond the split is a harmless read inside the cache allocation (an L2 hit of the split's first tile) and is never computed on
    const int last_tile = a.S_max / KT - 1;
    // per-lane byte offsets of this wave's pieces inside a K tile / a V^T tile, resolved ONCE: a piece of a tile is then (wave-uniform tile
    // base) + (32-bit lane offset) - one scalar add per piece instead of a 64-bit multiply-add per lane.  (The prologue's 16 pieces per wave
    // took 4 k cycles to issue; every cycle of it delays the moment the split's stream is fully requested.)
    uint32_t k_off[KPW], v_off[VPW];
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        const int piece = wave * KPW + i;
        const int row = piece * (64 / K_CPR) + lane / K_CPR;
        const int c = (lane % K_CPR) ^ swz16<2 * D>(row);
        k_off[i] = (uint32_t)(row * D + c * 8) * 2u;
    }
#pragma unroll
    for (int i = 0; i < VPW; ++i) {
        const int piece = wave * VPW + i;
        const int row = piece * 8 + (lane >> 3);
        const int c = (lane & 7) ^ swz16<2 * KT>(row);
        v_off[i] = ((uint32_t)row * (uint32_t)a.S_max + (uint32_t)(c * 8)) * 2u;
    }
    const unsigned char* kbytes = reinterpret_cast<const unsigned char*>(kbase);
    const unsigned char* vbytes = reinterpret_cast<const unsigned char*>(vbase);
    auto issue_tiles_aux = [&](int stage, int ts, int tile, auto aux_c) {
        constexpr int AUX = decltype(aux_c)::value;          // cache-policy bits of the K / V loads: 1 = sc0, 2 = nt, 16 = sc1
        const int k0 = min(tile, last_tile) * KT;
        unsigned char* ks = smem + stage * STAGE_BYTES + ts * TILE_BYTES;
        unsigned char* vs = ks + K_BYTES;
        const unsigned char* kt = kbytes + (size_t)k0 * D * 2;          // wave uniform
        const unsigned char* vt = vbytes + (size_t)k0 * 2;
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
            const int piece = wave * KPW + i;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kt + k_off[i]),
                                             (__attribute__((address_space(3))) void*)(ks + piece * 1024), 16, 0, AUX);
        }
#pragma unroll
        for (int i = 0; i < VPW; ++i) {
            const int piece = wave * VPW + i;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vt + v_off[i]),
                                             (__attribute__((address_space(3))) void*)(vs + piece * 1024), 16, 0, AUX);
        }
    };
    int sc1_from_q = 1 << 30;            // producer mode: tiles (offsets within the split) from this one on hold rows a producer wrote in THIS launch: sc1 loads
    // cache policy of the K / V stream: a COMPILE-time constant (LADE_KV_AUX; bits 1 = sc0, 2 = nt, 16 = sc1) - variants are separate builds
    auto issue_tiles = [&](int stage, int ts, int tile) {
        if (NPC > 0 && tile - base >= sc1_from_q) issue_tiles_aux(stage, ts, tile, std::integral_constant<int, 16>{});
        else issue_tiles_aux(stage, ts, tile, std::integral_constant<int, LADE_KV_AUX>{});
    };
    const int nt = (my_tiles + TPS - 1) / TPS;                                  // stages
    const int nt_issued = max(nt, NSTG);                                        // the first NSTG stages are always in flight
    // global index (64-key units) of tile ts of stage j; a stage's missing second tile re-reads the split's first tile (an L2
    // hit) so that every stage carries the same number of pieces for the counted vmcnt waits - it is never computed on
    auto tile_of = [&](int j, int ts) { const int q = j * TPS + ts; return q < my_tiles ? base + q * stride : base; };
    auto issue_stage = [&](int j, int stage) {
#pragma unroll
        for (int ts = 0; ts < TPS; ++ts) issue_tiles(stage, ts, tile_of(j, ts));
    };
    float pm_poison = 0.f;               // producer mode: NaN once this work-group's wait for its producers has timed out
    bool pm_wait_all = false;            // producer mode: q was requested AFTER the first stages - the loop's first wait covers everything
    int store_fence_at = -1;             // fused: the loop iteration behind which the first tile holding a row this work-group stored is requested
    if constexpr (NPC == 0) {
#pragma unroll
        for (int s = 0; s < NSTG; ++s) issue_stage(s, s);
    } else {
        // ---- fused RoPE + KV append ----
        if (pmode) {
            // producer mode: the ring stages that hold no new row are requested at once; q and the others after this KV head's flag
            const int first_new_q = m.P / KT - base;       // tile (within the split) of the first new key; <= 0: the split starts inside the new rows
            int early = 0;
#pragma unroll
            for (int s = 0; s < NSTG; ++s) {
                const int last_q = s * TPS < my_tiles ? min(s * TPS + TPS - 1, my_tiles - 1) : 0;
                if (early == s && last_q < first_new_q) early = s + 1;
            }
            if (early == NSTG) {
#pragma unroll
                for (int s = 0; s < NSTG; ++s) issue_stage(s, s);
            } else {
                for (int s = 0; s < early; ++s) issue_stage(s, s);
            }
            if (tid == 0) {                          // ONE lane polls (relaxed, device scope: an sc1 load), bounded: ~0.1 s
                int spins = 0;
                while (__hip_atomic_load(a.flags + kvh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.prod_chunks && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(2);
                *reinterpret_cast<volatile int*>(q_lds) = spins >= (1 << 20);
            }
            wg_barrier();
            // a poll that ran out must not end as plausible numbers: the work-group's row sums become NaN, and with them the merged output rows
            // (the Q tile is free until issue_q below: the word is read by every thread between two barriers)
            if (*reinterpret_cast<volatile int*>(q_lds)) pm_poison = __builtin_nanf("");
            wg_barrier();
            sc1_from_q = first_new_q;                // from here on the tiles that hold new rows are requested with sc1 loads
            issue_q(std::true_type{});
            for (int s = early; s < NSTG; ++s) issue_stage(s, s);
            pm_wait_all = true;
        } else {
        // the rows P .. P+T of K and V^T that fall into this work-group's key range are its own to produce: key k_lo .. k_hi
        const int k_lo = max(m.P, base * KT), k_hi = min(S_tot, (base + my_tiles) * KT);
        const int n_new = S_tot <= a.S_max ? max(0, k_hi - k_lo) : 0;         // (a device-side cache length past the cache: nothing is written, as in lade_rope_kv_append)
        const int t_lo = k_lo - m.P;
        auto q_finish = [&]() {                  // rotate the q rows and write them to the Q tile (swizzled like the DMA would have)
#pragma unroll
            for (int it = 0; it < QI; ++it) {
                int row, i, hg, t;
                const int kind = q_item(it, row, i, hg, t);
                if (kind == 0) continue;
                u32x4 o1 = u32x4{0u, 0u, 0u, 0u}, o2 = o1;
                if (kind == 1) rope_apply<T, NPI>(qit[it], a.n_parts, o1, o2);
                *reinterpret_cast<u32x4*>(q_lds + tile_off<2 * D>(row, i >> 3)) = o1;
                *reinterpret_cast<u32x4*>(q_lds + tile_off<2 * D>(row, (i + D / 2) >> 3)) = o2;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // published by the barrier of the loop's first iteration
        };
        if (n_new == 0) {
#pragma unroll
            for (int s = 0; s < NSTG; ++s) issue_stage(s, s);
            q_finish();
        } else {
            constexpr int CH = 16 * RG;                      // new K / V rows per pass
            constexpr int KI = (CH * VPH + NTHR - 1) / NTHR; // rope items of a pass per thread
            constexpr int VS = CH * (D / 4) / NTHR;          // float4 sites of a pass's V rows per thread
            constexpr int LDV = D + 2;                       // row stride (elements) of the V staging tile [CH][D]: it lives in the Q tile
            static_assert(VS >= 1 && (CH * (D / 4)) % NTHR == 0 && CH * LDV * 2 <= Q_BYTES, "V staging pass");
            const size_t k_col = (size_t)(a.H + kvh) * D, v_col = (size_t)(a.H + a.Hkv + kvh) * D;
            RopeItem<NPI> kit[KI];
            float4 vpart[NPI][VS];
            auto kv_load = [&](int c) {
#pragma unroll
                for (int it = 0; it < KI; ++it) {
                    const int idx = tid + it * NTHR, tt = idx / VPH, i = (idx - tt * VPH) * 8;
                    if (idx < CH * VPH && c * CH + tt < n_new) {
                        const int t = t_lo + c * CH + tt;
                        int trow = t;
                        if (a.positions) { trow = a.positions[t]; trow = trow < 0 ? 0 : (trow >= a.max_pos ? a.max_pos - 1 : trow); }
                        rope_load<NPI, D>(kit[it], a, (size_t)t * a.row_w + k_col + i, trow, i);
                    }
                }
#pragma unroll
                for (int sv = 0; sv < VS; ++sv) {
                    const int idx = tid + sv * NTHR, tt = idx / (D / 4), d4 = (idx - tt * (D / 4)) * 4;
                    if (c * CH + tt < n_new) {
                        const size_t e0 = (size_t)(t_lo + c * CH + tt) * a.row_w + v_col + d4;
#pragma unroll
                        for (int j = 0; j < NPI; ++j) vpart[j][sv] = *reinterpret_cast<const float4*>(a.parts + (size_t)min(j, a.n_parts - 1) * a.part_stride + e0);
                    }
                }
            };
            auto kv_store = [&](int c, uint16_t* stage_v) {
                // V: partials summed in split order, rounded once, staged [token][d] ...
#pragma unroll
                for (int sv = 0; sv < VS; ++sv) {
                    const int idx = tid + sv * NTHR, tt = idx / (D / 4), d4 = (idx - tt * (D / 4)) * 4;
                    if (c * CH + tt < n_new) {
                        float4 acc = float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int j = 0; j < NPI; ++j)
                            if (j < a.n_parts) { acc.x += vpart[j][sv].x; acc.y += vpart[j][sv].y; acc.z += vpart[j][sv].z; acc.w += vpart[j][sv].w; }
                        uint32_t* dst = reinterpret_cast<uint32_t*>(stage_v + tt * LDV + d4);
                        dst[0] = (uint32_t)from_f32<T>(acc.x) | ((uint32_t)from_f32<T>(acc.y) << 16);
                        dst[1] = (uint32_t)from_f32<T>(acc.z) | ((uint32_t)from_f32<T>(acc.w) << 16);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                wg_barrier();
                // K: rotated rows straight to cache row P + t
#pragma unroll
                for (int it = 0; it < KI; ++it) {
                    const int idx = tid + it * NTHR, tt = idx / VPH, i = (idx - tt * VPH) * 8;
                    if (idx < CH * VPH && c * CH + tt < n_new) {
                        u32x4 o1, o2;
                        rope_apply<T, NPI>(kit[it], a.n_parts, o1, o2);
                        uint16_t* dst = a.k_w + ((size_t)kvh * a.S_max + k_lo + c * CH + tt) * D + i;
                        *reinterpret_cast<u32x4*>(dst) = o1;
                        *reinterpret_cast<u32x4*>(dst + D / 2) = o2;
                    }
                }
                // ... and written transposed, the token index fastest: V^T[d][P + t]
                for (int idx = tid; idx < D * CH; idx += NTHR) {
                    const int dd = idx / CH, tt = idx - dd * CH;
                    if (c * CH + tt < n_new) a.vt_w[((size_t)kvh * D + dd) * a.S_max + k_lo + c * CH + tt] = stage_v[tt * LDV + dd];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                wg_barrier();                  // the staging tile is free again (next pass, or the q rows)
            };
            // the stages that hold none of those rows are requested now, the others once the rows are in memory
            const int first_new_q = m.P / KT - base;       // tile (within the split) of the first new key; <= 0: the split starts inside the new rows
            int early = 0;
#pragma unroll
            for (int s = 0; s < NSTG; ++s) {
                const int last_q = s * TPS < my_tiles ? min(s * TPS + TPS - 1, my_tiles - 1) : 0;
                if (early == s && last_q < first_new_q) early = s + 1;
            }
            const int n_pass = (n_new + CH - 1) / CH;
            if (early == NSTG) {
                // the usual case: the new rows lie behind the first NSTG stages.  Straight-line code: the partials are requested BEFORE the
                // stages, so the counted wait the compiler derives for them leaves every DMA piece in flight.  The V staging tile lives in
                // the Q tile, which is written last.
                kv_load(0);
#pragma unroll
                for (int s = 0; s < NSTG; ++s) issue_stage(s, s);
                kv_store(0, reinterpret_cast<uint16_t*>(q_lds));
                for (int c = 1; c < n_pass; ++c) { kv_load(c); kv_store(c, reinterpret_cast<uint16_t*>(q_lds)); }
                q_finish();
                // the stores must have landed before the first stage that holds one of those rows is requested: checked in the loop, where
                // that request is made (a wait here would also wait for the stages in flight)
                store_fence_at = max(NSTG, first_new_q / TPS) - NSTG;
            } else {
                // a split that reaches the new rows within its first NSTG stages (short caches): nothing is requested before the rows are in
                // memory; the q rows go first (their registers are free for the passes), the V staging tile lives in the idle ring
                q_finish();
                for (int c = 0; c < n_pass; ++c) { kv_load(c); kv_store(c, reinterpret_cast<uint16_t*>(smem)); }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                wg_barrier();
#pragma unroll
                for (int s = 0; s < NSTG; ++s) issue_stage(s, s);
            }
        }
        }
    }
    dbg_stamp(a, 1);

    // this lane's query row (overlaps the DMA flight)
    const int r = rbk * ROWS + rg * 32 + ql;
    const bool valid = r < n_rows;
    int hg = 0, t = 0;
    if (valid) split_row(r, hg, t);
    const int qh = kvh * n_rep + hg;
    const RowDesc rd = make_row(t, valid, m);
    const bool wave_rows = __builtin_amdgcn_ballot_w64(valid) != 0ull;     // any row in this wave's group

    u32x4 qf[KSTEPS];
    f32x16 oacc[DBLK];
#pragma unroll
    for (int i = 0; i < DBLK; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) oacc[i][j] = 0.f;
    float m_run = NEG_BIG, l_run = pm_poison;
    const int krow = 32 * kh + kappa(ql);          // LDS row (key within the tile) of this lane's MFMA row

    for (int i = 0; i < nt; ++i) {
        const int stage = i % NSTG;
        const unsigned char* ks = smem + stage * STAGE_BYTES + ts_mine * TILE_BYTES;
        const unsigned char* vs = ks + K_BYTES;
        // wait for stage i (and, the first time, the older Q pieces): the pieces of the younger stages stay in flight
        const int younger = min(nt_issued, i + NSTG) - (i + 1);
        if (NPC > 0 && pm_wait_all && i == 0) wait_vm<0>();
        else if (younger >= 2) wait_vm<2 * PIECES>();
        else if (younger == 1) wait_vm<PIECES>();
        else wait_vm<0>();
        wg_barrier();
        if (i == 0) {
            dbg_stamp(a, 2);
            // Q^T B-operand fragments: Q[row][kk*16 + hi*8 .. +8]
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk)
                qf[kk] = *reinterpret_cast<const u32x4*>(q_lds + tile_off<2 * D>(rg * 32 + ql, kk * 2 + hi));
        }

#pragma unroll
        for (int ts = 0; ts < TPS; ++ts) {
            const int kt0 = tile_of(i, ts) * KT;
            if (i * TPS + ts < my_tiles && kt0 + KT > S_tot) {          // work-group uniform
                // last tile: V^T columns of keys >= P+T hold stale bytes; zero them (0 * NaN would poison O)
                unsigned char* vw = smem + stage * STAGE_BYTES + ts * TILE_BYTES + K_BYTES;
                const int first = S_tot - kt0;
                for (int idx = tid; idx < D * (KT - first); idx += NTHR) {
                    const int row = idx / (KT - first), key = first + idx % (KT - first);
                    *reinterpret_cast<uint16_t*>(vw + tile_off<2 * KT>(row, key >> 3) + (key & 7) * 2) = 0;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                wg_barrier();
            }
        }
        const bool have_tile = i * TPS + ts_mine < my_tiles;           // wave uniform
        const int k0 = tile_of(i, ts_mine) * KT;
        const bool full = (k0 + KT <= m.P);           // whole tile in the cache: every key visible
        uint32_t bits = 0xffffu;                      // visibility of this lane's 16 keys 32kh + 16hi + e
        if (!full) bits = (uint32_t)(vis_bits(k0 - m.P, rd, m) >> (32 * kh + 16 * hi)) & 0xffffu;
        else if (!valid) bits = 0u;
        if (have_tile && wave_rows && __builtin_amdgcn_ballot_w64(bits != 0u) != 0ull) {
            // ---- S^T = K Q^T on this wave's 32-key part ----
            f32x16 sacc;
#pragma unroll
            for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
            {
                u32x4 kf[KSTEPS];                   // all fragments first: LDS latency overlaps the MFMAs
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk) kf[kk] = *reinterpret_cast<const u32x4*>(ks + tile_off<2 * D>(krow, kk * 2 + hi));
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk) sacc = Mfma<T>::run(kf[kk], qf[kk], sacc);
            }
            // V^T fragments of the first PV k-step are requested before the softmax VALU work
            u32x4 vf0[DBLK];
#pragma unroll
            for (int db = 0; db < DBLK; ++db) vf0[db] = *reinterpret_cast<const u32x4*>(vs + tile_off<2 * KT>(db * 32 + ql, 4 * kh + 2 * hi));
            // ---- online softmax in the log2 domain; lane (q, hi) holds keys 32kh + 16hi + e ----
            float tmax = NEG_BIG;
            if (full) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    sacc[e] *= a.scale_log2;
                    tmax = fmaxf(tmax, sacc[e]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float sv = ((bits >> e) & 1u) ? sacc[e] * a.scale_log2 : MASKED;
                    sacc[e] = sv;
                    tmax = fmaxf(tmax, sv);
                }
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
            // raise the running max only when it grows by more than RESCALE_THR (wave-uniform decision):
            // p stays <= 2^THR, harmless in the fp32 accumulators, and the O^T rescale is skipped on most tiles
            if (__builtin_amdgcn_ballot_w64(tmax > m_run + RESCALE_THR) != 0ull) {
                const float m_new = fmaxf(m_run, tmax);
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                m_run = m_new;
                l_run *= alpha;
#pragma unroll
                for (int ii = 0; ii < DBLK; ++ii)
#pragma unroll
                    for (int e = 0; e < 16; ++e) oacc[ii][e] *= alpha;
            }
            float psum = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float p = __builtin_amdgcn_exp2f(sacc[e] - m_run);   // masked: 2^(-3e30 - m) = 0
                sacc[e] = p;
                psum += p;
            }
            l_run += psum;
            // ---- O^T += V^T P^T : k-step jj = keys 32kh + 16hi + 8jj + 0..7 ----
            u32x4 vf1[DBLK];
#pragma unroll
            for (int db = 0; db < DBLK; ++db) vf1[db] = *reinterpret_cast<const u32x4*>(vs + tile_off<2 * KT>(db * 32 + ql, 4 * kh + 2 * hi + 1));
            u32x4 pf0, pf1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pf0[e] = pack2<T>(sacc[2 * e], sacc[2 * e + 1]);
                pf1[e] = pack2<T>(sacc[8 + 2 * e], sacc[8 + 2 * e + 1]);
            }
#pragma unroll
            for (int db = 0; db < DBLK; ++db) oacc[db] = Mfma<T>::run(vf0[db], pf0, oacc[db]);
#pragma unroll
            for (int db = 0; db < DBLK; ++db) oacc[db] = Mfma<T>::run(vf1[db], pf1, oacc[db]);
        }
        if (i + NSTG < nt) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (NPC > 0) {
                // fused: this work-group's own K / V rows have reached memory before a stage that holds them is requested (once per launch,
                // work-group uniform; every wave waits for its own stores, the barrier covers the others')
                if (i == store_fence_at) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            wg_barrier();              // every wave is done reading this stage
            issue_stage(i + NSTG, stage);
        }
    }
    dbg_stamp(a, 3);

    // ---- merge the key parts ----
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    wg_barrier();                      // ring no longer read or written: reuse it
    constexpr int RS = 2 * D + 16;                     // staging row stride (bytes), 16-B aligned
    unsigned char* stg_base;                           // normalised rows of the block in the model dtype, 32 rows per row group
    size_t stg_stride;
    if constexpr (KQ == 2) {
        // Two key parts (the default shape): the two waves of a row group SWAP halves of O^T - wave (rg, kq) keeps the d-blocks
        // [kq*HALF, kq*HALF + HALF) and hands the other half plus its (m, l) to its partner - so that each merges, normalises, packs and
        // stages half of the columns.  All fragment reads of the merge are requested before the first use (the compiler had chained them
        // one LDS latency after the other: 16 x ~100 cycles on the critical path of every launch).
        constexpr int HALF = DBLK / 2;
        constexpr int SX_F4 = HALF * 4 + 1;            // float4 rows per lane of a hand-over slot: HALF d-blocks + (m, l, -, -)
        constexpr size_t SLOT = (size_t)SX_F4 * 64 * 16;
        constexpr size_t STG_OFF = SLOT * RG * 2;
        static_assert(STG_OFF + (size_t)RG * 32 * RS <= (size_t)NSTG * STAGE_BYTES + Q_BYTES, "hand-over slots + staging rows must fit in the ring + Q tile");
        stg_base = smem + STG_OFF;
        stg_stride = (size_t)32 * RS;
        // kq is wave uniform: the two roles are two straight-line instantiations (a run-time select between register arrays would be
        // compiled into a scratch-memory round trip)
        auto swap_merge = [&](auto first_part) {
            constexpr bool P0 = decltype(first_part)::value;          // this wave holds key part 0 and keeps the LOWER d-blocks
            constexpr int KEEP = P0 ? 0 : HALF, SEND = P0 ? HALF : 0;
            float4* mine = reinterpret_cast<float4*>(smem + SLOT * (rg * 2 + (P0 ? 0 : 1)));
            const float4* theirs = reinterpret_cast<const float4*>(smem + SLOT * (rg * 2 + (P0 ? 1 : 0)));
#pragma unroll
            for (int h = 0; h < HALF; ++h)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
                    mine[(h * 4 + g4) * 64 + lane] = float4{oacc[SEND + h][4 * g4], oacc[SEND + h][4 * g4 + 1], oacc[SEND + h][4 * g4 + 2], oacc[SEND + h][4 * g4 + 3]};
            mine[(HALF * 4) * 64 + lane] = float4{m_run, l_run, 0.f, 0.f};
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            wg_barrier();
            float4 in[HALF * 4];
            const float4 mlo = theirs[(HALF * 4) * 64 + lane];
#pragma unroll
            for (int j = 0; j < HALF * 4; ++j) in[j] = theirs[j * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);             // every read is in flight before the first one is consumed
            // the same expression in both waves, always in key-part order (part 0, part 1): identical m and l on both sides
            const float m0 = P0 ? m_run : mlo.x, m1 = P0 ? mlo.x : m_run;
            const float l0 = P0 ? l_run : mlo.y, l1 = P0 ? mlo.y : l_run;
            const float mm = fmaxf(m0, m1);
            const float a0 = __builtin_amdgcn_exp2f(m0 - mm), a1 = __builtin_amdgcn_exp2f(m1 - mm);
            float lsum = __builtin_fmaf(l0, a0, l1 * a1);
            dbg_stamp(a, 4);
            lsum += __shfl_xor(lsum, 32);
            const float inv = lsum > 0.f ? __builtin_amdgcn_rcpf(lsum) : 0.f;      // 1 ulp; the result is rounded to 16 bits
            const float s_own = (P0 ? a0 : a1) * inv, s_oth = (P0 ? a1 : a0) * inv;
            unsigned char* stg = stg_base + (size_t)rg * stg_stride;
#pragma unroll
            for (int h = 0; h < HALF; ++h)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const float4 o1 = in[h * 4 + g4];
                    u32x2 w;
                    w[0] = pack2<T>(__builtin_fmaf(oacc[KEEP + h][4 * g4 + 0], s_own, o1.x * s_oth), __builtin_fmaf(oacc[KEEP + h][4 * g4 + 1], s_own, o1.y * s_oth));
                    w[1] = pack2<T>(__builtin_fmaf(oacc[KEEP + h][4 * g4 + 2], s_own, o1.z * s_oth), __builtin_fmaf(oacc[KEEP + h][4 * g4 + 3], s_own, o1.w * s_oth));
                    *reinterpret_cast<u32x2*>(stg + ql * RS + ((KEEP + h) * 32 + 8 * g4 + 4 * hi) * 2) = w;
                }
            if (P0 && a.n_splits > 1 && valid && hi == 0) {
                const size_t prow = ((size_t)sp * a.H + qh) * m.T + t;
                *reinterpret_cast<float2*>(a.part_ml + prow * 2) = float2{mm, lsum};
            }
        };
        if (kq == 0) swap_merge(std::true_type{});
        else swap_merge(std::false_type{});
    } else {
        // more than two key parts (the 64- and 32-row shapes): wave (rg, kq > 0) hands its state to wave (rg, 0), lane to lane
        constexpr int EX_F4 = DBLK * 4 + 1;            // float4 slots per lane: O^T (16*DBLK floats) + (m, l, -, -)
        static_assert((size_t)(KQ - 1) * RG * EX_F4 * 64 * 16 <= (size_t)NSTG * STAGE_BYTES, "merge slots must fit in the ring");
        float4* ex_all = reinterpret_cast<float4*>(smem);
        if (kq > 0) {
            float4* ex = ex_all + (size_t)((kq - 1) * RG + rg) * EX_F4 * 64;
#pragma unroll
            for (int db = 0; db < DBLK; ++db)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
                    ex[(db * 4 + g4) * 64 + lane] = float4{oacc[db][4 * g4], oacc[db][4 * g4 + 1], oacc[db][4 * g4 + 2], oacc[db][4 * g4 + 3]};
            ex[(DBLK * 4) * 64 + lane] = float4{m_run, l_run, 0.f, 0.f};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wg_barrier();
        // staging rows of row group rg: the merge slot of (kq = 1, rg), which only wave (rg, 0) reads (LDS operations of one wave
        // execute in order, so its staging writes cannot overtake its own merge reads)
        stg_base = reinterpret_cast<unsigned char*>(ex_all);
        stg_stride = (size_t)EX_F4 * 64 * 16;
        static_assert(32 * RS <= (size_t)EX_F4 * 64 * 16, "a row group's staging rows fit in its merge slot");
        if (kq == 0) {
#pragma unroll
            for (int pq = 1; pq < KQ; ++pq) {
                const float4* ex = ex_all + (size_t)((pq - 1) * RG + rg) * EX_F4 * 64;
                const float4 ml1 = ex[(DBLK * 4) * 64 + lane];
                const float mm = fmaxf(m_run, ml1.x);
                const float a0 = __builtin_amdgcn_exp2f(m_run - mm), a1 = __builtin_amdgcn_exp2f(ml1.x - mm);
#pragma unroll
                for (int db = 0; db < DBLK; ++db)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const float4 o1 = ex[(db * 4 + g4) * 64 + lane];
                        oacc[db][4 * g4 + 0] = oacc[db][4 * g4 + 0] * a0 + o1.x * a1;
                        oacc[db][4 * g4 + 1] = oacc[db][4 * g4 + 1] * a0 + o1.y * a1;
                        oacc[db][4 * g4 + 2] = oacc[db][4 * g4 + 2] * a0 + o1.z * a1;
                        oacc[db][4 * g4 + 3] = oacc[db][4 * g4 + 3] * a0 + o1.w * a1;
                    }
                l_run = l_run * a0 + ml1.y * a1;
                m_run = mm;
            }
            // ---- normalise, transpose through LDS ----
            dbg_stamp(a, 4);
            l_run += __shfl_xor(l_run, 32);
            unsigned char* stg = stg_base + (size_t)rg * stg_stride;
            const float inv = l_run > 0.f ? __builtin_amdgcn_rcpf(l_run) : 0.f;      // 1 ulp; the result is rounded to 16 bits
#pragma unroll
            for (int db = 0; db < DBLK; ++db)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    u32x2 w;
                    w[0] = pack2<T>(oacc[db][4 * g4 + 0] * inv, oacc[db][4 * g4 + 1] * inv);
                    w[1] = pack2<T>(oacc[db][4 * g4 + 2] * inv, oacc[db][4 * g4 + 3] * inv);
                    *reinterpret_cast<u32x2*>(stg + ql * RS + (db * 32 + 8 * g4 + 4 * hi) * 2) = w;
                }
            if (a.n_splits > 1 && valid && hi == 0) {
                const size_t prow = ((size_t)sp * a.H + qh) * m.T + t;
                *reinterpret_cast<float2*>(a.part_ml + prow * 2) = float2{m_run, l_run};
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wg_barrier();
    dbg_stamp(a, 7);
    // ---- every wave stores whole 2D-byte rows (16 bytes per lane): the store tail is issue bound, so it is spread over all waves
    constexpr int CPR = 2 * D / 16;                    // 16-B chunks per output row
    uint16_t* obase = a.n_splits == 1 ? a.out : a.part_o + (size_t)sp * m.T * a.H * D;
    const int64_t ostride = a.n_splits == 1 ? a.out_row_stride : (int64_t)a.H * D;
    const int row0 = rbk * ROWS;
    const int n_store = min(ROWS, n_rows - row0) * CPR;
    for (int idx = tid; idx < n_store; idx += NTHR) {
        const int row = idx / CPR, c = idx % CPR;
        int hg2, t2;
        split_row(row0 + row, hg2, t2);
        const u32x4 v = *reinterpret_cast<const u32x4*>(stg_base + (size_t)(row >> 5) * stg_stride + (row & 31) * RS + c * 16);
        uint16_t* dstp = obase + (size_t)t2 * ostride + (size_t)(kvh * n_rep + hg2) * D + c * 8;
#if LADE_PO_WT
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dstp), "v"(v) : "memory");          // write-through: nothing dirty left for the boundary in front of the merge
#else
        *reinterpret_cast<u32x4*>(dstp) = v;
#endif
    }
    dbg_stamp(a, 5);
#ifdef LADE_ATTN_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    dbg_stamp(a, 6);
#endif
}

// merges split-KV partials: out = sum_s w_s o_s / sum_s w_s with w_s = l_s 2^(m_s - m); o_s are the
// normalised per-split outputs in the model dtype.  One thread per 16 bytes (8 values) of one (token, head);
// the per-split (m, l) pairs are read first so the partial-output loads are independent.
template <typename T>
__global__ __launch_bounds__(256) void attn_combine_kernel(AttnK a, int D) {
    const int t = blockIdx.x, qh = blockIdx.y * blockDim.y + threadIdx.y;
    const int Tn = a.m.T;
    const int d0 = threadIdx.x * 8;
    PlainPartials ld;
    ld.ml_p = a.part_ml + ((size_t)qh * Tn + t) * 2;
    ld.ml_stride = (size_t)a.H * Tn * 2;
    ld.po_p = a.part_o + ((size_t)t * a.H + qh) * D + d0;
    ld.po_stride = (size_t)Tn * a.H * D;
    const u32x4 wv = merge_splits<T>(a.n_splits, ld);
    *reinterpret_cast<u32x4*>(a.out + (size_t)t * a.out_row_stride + (size_t)qh * D + d0) = wv;
    // producer mode of the fused form: every attention work-group of the launch before this one is done - the flags are zero again for the next
    if (a.flags && blockIdx.x == 0 && blockIdx.y == 0) {
        const int lt = threadIdx.y * blockDim.x + threadIdx.x;
        if (lt < a.Hkv) a.flags[lt] = 0;
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
    LADE_REQUIRE((a->q || a->n_parts != 0) && a->k_cache && a->vt_cache && a->out, LADE_E_ARG, "lade_attn: null tensor pointer");
    LADE_REQUIRE(a->H > 0 && a->Hkv > 0 && a->H % a->Hkv == 0, LADE_E_ARG, "lade_attn: H=%d Hkv=%d", a->H, a->Hkv);
    LADE_REQUIRE(a->mask.T > 0 && a->mask.P >= 0, LADE_E_ARG, "lade_attn: T=%d P=%d", a->mask.T, a->mask.P);
    LADE_REQUIRE(a->S_max % 64 == 0 && a->mask.P + a->mask.T <= a->S_max, LADE_E_ARG,
                 "lade_attn: S_max=%d must be a multiple of 64 and >= P+T=%d", a->S_max, a->mask.P + a->mask.T);
    LADE_REQUIRE(a->n_splits >= 1 && a->n_splits <= 32, LADE_E_ARG, "lade_attn: n_splits=%d (1..32)", a->n_splits);
    LADE_REQUIRE(a->n_splits == 1 || (a->part_o && a->part_ml), LADE_E_ARG, "lade_attn: split-KV needs partial buffers");
    LADE_REQUIRE(a->wg_rows == 0 || a->wg_rows == 32 || a->wg_rows == 64 || a->wg_rows == 128, LADE_E_ARG, "lade_attn: wg_rows=%d (0, 32, 64, 128)", a->wg_rows);
    if (a->n_parts != 0) {
        LADE_REQUIRE(a->n_parts >= 1 && a->n_parts <= 4, LADE_E_LIMIT, "lade_attn: fused RoPE takes 1..4 split-K partials, got %d", a->n_parts);
        LADE_REQUIRE(a->qkv_parts && a->cos_tab && a->sin_tab && a->max_pos > 0, LADE_E_ARG, "lade_attn: fused RoPE needs qkv_parts, cos_tab, sin_tab, max_pos");
        LADE_REQUIRE(a->positions || a->max_pos >= a->mask.T, LADE_E_ARG, "lade_attn: fused RoPE without positions reads table rows 0..T-1 (max_pos=%d, T=%d)", a->max_pos, a->mask.T);
        LADE_REQUIRE(a->dtype == LADE_BF16 || a->dtype == LADE_F16, LADE_E_DTYPE, "lade_attn: fused RoPE is built for bf16 / f16 (dtype=%d)", a->dtype);
        LADE_REQUIRE(a->part_stride >= (int64_t)a->mask.T * (a->H + 2 * a->Hkv) * a->d, LADE_E_ARG, "lade_attn: part_stride %lld is shorter than one partial", (long long)a->part_stride);
        LADE_REQUIRE(!a->sync_flags || a->Hkv <= 256, LADE_E_LIMIT, "lade_attn: the producer mode's flags are reset by one 256-thread block of lade_attn_combine: Hkv=%d > 256", a->Hkv);
        LADE_REQUIRE(!a->sync_flags || (a->q && a->n_splits > 1 && a->q_row_stride % 8 == 0), LADE_E_ARG,
                     "lade_attn: the producer mode of the fused RoPE (sync_flags) writes the rotated q rows to `q` and is reset by lade_attn_combine: it needs q and n_splits > 1");
    } else {
        LADE_REQUIRE(!a->sync_flags, LADE_E_ARG, "lade_attn: sync_flags without the fused RoPE (n_parts = 0)");
    }
    if (!a->mask.is_prefill) {
        const lade_mask_params& m = a->mask;
        LADE_REQUIRE(m.s >= 0 && m.gs > 0 && m.lguess >= 0 && m.lguess % m.gs == 0 && m.level_offset >= 0 && m.dist_offset >= 0,
                     LADE_E_ARG, "lade_attn: bad mask params s=%d gs=%d lguess=%d lo=%d do=%d", m.s, m.gs, m.lguess,
                     m.level_offset, m.dist_offset);
        // s == 0: a lookahead-parallel rank that owns no window column feeds only the L0 prefix (all causal rows)
        const int body = m.T - m.lguess - (m.level_offset + m.dist_offset);
        LADE_REQUIRE(m.layout == 0 || m.layout == 1, LADE_E_ARG, "lade_attn: mask layout=%d", m.layout);
        LADE_REQUIRE(body >= 0 && (m.s > 0 ? body % m.s == 0 : body == 0), LADE_E_ARG,
                     "lade_attn: T=%d is not offsets(%d)+k*s(%d)+lguess(%d)", m.T, m.level_offset + m.dist_offset, m.s, m.lguess);
    }
    return LADE_OK;
}

// floor(x / d) == umulhi(x, magic_for(d)) for x * d < 2^32 (d = 1: magic 0, see div_magic)
static uint32_t magic_for(int d) { return d <= 1 ? 0u : (uint32_t)(((1ull << 32) + (uint64_t)d - 1) / (uint64_t)d); }

static AttnK make_k(const lade_attn_args* a) {
    AttnK k;
    k.q = (const uint16_t*)a->q; k.k = (const uint16_t*)a->k_cache; k.vt = (const uint16_t*)a->vt_cache;
    k.out = (uint16_t*)a->out; k.part_o = (uint16_t*)a->part_o; k.part_ml = a->part_ml; k.dyn_P = a->dyn_P;
    k.q_row_stride = a->q_row_stride; k.out_row_stride = a->out_row_stride;
    k.H = a->H; k.Hkv = a->Hkv; k.S_max = a->S_max; k.n_splits = a->n_splits; k.n_rep = a->H / a->Hkv;
    k.ns_magic = magic_for(a->n_splits); k.hkv_magic = magic_for(a->Hkv); k.t_magic = magic_for(a->mask.T);
    k.inv_T = 1.0f / (float)a->mask.T;
    k.n_groups = 0;                                 // set per work-group shape (launch_fwd_shape)
    k.scale_log2 = a->scale * 1.4426950408889634f;
    { static int dbg = -1; if (dbg < 0) dbg = debug_int("attn_dbg"); k.dbg = dbg; }
    k.m = a->mask;
    k.parts = a->qkv_parts; k.part_stride = (size_t)a->part_stride; k.positions = a->positions;
    k.cos_tab = (const uint16_t*)a->cos_tab; k.sin_tab = (const uint16_t*)a->sin_tab;
    k.k_w = (uint16_t*)a->k_cache; k.vt_w = (uint16_t*)a->vt_cache;
    k.n_parts = a->n_parts; k.max_pos = a->max_pos; k.row_w = (a->H + 2 * a->Hkv) * a->d;
    k.flags = a->n_parts ? a->sync_flags : nullptr;
    k.prod_chunks = cdiv(a->mask.T, PROD_TOK);
    k.n_prod = k.flags ? 8 * cdiv(a->Hkv * k.prod_chunks, 8) : 0;
    return k;
}

template <typename T, int D, int RG, int KQ, int NPC>
static int launch_fwd_shape(const lade_attn_args* a, hipStream_t st) {
    AttnK k = make_k(a);
    const int n_rep = a->H / a->Hkv;
    constexpr int ROWS = 32 * RG, TPS = KQ / 2, NSTG = TPS == 1 ? 3 : 2;
    k.n_groups = cdiv(n_rep * a->mask.T, ROWS) * a->Hkv;
    dim3 grid(k.n_prod + 8 * cdiv(k.n_groups, 8) * a->n_splits);       // [producers |] block_decode: b = x + 8 (q n_splits + sp), group = x + 8 q
    const size_t lds = (size_t)KT * D * 2 * 2 * TPS * NSTG + (size_t)ROWS * D * 2;
    static bool attr_set = false;          // > 64 KiB of dynamic LDS needs the opt-in once per kernel
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<T, D, RG, KQ, NPC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((attn_fwd_kernel<T, D, RG, KQ, NPC>), grid, dim3(64 * RG * KQ), lds, st, k);
    return check_launch("lade_attn_fwd");
}

// The work-group shape is a launch parameter (lade_attn_args.wg_rows: 128 | 64 | 32 query rows of the (head-in-group, token) space per
// work-group; 0 = 128): which one is fastest depends on the rows per KV head, the split count and the cache length, so the caller's autotune
// decides it per launch shape inside a step, like the GEMM configurations (StepEngine._refine_attn).  LADE_DEBUG=attn_shape=<rows> forces one (experiments).
template <typename T, int D, int NPC>
static int launch_fwd_npc(const lade_attn_args* a, hipStream_t st) {
    static int forced = -1;
    if (forced < 0) forced = debug_int("attn_shape");
    const int shape = forced ? forced : (a->wg_rows ? a->wg_rows : 128);
    if (shape == 32) return launch_fwd_shape<T, D, 1, 4, NPC>(a, st);
    // (fused RoPE from 3 or 4 partials: two q items of four partials each + a pass of K / V rows do not fit the 128-row shape's
    // register file - 359 spilled registers -, so a 128-row request runs as 64-row blocks)
    if (shape == 64 || NPC > 2) return launch_fwd_shape<T, D, 2, 4, NPC>(a, st);
    if constexpr (NPC <= 2) return launch_fwd_shape<T, D, 4, 2, NPC>(a, st);
    return LADE_E_LIMIT;
}

template <typename T, int D>
static int launch_fwd(const lade_attn_args* a, hipStream_t st) {
    if (a->n_parts == 0) return launch_fwd_npc<T, D, 0>(a, st);
#ifdef LADE_EXPERIMENTAL
    // RoPE + KV append inside the attention launch (two forms, both bit-identical to the two-launch form and both measured SLOWER at every
    // BASELINE shape: DESIGN 4.9, profiles/r5_fused_rope_ab.txt): built only with -DLADE_EXPERIMENTAL (make EXPERIMENTAL=1)
    return a->n_parts <= 2 ? launch_fwd_npc<T, D, 2>(a, st) : launch_fwd_npc<T, D, 4>(a, st);
#else
    LADE_REQUIRE(false, LADE_E_ARG, "lade_attn: the fused RoPE + KV append forms (n_parts > 0) are not in this build (make EXPERIMENTAL=1; lade_build_flags() bit 0)");
#endif
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
    LADE_REQUIRE((a->n_parts != 0 || a->q_row_stride % 8 == 0) && a->out_row_stride % 8 == 0, LADE_E_ARG, "lade_attn_fwd: row strides must keep 16-B alignment");
    if (a->dtype == LADE_BF16) return a->d == 128 ? launch_fwd<BF16, 128>(a, st) : launch_fwd<BF16, 64>(a, st);
    return a->d == 128 ? launch_fwd<F16, 128>(a, st) : launch_fwd<F16, 64>(a, st);
}

extern "C" int lade_attn_combine(const lade_attn_args* a, void* stream) {
    int rc = validate(a);
    if (rc) return rc;
    LADE_REQUIRE(a->n_splits > 1, LADE_E_ARG, "lade_attn_combine: n_splits=%d", a->n_splits);
    LADE_REQUIRE(a->dtype == LADE_BF16 || a->dtype == LADE_F16, LADE_E_DTYPE, "lade_attn_combine: dtype=%d", a->dtype);
    const AttnK k = make_k(a);
    LADE_REQUIRE(a->n_splits <= 32, LADE_E_LIMIT, "lade_attn_combine: n_splits=%d > 32", a->n_splits);
    static int hpb_env = -1;
    if (hpb_env < 0) hpb_env = debug_int("combine_hpb");
    int hpb = hpb_env > 0 ? hpb_env : 256 / (a->d / 8);     // heads per block
    while (hpb > 1 && a->H % hpb != 0) hpb >>= 1;
    dim3 grid(a->mask.T, a->H / hpb), block(a->d / 8, hpb);
    if (a->dtype == LADE_BF16) hipLaunchKernelGGL(attn_combine_kernel<BF16>, grid, block, 0, (hipStream_t)stream, k, a->d);
    else hipLaunchKernelGGL(attn_combine_kernel<F16>, grid, block, 0, (hipStream_t)stream, k, a->d);
    return check_launch("lade_attn_combine");
}

extern "C" int lade_mask_render(const lade_mask_params* m, uint8_t* out, void* stream) {
    LADE_REQUIRE(m && out && m->T > 0, LADE_E_ARG, "lade_mask_render: bad args");
    hipLaunchKernelGGL(mask_render_kernel, dim3(m->T), dim3(64), 0, (hipStream_t)stream, *m, out);
    return check_launch("lade_mask_render");
}
