// Logits warpers of the sampling path on the device: temperature -> top-k -> top-p in ONE launch.
//
// Replaces HF's TemperatureLogitsWarper / TopKLogitsWarper / TopPLogitsWarper as the reference applies them through
// LogitsProcessorList (lade/decoding.py:375-377, :443, :488) - there a division, a topk, a full sort of every [V] row, a softmax, a
// cumsum, a scatter and two masked_fills.  Here one work-group owns one row, keeps it in registers (blocked: a thread owns 32
// consecutive tokens) and finds the two cut-offs by bisection on the ORDER-PRESERVING 32-bit key of a float - 32 counting passes for
// the k-th largest value, 32 mass passes for the nucleus - no sort, nothing of size [V] besides the input and the output.
//
// Semantics reproduced (tests compare with the oracle's warp_logits = the HF warpers, including ties):
//   temperature  x / T                                         (true fp32 division, as `scores / self.temperature`)
//   top-k        remove x < (k-th largest x)                    (values tied with the k-th largest stay)
//   top-p        ascending stable order (ties by token index), p = softmax(x), remove while cumsum(p) <= 1 - top_p, keep the last
// The cumulative mass is accumulated in 64-bit FIXED POINT (exp(x - max) * 2^40): integer sums do not depend on the order of
// summation, so the cut is deterministic; against torch's fp32 cumsum (whose own order is unspecified) it can only differ when the
// mass at a token falls within ~1e-7 of the cut.
#include <stdlib.h>

#include "common.hpp"

namespace lade {

constexpr int WARP_THREADS = 1024;
constexpr int WARP_EPT = 32;                         // tokens per thread: V <= 32768
constexpr int WARP_WAVES = WARP_THREADS / 64;

__device__ __forceinline__ uint32_t float_key(float x) {        // ascending floats <-> ascending unsigned keys
    const uint32_t u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_float(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k); }

template <typename T> __device__ __forceinline__ float load_logit(const void* p, size_t i);
template <> __device__ __forceinline__ float load_logit<F32>(const void* p, size_t i) { return reinterpret_cast<const float*>(p)[i]; }
template <> __device__ __forceinline__ float load_logit<BF16>(const void* p, size_t i) { return to_f32<BF16>(reinterpret_cast<const uint16_t*>(p)[i]); }
template <> __device__ __forceinline__ float load_logit<F16>(const void* p, size_t i) { return to_f32<F16>(reinterpret_cast<const uint16_t*>(p)[i]); }

// Work-group reductions (every thread gets the result).  `red` holds 2 x WARP_WAVES slots used alternately (`turn` flips per call): the
// writes of call n+1 go to the other half, and a wave can only reach call n+2 through the barrier of call n+1 - one barrier per call.
template <typename U>
__device__ __forceinline__ U wg_sum(U v, U* red, int& turn) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    U* r = red + (turn & 1) * WARP_WAVES;
    turn ^= 1;
    if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = v;
    __syncthreads();
    U t = 0;
#pragma unroll
    for (int w = 0; w < WARP_WAVES; ++w) t += r[w];
    return t;
}

template <typename U>
__device__ __forceinline__ U wg_max(U v, U* red, int& turn) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const U w = __shfl_xor(v, o); v = w > v ? w : v; }
    U* r = red + (turn & 1) * WARP_WAVES;
    turn ^= 1;
    if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = v;
    __syncthreads();
    U t = r[0];
#pragma unroll
    for (int w = 1; w < WARP_WAVES; ++w) t = r[w] > t ? r[w] : t;
    return t;
}

template <typename T>
__global__ __launch_bounds__(WARP_THREADS) void warp_rows_kernel(const void* logits, int64_t ld, int V, float temperature, int top_k, float top_p,
                                                                 int skip, float* out) {
    __shared__ unsigned long long red64[2 * WARP_WAVES];
    __shared__ uint32_t red32[2 * WARP_WAVES];
    int t64 = 0, t32 = 0;
    const int row = blockIdx.x, tid = threadIdx.x;
    const int prow = row == 0 ? 0 : row + skip;       // logical row 0 = the out row, then the candidate rows behind `skip` window rows
    const int i0 = tid * WARP_EPT;
    const uint32_t KEY_NEG_INF = 0x007fffffu;         // float_key(-inf); padding slots (token >= V) carry key 0, below every real key

    // ---- load (blocked: a thread owns WARP_EPT consecutive tokens), temperature; the row lives on as order-preserving keys ----
    uint32_t key[WARP_EPT];
#pragma unroll
    for (int e = 0; e < WARP_EPT; ++e) {
        const int i = i0 + e;
        uint32_t k = 0u;
        if (i < V) {
            float v = load_logit<T>(logits, (size_t)prow * ld + i);
            if (temperature != 1.0f) v = v / temperature;
            k = float_key(v);
        }
        key[e] = k;
    }

    // ---- top-k: K = largest key with count(key >= K) >= k, found bit by bit; remove key < K (ties with the k-th largest stay) ----
    if (top_k > 0 && top_k < V) {
        uint32_t K = 0;
        for (int b = 31; b >= 0; --b) {
            const uint32_t t = K | (1u << b);
            uint32_t c = 0;
#pragma unroll
            for (int e = 0; e < WARP_EPT; ++e) c += key[e] >= t ? 1u : 0u;
            if (wg_sum<uint32_t>(c, red32, t32) >= (uint32_t)top_k) K = t;
        }
#pragma unroll
        for (int e = 0; e < WARP_EPT; ++e)
            if (key[e] != 0u && key[e] < K) key[e] = KEY_NEG_INF;
    }

    // ---- top-p ----
    if (top_p < 1.0f) {
        uint32_t kmax = 0;
#pragma unroll
        for (int e = 0; e < WARP_EPT; ++e) kmax = key[e] > kmax ? key[e] : kmax;
        kmax = wg_max<uint32_t>(kmax, red32, t32);
        const float mx = key_float(kmax);
        // fixed-point mass of every token: exp(x - max) * 2^40 (the largest token contributes exactly 2^40; -inf and padding contribute 0)
        unsigned long long E[WARP_EPT];
        unsigned long long z = 0;
#pragma unroll
        for (int e = 0; e < WARP_EPT; ++e) {
            const float ev = key[e] > KEY_NEG_INF ? __expf(key_float(key[e]) - mx) : 0.f;
            E[e] = (unsigned long long)((double)ev * 1099511627776.0 + 0.5);
            z += E[e];
        }
        const unsigned long long Z = wg_sum<unsigned long long>(z, red64, t64);
        // tokens are removed while their cumulative probability (ascending order) is <= 1 - top_p, the bound taken in fp32 as torch
        // compares it; capped one unit below the total: the last token of the order always stays (min_tokens_to_keep = 1)
        const float theta = (float)(1.0 - (double)top_p);
        unsigned long long Theta = (unsigned long long)((double)theta * (double)Z);
        if (Theta >= Z) Theta = Z - 1;
        // K = largest key with mass(key < K) <= Theta: every smaller key is removed entirely, K's own tie group partly
        uint32_t K = 0;
        for (int b = 31; b >= 0; --b) {
            const uint32_t t = K | (1u << b);
            unsigned long long mlt = 0;
#pragma unroll
            for (int e = 0; e < WARP_EPT; ++e) mlt += key[e] < t ? E[e] : 0ull;
            if (wg_sum<unsigned long long>(mlt, red64, t64) <= Theta) K = t;
        }
        unsigned long long below = 0, e_tie = 0;
        uint32_t n_tie_mine = 0;
#pragma unroll
        for (int e = 0; e < WARP_EPT; ++e) {
            below += key[e] < K ? E[e] : 0ull;
            if (key[e] == K) { n_tie_mine++; e_tie = E[e]; }
        }
        const unsigned long long B = wg_sum<unsigned long long>(below, red64, t64);
        const unsigned long long Et = wg_max<unsigned long long>(e_tie, red64, t64);       // the tie group's common mass (equal keys, equal masses)
        // in the stable ascending order the members of K's tie group come by token index: B + j Et <= Theta removes the first c of them
        const unsigned long long c_rm = (Et > 0 && Theta >= B) ? (Theta - B) / Et : 0ull;
        // exclusive prefix of the per-thread tie counts over the work-group (thread order = token order)
        uint32_t incl = n_tie_mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t w = __shfl_up(incl, o); if ((tid & 63) >= o) incl += w; }
        uint32_t* sc = red32 + (t32 & 1) * WARP_WAVES;
        t32 ^= 1;
        if ((tid & 63) == 63) sc[tid >> 6] = incl;
        __syncthreads();
        uint32_t seen = incl - n_tie_mine;
        for (int w = 0; w < (tid >> 6); ++w) seen += sc[w];
#pragma unroll
        for (int e = 0; e < WARP_EPT; ++e) {
            if (key[e] == 0u) continue;
            if (key[e] < K) key[e] = KEY_NEG_INF;
            else if (key[e] == K) { if ((unsigned long long)seen < c_rm) key[e] = KEY_NEG_INF; seen++; }
        }
    }

    // ---- store ----
    float* o = out + (size_t)row * V;
#pragma unroll
    for (int e = 0; e < WARP_EPT; ++e)
        if (i0 + e < V) o[i0 + e] = key_float(key[e]);
}

// ---- vocabularies beyond one work-group's registers (V > 32768: Llama-3-class) ------------------------------------------------------
// The same cut-offs, tie rules and fixed-point masses as above.  The row's keys live in the OUTPUT row (its fp32 slots hold the 32-bit keys
// until the last pass) instead of registers and stay in L2 between the passes; a thread owns groups of 4 consecutive tokens (group q
// belongs to thread q % 1024: 16-byte coalesced accesses when V % 4 == 0) and only ever touches its own; a mass is recomputed from its key
// where the register kernel keeps it; top-k's removals are applied on the fly (a key below the top-k cut reads as -inf).  Every pass over
// the row settles TWO bits of a cut-off - 3 thresholds counted at once, one work-group reduction of 3 values - so top-k and top-p take
// 16 passes each instead of 32.  The last pass walks the tokens in order and removes the first c members of the boundary's tie group: a
// work-group scan per 4096 tokens, taken only until c are found.
constexpr int BIG_BITS = 2;                          // bits of a cut-off settled per pass: 16 passes of 3 thresholds (4 bits = 15 thresholds
                                                     // made the passes VALU bound: 33 us each; 1 bit = 32 passes of loads and barriers)
constexpr int BIG_M = (1 << BIG_BITS) - 1;           // thresholds per pass

template <typename U>
__device__ __forceinline__ void wg_sum_vec(U (&v)[BIG_M], U* red, int& turn) {     // red: [2][WARP_WAVES][BIG_M]
#pragma unroll
    for (int n = 0; n < BIG_M; ++n)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[n] += __shfl_xor(v[n], o);
    U* r = red + (turn & 1) * WARP_WAVES * BIG_M;
    turn ^= 1;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int n = 0; n < BIG_M; ++n) r[(threadIdx.x >> 6) * BIG_M + n] = v[n];
    __syncthreads();
#pragma unroll
    for (int n = 0; n < BIG_M; ++n) {
        U t = 0;
#pragma unroll
        for (int w = 0; w < WARP_WAVES; ++w) t += r[w * BIG_M + n];
        v[n] = t;
    }
}

template <typename T>
__global__ __launch_bounds__(WARP_THREADS) void warp_rows_big_kernel(const void* logits, int64_t ld, int V, float temperature, int top_k, float top_p,
                                                                     int skip, float* out) {
    __shared__ unsigned long long red64[2 * WARP_WAVES];
    __shared__ uint32_t red32[2 * WARP_WAVES];
    __shared__ unsigned long long red64v[2 * WARP_WAVES * BIG_M];
    __shared__ uint32_t red32v[2 * WARP_WAVES * BIG_M];
    int t64 = 0, t32 = 0, t64v = 0, t32v = 0;
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int prow = row == 0 ? 0 : row + skip;
    const uint32_t KEY_NEG_INF = 0x007fffffu;         // float_key(-inf); slots behind the row's end read as key 0, below every real key
    uint32_t* kb = reinterpret_cast<uint32_t*>(out + (size_t)row * V);
    const int NQ = (V + 3) >> 2;                      // groups of 4 tokens
    const bool vec = (V & 3) == 0;                    // then every group is whole and 16-byte aligned (the output is a dense [rows][V] fp32 tensor)
    auto load4 = [&](int q) -> uint4 {
        if (vec) return *reinterpret_cast<const uint4*>(kb + 4 * (size_t)q);
        const int i = 4 * q;
        uint4 r;
        r.x = kb[i];
        r.y = i + 1 < V ? kb[i + 1] : 0u;
        r.z = i + 2 < V ? kb[i + 2] : 0u;
        r.w = i + 3 < V ? kb[i + 3] : 0u;
        return r;
    };

    // one pass over the thread's groups with 8 loads in flight (a pass is latency bound otherwise: 2 loads in flight per thread made it
    // 10 us instead of ~3); groups behind the end read as padding keys (0: counted by nothing)
    constexpr int UNR = 8;
    auto walk = [&](auto&& f) {
        for (int q0 = tid; q0 < NQ; q0 += UNR * WARP_THREADS) {
            uint4 kk[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int q = q0 + u * WARP_THREADS;
                kk[u] = q < NQ ? load4(q) : uint4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) { f(kk[u].x); f(kk[u].y); f(kk[u].z); f(kk[u].w); }
        }
    };

    // ---- load, temperature, keys ----
    uint32_t kmax = 0;
    for (int q = tid; q < NQ; q += WARP_THREADS) {
        uint32_t k[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = 4 * q + e;
            k[e] = 0u;
            if (i < V) {
                float v = load_logit<T>(logits, (size_t)prow * ld + i);
                if (temperature != 1.0f) v = v / temperature;
                k[e] = float_key(v);
                kmax = k[e] > kmax ? k[e] : kmax;
            }
        }
        if (vec) *reinterpret_cast<uint4*>(kb + 4 * (size_t)q) = uint4{k[0], k[1], k[2], k[3]};
        else
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * q + e < V) kb[4 * q + e] = k[e];
    }

    // ---- top-k: Ktop = largest key with count(key >= Ktop) >= k, BIG_BITS bits per pass; keys below it read as -inf from here on ----
    uint32_t Ktop = 0;
    if (top_k > 0 && top_k < V) {
        for (int b = 32 - BIG_BITS; b >= 0; b -= BIG_BITS) {
            uint32_t c[BIG_M];
#pragma unroll
            for (int m = 0; m < BIG_M; ++m) c[m] = 0u;
            walk([&](uint32_t k) {
#pragma unroll
                for (int m = 0; m < BIG_M; ++m) c[m] += k >= (Ktop | ((uint32_t)(m + 1) << b)) ? 1u : 0u;
            });
            wg_sum_vec<uint32_t>(c, red32v, t32v);
            uint32_t best = 0;                         // counts fall as the threshold rises: the largest m that still has k keys at or above it
#pragma unroll
            for (int m = 0; m < BIG_M; ++m)
                if (c[m] >= (uint32_t)top_k) best = (uint32_t)(m + 1);
            Ktop |= best << b;
        }
    }
    auto eff = [&](uint32_t k) { return k < Ktop ? (k == 0u ? 0u : KEY_NEG_INF) : k; };

    // ---- top-p: K = largest key with mass(key < K) <= Theta, BIG_BITS bits per pass ----
    uint32_t K = 0;
    unsigned long long c_rm = 0;
    const bool nucleus = top_p < 1.0f;
    if (nucleus) {
        kmax = wg_max<uint32_t>(kmax, red32, t32);     // (top-k never removes the largest key)
        const float mx = key_float(kmax);
        auto mass = [&](uint32_t k) {                  // of a key that is not removed: exp(x - max) * 2^40
            return (unsigned long long)((double)__expf(key_float(k) - mx) * 1099511627776.0 + 0.5);
        };
        unsigned long long z = 0;
        walk([&](uint32_t k) {
            const uint32_t ke = eff(k);
            if (ke > KEY_NEG_INF) z += mass(ke);
        });
        const unsigned long long Z = wg_sum<unsigned long long>(z, red64, t64);
        const float theta = (float)(1.0 - (double)top_p);
        unsigned long long Theta = (unsigned long long)((double)theta * (double)Z);
        if (Theta >= Z) Theta = Z - 1;
        unsigned long long below_K = 0;                // mass of the keys below K, exact (the chosen threshold's own sum)
        for (int b = 32 - BIG_BITS; b >= 0; b -= BIG_BITS) {
            unsigned long long mlt[BIG_M];             // mlt[m] = mass of the keys in [K, K | (m+1) << b)
#pragma unroll
            for (int m = 0; m < BIG_M; ++m) mlt[m] = 0ull;
            const uint32_t lo = K > KEY_NEG_INF ? K : KEY_NEG_INF + 1u, hi = K | ((uint32_t)BIG_M << b);
            walk([&](uint32_t k) {
                const uint32_t ke = eff(k);
                if (ke >= lo && ke < hi) {              // inside the interval still open: the only keys whose exp is needed in this pass
                    const unsigned long long ms = mass(ke);
#pragma unroll
                    for (int m = 0; m < BIG_M; ++m)
                        if (ke < (K | ((uint32_t)(m + 1) << b))) mlt[m] += ms;
                }
            });
            wg_sum_vec<unsigned long long>(mlt, red64v, t64v);
            uint32_t best = 0;                         // the mass below a threshold grows with it: the largest m still within Theta
#pragma unroll
            for (int m = 0; m < BIG_M; ++m)
                if (below_K + mlt[m] <= Theta) best = (uint32_t)(m + 1);
            if (best) below_K += mlt[best - 1];
            K |= best << b;
        }
        // K is a key that occurs (the first whose own mass takes the cumulative sum beyond Theta): its tie group's common mass is mass(K)
        const unsigned long long Et = K > KEY_NEG_INF ? mass(K) : 0ull;
        c_rm = (Et > 0 && Theta >= below_K) ? (Theta - below_K) / Et : 0ull;
    }

    // ---- last pass, in token order: keys -> values; below K removed; the first c_rm members of K's tie group removed ----
    unsigned long long base = 0;                      // tie members among the tokens walked so far (uniform)
    for (int q0 = 0; q0 < NQ; q0 += WARP_THREADS) {
        const int q = q0 + tid;
        const bool live = q < NQ;
        uint32_t k[4] = {0u, 0u, 0u, 0u};
        if (live) {
            const uint4 k4 = load4(q);
            k[0] = eff(k4.x); k[1] = eff(k4.y); k[2] = eff(k4.z); k[3] = eff(k4.w);
        }
        uint32_t pre = 0xffffffffu;
        if (base < c_rm) {                            // uniform: scan the tie members of these 4096 tokens only until c_rm have been found
            uint32_t cnt = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) cnt += (k[e] != 0u && k[e] == K) ? 1u : 0u;
            uint32_t incl = cnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t w = __shfl_up(incl, o); if (lane >= o) incl += w; }
            uint32_t* sc = red32 + (t32 & 1) * WARP_WAVES;
            t32 ^= 1;
            if (lane == 63) sc[wave] = incl;
            __syncthreads();
            pre = incl - cnt;
            uint32_t total = 0;
#pragma unroll
            for (int w = 0; w < WARP_WAVES; ++w) { pre += w < wave ? sc[w] : 0u; total += sc[w]; }
            const unsigned long long first = base + pre;      // this thread's first tie member is number `first` of the group
            uint32_t seen = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (k[e] != 0u && k[e] == K) { if (first + seen < c_rm) k[e] = KEY_NEG_INF; seen++; }
            base += total;
        }
        if (live) {
            float o4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (nucleus && k[e] < K) k[e] = KEY_NEG_INF;
                o4[e] = key_float(k[e] == 0u ? KEY_NEG_INF : k[e]);
            }
            float* o = reinterpret_cast<float*>(kb) + 4 * (size_t)q;
            if (vec) *reinterpret_cast<float4*>(o) = float4{o4[0], o4[1], o4[2], o4[3]};
            else
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * q + e < V) o[e] = o4[e];
        }
    }
}

}  // namespace lade

using namespace lade;

extern "C" int lade_warp_rows(const void* logits, int64_t ld, int32_t rows, int32_t V, int32_t dtype, float temperature, int32_t top_k,
                              float top_p, int32_t skip, float* out, void* stream) {
    LADE_REQUIRE(logits && out && rows >= 0 && V > 0 && ld >= V && temperature > 0.f && top_k >= 0 && top_p > 0.f && skip >= 0, LADE_E_ARG,
                 "lade_warp_rows: rows=%d V=%d T=%f top_k=%d top_p=%f skip=%d", rows, V, temperature, top_k, top_p, skip);
    LADE_REQUIRE(V <= (1 << 24), LADE_E_LIMIT, "lade_warp_rows: V=%d > %d", V, 1 << 24);
    if (rows == 0) return LADE_OK;
    hipStream_t st = (hipStream_t)stream;
    if (V > WARP_THREADS * WARP_EPT) {                  // the row does not fit one work-group's registers: keys in the output row
        if (dtype == LADE_F32) hipLaunchKernelGGL(warp_rows_big_kernel<F32>, dim3(rows), dim3(WARP_THREADS), 0, st, logits, ld, V, temperature, top_k, top_p, skip, out);
        else if (dtype == LADE_BF16) hipLaunchKernelGGL(warp_rows_big_kernel<BF16>, dim3(rows), dim3(WARP_THREADS), 0, st, logits, ld, V, temperature, top_k, top_p, skip, out);
        else if (dtype == LADE_F16) hipLaunchKernelGGL(warp_rows_big_kernel<F16>, dim3(rows), dim3(WARP_THREADS), 0, st, logits, ld, V, temperature, top_k, top_p, skip, out);
        else LADE_REQUIRE(false, LADE_E_DTYPE, "lade_warp_rows: dtype=%d", dtype);
        return check_launch("lade_warp_rows");
    }
    if (dtype == LADE_F32) hipLaunchKernelGGL(warp_rows_kernel<F32>, dim3(rows), dim3(WARP_THREADS), 0, st, logits, ld, V, temperature, top_k, top_p, skip, out);
    else if (dtype == LADE_BF16) hipLaunchKernelGGL(warp_rows_kernel<BF16>, dim3(rows), dim3(WARP_THREADS), 0, st, logits, ld, V, temperature, top_k, top_p, skip, out);
    else if (dtype == LADE_F16) hipLaunchKernelGGL(warp_rows_kernel<F16>, dim3(rows), dim3(WARP_THREADS), 0, st, logits, ld, V, temperature, top_k, top_p, skip, out);
    else LADE_REQUIRE(false, LADE_E_DTYPE, "lade_warp_rows: dtype=%d", dtype);
    return check_launch("lade_warp_rows");
}
