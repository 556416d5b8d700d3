// Integer side of one lookahead step, device resident and bit-exact against the reference's python
// list logic: n-gram pool (LRU order included), greedy verify, Jacobi window fill / roll, step
// input assembly, row argmax, and the fused post-step that chains them so a steady step leaves the
// GPU only as one small record.  Reference: lade/decoding.py:37-127, 948-954, 1038-1084, 1119-1177;
// lade/models/modeling_llama.py:1463-1511.
//
// These kernels move a few hundred bytes; they are latency bound, so each is a single 64-lane
// workgroup (one wavefront) with the sequential semantics of the reference preserved where it
// matters: pool inserts with duplicate keys inside one step must happen in column order.
#include <type_traits>

#include "common.hpp"

namespace lade {

// ---- n-gram pool --------------------------------------------------------------------------
// pool_tok [V][G][gs], pool_cnt [V].  The per-key python list (oldest first) is the slot array.
// One wavefront, lane = slot.  `tup` must be wave-uniform.  Ends with a barrier so the next
// insert (possibly to the same key) observes this one.
__device__ void lru_insert(int32_t* pool_tok, int32_t* pool_cnt, int V, int G, int gs, int key, const int32_t* tup) {
    const int lane = threadIdx.x;
    if (G <= 0) return;      // GUESS_SET_SIZE <= 0: the reference never reads the pool (lade/decoding.py:948), nothing to keep
    if (key >= 0 && key < V) {
        const int cnt = pool_cnt[key];
        int32_t* slots = pool_tok + (size_t)key * G * gs;
        bool match = lane < cnt;
        for (int j = 0; j < gs && match; ++j) match = slots[lane * gs + j] == tup[j];
        const uint64_t ball = __ballot(match);
        if (ball == 0ull && cnt < G) {                       // append
            if (lane < gs) slots[cnt * gs + lane] = tup[lane];
            if (lane == 0) pool_cnt[key] = cnt + 1;
        } else {
            // hit at slot `from`: remove it and append (move to end); full without hit: drop the head
            const int from = ball ? (__ffsll((unsigned long long)ball) - 1) : 0;
            int32_t mv[LADE_MAX_LEVEL];
            const bool shifts = lane >= from && lane + 1 < cnt;
            if (shifts)
                for (int j = 0; j < gs; ++j) mv[j] = slots[(lane + 1) * gs + j];
            __syncthreads();
            if (shifts)
                for (int j = 0; j < gs; ++j) slots[lane * gs + j] = mv[j];
            if (lane < gs) slots[(cnt - 1) * gs + lane] = tup[lane];
        }
    }
    __syncthreads();
}

// update_token_map (lade/decoding.py:37-63): W sequential inserts, column i:
// key = lst_token (i = 0) or L0[i-1]; tuple = (L1[i], ..., L(N-2)[i], new_results[i])
__device__ void pool_insert_window(int32_t* pool_tok, int32_t* pool_cnt, int V, int G, int gs, int lst_token,
                                   const int32_t* window, int wcap, const int32_t* new_results, int W, int N, int32_t* tup_sm) {
    for (int i = 0; i < W; ++i) {
        const int key = (i == 0) ? lst_token : window[i - 1];
        if (threadIdx.x < gs) {
            const int j = threadIdx.x;
            tup_sm[j] = (j < gs - 1) ? window[(j + 1) * wcap + i] : new_results[i];
        }
        __syncthreads();
        lru_insert(pool_tok, pool_cnt, V, G, gs, key, tup_sm);
    }
}

__global__ __launch_bounds__(64) void pool_insert_window_kernel(int32_t* pool_tok, int32_t* pool_cnt, int V, int G, int gs,
                                                                const int32_t* lst_token, const int32_t* window, int wcap,
                                                                const int32_t* new_results, int W, int N) {
    __shared__ int32_t tup[LADE_MAX_LEVEL];
    pool_insert_window(pool_tok, pool_cnt, V, G, gs, *lst_token, window, wcap, new_results, W, N, tup);
}

// mode 0: ngrams [n][gs+1] (key, tuple);  mode 1: sliding windows over tokens[0..n) (fill_pool_with_prompt)
__global__ __launch_bounds__(64) void pool_insert_ngrams_kernel(int32_t* pool_tok, int32_t* pool_cnt, int V, int G, int gs,
                                                                const int32_t* src, int n, int mode) {
    __shared__ int32_t tup[LADE_MAX_LEVEL];
    const int N = gs + 1;
    const int count = mode == 0 ? n : n - N + 1;
    for (int i = 0; i < count; ++i) {
        const int32_t* g = mode == 0 ? src + (size_t)i * N : src + i;
        if (threadIdx.x < gs) tup[threadIdx.x] = g[1 + threadIdx.x];
        __syncthreads();
        lru_insert(pool_tok, pool_cnt, V, G, gs, g[0], tup);
    }
}

// lade/decoding.py:948-954: flatten the key's tuples in list order
__device__ int pool_lookup(const int32_t* pool_tok, const int32_t* pool_cnt, int V, int G, int gs, int key, int32_t* guess_out) {
    int cnt = 0;
    if (key >= 0 && key < V && G > 0) cnt = pool_cnt[key];
    const int32_t* slots = pool_tok + (size_t)(key >= 0 && key < V ? key : 0) * G * gs;
    // the row is read whatever the count says (any valid key's row is in bounds): count and row travel together, one memory latency
    for (int idx = threadIdx.x; idx < G * gs; idx += blockDim.x) {
        const int32_t v = slots[idx];
        guess_out[idx] = idx < cnt * gs ? v : 0;
    }
    return cnt;
}

__global__ __launch_bounds__(64) void pool_lookup_kernel(const int32_t* pool_tok, const int32_t* pool_cnt, int V, int G, int gs,
                                                         const int32_t* key, int32_t* guess_out, int32_t* g_out) {
    const int cnt = pool_lookup(pool_tok, pool_cnt, V, G, gs, *key, guess_out);
    if (threadIdx.x == 0) *g_out = cnt;
}

// ---- greedy verify (lade/decoding.py:1071-1084) ------------------------------------------------
// lane e = candidate e.  gg = first index where guess differs from [first_guess]+argmax rows, or
// gs-1 when none differ; the best is the FIRST candidate with the largest gg, and only gg > 0 counts.
__device__ void verify_greedy(int first_guess, const int32_t* guess, const int32_t* guess_argmax, int g, int gs,
                              int* max_hit_out, int* max_hit_idx_out, int32_t* hits /*[gs], wave-visible*/) {
    const int lane = threadIdx.x;
    int gg = 0;
    if (lane < g) {
        // every load of the candidate first (a break inside the loop made them gs dependent memory latencies), then the comparison
        int32_t gv[LADE_MAX_LEVEL], cv[LADE_MAX_LEVEL];
#pragma unroll
        for (int j = 0; j < LADE_MAX_LEVEL; ++j) {
            gv[j] = j < gs ? guess[lane * gs + j] : 0;
            cv[j] = (j == 0) ? first_guess : (j < gs ? guess_argmax[lane * gs + j - 1] : 0);
        }
        gg = gs - 1;
#pragma unroll
        for (int j = LADE_MAX_LEVEL - 1; j >= 0; --j)
            if (j < gs && gv[j] != cv[j]) gg = j;
    }
    int best = gg;
    for (int o = 32; o > 0; o >>= 1) best = max(best, __shfl_xor(best, o));
    const uint64_t who = __ballot(gg == best && lane < g);
    const int idx = (best > 0 && who) ? (__ffsll((unsigned long long)who) - 1) : 0;
    if (lane < gs) {
        int h = 0;
        if (lane == 0) h = first_guess;
        else if (lane <= best && best > 0) h = guess_argmax[idx * gs + lane - 1];
        hits[lane] = h;
    }
    *max_hit_out = best;
    *max_hit_idx_out = idx;
}

__global__ __launch_bounds__(64) void verify_greedy_kernel(const int32_t* first_guess, const int32_t* guess,
                                                           const int32_t* guess_argmax, int g, int gs, int32_t* out2,
                                                           int32_t* hits) {
    int mh, mi;
    verify_greedy(*first_guess, guess, guess_argmax, g, gs, &mh, &mi, hits);
    if (threadIdx.x == 0) { out2[0] = mh; out2[1] = mi; }
}

// ---- window (lade/decoding.py:1038-1066, 1119-1124) ---------------------------------------------
// window [N-1][wcap]; level lengths in ctl[LADE_CTL_WLEN + l]
__device__ void window_fill_first(int32_t* window, int wcap, int32_t* ctl, const int32_t* inp_argmax, int n) {
    // L0 <- L0[1:], L1 <- argmax(inp_logits)
    const int len0 = ctl[LADE_CTL_WLEN];
    int32_t v = 0;
    for (int base = 0; base < len0 - 1; base += 64) {
        const int i = base + threadIdx.x;
        if (i < len0 - 1) v = window[i + 1];
        __syncthreads();
        if (i < len0 - 1) window[i] = v;
        __syncthreads();
    }
    for (int i = threadIdx.x; i < n; i += 64) window[wcap + i] = inp_argmax[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        ctl[LADE_CTL_WLEN] = len0 - 1;
        ctl[LADE_CTL_WLEN + 1] = n;
        ctl[LADE_CTL_FILL_LEVEL] = 1;
    }
    __syncthreads();
}

__device__ void window_fill(int32_t* window, int wcap, int32_t* ctl, int fill_level, const int32_t* inp_argmax, int n) {
    // levels 0..fill_level drop their head; level fill_level+1 <- argmax(inp_logits)[1:]
    for (int l = 0; l <= fill_level; ++l) {
        const int len = ctl[LADE_CTL_WLEN + l];
        int32_t v = 0;
        for (int base = 0; base < len - 1; base += 64) {
            const int i = base + threadIdx.x;
            if (i < len - 1) v = window[l * wcap + i + 1];
            __syncthreads();
            if (i < len - 1) window[l * wcap + i] = v;
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i + 1 < n; i += 64) window[(fill_level + 1) * wcap + i] = inp_argmax[i + 1];
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int l = 0; l <= fill_level; ++l) ctl[LADE_CTL_WLEN + l] -= 1;
        ctl[LADE_CTL_WLEN + fill_level + 1] = n - 1;
        ctl[LADE_CTL_FILL_LEVEL] = fill_level + 1;
    }
    __syncthreads();
}

__global__ __launch_bounds__(64) void window_fill_first_kernel(int32_t* window, int wcap, int32_t* ctl, const int32_t* inp_argmax, int n) {
    window_fill_first(window, wcap, ctl, inp_argmax, n);
}

__global__ __launch_bounds__(64) void window_fill_kernel(int32_t* window, int wcap, int32_t* ctl, int fill_level,
                                                         const int32_t* inp_argmax, int n) {
    window_fill(window, wcap, ctl, fill_level, inp_argmax, n);
}

// L0 <- L1[1:], Lj <- Lj+1 (1 <= j <= N-3), L(N-2) <- new_results
__device__ void window_roll(int32_t* window, int wcap, int32_t* ctl, const int32_t* new_results, int W, int N) {
    for (int l = 0; l < N - 2; ++l) {
        const int off = (l == 0) ? 1 : 0;
        const int len = W - off;
        for (int base = 0; base < len; base += 64) {
            const int i = base + threadIdx.x;
            if (i < len) window[l * wcap + i] = window[(l + 1) * wcap + i + off];
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < W; i += 64) window[(N - 2) * wcap + i] = new_results[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        ctl[LADE_CTL_WLEN] = W - 1;
        for (int l = 1; l < N - 1; ++l) ctl[LADE_CTL_WLEN + l] = W;
    }
}

__global__ __launch_bounds__(64) void window_roll_kernel(int32_t* window, int wcap, int32_t* ctl, const int32_t* new_results, int W, int N) {
    window_roll(window, wcap, ctl, new_results, W, N);
}

// ---- step input assembly (lade/models/modeling_llama.py:1487-1511) ------------------------------
// [inputs | L0[0:c1-1] | L1[c0:c1] | ... | L_fill[c0:c1] | candidates | zero padding to pad_to_T]
// (c0,c1) = this rank's window columns under lookahead parallelism (lade/decoding.py:973-984);
// c1 < 0 = everything.  g < 0: read ctl[LADE_CTL_G].
__global__ __launch_bounds__(256) void build_inputs_kernel(const int32_t* in_ids, const int32_t* in_pos, int n_input,
                                                           const int32_t* window, int wcap, const int32_t* ctl, int fill_level,
                                                           int c0, int c1, const int32_t* guess, int g, int gs, int cand_rows,
                                                           int32_t* ids, int32_t* pos, int32_t* out_T, int lp_rank, int lp_world) {
    if (g < 0) {
        g = ctl[LADE_CTL_G];
        if (lp_world > 1) {        // this rank's share of the candidates, decided on the device (lade/decoding.py:956-963): graph replay
            const int cnt = (g + lp_world - 1) / lp_world;
            const int glo = min(cnt * lp_rank, g), ghi = min(cnt * (lp_rank + 1), g);
            guess += glo * gs;
            g = ghi - glo;
        }
    }
    // inputs default to the control block: the accepted tokens of the last step (hits[0..n_input), or
    // lst_token alone) at positions ending in ctl[LST_POS]
    const int lst_id = in_pos ? in_pos[n_input - 1] : ctl[LADE_CTL_LST_POS];
    for (int i = threadIdx.x; i < n_input; i += blockDim.x) {
        ids[i] = in_ids ? in_ids[i] : (n_input == 1 ? ctl[LADE_CTL_LST_TOKEN] : ctl[LADE_CTL_HITS + i]);
        pos[i] = in_pos ? in_pos[i] : lst_id - (n_input - 1) + i;
    }
    int base = n_input;
    int len0_inp = 0;
    for (int l = 0; l <= fill_level; ++l) {
        const int len = ctl[LADE_CTL_WLEN + l];
        int a, b;
        if (c1 < 0) { a = 0; b = len; }
        else if (l == 0) { a = 0; b = min(c1 - 1, len); }
        else { a = min(c0, len); b = min(c1, len); }
        const int n = max(b - a, 0);
        if (l == 0) len0_inp = n;
        const int off = (l == 0) ? 1 : l + (len0_inp + 1 - n);
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            ids[base + i] = window[l * wcap + a + i];
            pos[base + i] = lst_id + off + i;
        }
        base += n;
    }
    const int rows = cand_rows >= 0 ? cand_rows : g * gs;        // padded mode: always cand_rows rows
    for (int i = threadIdx.x; i < rows; i += blockDim.x) {
        ids[base + i] = i < g * gs ? guess[i] : 0;
        pos[base + i] = lst_id + 1 + (i % gs);
    }
    if (threadIdx.x == 0 && out_T) *out_T = base + rows;
}

// ---- row argmax: first index wins ties (torch.argmax) -------------------------------------------
template <typename T>
__device__ __forceinline__ float ld_logit(const void* p, size_t i);
template <> __device__ __forceinline__ float ld_logit<BF16>(const void* p, size_t i) { return to_f32<BF16>(((const uint16_t*)p)[i]); }
template <> __device__ __forceinline__ float ld_logit<F16>(const void* p, size_t i) { return to_f32<F16>(((const uint16_t*)p)[i]); }
template <> __device__ __forceinline__ float ld_logit<F32>(const void* p, size_t i) { return ((const float*)p)[i]; }

// one block of 1024 threads per row; 16-byte loads when the row is 16-byte aligned
template <typename T>
__global__ __launch_bounds__(1024) void argmax_rows_kernel(const void* logits, int64_t ld, int V, int32_t* out) {
    constexpr int ES = sizeof(T) == 0 ? 2 : 2;
    const int row = blockIdx.x;
    const size_t base = (size_t)row * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    auto upd = [&](float v, int i) {
        if (argmax_better(v, i, best, bi)) { best = v; bi = i; }
    };
    constexpr bool is16 = !__is_same(T, F32);
    constexpr int VEC = is16 ? 8 : 4;
    const size_t ebytes = is16 ? 2 : 4;
    const bool aligned = (((size_t)logits + base * ebytes) & 15) == 0;
    int vend = 0;
    if (aligned) {
        vend = (V / VEC) * VEC;
        // four 16-byte loads of a thread in flight at a time (a row of 32000 logits is four strides of the block: one memory latency
        // instead of four)
        for (int i0 = threadIdx.x * VEC; i0 < vend; i0 += 4 * 1024 * VEC) {
            uint4 uu[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = min(i0 + k * 1024 * VEC, vend - VEC);          // past the end: the row's last chunk again (not used)
                uu[k] = *reinterpret_cast<const uint4*>((const char*)logits + (base + i) * ebytes);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + k * 1024 * VEC;
                if (i < vend) {
                    const uint32_t w[4] = {uu[k].x, uu[k].y, uu[k].z, uu[k].w};
                    if (is16) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            upd(to_f32<typename std::conditional<is16, T, BF16>::type>((uint16_t)(w[e] & 0xffffu)), i + 2 * e);
                            upd(to_f32<typename std::conditional<is16, T, BF16>::type>((uint16_t)(w[e] >> 16)), i + 2 * e + 1);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) upd(__uint_as_float(w[e]), i + e);
                    }
                }
            }
        }
    }
    for (int i = vend + threadIdx.x; i < V; i += 1024) upd(ld_logit<T>(logits, base + i), i);
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (oi != 0x7fffffff && argmax_better(ob, oi, best, bi)) { best = ob; bi = oi; }
    }
    __shared__ float sb[16];
    __shared__ int si[16];
    if ((threadIdx.x & 63) == 0) { sb[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            if (si[w] != 0x7fffffff && argmax_better(sb[w], si[w], best, bi)) { best = sb[w]; bi = si[w]; }
        out[row] = bi;
    }
}

// merge of the (value, column) pairs the lm_head GEMM's argmax epilogue leaves per row and column block (gemm_kernel.hpp, epi == 2):
// one wave per row, every pair of the row requested before the first compare; the lowest column wins among equal values
__global__ __launch_bounds__(64) void argmax_pairs_kernel(const float2* pairs, int nb, int32_t* out) {
    const int row = blockIdx.x, lane = threadIdx.x;
    const float2* p = pairs + (size_t)row * nb;
    constexpr int R = 8;                                   // 512 column blocks per pass (V = 32000 at 96 columns per block: 334)
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int b0 = 0; b0 < nb; b0 += 64 * R) {
        float2 v[R];
#pragma unroll
        for (int k = 0; k < R; ++k) v[k] = p[min(b0 + k * 64 + lane, nb - 1)];
#pragma unroll
        for (int k = 0; k < R; ++k)
            if (b0 + k * 64 + lane < nb) {
                const int oi = __float_as_int(v[k].y);
                if (argmax_better(v[k].x, oi, best, bi)) { best = v[k].x; bi = oi; }
            }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (oi != 0x7fffffff && argmax_better(ob, oi, best, bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) out[row] = bi;
}

// ---- fused post-step of one single-rank greedy step ------------------------------------------------
// am = [out row | n_inp inp rows | cand_rows guess rows] argmax ids.  See include/lade_hip.h.
// phase 0 = prefill step, 1 = window-fill step, 2 = steady step (lade/decoding.py:1038-1130).
// tail: [0] = length, [1..N] = the last <= N tokens of the reference's `all_old_tokens`
// (which, in the greedy path, receives hits[max_hit] once per accepted index: decoding.py:1175).

// EOS scan + POOL_FROM_PROMPT appends (lade/decoding.py:1167-1177); returns n_accept, sets *finished
__device__ int accept_scan(const int32_t* hits, int max_hit, int eos, int pool_from_prompt, int32_t* tail, int32_t* ng,
                           int32_t* pool_tok, int32_t* pool_cnt, int V, int G, int N, int sample_mode, int* finished) {
    const int lane = threadIdx.x;
    const int gs = N - 1;
    int n_accept = max_hit + 1;
    *finished = 0;
    for (int hit_idx = 0; hit_idx <= max_hit; ++hit_idx) {
        if (eos >= 0 && hits[hit_idx] == eos) { n_accept = hit_idx + 1; *finished = 1; break; }
        // greedy appends hits[max_hit] for every accepted index (decoding.py:1175), sampling hits[hit_idx] (:601)
        const int new_lst = sample_mode ? hits[hit_idx] : hits[max_hit];
        if (pool_from_prompt) {
            // all_old_tokens.append(hits[max_hit]); append_new_generated_pool(all_old_tokens[-N:])
            __syncthreads();
            const int len = tail[0];
            int32_t v = 0;
            if (len == N && lane + 1 < N) v = tail[2 + lane];
            __syncthreads();
            if (len == N) { if (lane + 1 < N) tail[1 + lane] = v; if (lane == 0) tail[N] = new_lst; }
            else if (lane == 0) { tail[1 + len] = new_lst; tail[0] = len + 1; }
            __syncthreads();
            if (tail[0] == N) {
                if (lane < N) ng[lane] = tail[1 + lane];
                __syncthreads();
                lru_insert(pool_tok, pool_cnt, V, G, gs, ng[0], ng + 1);
            }
        }
    }
    __syncthreads();
    return n_accept;
}

__global__ __launch_bounds__(64) void greedy_post_step_kernel(int32_t* ctl, int32_t* window, int wcap, int32_t* pool_tok,
                                                              int32_t* pool_cnt, int V, int W, int N, int G, const int32_t* am,
                                                              int n_inp, int32_t* guess, int T_step, int cand_rows, int phase,
                                                              int pool_from_prompt, int32_t* tail, int eos, const int32_t* forced,
                                                              const int32_t* level_override, int32_t* record, int32_t* record_host) {
    __shared__ int32_t tup[LADE_MAX_LEVEL];
    __shared__ int32_t hits[LADE_MAX_LEVEL];
    __shared__ int32_t ng[LADE_MAX_LEVEL + 1];
    __shared__ int32_t rec_s[LADE_REC_WORDS];
    const int gs = N - 1;
    const int lane = threadIdx.x;
    const int g = ctl[LADE_CTL_G];
    const int P = ctl[LADE_CTL_P];
    const int n_input = ctl[LADE_CTL_N_INPUT];
    const int lst_token = ctl[LADE_CTL_LST_TOKEN];
    const int lst_pos = ctl[LADE_CTL_LST_POS];
    const int fill_level = ctl[LADE_CTL_FILL_LEVEL];
    const int first_guess = am[0];
    const int32_t* inp_am = am + 1;
    const int32_t* am_guess = am + 1 + n_inp;
    int max_hit = 0, max_hit_idx = 0;
    // `forced` = {max_hit, max_hit_idx, hits[gs]} decided on the host (sampling verify, lade/decoding.py:484-540)
    if (forced) {
        max_hit = forced[0];
        max_hit_idx = forced[1];
        if (lane < gs) hits[lane] = forced[2 + lane];
        __syncthreads();
    }
    if (phase == 2) {
        if (!forced) {
            verify_greedy(first_guess, guess, am_guess, g, gs, &max_hit, &max_hit_idx, hits);
            __syncthreads();
        }
        // (round 4 staged the W pool rows and the window in LDS - all loads in one latency, the sequential inserts on LDS: measured 26-30 us
        // against 16.6 us for this form; the in-kernel timeline put 2.2 k cycles on every LDS insert (gs is a run-time value: the unrolled
        // guards cost more than the global round trips they replaced) and 3 us on the write-back.  Dropped: profiles/r4_post_step_bench.txt)
        pool_insert_window(pool_tok, pool_cnt, V, G, gs, lst_token, window, wcap, inp_am, W, N, tup);
        window_roll(window, wcap, ctl, inp_am, W, N);
        if (level_override) {                                   // filter_window (lade/decoding.py:131-135, :578-580)
            for (int i = lane; i < W; i += 64)
                if (level_override[i] >= 0) window[(N - 2) * wcap + i] = level_override[i];
            __syncthreads();
        }
    } else {
        if (!forced && lane < gs) hits[lane] = lane == 0 ? first_guess : 0;
        __syncthreads();
        if (phase == 0) window_fill_first(window, wcap, ctl, inp_am, n_inp);
        else window_fill(window, wcap, ctl, fill_level, inp_am, n_inp);
    }
    const int kvcache_len = P + n_input;                 // lade/decoding.py:1154-1165
    const int new_lst = hits[max_hit];
    int finished;
    const int n_accept = accept_scan(hits, max_hit, eos, pool_from_prompt, tail, ng, pool_tok, pool_cnt, V, G, N, forced != nullptr, &finished);
    // next step's candidates (lade/decoding.py:948-954): only once the window is full
    const bool window_full = phase == 2 || ctl[LADE_CTL_FILL_LEVEL] >= N - 2;
    int g_next = 0;
    if (window_full) g_next = pool_lookup(pool_tok, pool_cnt, V, G, gs, new_lst, guess);
    __syncthreads();
    int step_no = 0;
    if (lane == 0) {
        ctl[LADE_CTL_MAX_HIT] = max_hit;
        ctl[LADE_CTL_MAX_HIT_IDX] = max_hit_idx;
        ctl[LADE_CTL_N_ACCEPT] = n_accept;
        ctl[LADE_CTL_FIRST_GUESS] = first_guess;
        ctl[LADE_CTL_KV_SRC] = P + T_step - cand_rows + max_hit_idx * gs;
        ctl[LADE_CTL_KV_DST] = kvcache_len;
        ctl[LADE_CTL_KV_CNT] = max_hit;
        ctl[LADE_CTL_P] = kvcache_len + max_hit;
        ctl[LADE_CTL_LST_TOKEN] = new_lst;
        ctl[LADE_CTL_LST_POS] = lst_pos + max_hit + 1;
        ctl[LADE_CTL_N_INPUT] = 1;
        ctl[LADE_CTL_G] = g_next;
        step_no = ctl[LADE_CTL_STEP] + 1;
        ctl[LADE_CTL_STEP] = step_no;
        rec_s[0] = max_hit;
        rec_s[1] = n_accept;
        rec_s[2] = finished;
        rec_s[3] = g_next;
        rec_s[4] = kvcache_len + max_hit;
        rec_s[5] = max_hit_idx;
        rec_s[6] = first_guess;
        rec_s[7] = step_no;
    }
    if (lane < LADE_MAX_LEVEL) rec_s[8 + lane] = lane < gs ? hits[lane] : 0;
    if (lane < gs) ctl[LADE_CTL_HITS + lane] = hits[lane];
    __syncthreads();
    // the record's last word seals it: a host that POLLS its pinned copy (no stream synchronisation: the copy lands word by word) accepts
    // the record only when the seal matches the step number it waits for and the other 23 words it has read
    if (lane == 0) rec_s[LADE_REC_WORDS - 1] = (int32_t)seal_words((const uint32_t*)rec_s, (uint32_t)step_no);
    __syncthreads();
    if (lane < LADE_REC_WORDS) {
        record[lane] = rec_s[lane];
        // pinned host memory mapped into the device: the stores leave the chip as they are issued (fine-grained host memory is not
        // cached in L2) - no system-scope fence (it would write the XCD's whole dirty L2 back first, several us); the seal is what
        // tells the polling host that all 24 words have landed
        if (record_host) record_host[lane] = rec_s[lane];
    }
}

// ---- lookahead parallelism: one fixed int32 record per rank per step ------------------------------
// rec = [first_guess, n_inp, g_local, 0 | new tokens[split] | argmax ids of the rank's g_local*gs candidate rows]
// (lade/decoding.py:1023-1024, 1088-1107).  The rank does NOT verify its candidates itself: the reference verifies
// against `next_tokens` broadcast from rank 0 (:1024), so the verification runs on the gathered records, against
// rank 0's first token, identically on every rank (16-bit logits may round differently from rank to rank).
__global__ __launch_bounds__(64) void lp_pack_kernel(const int32_t* am_out, const int32_t* am_inp, int n_inp,
                                                     const int32_t* am_guess, int g_local, int gs, int split, int32_t* rec,
                                                     const int32_t* ctl, int lp_rank, int lp_world) {
    if (ctl) {                     // graph replay: the rank's candidate count follows the device-side candidate count
        const int g = ctl[LADE_CTL_G];
        const int cnt = (g + lp_world - 1) / lp_world;
        g_local = min(cnt * (lp_rank + 1), g) - min(cnt * lp_rank, g);
    }
    if (threadIdx.x == 0) { rec[0] = *am_out; rec[1] = n_inp; rec[2] = g_local; rec[3] = 0; }
    for (int i = threadIdx.x; i < split; i += 64) rec[4 + i] = i < n_inp ? am_inp[i] : 0;
    for (int i = threadIdx.x; i < g_local * gs; i += 64) rec[4 + split + i] = am_guess[i];
}

// Every rank runs the same reduction over the gathered records: first_guess from rank 0
// (decoding.py:1024), winner = lowest rank with the largest max_hit (:1090-1096), new tokens = the
// last rank's list in the prefill step (:1045) or the concatenation of all ranks' columns (:1057,
// :1106); then the same window fill / pool insert / roll as the single-rank path.  On a hit the KV
// cache is NOT patched: it is cut back to kvcache_len and the 1+max_hit accepted tokens are re-fed
// next step (lade/decoding.py:1148-1153).
__global__ __launch_bounds__(64) void lp_reduce_apply_kernel(const int32_t* all_rec, int R, int rec_words, int split, int32_t* ctl,
                                                             int32_t* window, int wcap, int32_t* pool_tok, int32_t* pool_cnt, int V,
                                                             int W, int N, int G, int phase, int32_t* guess_all, int32_t* scratch,
                                                             int32_t* record, int pool_from_prompt, int32_t* tail, int eos) {
    __shared__ int32_t tup[LADE_MAX_LEVEL];
    __shared__ int32_t hits[LADE_MAX_LEVEL];
    __shared__ int32_t ng[LADE_MAX_LEVEL + 1];
    const int gs = N - 1;
    const int lane = threadIdx.x;
    const int P = ctl[LADE_CTL_P];
    const int n_input = ctl[LADE_CTL_N_INPUT];
    const int lst_token = ctl[LADE_CTL_LST_TOKEN];
    const int lst_pos = ctl[LADE_CTL_LST_POS];
    const int fill_level = ctl[LADE_CTL_FILL_LEVEL];
    const int first_guess = all_rec[0];                          // rank 0's first token (decoding.py:1024)
    int total = 0;
    for (int r = (phase == 0 ? R - 1 : 0); r < R; ++r) {
        const int n = all_rec[(size_t)r * rec_words + 1];
        for (int i = lane; i < n; i += 64) scratch[total + i] = all_rec[(size_t)r * rec_words + 4 + i];
        total += n;
    }
    // the ranks' candidate shards are consecutive (:956-963): their argmax rows concatenate to the rows of all g candidates
    int32_t* am_all = scratch + R * split;
    int g_tot = 0;
    if (phase == 2) {
        for (int r = 0; r < R; ++r) {
            const int gl = all_rec[(size_t)r * rec_words + 2];
            for (int i = lane; i < gl * gs; i += 64) am_all[g_tot * gs + i] = all_rec[(size_t)r * rec_words + 4 + split + i];
            g_tot += gl;
        }
    }
    __syncthreads();
    // one scan over the candidates in order with "strictly greater wins" = every rank's best, then the lowest rank among
    // the best (:1071-1096)
    int max_hit = 0, win_idx = 0;
    verify_greedy(first_guess, guess_all, am_all, g_tot, gs, &max_hit, &win_idx, hits);
    __syncthreads();
    const int cnt = (g_tot + R - 1) / R;
    const int win = (max_hit > 0 && cnt > 0) ? win_idx / cnt : 0;
    const int kvcache_len = P + n_input;
    const int new_lst = hits[max_hit];
    if (phase == 2) {
        pool_insert_window(pool_tok, pool_cnt, V, G, gs, lst_token, window, wcap, scratch, W, N, tup);
        window_roll(window, wcap, ctl, scratch, W, N);
    } else if (phase == 0) {
        window_fill_first(window, wcap, ctl, scratch, total);
    } else {
        window_fill(window, wcap, ctl, fill_level, scratch, total);
    }
    // EOS scan + POOL_FROM_PROMPT appends, identical on every rank (lade/decoding.py:1167-1177)
    int finished;
    const int n_accept = accept_scan(hits, max_hit, eos, pool_from_prompt, tail, ng, pool_tok, pool_cnt, V, G, N, 0, &finished);
    const bool window_full = phase == 2 || ctl[LADE_CTL_FILL_LEVEL] >= N - 2;
    int g_next = 0;
    if (window_full) g_next = pool_lookup(pool_tok, pool_cnt, V, G, gs, new_lst, guess_all);
    __syncthreads();
    if (lane == 0) {
        ctl[LADE_CTL_MAX_HIT] = max_hit;
        ctl[LADE_CTL_MAX_HIT_IDX] = 0;
        ctl[LADE_CTL_N_ACCEPT] = n_accept;
        ctl[LADE_CTL_FIRST_GUESS] = first_guess;
        ctl[LADE_CTL_KV_CNT] = 0;
        ctl[LADE_CTL_P] = kvcache_len;                           // truncate; the hits are re-fed
        ctl[LADE_CTL_LST_TOKEN] = new_lst;
        ctl[LADE_CTL_LST_POS] = lst_pos + max_hit + 1;
        ctl[LADE_CTL_N_INPUT] = 1 + max_hit;
        ctl[LADE_CTL_G] = g_next;
        ctl[LADE_CTL_STEP] += 1;
        record[0] = max_hit;
        record[1] = n_accept;
        record[2] = finished;
        record[3] = g_next;
        record[4] = kvcache_len;
        record[5] = win;
        record[6] = first_guess;
        record[7] = total;
    }
    if (lane < gs) { ctl[LADE_CTL_HITS + lane] = hits[lane]; record[8 + lane] = hits[lane]; }
}

}  // namespace lade

using namespace lade;

#define POOL_ARGS_OK(fn) \
    LADE_REQUIRE(pool_tok && pool_cnt && V > 0 && G >= 0 && G <= LADE_MAX_GUESS_SET && gs > 0 && gs < LADE_MAX_LEVEL, LADE_E_ARG, \
                 fn ": V=%d G=%d gs=%d (limits: G<=%d, gs<%d)", V, G, gs, LADE_MAX_GUESS_SET, LADE_MAX_LEVEL)

extern "C" int lade_pool_insert_window(int32_t* pool_tok, int32_t* pool_cnt, int32_t V, int32_t G, int32_t gs,
                                       const int32_t* lst_token, const int32_t* window, int32_t wcap,
                                       const int32_t* new_results, int32_t W, int32_t N, void* stream) {
    POOL_ARGS_OK("lade_pool_insert_window");
    LADE_REQUIRE(lst_token && window && new_results && W > 0 && N == gs + 1 && wcap >= W, LADE_E_ARG, "lade_pool_insert_window: W=%d N=%d wcap=%d", W, N, wcap);
    hipLaunchKernelGGL(pool_insert_window_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, pool_tok, pool_cnt, V, G, gs, lst_token, window, wcap, new_results, W, N);
    return check_launch("lade_pool_insert_window");
}

extern "C" int lade_pool_insert_ngrams(int32_t* pool_tok, int32_t* pool_cnt, int32_t V, int32_t G, int32_t gs,
                                       const int32_t* ngrams, int32_t n, void* stream) {
    POOL_ARGS_OK("lade_pool_insert_ngrams");
    LADE_REQUIRE(ngrams && n >= 0, LADE_E_ARG, "lade_pool_insert_ngrams: n=%d", n);
    if (n == 0) return LADE_OK;
    hipLaunchKernelGGL(pool_insert_ngrams_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, pool_tok, pool_cnt, V, G, gs, ngrams, n, 0);
    return check_launch("lade_pool_insert_ngrams");
}

extern "C" int lade_pool_fill_prompt(int32_t* pool_tok, int32_t* pool_cnt, int32_t V, int32_t G, int32_t gs,
                                     const int32_t* tokens, int32_t len, void* stream) {
    POOL_ARGS_OK("lade_pool_fill_prompt");
    LADE_REQUIRE(tokens && len >= 0, LADE_E_ARG, "lade_pool_fill_prompt: len=%d", len);
    if (len < gs + 1) return LADE_OK;
    hipLaunchKernelGGL(pool_insert_ngrams_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, pool_tok, pool_cnt, V, G, gs, tokens, len, 1);
    return check_launch("lade_pool_fill_prompt");
}

extern "C" int lade_pool_lookup(const int32_t* pool_tok, const int32_t* pool_cnt, int32_t V, int32_t G, int32_t gs,
                                const int32_t* key, int32_t* guess_out, int32_t* g_out, void* stream) {
    POOL_ARGS_OK("lade_pool_lookup");
    LADE_REQUIRE(key && guess_out && g_out, LADE_E_ARG, "lade_pool_lookup: null pointer");
    hipLaunchKernelGGL(pool_lookup_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, pool_tok, pool_cnt, V, G, gs, key, guess_out, g_out);
    return check_launch("lade_pool_lookup");
}

extern "C" int lade_verify_greedy(const int32_t* first_guess, const int32_t* guess, const int32_t* guess_argmax, int32_t g,
                                  int32_t gs, int32_t* out2, int32_t* hits, void* stream) {
    LADE_REQUIRE(first_guess && out2 && hits && g >= 0 && g <= LADE_MAX_GUESS_SET && gs > 0 && gs < LADE_MAX_LEVEL && (g == 0 || (guess && guess_argmax)),
                 LADE_E_ARG, "lade_verify_greedy: g=%d gs=%d", g, gs);
    hipLaunchKernelGGL(verify_greedy_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, first_guess, guess, guess_argmax, g, gs, out2, hits);
    return check_launch("lade_verify_greedy");
}

extern "C" int lade_window_fill_first(int32_t* window, int32_t wcap, int32_t* ctl, const int32_t* inp_argmax, int32_t n, void* stream) {
    LADE_REQUIRE(window && ctl && inp_argmax && n > 0 && n <= wcap, LADE_E_ARG, "lade_window_fill_first: n=%d wcap=%d", n, wcap);
    hipLaunchKernelGGL(window_fill_first_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, window, wcap, ctl, inp_argmax, n);
    return check_launch("lade_window_fill_first");
}

extern "C" int lade_window_fill(int32_t* window, int32_t wcap, int32_t* ctl, int32_t fill_level, const int32_t* inp_argmax,
                                int32_t n, void* stream) {
    LADE_REQUIRE(window && ctl && inp_argmax && n > 0 && n - 1 <= wcap && fill_level >= 1 && fill_level + 1 < LADE_MAX_LEVEL, LADE_E_ARG,
                 "lade_window_fill: n=%d wcap=%d fill_level=%d", n, wcap, fill_level);
    hipLaunchKernelGGL(window_fill_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, window, wcap, ctl, fill_level, inp_argmax, n);
    return check_launch("lade_window_fill");
}

extern "C" int lade_window_roll(int32_t* window, int32_t wcap, int32_t* ctl, const int32_t* new_results, int32_t W, int32_t N, void* stream) {
    LADE_REQUIRE(window && ctl && new_results && W > 0 && W <= wcap && N >= 3 && N <= LADE_MAX_LEVEL, LADE_E_ARG, "lade_window_roll: W=%d N=%d wcap=%d", W, N, wcap);
    hipLaunchKernelGGL(window_roll_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, window, wcap, ctl, new_results, W, N);
    return check_launch("lade_window_roll");
}

extern "C" int lade_build_inputs(const int32_t* in_ids, const int32_t* in_pos, int32_t n_input, const int32_t* window, int32_t wcap,
                                 const int32_t* ctl, int32_t fill_level, int32_t c0, int32_t c1, const int32_t* guess, int32_t g,
                                 int32_t gs, int32_t cand_rows, int32_t* ids, int32_t* pos, int32_t* out_T, int32_t lp_rank, int32_t lp_world,
                                 void* stream) {
    LADE_REQUIRE(window && ctl && ids && pos && n_input > 0 && gs > 0 && fill_level >= 0 && fill_level < LADE_MAX_LEVEL,
                 LADE_E_ARG, "lade_build_inputs: n_input=%d gs=%d fill_level=%d", n_input, gs, fill_level);
    LADE_REQUIRE(lp_world >= 1 && lp_rank >= 0 && lp_rank < lp_world, LADE_E_ARG, "lade_build_inputs: lp_rank=%d lp_world=%d", lp_rank, lp_world);
    LADE_REQUIRE(in_ids || n_input <= LADE_MAX_LEVEL, LADE_E_ARG, "lade_build_inputs: n_input=%d needs explicit in_ids", n_input);
    LADE_REQUIRE(guess || (g == 0 && cand_rows <= 0), LADE_E_ARG, "lade_build_inputs: candidates requested without a guess buffer");
    hipLaunchKernelGGL(build_inputs_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, in_ids, in_pos, n_input, window, wcap, ctl, fill_level,
                       c0, c1, guess, g, gs, cand_rows, ids, pos, out_T, lp_rank, lp_world);
    return check_launch("lade_build_inputs");
}

extern "C" int lade_argmax_rows(const void* logits, int64_t ld, int32_t rows, int32_t V, int32_t dtype, int32_t* out, void* stream) {
    LADE_REQUIRE(logits && out && rows >= 0 && V > 0 && ld >= V, LADE_E_ARG, "lade_argmax_rows: rows=%d V=%d ld=%lld", rows, V, (long long)ld);
    if (rows == 0) return LADE_OK;
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case LADE_BF16: hipLaunchKernelGGL(argmax_rows_kernel<BF16>, dim3(rows), dim3(1024), 0, st, logits, ld, V, out); break;
        case LADE_F16: hipLaunchKernelGGL(argmax_rows_kernel<F16>, dim3(rows), dim3(1024), 0, st, logits, ld, V, out); break;
        case LADE_F32: hipLaunchKernelGGL(argmax_rows_kernel<F32>, dim3(rows), dim3(1024), 0, st, logits, ld, V, out); break;
        default: LADE_REQUIRE(false, LADE_E_DTYPE, "lade_argmax_rows: dtype=%d", dtype);
    }
    return check_launch("lade_argmax_rows");
}

extern "C" int lade_argmax_pairs(const float* pairs, int32_t rows, int32_t n_blocks, int32_t* out, void* stream) {
    LADE_REQUIRE(pairs && out && rows >= 0 && n_blocks > 0 && ((size_t)pairs & 7) == 0, LADE_E_ARG, "lade_argmax_pairs: rows=%d n_blocks=%d", rows, n_blocks);
    if (rows == 0) return LADE_OK;
    hipLaunchKernelGGL(argmax_pairs_kernel, dim3(rows), dim3(64), 0, (hipStream_t)stream, (const float2*)pairs, n_blocks, out);
    return check_launch("lade_argmax_pairs");
}

extern "C" uint32_t lade_record_seal(const uint32_t* rec, uint32_t step_no) { return rec ? seal_words(rec, step_no) : 0u; }

extern "C" int lade_greedy_post_step(int32_t* ctl, int32_t* window, int32_t wcap, int32_t* pool_tok, int32_t* pool_cnt, int32_t V,
                                     int32_t W, int32_t N, int32_t G, const int32_t* am, int32_t n_inp, int32_t* guess, int32_t T_step,
                                     int32_t cand_rows, int32_t phase, int32_t pool_from_prompt, int32_t* tail, int32_t eos,
                                     const int32_t* forced, const int32_t* level_override, int32_t* record, int32_t* record_host, void* stream) {
    const int gs = N - 1;
    POOL_ARGS_OK("lade_greedy_post_step");
    LADE_REQUIRE(ctl && window && am && guess && record && W > 0 && W + N - 3 <= wcap && N >= 3 && N <= LADE_MAX_LEVEL && T_step > 0 && cand_rows >= 0 &&
                     phase >= 0 && phase <= 2 && n_inp > 0 && n_inp <= wcap && (phase != 2 || n_inp == W),
                 LADE_E_ARG, "lade_greedy_post_step: W=%d N=%d wcap=%d T=%d phase=%d n_inp=%d", W, N, wcap, T_step, phase, n_inp);
    LADE_REQUIRE(!pool_from_prompt || tail, LADE_E_ARG, "lade_greedy_post_step: POOL_FROM_PROMPT needs the tail buffer");
    hipLaunchKernelGGL(greedy_post_step_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ctl, window, wcap, pool_tok, pool_cnt, V, W, N, G,
                       am, n_inp, guess, T_step, cand_rows, phase, pool_from_prompt, tail, eos, forced, level_override, record, record_host);
    return check_launch("lade_greedy_post_step");
}

extern "C" int lade_lp_pack(const int32_t* am_out, const int32_t* am_inp, int32_t n_inp, const int32_t* am_guess, int32_t g_local,
                            int32_t gs, int32_t split, int32_t* rec, int32_t rec_words, const int32_t* ctl, int32_t lp_rank, int32_t lp_world,
                            void* stream) {
    LADE_REQUIRE(lp_world >= 1 && lp_rank >= 0 && lp_rank < lp_world, LADE_E_ARG, "lade_lp_pack: lp_rank=%d lp_world=%d", lp_rank, lp_world);
    LADE_REQUIRE(am_out && am_inp && rec && n_inp >= 0 && n_inp <= split && gs > 0 && gs < LADE_MAX_LEVEL &&
                     g_local >= 0 && g_local <= LADE_MAX_GUESS_SET && rec_words >= 4 + split + g_local * gs && (g_local == 0 || am_guess),
                 LADE_E_ARG, "lade_lp_pack: n_inp=%d split=%d gs=%d rec_words=%d g=%d", n_inp, split, gs, rec_words, g_local);
    hipLaunchKernelGGL(lp_pack_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, am_out, am_inp, n_inp, am_guess, g_local, gs, split, rec, ctl,
                       lp_rank, lp_world);
    return check_launch("lade_lp_pack");
}

extern "C" int lade_lp_reduce_apply(const int32_t* all_rec, int32_t R, int32_t rec_words, int32_t split, int32_t* ctl, int32_t* window,
                                    int32_t wcap, int32_t* pool_tok, int32_t* pool_cnt, int32_t V, int32_t W, int32_t N, int32_t G,
                                    int32_t phase, int32_t* guess_all, int32_t* scratch, int32_t* record, int32_t pool_from_prompt,
                                    int32_t* tail, int32_t eos, void* stream) {
    const int gs = N - 1;
    POOL_ARGS_OK("lade_lp_reduce_apply");
    LADE_REQUIRE(!pool_from_prompt || tail, LADE_E_ARG, "lade_lp_reduce_apply: POOL_FROM_PROMPT needs the tail buffer");
    LADE_REQUIRE(all_rec && ctl && window && guess_all && scratch && record && R > 0 && rec_words >= 4 + split && W <= wcap && N >= 3 && N <= LADE_MAX_LEVEL,
                 LADE_E_ARG, "lade_lp_reduce_apply: R=%d rec_words=%d split=%d", R, rec_words, split);
    hipLaunchKernelGGL(lp_reduce_apply_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, all_rec, R, rec_words, split, ctl, window, wcap,
                       pool_tok, pool_cnt, V, W, N, G, phase, guess_all, scratch, record, pool_from_prompt, tail, eos);
    return check_launch("lade_lp_reduce_apply");
}
