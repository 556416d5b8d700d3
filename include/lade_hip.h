/*
 * lade_hip.h -- C ABI of liblade_hip.so, the MI355X (gfx950) native hot path of lookahead decoding.
 *
 * Boundary contract:
 *   - plain C: pointers + sizes, no torch / C++ types.  Device pointers are raw HBM addresses
 *     (tensor.data_ptr()), `stream` is a hipStream_t passed as void* (torch's current stream).
 *   - the caller owns every buffer; the library allocates nothing persistent.
 *   - every entry point returns 0 on success or a negative LADE_E_* code and never throws;
 *     arguments are validated on the host before any launch; lade_last_error_string() gives the
 *     text of the last failure on the calling thread.  Asynchronous kernel faults surface at the
 *     caller's next synchronisation, as with any HIP launch.
 *   - all entry points are stateless and re-entrant; ordering is only through `stream`.
 *     Everything is hipGraph-capturable (no allocation / sync inside).
 *
 * What each entry point replaces in the reference (paths relative to the reference repo):
 *   lade_attn_fwd            flash_attn_lade.flash_attn_func(q,k,v,...,lookahead=[7 ints])
 *                            lade/models/modeling_llama.py:705-713 (flash) and the eager op
 *                            :520-541 + the dense mask of :115-207 (evaluated in-kernel here)
 *   lade_attn_combine        (second pass of the split-KV attention; no reference counterpart)
 *   lade_rope_kv_append      apply_rotary_pos_emb + torch.cat KV append
 *                            lade/models/modeling_llama.py:321-346, :510-516
 *   lade_kv_commit           lade/decoding.py:1154-1163 (greedy) / :582-590 (sample)
 *   lade_kv_pack_bshd        (adapter only) K / V as `flash_attn_func` receives them -> this library's cache layout
 *   lade_build_inputs        lade/models/modeling_llama.py:1463-1511 (ids / position_ids assembly)
 *   lade_argmax_rows / lade_argmax_pairs   torch.argmax(outputs.*_logits, dim=-1)  lade/decoding.py:1021,1041,1052,1072,1102
 *                            (pairs: on what the lm_head GEMM's argmax epilogue leaves instead of the logits)
 *   lade_verify_greedy       lade/decoding.py:1071-1084
 *   lade_pool_insert_window  update_token_map             lade/decoding.py:37-63
 *   lade_pool_insert_ngrams  fill_pool_with_prompt / append_new_generated_pool  lade/decoding.py:80-127
 *   lade_pool_lookup         lade/decoding.py:948-954
 *   lade_window_fill_first / lade_window_fill / lade_window_roll   lade/decoding.py:1038-1066, :1119-1124
 *   lade_greedy_post_step    the fused single-rank tail of one steady step: verify + pool insert +
 *                            roll + next lookup + control update   lade/decoding.py:1071-1130,1165
 *   lade_lp_unique_id / lade_lp_comm_create / lade_lp_allgather / lade_lp_comm_count / lade_lp_comm_destroy   RCCL communicator + the step's one
 *                            collective   lade/utils.py:28-33, lade/decoding.py:1024,1057,1090,1096,1106
 *   lade_lp_pack / lade_lp_reduce_apply   the per-step lookahead-parallel exchange record
 *                            lade/decoding.py:1023-1024, 1088-1107 (four pickled object collectives
 *                            -> one fixed int32 all-gather issued by the host through RCCL)
 *   lade_softmax_rows / lade_softmax_gather   probabilities of the sampling verify: one full row (the distribution a token is
 *                            finally drawn from) / the per-candidate draft probabilities of the acceptance loop, gathered on
 *                            the device without materialising guess_probs   lade/decoding.py:484-540
 *   lade_warp_rows           temperature / top-k / top-p warpers in one launch   lade/decoding.py:375-377, :443, :488
 *   lade_rmsnorm / lade_embed_rmsnorm / lade_add_rmsnorm / lade_silu_mul / lade_gather_rows   LlamaRMSNorm (+ residual add), SwiGLU,
 *                            embedding / logits-row gather around the GEMMs
 *                            lade/models/modeling_llama.py:222-227, :360-380, :1164 ("next" row)
 *   lade_gemm_skinny / lade_gemm_skinny_kt / lade_weight_to_ktile / lade_weight_from_ktile / lade_splitk_reduce   the nn.Linear projections of the step at M = T <= 128 rows
 *                            lade/models/modeling_llama.py:360-380 (MLP), :492-494, :558 (q/k/v/o)
 *   lade_rope_kv_append_parts / lade_add_rmsnorm_parts / lade_silu_mul_parts   the same glue ops taking that
 *                            GEMM's fp32 split-K partials as input (the reduction is fused into the consumer)
 *   lade_time_attn / lade_time_attn_rot   hipEvent timing helpers for bench.py (no reference counterpart)
 */
#ifndef LADE_HIP_H
#define LADE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LADE_ABI_VERSION 3      /* 2: lade_attn_args grew (wg_rows, fused RoPE), lade_gemm_skinny* take `ring`, lade_greedy_post_step takes record_host;
                                 * 3: + lade_gemm_ra_kt, lade_build_flags (nothing that existed changed) */

/* error codes */
#define LADE_OK 0
#define LADE_E_ARG (-1)     /* invalid argument (null pointer, unsupported size ...) */
#define LADE_E_DTYPE (-2)   /* unsupported dtype / head_dim for this kernel */
#define LADE_E_LAUNCH (-3)  /* hipLaunch failed (text in lade_last_error_string) */
#define LADE_E_LIMIT (-4)   /* a compiled-in limit exceeded (LADE_MAX_*) */

/* element types of floating-point tensors */
#define LADE_BF16 0
#define LADE_F16 1
#define LADE_F32 2

/* compiled-in limits of the integer kernels */
#define LADE_MAX_LEVEL 16        /* N  (n-gram size) */
#define LADE_MAX_WINDOW 128      /* W + N - 3 */
#define LADE_MAX_GUESS_SET 64    /* G  (candidates per key; one wave lane per slot) */
#define LADE_REC_WORDS (8 + LADE_MAX_LEVEL)      /* int32 words of a step's record */
/* Seal of a step record (its last word), so that a host polling a pinned copy can tell a complete record of step `step_no` from a stale
 * or half-landed one: a multiplicative hash of the step number xor-folded with the other words, each times an odd constant (the
 * definition is seal_words() in csrc/common.hpp, shared by the kernel and this host entry point).  rec: LADE_REC_WORDS words in HOST memory. */
uint32_t lade_record_seal(const uint32_t* rec, uint32_t step_no);

/* ---- control block ------------------------------------------------------------------
 * One int32 array in HBM carries the dynamic state of a sequence between kernels, so a steady
 * step needs no host round trip except reading the small `record` it leaves behind.
 * Layout (indices into int32_t ctl[LADE_CTL_WORDS]): */
#define LADE_CTL_P 0            /* committed KV rows (= past_key_values_length of the coming step) */
#define LADE_CTL_LST_TOKEN 1    /* lst_token: last accepted token = the next input token */
#define LADE_CTL_LST_POS 2      /* position id of that token (= attention_mask length - 1) */
#define LADE_CTL_N_INPUT 3      /* true input tokens fed at the head of the step (1 + guess_skip_dist) */
#define LADE_CTL_G 4            /* candidates to verify in the coming step (0..G) */
#define LADE_CTL_FILL_LEVEL 5   /* fill_level */
#define LADE_CTL_MAX_HIT 6      /* results of the last step */
#define LADE_CTL_MAX_HIT_IDX 7
#define LADE_CTL_N_ACCEPT 8     /* max_hit + 1 */
#define LADE_CTL_STEP 9         /* steps executed */
#define LADE_CTL_KV_SRC 10      /* kv_commit: first source row (step_len - lguess + idx*gs) */
#define LADE_CTL_KV_DST 11      /* kv_commit: first destination row (kvcache_len) */
#define LADE_CTL_KV_CNT 12      /* kv_commit: rows to copy (max_hit, 0 under lookahead parallelism) */
#define LADE_CTL_FIRST_GUESS 13
#define LADE_CTL_HITS 16        /* hits[0..gs) */
#define LADE_CTL_WLEN 32        /* lengths of window levels 0..N-2 */
#define LADE_CTL_WORDS 64

/* ---- attention ----------------------------------------------------------------------- */

/* The lookahead mask in closed form (reference: j_make_causal_mask_multilevel,
 * lade/models/modeling_llama.py:115-207; its flash counterpart gets the same information as
 * lookahead=[window, level, n_guess, kv_cache, fill_offset, guess_offset, 0], :1184-1187).
 * Token order of the T new rows: [n_input inputs | L0 | L1 .. | g*gs candidate tokens] (layout 0, the eager
 * path's order), or with levels >= 1 interleaved column-major [.. | L0 | L1[0] L2[0] .. | L1[1] L2[1] .. | ..]
 * (layout 1: the order the reference feeds its flash kernel, modeling_llama.py:1471-1485); q rows, the new
 * K/V rows and the output rows all follow the chosen order. */
typedef struct lade_mask_params {
    int32_t T;            /* new tokens this step (query rows; keys P..P+T) */
    int32_t P;            /* cached keys, visible to every row */
    int32_t is_prefill;   /* 1: plain causal over the T new tokens (:124-130) */
    int32_t s;            /* level_sizes[-1]                       ("window") */
    int32_t lguess;       /* g*gs candidate rows at the tail */
    int32_t gs;           /* tokens per candidate = N-1 */
    int32_t level_offset; /* n_input-1 = guess_offset */
    int32_t dist_offset;  /* 1 + level_sizes[0] - level_sizes[-1]  (fill_offset = level_offset+dist_offset) */
    int32_t layout;       /* 0 = level-major rows (eager order), 1 = levels >= 1 column-major (flash order) */
} lade_mask_params;

typedef struct lade_attn_args {
    const void* q;          /* [T][H][d] rows, element stride q_row_stride between tokens */
    const void* k_cache;    /* [Hkv][S_max][d]   keys, row-major            */
    const void* vt_cache;   /* [Hkv][d][S_max]   values, TRANSPOSED (key index fastest) */
    void* out;              /* [T][H][d], token stride out_row_stride */
    void* part_o;           /* split-KV partial outputs [n_splits][T][H][d], model dtype (n_splits>1) */
    float* part_ml;         /* [n_splits][H][T][2]  (running max in log2 units, running sum) */
    const int32_t* dyn_P;   /* optional device int32: overrides mask.P at run time (graph replay) */
    int64_t q_row_stride, out_row_stride; /* in elements */
    int32_t H, Hkv, d, S_max;
    int32_t dtype;          /* LADE_BF16 / LADE_F16 (MFMA kernel, d = 64 or 128), LADE_F32 (VALU kernel, d <= 256) */
    int32_t n_splits;       /* >= 1 ; work-groups = ceil(n_rep*T/128) x H/n_rep x n_splits (1-D grid, XCD-aware order) */
    float scale;            /* softmax scale, 1/sqrt(d) */
    lade_mask_params mask;
    /* ---- since LADE_ABI_VERSION 2 ---- */
    int32_t wg_rows;        /* work-group shape: query rows of the (head-in-group, token) row space per work-group - 128 (4 x 2 waves, one 64-key
                             * tile per stage), 64 or 32 (two tiles per stage, every wave computes); 0 = 128.  A launch parameter like the GEMM's
                             * tile shape: the caller's in-step autotune picks it per launch shape together with n_splits */
    /* Fused RoPE + KV append (n_parts > 0; bf16 / f16): q and the step's new K / V rows are taken from the qkv projection's fp32 split-K
     * partials instead of from `q` and the cache - the work of lade_rope_kv_append_parts (lade/models/modeling_llama.py:321-346, :510-516)
     * inside this launch, operation for operation: the rows P .. P+T of k_cache / vt_cache are WRITTEN (by the work-groups whose key range
     * holds them) and the results are bit-identical to the two-launch form.  `q` may be null. */
    int32_t n_parts;        /* 0 = off; 1..4 partials */
    const float* qkv_parts; /* [n_parts][T][(H + 2 Hkv) d], part_stride elements apart */
    int64_t part_stride;
    const int32_t* positions;   /* [T] position ids (rows of the tables), or null: token t uses table row t (per-step gathered rows) */
    const void* cos_tab;    /* [max_pos][d], model dtype; the fused form needs tables whose two halves are equal (emb = cat(freqs, freqs)) */
    const void* sin_tab;
    int32_t max_pos;
    /* Producer mode of the fused form (non-null; needs n_splits > 1 and `q`): instead of every KV split rebuilding its head's q rows, the
     * first work-groups of the grid do the RoPE + append work ONCE per (KV head, 32 tokens) - rotated q rows to `q`, K / V rows to the
     * caches, write-through - and raise sync_flags[kvh]; the attention work-groups of that head request the cache tiles that hold no new
     * row, poll the flag from one lane (bounded: a launch never hangs) and then fetch q and the other tiles.  sync_flags: device
     * int32[Hkv], zero before the first launch; lade_attn_combine (which must follow) zeroes it again. */
    int32_t* sync_flags;
} lade_attn_args;

int lade_attn_fwd(const lade_attn_args* a, void* stream);
/* merges the n_splits partials of lade_attn_fwd into `out` (call only when n_splits > 1) */
int lade_attn_combine(const lade_attn_args* a, void* stream);
/* dense 0/1 rendering of the mask predicate, [T][P+T] bytes on the device (test / debug aid) */
int lade_mask_render(const lade_mask_params* m, uint8_t* out, void* stream);

/* ---- RoPE + KV append, KV commit ----------------------------------------------------- */

/* rope_scaling "dynamic" (LlamaDynamicNTKScalingRotaryEmbedding, lade/models/modeling_llama.py:292-318): the cos / sin rows of ONE step.
 * The reference rebuilds its tables with a sequence-dependent base whenever kv_seq_len = P + T (:502-510) exceeds the longest length seen;
 * here state[0] (device int32, initialised to max_position_embeddings, reset with the sequence) carries that length, inv_tab
 * [n_len][d/2] (fp32, host-built like the reference builds inv_freq) holds the inverse frequencies of a rebuild at length
 * max_position_embeddings + i, and the kernel writes cos_rows / sin_rows [T][d] (model dtype) for the step's T positions - the rope
 * entry points below are then called with positions = 0..T-1 and these rows as their tables.  len_hint: a longer length the step
 * belongs to (chunked prefill: the reference sees the whole prompt at once), 0 otherwise.  P from dyn_P when non-null.  g_dev (nullable):
 * the step is a hipGraph step padded to gcap candidates of gs rows of which only *g_dev exist - the length counts the real rows. */
int lade_rope_rows_dynamic(const int32_t* positions, int32_t T, int32_t P, const int32_t* dyn_P, int32_t len_hint, int32_t* state,
                           int32_t max_position_embeddings, const float* inv_tab, int32_t n_len, int32_t d, void* cos_rows, void* sin_rows,
                           int32_t dtype, const int32_t* g_dev, int32_t gcap, int32_t gs, void* stream);

/* qkv: [T][(H+2*Hkv)*d] fused projection output.  Rotates q in place and writes the rotated k
 * and v of token t into cache row P+t.  cos/sin: [max_pos][d] tables in the model dtype, built
 * as the reference builds them (fp32 math, then cast; modeling_llama.py:248-256); the rotation
 * reproduces the reference's per-op rounding: round(q*cos) + round(rotate_half(q)*sin), rounded.
 * positions: device int32[T] (explicit, non-monotone position ids).  P from `dyn_P` if non-null. */
int lade_rope_kv_append(void* qkv, const int32_t* positions, const void* cos_tab, const void* sin_tab,
                        void* k_cache, void* vt_cache, int32_t T, int32_t P, const int32_t* dyn_P,
                        int32_t H, int32_t Hkv, int32_t d, int32_t S_max, int32_t max_pos, int32_t dtype,
                        void* stream);

/* Copies `cnt` K/V rows src..src+cnt -> dst..dst+cnt in every layer of a [L][2] cache whose
 * K part is [Hkv][S_max][d] and V part [Hkv][d][S_max]; layer_stride / v_offset in elements.
 * If `ctl` is non-null, (src,dst,cnt) are read from ctl[LADE_CTL_KV_SRC..KV_CNT] on the device.  max_cnt is reserved
 * (pass 0). */
int lade_kv_commit(void* cache, int64_t layer_stride, int64_t v_offset, int32_t L, int32_t Hkv, int32_t d,
                   int32_t S_max, int32_t src, int32_t dst, int32_t cnt, const int32_t* ctl, int32_t max_cnt,
                   int32_t elem_bytes, void* stream);

/* K / V in the layout the reference hands its flash kernel - [S][Hkv][d], token-major: `flash_attn_func(q, k, v, ...)` at
 * lade/models/modeling_llama.py:705-713 after the transposes of :636-638 - re-laid into this library's cache layout: k_cache
 * [Hkv][S_max][d], vt_cache [Hkv][d][S_max] (rows >= S untouched).  k and v share their strides (elements): tok_stride between tokens,
 * head_stride between heads - Hkv*d and d for a contiguous [S][Hkv][d] tensor, d and S*d for the transposed VIEW of [Hkv][S][d] the
 * reference really passes (:636-638 transpose without a copy).  One launch, every byte read and written once: the adapter
 * `lookaheaddecoding_amd/flash_attn_lade.py:flash_attn_func` calls it per attention call, which costs about what the reference's own
 * per-layer torch.cat of the whole cache costs (:626-629); the step engine never does - its cache is resident in this layout. */
int lade_kv_pack_bshd(const void* k, const void* v, int64_t tok_stride, int64_t head_stride, void* k_cache, void* vt_cache, int32_t S,
                      int32_t Hkv, int32_t d, int32_t S_max, int32_t elem_bytes, void* stream);

/* ---- integer path (all bit-exact against the reference's python logic) ---------------- */

/* window: int32 [N-1][wcap] (wcap >= W+N-3), level lengths in ctl[LADE_CTL_WLEN+l].
 * pool_tok: int32 [V][G][gs], pool_cnt: int32 [V]. */

/* ids/pos <- [n_input inputs | L0[0:c1-1] | L1[c0:c1] .. L_fill[c0:c1] | candidates].  The input tokens
 * and their positions are read from in_ids/in_pos (device); when null they come from the control block:
 * ids = lst_token (n_input = 1) or hits[0..n_input) (re-fed accepted tokens under lookahead parallelism),
 * positions = the n_input positions ending at ctl[LADE_CTL_LST_POS].  (c0,c1) = this rank's window columns under lookahead
 * parallelism (lade/decoding.py:973-984), c1 < 0 = all columns.  g < 0 reads ctl[LADE_CTL_G].
 * cand_rows >= 0 emits exactly that many candidate rows (zero tokens beyond g*gs: fixed-shape graph
 * replay), cand_rows < 0 emits g*gs.  out_T[0] = total tokens written (may be null).  lp_world > 1 with g < 0: the candidates
 * emitted are rank lp_rank's share of ctl[LADE_CTL_G] (lade/decoding.py:956-963), decided on the device; otherwise pass 0, 1. */
int lade_build_inputs(const int32_t* in_ids, const int32_t* in_pos, int32_t n_input, const int32_t* window,
                      int32_t wcap, const int32_t* ctl, int32_t fill_level, int32_t c0, int32_t c1,
                      const int32_t* guess, int32_t g, int32_t gs, int32_t cand_rows, int32_t* ids, int32_t* pos,
                      int32_t* out_T, int32_t lp_rank, int32_t lp_world,
                      void* stream);

/* one argmax per row, first index wins ties (torch.argmax semantics); logits [rows][V] with row
 * stride `ld` elements. */
int lade_argmax_rows(const void* logits, int64_t ld, int32_t rows, int32_t V, int32_t dtype, int32_t* out,
                     void* stream);

/* out[r] = column of the best of row r's n_blocks {value, column} pairs (lade_gemm_skinny, epilogue 2): the same ids lade_argmax_rows
 * returns on the materialised logits (first index wins ties). */
int lade_argmax_pairs(const float* pairs, int32_t rows, int32_t n_blocks, int32_t* out, void* stream);

/* out3 = {max_hit, max_hit_idx}, hits[gs]; g may also come from ctl[LADE_CTL_G] when dyn != 0 */
int lade_verify_greedy(const int32_t* first_guess, const int32_t* guess, const int32_t* guess_argmax, int32_t g,
                       int32_t gs, int32_t* out2, int32_t* hits, void* stream);

int lade_pool_insert_window(int32_t* pool_tok, int32_t* pool_cnt, int32_t V, int32_t G, int32_t gs,
                            const int32_t* lst_token, const int32_t* window, int32_t wcap, const int32_t* new_results,
                            int32_t W, int32_t N, void* stream);
/* n sequential inserts; ngrams [n][N]: key then gs tokens */
int lade_pool_insert_ngrams(int32_t* pool_tok, int32_t* pool_cnt, int32_t V, int32_t G, int32_t gs,
                            const int32_t* ngrams, int32_t n, void* stream);
/* sliding N-grams over tokens[0..len) == fill_pool_with_prompt */
int lade_pool_fill_prompt(int32_t* pool_tok, int32_t* pool_cnt, int32_t V, int32_t G, int32_t gs,
                          const int32_t* tokens, int32_t len, void* stream);
/* guess_out [G*gs], g_out[0] = number of tuples stored under *key (0 if none) */
int lade_pool_lookup(const int32_t* pool_tok, const int32_t* pool_cnt, int32_t V, int32_t G, int32_t gs,
                     const int32_t* key, int32_t* guess_out, int32_t* g_out, void* stream);

int lade_window_fill_first(int32_t* window, int32_t wcap, int32_t* ctl, const int32_t* inp_argmax, int32_t n,
                           void* stream);
int lade_window_fill(int32_t* window, int32_t wcap, int32_t* ctl, int32_t fill_level, const int32_t* inp_argmax,
                     int32_t n, void* stream);
int lade_window_roll(int32_t* window, int32_t wcap, int32_t* ctl, const int32_t* new_results, int32_t W, int32_t N,
                     void* stream);

/* Fused tail of one single-rank greedy step.  am = argmax rows in the order
 * [out row | n_inp inp rows | cand_rows guess rows].  phase 0 = prefill step (L1 <- inp rows),
 * 1 = window-fill step, 2 = steady step.  Does: verify (g = ctl[G], phase 2) -> pool insert (W n-grams)
 * -> window fill / roll -> EOS scan + POOL_FROM_PROMPT appends (`tail` = [len, last <=N tokens of the
 * reference's all_old_tokens]) -> lookup of the next step's candidates into `guess` / ctl[G] -> ctl update
 * (P, lst_token, lst_pos, kv-commit triple, hits, step).
 * record = {max_hit, n_accept, finished_by_eos, g_next, P_next, max_hit_idx, first_guess, step number (ctl[STEP] after this step),
 * hits[gs], 0 ..., seal} - LADE_REC_WORDS words, the last one lade_record_seal() of the others.  record_host (nullable): the same
 * words are ALSO stored to this address - pinned host memory mapped into the device - so that the host can poll for the step's record
 * instead of synchronising the stream (no copy node, no blocking wait in a steady step).  NO fence follows those stores (fine-grained host
 * memory takes them as they are issued, in any order): the seal, computed over all the other words, is the only ordering guarantee - a
 * reader accepts a record only when the step number AND the seal match what it read, and must bound its wait (memory the device's stores do
 * not reach coherently never shows the record).
 * eos < 0 disables the EOS scan; the scan follows lade/decoding.py:1167-1177.
 * Sampling (lade/decoding.py:137-692): the rejection-sampling verify runs on the host (it consumes the python
 * and torch RNG streams in the reference's order); its result is passed as forced = {max_hit, max_hit_idx,
 * hits[gs]} (device) and replaces the greedy verify; level_override[W] (device, -1 = keep) carries the
 * filter_window() replacements of EOS tokens in the newest window level.  Both are null for greedy. */
int lade_greedy_post_step(int32_t* ctl, int32_t* window, int32_t wcap, int32_t* pool_tok, int32_t* pool_cnt,
                          int32_t V, int32_t W, int32_t N, int32_t G, const int32_t* am, int32_t n_inp,
                          int32_t* guess, int32_t T_step, int32_t cand_rows, int32_t phase, int32_t pool_from_prompt,
                          int32_t* tail, int32_t eos, const int32_t* forced, const int32_t* level_override,
                          int32_t* record, int32_t* record_host, void* stream);

/* lookahead parallelism: record packing, then (after the all-gather of rec_words int32 per rank, lade_lp_allgather
 * or the host's collective) the deterministic reduction every rank applies (same phases as above).
 * rec = [first_guess, n_inp, g_local, 0 | new tokens[split] | argmax ids of the rank's g_local*gs candidate rows];
 * rec_words >= 4 + split + g_local*gs.  The candidates are verified inside lade_lp_reduce_apply, on the gathered rows,
 * against RANK 0's first token - the reference broadcasts `next_tokens` from rank 0 before verifying
 * (lade/decoding.py:1024, :1071-1096) - so every rank takes the same decision even when 16-bit logits round
 * differently from rank to rank.  scratch: int32[R*split + G*gs].  pool_from_prompt / tail / eos: as in
 * lade_greedy_post_step (the EOS scan and the POOL_FROM_PROMPT appends of lade/decoding.py:1167-1177 run identically on
 * every rank); record[1] = n_accept, record[2] = finished, record[5] = winning rank.  lade_lp_pack with ctl != NULL derives g_local
 * from ctl[LADE_CTL_G] and (lp_rank, lp_world) on the device (graph replay; pass g_local = the padded row count's candidates). */
int lade_lp_pack(const int32_t* am_out, const int32_t* am_inp, int32_t n_inp, const int32_t* am_guess, int32_t g_local,
                 int32_t gs, int32_t split, int32_t* rec, int32_t rec_words, const int32_t* ctl, int32_t lp_rank, int32_t lp_world,
                 void* stream);
int lade_lp_reduce_apply(const int32_t* all_rec, int32_t R, int32_t rec_words, int32_t split, int32_t* ctl,
                         int32_t* window, int32_t wcap, int32_t* pool_tok, int32_t* pool_cnt, int32_t V, int32_t W,
                         int32_t N, int32_t G, int32_t phase, int32_t* guess_all, int32_t* scratch, int32_t* record,
                         int32_t pool_from_prompt, int32_t* tail, int32_t eos, void* stream);

/* The step's one collective and its communicator: RCCL over xGMI, bound at run time by soname (the process
 * keeps the RCCL instance it already carries, e.g. torch's).  Replaces dist.init_process_group (lade/utils.py:28-33) and the
 * per-step object collectives of lade/decoding.py:1024, :1057, :1090, :1096, :1106 for a caller without torch.distributed.
 *   lade_lp_unique_id    rank 0: 128 opaque bytes (ncclUniqueId) to hand to every rank over the host's own channel
 *   lade_lp_comm_create  every rank, after hipSetDevice on its GPU; *comm_out = opaque handle, the only persistent allocation of
 *                        the library; one handle per rank / process
 *   lade_lp_allgather    recv[r*words .. ] = rank r's send[0..words): ordered on `stream` (after lade_lp_pack, before
 *                        lade_lp_reduce_apply), in place allowed when send == recv + rank*words
 *   lade_lp_comm_count   number of ranks the communicator spans (ncclCommCount) - what `dist.get_world_size()` is for a caller of the
 *                        reference (lade/utils.py:33 asserts it against DIST_WORKERS); lade_lp_comm_create already refuses a
 *                        communicator whose span differs from `world`
 *   lade_lp_comm_destroy explicit release (the handle is freed even when RCCL reports an error)
 * Without an RCCL library in the process they return LADE_E_LIMIT. */
int lade_lp_unique_id(void* id128);
int lade_lp_comm_create(const void* id128, int32_t rank, int32_t world, void** comm_out);
int lade_lp_allgather(void* comm, const int32_t* send, int32_t* recv, int32_t words_per_rank, void* stream);
int lade_lp_comm_count(void* comm, int32_t* ranks_out);
int lade_lp_comm_destroy(void* comm);

/* ---- sampling helpers ------------------------------------------------------------------ */
/* probs[r][:] = softmax(logits[r][:] / temperature) in fp32 */
int lade_softmax_rows(const void* logits, int64_t ld, int32_t rows, int32_t V, int32_t dtype, float temperature,
                      float* probs, void* stream);
/* Device side of the sampling verify (lade/decoding.py:484-540; K11 of SURVEY 8b).  logits: [..][ld] with logical row 0 = the
 * out row (physical row 0) and logical row 1 + c*gs + j = position j of candidate c (physical row 1 + c*gs + j + skip: the step's
 * logits hold the `skip` window rows between the out row and the candidate rows); padded candidate slots are ignored.  Per row the
 * softmax statistics of logits/temperature go to stats[row] = {max, sum of exponentials}; the probabilities the acceptance
 * loop consults are gathered into scal[row][g_cap]: row 0 -> P(first token of candidate c), row 1 + c'*gs + j (j < gs-1) ->
 * P(token j+1 of candidate c | the prefix candidate c' shares up to j).  guess: the candidates' tokens [g][gs] (device);
 * g is read from *g_dev when that is not null (hipGraph steps), capped by g_cap.  Nothing of size [rows][V] is written. */
int lade_softmax_gather(const void* logits, int64_t ld, int32_t rows, int32_t V, int32_t dtype, float temperature, int32_t skip,
                        const int32_t* guess, const int32_t* g_dev, int32_t g, int32_t gs, int32_t g_cap, float* scal,
                        float* stats, void* stream);

/* Logits warpers of the sampling path in one launch (lade/decoding.py:375-377: the reference admits exactly HF's Temperature, TopK and
 * TopP warpers, applied in that order through LogitsProcessorList at :443 / :488):  out[r][:] = top_p(top_k(logits[r'][:] / temperature))
 * in fp32 with removed tokens set to -inf, r' = r for r = 0 and r + skip behind it (the same row addressing as lade_softmax_gather).
 * top_k = 0 and top_p >= 1 switch the respective filter off.  Tie rules of the HF warpers: values equal to the k-th largest stay; the
 * nucleus cut walks the ascending order with ties in token order and always keeps the last token.  Up to V = 32768 one work-group holds a
 * row in registers; a larger vocabulary (Llama-3 class) keeps the row's keys in the output row and settles two bits of a cut-off per pass
 * (same results; 0.4 ms for 31 rows of 128256 against 0.8 ms of torch topk / sort).  V <= 2^24. */
int lade_warp_rows(const void* logits, int64_t ld, int32_t rows, int32_t V, int32_t dtype, float temperature, int32_t top_k,
                   float top_p, int32_t skip, float* out, void* stream);

/* ---- glue around the GEMMs -------------------------------------------------------------- */
/* y = weight * (x * rsqrt(mean(x^2) + eps)) with the reference's rounding (fp32 norm, cast, then * weight) */
int lade_rmsnorm(const void* x, const void* weight, void* y, int32_t rows, int32_t hidden, float eps, int32_t dtype,
                 void* stream);
/* the embedding lookup of a step (lade/models/modeling_llama.py:1413 `embed_tokens(input_ids)`) fused with the first layer's input norm
 * (:857): x[r] = table[ids[r]] (ids clamped to the table), y[r] = rmsnorm(x[r]) - one launch instead of a row gather and a norm */
int lade_embed_rmsnorm(const void* table, int32_t table_rows, const int32_t* ids, void* x, const void* weight, void* y, int32_t rows,
                       int32_t hidden, float eps, int32_t dtype, void* stream);
/* y = x + r (residual add) fused with the norm of the sum: x <- x + r ; y = rmsnorm(x) */
int lade_add_rmsnorm(void* x, const void* r, const void* weight, void* y, int32_t rows, int32_t hidden, float eps,
                     int32_t dtype, void* stream);
/* The row-pruned tail of a step (the reference norms and projects all T rows, lade/models/modeling_llama.py:1541-1544; only the out row,
 * the last level's rows and the candidate rows are ever read): y[j] = rmsnorm(x[sel[j]] + res[sel[j]]) for j < n_sel, res = r or the sum of
 * n_parts fp32 split-K partials (part_stride elements apart) in split order; x is NOT written back.  One launch instead of a partial
 * fold, two row gathers and a norm. */
int lade_add_rmsnorm_rows(const void* x, const void* r, const float* parts, int32_t n_parts, int64_t part_stride, const int32_t* sel,
                          const void* weight, void* y, int32_t n_sel, int32_t src_rows, int32_t hidden, float eps, int32_t dtype,
                          void* stream);
/* SwiGLU of LlamaMLP (lade/models/modeling_llama.py:360-380) on the output of the fused gate/up GEMM, rounded like the separate torch ops.
 * layout 0: gu rows are [gate (inter) | up (inter)]: out[r][i] = silu(gu[r][i]) * gu[r][inter + i].
 * layout 1: groups of 16 - [16 gate | their 16 up] per 32 columns: out[r][i] = silu(gu[r][32*(i/16) + i%16]) * gu[r][32*(i/16) + 16 + i%16]
 *           (the row order in which the step engine fuses the two weights, so that lade_gemm_skinny's SwiGLU epilogue is lane-local). */
int lade_silu_mul(const void* gu, void* out, int32_t rows, int32_t inter, int32_t layout, int32_t dtype, void* stream);
/* dst[r][:] = src[idx[r]][:]  (row gather: embedding lookup, logits-row selection) */
int lade_gather_rows(const void* src, const int32_t* idx, void* dst, int32_t rows, int32_t width, int32_t elem_bytes,
                     int32_t src_rows, void* stream);

/* ---- skinny weight-streaming GEMM -------------------------------------
 * C[M,N] = A[M,K] . W[N,K]^T for the projections of a decode step (M = T <= ~256 rows), bf16 / f16, fp32 accumulate.
 * n_split == 1: writes C (model dtype).  n_split > 1: writes fp32 partials Cpart[n_split][M][N] (summed in split
 * order - deterministic - by lade_splitk_reduce).  bn = weight rows per work-group (32..256), mb = 32-row activation
 * blocks per work-group (1: 32 rows, 2: 64, 3: 96, 4: 128, 0: by M; larger M runs as several row blocks).  mt = 32-row activation blocks per WAVE (0 | 1..4, divides
 * mb) and nt = 32-row weight tiles per wave (0 = fewest): the waves form an (mb/mt) x (bn/32/nt) grid; larger wave tiles
 * re-read less from LDS per weight byte.  ring = stages of the LDS ring the tiles arrive in (0 = default: 4 where they fit; 2, 3, 4, 5, 6, 8): a
 * deeper ring keeps more bytes in flight per work-group and leaves room for fewer co-resident work-groups - chosen per projection by the
 * caller's autotune.  Unsupported shapes return LADE_E_ARG.  K % 64 == 0.  M > 256 runs as several 256-row blocks per launch (bn = 256,
 * mb = 8, nt = 2 | 4 is a compute-shaped 256 x 256 tile on a double buffer).
 * epilogue (n_split == 1 only): 0 = none; 1 = SwiGLU - W is the fused gate/up weight in the 16-row interleaved order of
 * lade_silu_mul's layout 1 and C is [M][N/2] = silu(gate) * up (the SwiGLU kernel and the fp32 partials of a split-K
 * gate/up GEMM disappear from the step); 2 = row argmax (the lm_head of a greedy step, lade/models/modeling_llama.py:1541-1544 +
 * torch.argmax at lade/decoding.py:1021): C is not written (may be null), Cpart receives one {float value, int32 column} pair per row
 * and column block - Cpart[(m * ceil(N / bn) + block) * 2] - the value rounded to the model dtype, the lowest column among equal
 * values; lade_argmax_pairs merges a row's pairs.  The [rows, V] logits never reach HBM. */
int lade_gemm_skinny(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, float* Cpart,
                     int32_t M, int32_t N, int32_t K, int32_t n_split, int32_t bn, int32_t mb, int32_t mt, int32_t nt,
                     int32_t ring, int32_t epilogue, int32_t dtype, void* stream);
/* The same GEMM on a weight stored K-TILE-MAJOR: Wkt[K/64][N][64], i.e. Wkt[kt][n][j] = W[n][64 kt + j] - the 128-byte segments of all N
 * rows of one 64-deep K tile are contiguous.  Row-major nn.Linear weights (lade/models/modeling_llama.py:360-380, 492-494, 558) make a
 * work-group's K tile BN separate 128-byte reads K elements apart; K-tile-major makes it ONE contiguous BN x 128 bytes, and the
 * work-groups of a split sweep memory linearly as they walk along K.  Same arithmetic in the same order: results are bit-identical to
 * lade_gemm_skinny.  lade_weight_to_ktile builds the copy once at load time (out of place; K % 64 == 0). */
int lade_gemm_skinny_kt(const void* A, int64_t lda, const void* Wkt, void* C, int64_t ldc, float* Cpart,
                        int32_t M, int32_t N, int32_t K, int32_t n_split, int32_t bn, int32_t mb, int32_t mt, int32_t nt,
                        int32_t ring, int32_t epilogue, int32_t dtype, void* stream);
/* Split-K GEMM with REGISTER-RESIDENT activations for steps of M <= 128 rows (the same projections, lade/models/modeling_llama.py:360-380,
 * 492-494, 558): a work-group is persistent over the weight rows of its column group and keeps the activation fragments of its K slice
 * (<= 20 K tiles = 1280 deep: ceil(K / 64 / n_split) <= 20) in registers, so the activations pass through a CU once per work-group and the
 * whole LDS is a ring of weight tiles.  Writes fp32 partials Cpart[n_split][M][N] for a `*_parts` consumer - bit-identical to
 * lade_gemm_skinny_kt's with the same n_split.  cs = 32-row weight strips per chunk (2 | 4; N % (32 cs) == 0), n_groups = column groups
 * per split (0 = CUs / n_split; the launch has n_groups * n_split work-groups, one per CU). */
int lade_gemm_ra_kt(const void* A, int64_t lda, const void* Wkt, float* Cpart, int32_t M, int32_t N, int32_t K, int32_t n_split,
                    int32_t cs, int32_t n_groups, int32_t dtype, void* stream);
int lade_weight_to_ktile(const void* W, int64_t ldw, void* Wkt, int32_t N, int32_t K, int32_t dtype, void* stream);
/* the inverse, into a caller-provided row-major scratch W[N][ldw]: what a library GEMM needs (prefill chunks wider than 256 rows) when a
 * model too large to be held twice keeps its projection weights K-tile-major only */
int lade_weight_from_ktile(const void* Wkt, void* W, int64_t ldw, int32_t N, int32_t K, int32_t dtype, void* stream);
/* consumers that take a GEMM output as n_parts fp32 split-K partials [n_parts][rows][width] (part_stride elements
 * apart), sum them in split order and round once to the model dtype - so the split-K GEMM needs no reduce pass */
int lade_add_rmsnorm_parts(void* x, const float* parts, int32_t n_parts, int64_t part_stride, const void* weight, void* y,
                           int32_t rows, int32_t hidden, float eps, int32_t dtype, void* stream);
int lade_silu_mul_parts(const float* parts, int32_t n_parts, int64_t part_stride, void* out, int32_t rows, int32_t inter,
                        int32_t layout, int32_t dtype, void* stream);
int lade_rope_kv_append_parts(const float* parts, int32_t n_parts, int64_t part_stride, void* q_out, const int32_t* positions,
                              const void* cos_tab, const void* sin_tab, void* k_cache, void* vt_cache, int32_t T, int32_t P,
                              const int32_t* dyn_P, int32_t H, int32_t Hkv, int32_t d, int32_t S_max, int32_t max_pos,
                              int32_t dtype, void* stream);
int lade_splitk_reduce(const float* part, void* C, int64_t ldc, int32_t M, int32_t N, int32_t n_split, int32_t dtype,
                       void* stream);

/* ---- misc ------------------------------------------------------------------------------- */
int lade_version(void);
/* what this build of the library contains: bit 0 = the experimental kernels (`make EXPERIMENTAL=1`): the attention forms with RoPE + KV append
 * inside the launch (lade_attn_args.n_parts > 0, sync_flags), lade_gemm_ra_kt, the ping-pong K loop of lade_gemm_skinny* (ring >= 10), its
 * 16-row-granular tiles (mt = 16) - all
 * bit-identical or within rounding of the default forms, all measured slower or equal at every BASELINE shape; the default build answers
 * them with LADE_E_ARG */
int lade_build_flags(void);
const char* lade_last_error_string(void);
/* kernel timing helper for bench.py: runs `reps` launches of lade_attn_fwd (+combine) on `stream`
 * bracketed by hipEvents and returns the mean duration of one launch pair in microseconds. */
int lade_time_attn(const lade_attn_args* a, int32_t reps, float* mean_us, void* stream);
/* the same over n argument sets used round-robin (a[i % n] in repetition i): n K/V caches larger than the Infinity
 * Cache in total make every launch stream from HBM, as consecutive layers of a decode step do */
int lade_time_attn_rot(const lade_attn_args* a, int32_t n, int32_t reps, float* mean_us, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LADE_HIP_H */
