"""CPU: host-side logic that needs no GPU - RoPE tables (default / linear / llama3 scaling) against HuggingFace's own rotary
embedding, and the guards of the HF drop-in (config_from_hf refuses what the HIP step does not implement; the reference
itself only patches the Llama classes, lade/utils.py:40-56)."""
import pytest
import torch


def _hf_cos_sin(rope_parameters, d, n_pos, max_pos=4096):
    from transformers import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaRotaryEmbedding
    cfg = LlamaConfig(hidden_size=d * 2, num_attention_heads=2, num_hidden_layers=1, intermediate_size=32, vocab_size=32,
                      max_position_embeddings=max_pos, rope_parameters=rope_parameters)
    rot = LlamaRotaryEmbedding(cfg)
    pos = torch.arange(n_pos)[None]
    cos, sin = rot(torch.zeros(1, n_pos, d), pos)
    return cos[0], sin[0]


@pytest.mark.parametrize("scaling", [None,
                                     {"rope_type": "linear", "factor": 4.0},
                                     {"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                                      "original_max_position_embeddings": 512}])
def test_rope_tables_match_huggingface(scaling):
    from lookaheaddecoding_amd.engine import rope_tables
    d, n_pos, theta = 64, 1500, 10000.0
    rp = {"rope_type": "default", "rope_theta": theta} if scaling is None else dict(scaling, rope_theta=theta)
    cos_ref, sin_ref = _hf_cos_sin(rp, d, n_pos)
    cos, sin = rope_tables(d, n_pos, theta, torch.float32, "cpu", scaling)
    assert torch.allclose(cos, cos_ref, atol=2e-5) and torch.allclose(sin, sin_ref, atol=2e-5)
    if scaling is not None:            # the scaling really changes the table
        c0, _ = rope_tables(d, n_pos, theta, torch.float32, "cpu", None)
        assert not torch.allclose(c0, cos, atol=1e-3)


def test_unknown_rope_scaling_is_refused():
    from lookaheaddecoding_amd import cabi
    from lookaheaddecoding_amd.engine import rope_tables
    with pytest.raises(cabi.LadeHipError):
        rope_tables(64, 128, 10000.0, torch.float32, "cpu", {"rope_type": "yarn", "factor": 2.0})
    # 'dynamic' is implemented since round 4 (per-step rows, lade_rope_rows_dynamic); its host-built inverse-frequency table is the
    # reference's formula (lade/models/modeling_llama.py:302-308): row 0 the original base, row i a rebuild at length max_pos + i
    from lookaheaddecoding_amd.engine import ntk_inv_freq_table
    tab = ntk_inv_freq_table(64, 10000.0, 2.0, 32, 40)
    assert tab.shape == (9, 32) and torch.equal(tab[0], 1.0 / (10000.0 ** (torch.arange(0, 64, 2).float() / 64)))
    base = 10000.0 * ((2.0 * 40 / 32) - 1.0) ** (64 / 62)
    assert torch.equal(tab[8], 1.0 / (base ** (torch.arange(0, 64, 2).float() / 64))) and bool((tab[8] <= tab[0]).all())


def test_config_from_hf_guards():
    from transformers import LlamaConfig, LlamaForCausalLM, MistralConfig, MistralForCausalLM
    from lookaheaddecoding_amd import cabi, hf
    small = dict(vocab_size=64, hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=1)
    m = LlamaForCausalLM(LlamaConfig(**small, max_position_embeddings=256))
    c = hf.config_from_hf(m)
    assert c["heads"] == 2 and c["kv_heads"] == 1 and c["head_dim"] == 16 and c["rope_scaling"] is None and c["rope_theta"] == 10000.0
    m3 = LlamaForCausalLM(LlamaConfig(**small, max_position_embeddings=256,
                                      rope_parameters={"rope_type": "llama3", "rope_theta": 500000.0, "factor": 8.0, "low_freq_factor": 1.0,
                                                       "high_freq_factor": 4.0, "original_max_position_embeddings": 128}))
    c3 = hf.config_from_hf(m3)
    assert c3["rope_scaling"]["rope_type"] == "llama3" and c3["rope_theta"] == 500000.0
    with pytest.raises(cabi.LadeHipError):       # another architecture: Llama math would silently produce wrong tokens
        hf.config_from_hf(MistralForCausalLM(MistralConfig(**small, max_position_embeddings=256, sliding_window=None)))
    my = LlamaForCausalLM(LlamaConfig(**small, max_position_embeddings=256,
                                      rope_parameters={"rope_type": "yarn", "rope_theta": 10000.0, "factor": 2.0, "original_max_position_embeddings": 128}))
    with pytest.raises(cabi.LadeHipError):
        hf.config_from_hf(my)
    mb = LlamaForCausalLM(LlamaConfig(**small, max_position_embeddings=256, attention_bias=True))
    with pytest.raises(cabi.LadeHipError):
        hf.config_from_hf(mb)


def test_engine_refuses_cpu_and_missing_library(monkeypatch):
    """the product path has no CPU fallback: without a GPU the engine does not construct"""
    from lookaheaddecoding_amd import cabi
    from lookaheaddecoding_amd.engine import StepEngine
    from lookaheaddecoding_amd.weights import make_config, random_weights_numpy
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = make_config("tiny-d16")
    w = {k: torch.as_tensor(v) for k, v in random_weights_numpy(cfg).items()}
    with pytest.raises(cabi.LadeHipError):
        StepEngine(cfg, w, dtype=torch.float32, device="cuda")


def test_resolve_drafts_equals_the_oracle_verify_loop():
    """sampling.resolve_drafts (scalars only) against the oracle's sample_verify (full probability vectors, a restatement of
    lade/decoding.py:484-540): same accepted drafts, same winner, same distribution for the final draw, same number of
    uniform() draws - over randomised candidate sets with shared prefixes and duplicate drafts."""
    import random

    import lade_oracle as O
    from lookaheaddecoding_amd.sampling import final_distribution, resolve_drafts
    rnd = random.Random(5)
    V = 50
    for trial in range(300):
        gs, g = rnd.choice([2, 3, 4]), rnd.choice([1, 2, 3, 5, 8])
        drafts = []
        for c in range(g):
            if c and rnd.random() < 0.5:                       # share a prefix with an earlier candidate
                src = rnd.randrange(c)
                keep = rnd.randrange(1, gs + 1)
                drafts += drafts[src * gs:src * gs + keep] + [rnd.randrange(6) for _ in range(gs - keep)]
            else:
                drafts += [rnd.randrange(6) for _ in range(gs)]
        gtorch = torch.Generator().manual_seed(trial)
        sharp = rnd.choice([0.3, 1.0, 3.0])
        probs_next = torch.softmax(torch.randn(V, generator=gtorch) * sharp + torch.nn.functional.one_hot(torch.tensor(drafts[0]), V) * 2.0, -1)
        guess_probs = torch.softmax(torch.randn(g * gs, V, generator=gtorch) * sharp, -1)
        for c in range(g):                                       # candidates that share a prefix see the same distributions
            for j in range(gs):
                for c2 in range(c):
                    if drafts[c2 * gs:c2 * gs + j + 1] == drafts[c * gs:c * gs + j + 1]:
                        guess_probs[c * gs + j] = guess_probs[c2 * gs + j]
                        break
        table = [[float(probs_next[drafts[c * gs]]) for c in range(g)]]
        for c2 in range(g):
            for j in range(gs):
                table.append([float(guess_probs[c2 * gs + j][drafts[c * gs + j + 1]]) if j + 1 < gs else 0.0 for c in range(g)])
        seed = rnd.randrange(1 << 30)
        r1, r2 = random.Random(seed), random.Random(seed)
        final = {}

        def fake_multinomial(p):
            final["p"] = p.clone()
            return int(torch.argmax(p))

        hits_ref, idx_ref = O.sample_verify(probs_next.clone(), guess_probs, drafts, gs, r1, fake_multinomial)
        v = resolve_drafts(table, drafts, g, gs, r2.random)
        assert r1.random() == r2.random(), "different number of uniform draws"
        if v.final_row is None:
            assert hits_ref == v.accepted and len(hits_ref) == gs
        else:
            assert hits_ref[:-1] == v.accepted
            base = probs_next if v.final_row == 0 else guess_probs[v.final_row - 1]
            mine = final_distribution(base.clone(), v.struck)
            assert torch.allclose(mine, final["p"], atol=1e-6), trial
        if v.accepted:
            assert idx_ref == v.winner


def test_file_channel_hands_rank0s_bytes_to_the_other_ranks(tmp_path):
    """parallel.FileChannel: the byte channel a caller without torch.distributed gives RcclComm for the 128-byte unique id."""
    import threading
    from lookaheaddecoding_amd.parallel import FileChannel
    blob = bytes(range(128))
    got = {}

    def rank(r):
        ch = FileChannel(str(tmp_path / "chan"), r, timeout_s=20)
        got[r] = ch(blob if r == 0 else None)

    ts = [threading.Thread(target=rank, args=(r,)) for r in (2, 1, 0)]      # readers first: they poll until rank 0 has published
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert got == {0: blob, 1: blob, 2: blob}
    import pytest
    with pytest.raises(TimeoutError):
        FileChannel(str(tmp_path / "empty"), 1, timeout_s=0.05)(None)


def test_gemm_tune_table_round_trips_through_its_int32_block():
    """rank 0's GEMM autotune decisions travel to the other ranks as a fixed int32 block through the step's own all-gather
    (parallel.encode_tune_table / decode_tune_table): None (library GEMM) and every (mb, bn, S, mt, nt, ring) tuple survive."""
    from lookaheaddecoding_amd.parallel import decode_tune_table, encode_tune_table
    names, classes = ("wqkv", "wo", "wgu", "wd"), (32, 64, 96, 128, 192, 256)
    table = {f"{n}:{m}": None for n in names for m in classes}
    table["wqkv:64"] = (1, 128, 5, 1, 0, 4)
    table["wgu:64"] = (2, 96, 1, 1, 1, 6)
    table["wd:256"] = (8, 128, 2, 4, 2, 3)
    words = encode_tune_table(table, names, classes)
    assert len(words) == 7 * len(names) * len(classes) and all(isinstance(w, int) for w in words)
    assert decode_tune_table(words, names, classes) == table


def test_graph_buckets_follow_the_gemm_row_classes():
    """LookaheadDecoder._buckets: the candidate counts a steady-step hipGraph is captured for = the most candidates that still fit each
    32-row GEMM class of the engine, + G/4, + {0, G}: a step with one candidate pads to 64 rows (config 2) / 126 rows (config 4), not to
    76 / 150; beyond the last class (the reference's default W=60 N=8 G=60: 420+ rows) the quartiles."""
    from types import SimpleNamespace
    from lookaheaddecoding_amd.decoding import LookaheadDecoder
    classes = (32, 64, 96, 128, 192, 256)

    def buckets(W, N, G):
        return LookaheadDecoder._buckets(SimpleNamespace(W=W, N=N, G=G, e=SimpleNamespace(ROW_CLASSES=classes)))

    assert buckets(15, 5, 15) == [0, 1, 4, 9, 15]
    assert [(5 - 1) * (15 + b) for b in buckets(15, 5, 15)] == [60, 64, 76, 96, 120]
    assert buckets(20, 7, 20) == [0, 1, 5, 12, 20]
    assert buckets(60, 8, 60) == [0, 15, 30, 60]
    assert buckets(15, 5, 0) == [0]
    for W, N, G in ((5, 3, 5), (7, 4, 7), (10, 5, 10), (5, 4, 5), (15, 5, 15), (20, 7, 20)):
        b = buckets(W, N, G)
        assert b[0] == 0 and b[-1] == G and b == sorted(set(b))
    # round 6: the 160-row class adds a bucket where it holds more candidates than the 128-row class (config 4: 6 candidates = 156 rows)
    classes = (32, 64, 96, 128, 160, 192, 256)
    assert buckets(15, 5, 15) == [0, 1, 4, 9, 15] and buckets(20, 7, 20) == [0, 1, 5, 6, 12, 20]


def test_rendezvous_ports_are_picked_below_the_ephemeral_range():
    """conftest.free_port: a port the kernel cannot hand to an outgoing connection between the probe and the TCPStore's bind (the diagnosed
    cause of the 1-in-50 start-up stall: EADDRINUSE at rank 0, the other ranks waiting for a store that never comes)."""
    import socket
    from conftest import free_port
    try:
        with open("/proc/sys/net/ipv4/ip_local_port_range") as f:
            lo = int(f.read().split()[0])
    except OSError:
        lo = 32768
    for _ in range(8):
        p = free_port()
        assert 10000 <= p < max(lo, 14000)
        s = socket.socket()
        s.bind(("127.0.0.1", p))          # really free
        s.close()


def test_flash_lookahead_tuple_maps_onto_the_closed_form_mask():
    """`flash_attn_func(..., lookahead=[window, level, n_guess, kv_cache, fill_offset, guess_offset, 0])` (lade/models/modeling_llama.py:705-713,
    tuple built at :1184-1187): the adapter's mapping of the tuple equals the mask description the engine derives from the step's level
    sizes, for every step of the reference's own greedy runs, and the reference's seqlen identity (:706-709) is enforced."""
    import json
    import os
    from conftest import GOLDEN
    from lookaheaddecoding_amd.flash_attn_lade import lookahead_tuple, mask_from_lookahead
    from lookaheaddecoding_amd.ops import StepMask
    n = 0
    for name in ("e2e_greedy.json", "e2e_greedy_wide.json"):
        with open(os.path.join(GOLDEN, name)) as f:
            runs = json.load(f)["runs"]
        for run in runs:
            gs = run["N"] - 1
            for tr in run["trace"]:
                ls, lg, ni, P = tr["level_sizes"], tr["lguess"], tr["n_input"], tr["P"]
                if len(ls) > 1 and any(x != ls[1] for x in ls[1:]):
                    continue                      # ragged levels: the reference cannot build its flash row order (:1483)
                T = ni + sum(ls) + lg
                tup = lookahead_tuple(ni, ls, lg // gs, P)
                assert tup == [ls[-1], len(ls) + 1, lg // gs, P, ni - 1 + 1 + ls[0] - ls[-1], ni - 1, 0]
                m = mask_from_lookahead(tup, T, P + T)
                want = StepMask.from_levels(ni, ls, lg, gs, P, layout=1)
                if lg == 0:
                    want.gs = m.gs                # without candidates `level - 1` follows the fill level, and gs is not read
                assert m == want, (tup, m, want)
                with pytest.raises(AssertionError, match="Setups"):
                    mask_from_lookahead(tup, T + 1, P + T + 1)
                n += 1
    assert n > 300
    assert mask_from_lookahead([0, 0, 0, 40, 0, 0, 0], 9, 49) == StepMask(T=9, P=40, is_prefill=True)
    assert mask_from_lookahead(None, 9, 49) == StepMask(T=9, P=40, is_prefill=True)


def test_multinomial_one_is_torch_multinomial():
    """the device draw's replacement for torch.multinomial(p, 1) must take the same token from the same generator state and leave the
    generator where torch leaves it (CPU generators here; tests/test_gpu_kernels.py repeats it with device generators)"""
    import torch
    from lookaheaddecoding_amd.sampling import multinomial_one
    for V in (5, 257, 32000):
        p = torch.softmax(torch.randn(V, generator=torch.Generator().manual_seed(V)) * 3, 0)
        p[V // 2] = 0.0
        for seed in range(60):
            g1, g2 = torch.Generator().manual_seed(seed), torch.Generator().manual_seed(seed)
            assert torch.multinomial(p, 1, generator=g1).item() == multinomial_one(p, g2).item()
            assert torch.equal(g1.get_state(), g2.get_state())


def test_lade_debug_list_is_parsed_per_call_and_by_the_library_the_same_way(monkeypatch):
    """LADE_DEBUG=name[=value],... : ONE list for the experiment switches (round 6).  Python side read at every call; the library's own parser
    (lade::debug_int, csrc/cabi.cpp) is not exported - its grammar is pinned here by the python twin and by the A/B tools that use both."""
    from lookaheaddecoding_amd import cabi
    monkeypatch.delenv("LADE_DEBUG", raising=False)
    assert cabi.debug("gemm_dbg") is None and cabi.debug("fuse_tail", "1") == "1"
    monkeypatch.setenv("LADE_DEBUG", "gemm_dbg=4, attn_splits=6,tune_verbose,row_classes=r5")
    assert cabi.debug("gemm_dbg") == "4" and cabi.debug("attn_splits", "0") == "6" and cabi.debug("tune_verbose") == "1"
    assert cabi.debug("row_classes") == "r5" and cabi.debug("gemm") is None and cabi.debug("attn_split", "x") == "x"      # no prefix matches
    monkeypatch.setenv("LADE_DEBUG", "fuse_tail=0")
    assert cabi.debug("fuse_tail", "1") == "0"


def test_pmc_traffic_falls_back_to_the_nearest_profiled_split_count():
    """bench.py's roofline.traffic (round 6): the exact profiled launch shape when profiles/ holds it, else the same shape at the nearest split count
    corrected by one partial written + read per split and marked ESTIMATE - never null for a BASELINE shape (the round-5 driver line had 7 splits, the
    profiles 6, and carried no traffic)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cfg = dict(heads=32, kv_heads=32, head_dim=128)
    exact, src = bench.pmc_traffic(cfg, 60, 2219, 6, 128)
    assert exact and "ESTIMATE" not in src and "profiles/r6_attn_pmc.json" in src
    est, src7 = bench.pmc_traffic(cfg, 60, 2219, 7, 128)
    per_split = 2 * (32 * 60 * 128 * 2 + 32 * 60 * 2 * 4)
    assert src7.startswith("ESTIMATE") and est == exact + per_split
    est5, _ = bench.pmc_traffic(cfg, 60, 2219, 5, 128)
    assert est5 == exact - per_split
    assert bench.pmc_traffic(cfg, 61, 2219, 6, 128) == (None, None)                   # another row count: nothing to go by
    gqa, srcg = bench.pmc_traffic(dict(heads=64, kv_heads=8, head_dim=128), 60, 2219, 4, 64)      # the GQA launch adopted in round 6
    alg = 2 * (2 * 8 * (2219 + 60) * 128 + 2 * 64 * 60 * 128)
    assert gqa and 1.7 < gqa / alg < 2.0 and "ESTIMATE" not in srcg                   # 1.83 x algorithmic (2.25 x at the 128-row x 6-split launch of round 5)
