"""What lookahead parallelism (lade/decoding.py:956-986) does to the step time at R = 1 / 2 / 4 / 8 ranks - measured per rank on ONE GPU.

Under LP every rank holds a full replica and feeds only its window columns and its share of the candidates, so a rank's step is a
forward over T_r < T rows; the step of the group is the slowest rank's forward + one record all-gather (tens of bytes: latency) +
`lade_lp_reduce_apply`.  No multi-GPU box was available to any round, so this tool measures what CAN be measured: for every R, the
steady-state forward of EVERY rank's shard (its own row count, its own mask: `dist_offset` = first owned column) as a hipGraph on this
GPU, cold regime (g = 0) and with a full candidate set (g = G), and prints max-over-ranks per R - the expected curve minus the
collective.  `python tools/lp_curve.py [7b|13b|70b] W N G [layers]`

The reference's default configuration W = 60, N = 8, G = 60 (lade/decoding.py:854-862) feeds 420..840 rows on one rank - beyond the
weight-streaming regime - and 100..130 per rank at R = 8: that is where LP is meant to pay.  At BASELINE's W = 15 the one-rank step is
already a weight stream and the curve is flat."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lookaheaddecoding_amd import ops
from lookaheaddecoding_amd.engine import StepEngine
from lookaheaddecoding_amd.parallel import guess_shard, shard_level_sizes, window_shard
from lookaheaddecoding_amd.weights import make_config, random_weights_torch

MODEL = {"7b": "llama2-7b", "13b": "codellama-13b", "70b": "llama2-70b"}[sys.argv[1] if len(sys.argv) > 1 else "7b"]
W, N, G = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (60, 8, 60)
LAYERS = int(sys.argv[5]) if len(sys.argv) > 5 else 0
P = 2048
gs = N - 1
cfg = make_config(MODEL)
if LAYERS:
    cfg["layers"] = LAYERS
dev = torch.device("cuda", 0)
w = random_weights_torch(cfg, seed=0, dtype=torch.bfloat16, device=dev)
T_max = (N - 1) * (W + G) + gs
eng = StepEngine(cfg, w, dtype=torch.bfloat16, device=dev, max_seq=P + 2 * T_max + 64, max_T=max(T_max, 512), consume_weights=True)
del w
L_full = make_config(MODEL)["layers"]


def time_forward(level_sizes, n_input, lguess):
    mask = ops.StepMask.from_levels(n_input, level_sizes, lguess, gs, P)
    T = mask.T
    ids = torch.randint(3, cfg["vocab"], (T,), device=dev, dtype=torch.int32)
    pos = torch.arange(P, P + T, device=dev, dtype=torch.int32)
    n_sel = 1 + level_sizes[-1] + lguess
    sel = torch.arange(T - n_sel, T, device=dev, dtype=torch.int32)

    def run():
        ops.argmax_rows(eng.forward(ids, pos, mask, sel, n_sel))

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 4)
    return T, best * L_full / cfg["layers"]          # scaled to the full depth when run on fewer layers (stated in the output)


level_lens = [W - 1] + [W] * (N - 2)
print(f"{MODEL} ({cfg['layers']} of {L_full} layers timed, scaled to {L_full}) W={W} N={N} G={G} P={P}: forward + argmax of every rank's shard, ms per step", flush=True)
base = {}
for R in (1, 2, 4, 8):
    if R > W:
        continue
    for label, g in (("cold g=0", 0), (f"hot g={G}", G)):
        rows = []
        for r in range(R):
            c0, c1 = window_shard(level_lens[0] + 1, R, r)
            ls = shard_level_sizes(level_lens, c0, c1)
            glo, ghi = guess_shard(g, R, r)
            # a rank that owns no column of the higher levels feeds its L0 prefix only
            T_r, ms = time_forward(ls, 1, (ghi - glo) * gs)
            rows.append((T_r, ms))
        worst = max(ms for _, ms in rows)
        base.setdefault(label, worst)
        print(f"  R={R} {label:10s}: rows per rank {[t for t, _ in rows]}  ms per rank {[round(m, 3) for _, m in rows]}  -> step >= {worst:.3f} ms "
              f"({base[label] / worst:.2f} x the one-rank step; + the record all-gather and lade_lp_reduce_apply, ~30-40 us)", flush=True)
