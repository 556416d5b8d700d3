"""What the engine's GEMM autotune sees: every (projection, row class) of a model shape, the library time, the winner and the five best
hand-written candidates (times include the tuner's consumer-tail model for gate/up):
    python tools/gemm_tune_probe.py [7b|13b|70b] [row classes ...]"""
import os
import sys

os.environ["LADE_DEBUG"] = ",".join(x for x in (os.environ.get("LADE_DEBUG"), "tune_verbose") if x)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lookaheaddecoding_amd.engine import StepEngine
from lookaheaddecoding_amd.weights import make_config, random_weights_torch

name = sys.argv[1] if len(sys.argv) > 1 else "7b"
classes = [int(x) for x in sys.argv[2:]] or list(StepEngine.ROW_CLASSES)
cfg = make_config({"7b": "llama2-7b", "13b": "codellama-13b", "70b": "llama2-70b"}[name], layers=8)
eng = StepEngine(cfg, random_weights_torch(cfg, seed=0), max_seq=256, max_T=256)
for m in classes:
    for proj in StepEngine.GEMM_NAMES:
        eng._tune(proj, m)
