"""Writes the kernel-decision table that is SHIPPED with the package for this GPU model (lookaheaddecoding_amd/tuned/<isa>_<CUs>cu.json): every
row class of the BASELINE model shapes is tuned once here (isolated pass + in-step pass + lm_head), so that every MI355X launches the same
kernels for them.  Run on the GPU box with the FINAL library build; commit the result.
    python tools/make_tune_table.py gpurun_out/tuned.json 7b:bf16 7b:f16 13b:bf16 [70b:bf16]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.abspath(sys.argv[1])
os.environ["LADE_TUNE_FILE"] = out
import torch

from lookaheaddecoding_amd.engine import StepEngine
from lookaheaddecoding_amd.weights import make_config, random_weights_torch

NAMES = {"7b": "llama2-7b", "13b": "codellama-13b", "70b": "llama2-70b"}
for spec in sys.argv[2:]:
    model, dt = spec.split(":")
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16}[dt]
    cfg = make_config(NAMES[model])
    t0 = time.time()
    w = random_weights_torch(cfg, seed=0, dtype=dtype, device="cuda")
    eng = StepEngine(cfg, w, dtype=dtype, device="cuda", max_seq=4096, max_T=2304, consume_weights=True)
    del w
    already = sorted(eng.tune_loaded)
    table = eng.tune_all()
    eng.save_tune_file()
    print(f"{spec}: {len(table)} decisions in {time.time() - t0:.0f} s (classes already in the file: {already}) -> {out}", flush=True)
    for m in eng.ROW_CLASSES:
        print("   ", m, {n: table.get(f"{n}:{m}") for n in eng.GEMM_NAMES}, flush=True)
    del eng
    torch.cuda.empty_cache()
print("device:", torch.cuda.get_device_name(0), "->", os.path.basename(__import__("lookaheaddecoding_amd.engine", fromlist=["x"]).shipped_tune_table(torch.device("cuda", 0))))
