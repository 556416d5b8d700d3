"""A short sampling run with temperature + top-k + top-p on a 7B-width (2-layer) engine, to be run under
`rocprofv3 --kernel-trace --stats`: the kernel list of its steps shows the warpers as ONE lade::warp_rows_kernel launch per step and
no at::native sort / topk / cumsum kernel (run on the GPU box: python tools/sample_trace.py)."""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lookaheaddecoding_amd.decoding import LookaheadDecoder
from lookaheaddecoding_amd.engine import StepEngine
from lookaheaddecoding_amd.sampling import make_warper
from lookaheaddecoding_amd.weights import make_config, random_weights_torch

cfg = make_config("llama2-7b", layers=2)
w = random_weights_torch(cfg, seed=0, dtype=torch.bfloat16, device="cuda")
w["embed"] = (w["embed"].float() * 50).bfloat16()
w["lm_head"] = w["embed"]
eng = StepEngine(cfg, w, dtype=torch.bfloat16, device="cuda", max_seq=1024, max_T=512)
prompt = [(7 * i) % 50 + 3 for i in range(96)]
dec = LookaheadDecoder(eng, 15, 5, 15, pool_from_prompt=True, use_graph=True)
out = dec.sample(prompt, len(prompt) + 64, warp=make_warper(temperature=0.8, top_k=50, top_p=0.9), rng=random.Random(1),
                 torch_gen=torch.Generator(device="cuda").manual_seed(1))
print("generated", out.generated, "steps", out.steps)
