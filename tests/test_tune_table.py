"""The kernel-decision table shipped with the package (lookaheaddecoding_amd/tuned/gfx950_256cu.json, tools/make_tune_table.py): well formed, written
for this library generation, complete for the BASELINE shapes - and every decision names a kernel the shape table builds."""
import json
import os
import re

from lookaheaddecoding_amd import cabi
from lookaheaddecoding_amd.engine import StepEngine

HERE = os.path.dirname(os.path.abspath(__file__))
TABLE = os.path.join(HERE, "..", "lookaheaddecoding_amd", "tuned", "gfx950_256cu.json")


def _built_shapes():
    src = open(os.path.join(HERE, "..", "lookaheaddecoding_amd", "csrc", "gemm_kernel.hpp")).read()
    return {tuple(int(x) for x in m) for m in re.findall(r"SHAPE\(TT,(\d),(\d),(\d),(\d)\)", src)}


def test_shipped_table_is_for_this_library_and_names_built_kernels():
    doc = json.load(open(TABLE))
    assert doc["header"] == {"version": StepEngine.TUNE_FILE_VERSION, "abi": cabi.ABI_VERSION, "device": "gfx950", "n_cu": 256,
                             "row_classes": list(StepEngine.ROW_CLASSES)}
    shapes = _built_shapes()
    keys = [json.loads(k) for k in doc["models"]]
    assert [4096, 11008, 32, 32, 128, 32, 32000, "torch.bfloat16"] in [k[:8] for k in keys]          # BASELINE configs 2 / 3
    assert [5120, 13824, 40, 40, 128, 40, 32016, "torch.bfloat16"] in [k[:8] for k in keys]          # config 4
    n = 0
    assert [8192, 28672, 64, 8, 128, 80, 32000, "torch.bfloat16"] in [k[:8] for k in keys]           # config 5
    for key, ent in doc["models"].items():
        H, Hkv = json.loads(key)[2:4]
        assert sorted(int(m) for m in ent) == sorted(StepEngine.ROW_CLASSES)
        for m, row in ent.items():
            assert set(row) >= set(StepEngine.LAYER_GEMMS) | {"attn"}
            # the attention launch is frozen: 128-row work-groups, sqrt split rule - 64-row ones where >= 8 heads stack > 128 rows on a KV head
            assert row["attn"][:3] == [0, 64 if (H // Hkv >= 8 and (H // Hkv) * (int(m) - 31) > 128) else 128, 0], (key, m, row["attn"])
            for name in StepEngine.GEMM_NAMES:
                c = row.get(name)
                if c is None:
                    continue                                        # the library GEMM
                mb, bn, S, mt, nt, ring = c
                assert mb * 32 == int(m) and S >= 1 and ring in (0, 2, 3, 4, 5, 6, 8)
                if nt:
                    assert (mb // mt, mt, bn // 32 // nt, nt) in shapes, (name, m, c)
                n += 1
    assert n >= 60
