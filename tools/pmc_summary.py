"""Summarises two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes)
of tools/attn_bench.py into profiles/r1_attn_pmc.json:
    python tools/pmc_summary.py fetch.csv write.csv T P n_splits out.json [H Hkv d [wg_rows]]
FETCH_SIZE is doubled (gfx950 reports 1/2 of wide coalesced reads), WRITE_SIZE is taken as is; KiB -> bytes."""
import collections
import csv
import json
import sys


def per_kernel(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "attn_fwd_kernel" in n:
            d["fwd"].append(float(r["Counter_Value"]))
        elif "attn_combine_kernel" in n:
            d["combine"].append(float(r["Counter_Value"]))
    return {k: sorted(v)[len(v) // 2] for k, v in d.items()}      # median launch


fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
T, P, ns = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
H, Hkv, d = (int(sys.argv[7]), int(sys.argv[8]), int(sys.argv[9])) if len(sys.argv) > 9 else (32, 32, 128)
alg = 2 * (2 * Hkv * (P + T) * d + 2 * H * T * d)
wg = int(sys.argv[10]) if len(sys.argv) > 10 else 128
entry = {"H": H, "Hkv": Hkv, "d": d, "T": T, "P": P, "n_splits": ns, "wg_rows": wg,
         "fetch_kib_raw": fetch, "write_kib_raw": write,
         "traffic_bytes": int((2 * (fetch.get("fwd", 0) + fetch.get("combine", 0)) + write.get("fwd", 0) + write.get("combine", 0)) * 1024),
         "algorithmic_bytes": alg,
         "note": "HBM bytes per launch pair = 2*FETCH_SIZE + WRITE_SIZE (KiB) over attn_fwd + attn_combine; the excess over the algorithmic "
                 "bytes is the split-KV partial round trip (written by attn_fwd, read by attn_combine)"}
entry["traffic_over_algorithmic"] = round(entry["traffic_bytes"] / alg, 3)
out = sys.argv[6]
try:
    doc = json.load(open(out))
except Exception:
    doc = {"entries": []}
doc["entries"] = [e for e in doc["entries"] if (e.get("H", 32), e.get("Hkv", 32), e["T"], e["P"], e["n_splits"]) != (H, Hkv, T, P, ns)] + [entry]
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps(entry))
