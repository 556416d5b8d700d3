"""Per-launch medians of rocprofv3 --pmc passes, per kernel AND grid size (the same GEMM kernel serves several projections):
    python tools/pmc_table.py out.json label=path.csv [label=path.csv ...] [--match substr ...] [--trim dir]
Writes {label: {"kernel @ grid": {counter: median, "launches": n, "median_us": kernel duration under the profiler}}} for the kernels
whose name contains one of the --match strings; --trim also writes the matching rows (few columns) as small CSVs into dir."""
import collections
import csv
import json
import os
import sys

args = sys.argv[1:]
out = args[0]
match, files, trim = [], [], None
i = 1
while i < len(args):
    if args[i] == "--match":
        match.append(args[i + 1]); i += 2
    elif args[i] == "--trim":
        trim = args[i + 1]; i += 2
    else:
        files.append(args[i].split("=", 1)); i += 1
match = match or ["attn_fwd_kernel", "attn_combine_kernel", "gemm_skinny_kernel"]
doc = {}
for label, path in files:
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    rows = []
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if not any(m in n for m in match):
            continue
        short = n.split("(")[0].replace("void ", "").strip()
        key = f"{short} @ grid {r['Grid_Size']}"
        per[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        rows.append([r["Dispatch_Id"], short, r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["Counter_Name"], r["Counter_Value"],
                     r["Start_Timestamp"], r["End_Timestamp"]])
    doc[label] = {k: dict({c: sorted(v)[len(v) // 2] for c, v in cs.items()}, launches=max(len(v) for v in cs.values()),
                          median_us=round(sorted(dur[k])[len(dur[k]) // 2], 2)) for k, cs in per.items()}
    if trim:
        os.makedirs(trim, exist_ok=True)
        with open(os.path.join(trim, os.path.basename(path)), "w", newline="") as f:
            wr = csv.writer(f)
            wr.writerow(["Dispatch_Id", "Kernel", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"])
            wr.writerows(rows[:480])
json.dump(doc, open(out, "w"), indent=1)
