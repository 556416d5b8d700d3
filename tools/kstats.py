"""Prints a compact table from a rocprofv3 *_kernel_stats.csv: python tools/kstats.py file.csv [n]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
for r in rows[:n]:
    name = r["Name"]
    if name.startswith("Cijk"):
        i = name.find("MT")
        name = "hipBLASLt GEMM " + name[i:i + 14]
    name = name.replace("void lade::", "lade::")[:64]
    print(f"{name:64s} calls {int(r['Calls']):6d}  avg {float(r['AverageNs']) / 1e3:8.2f} us  total {float(r['TotalDurationNs']) / 1e6:8.2f} ms  {float(r['Percentage']):5.1f}%")
