"""Per-kernel medians of a rocprofv3 kernel trace (`*_kernel_trace.csv`), steady decode steps included as they are:
python tools/trace_medians.py gpurun_out/.../NNN_kernel_trace.csv
The recorded duration of a kernel inside the step graph starts when its predecessor ends, so it includes the dispatch
latency of a dependent launch (no idle gaps appear between the kernels of a layer)."""
import collections
import csv
import statistics
import sys


def short(name: str) -> str:
    name = name.replace("void ", "")
    head = name.split("(")[0]
    return head if len(head) <= 72 else head[:69] + "..."


def main(path: str) -> None:
    by = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            by[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    rows = sorted(by.items(), key=lambda kv: -sum(kv[1]))
    total = sum(sum(v) for _, v in rows)
    for name, v in rows[:24]:
        v.sort()
        print(f"{name:72s} calls {len(v):6d}  median {statistics.median(v):8.2f} us  p10 {v[len(v) // 10]:8.2f}  p90 {v[len(v) * 9 // 10]:8.2f}  "
              f"share {100 * sum(v) / total:5.1f}%")


if __name__ == "__main__":
    main(sys.argv[1])
