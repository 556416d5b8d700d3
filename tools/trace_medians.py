"""Per-kernel medians of a rocprofv3 kernel trace (`*_kernel_trace.csv`) - EVERY kernel, keyed by (name, grid, work-group) so that
launches of one template with different grids (the qkv and the gate/up GEMM, ...) stay apart - and, with --steps, the steady decode
step taken apart launch by launch: the trace is cut at every `greedy_post_step_kernel`, the most frequent launch sequence is the
steady step, and every position of it gets its median duration, its median gap to the previous launch's end and the running sum.

    python tools/trace_medians.py gpurun_out/.../NNN_kernel_trace.csv [--steps] [--top N]

The recorded duration of a kernel inside the step graph starts when its predecessor ends, so it includes the dispatch latency of a
dependent launch; the gaps column shows what is left between launches."""
import collections
import csv
import statistics
import sys


def short(name: str) -> str:
    name = name.replace("void ", "")
    head = name.split("(")[0]
    return head if len(head) <= 64 else head[:61] + "..."


def load(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            grid = tuple(int(r.get(k, 0) or 0) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
            wg = tuple(int(r.get(k, 0) or 0) for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z"))
            n_wg = 1
            for g, w in zip(grid, wg):
                n_wg *= max(1, g // max(1, w))
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), n_wg, wg[0] * max(1, wg[1]) * max(1, wg[2])))
    rows.sort()
    return rows


def table(rows, top):
    by = collections.defaultdict(list)
    for s, e, name, n_wg, thr in rows:
        by[(name, n_wg, thr)].append((e - s) / 1e3)
    items = sorted(by.items(), key=lambda kv: -sum(kv[1]))
    total = sum(sum(v) for _, v in items)
    for (name, n_wg, thr), v in (items if top <= 0 else items[:top]):
        v.sort()
        print(f"{name:64s} wgs {n_wg:6d} x {thr:4d}  calls {len(v):6d}  median {statistics.median(v):8.2f} us  p10 {v[len(v) // 10]:8.2f}  "
              f"p90 {v[len(v) * 9 // 10]:8.2f}  share {100 * sum(v) / total:5.1f}%")


def steps(rows, marker="greedy_post_step_kernel"):
    cuts, cur = [], []
    for r in rows:
        cur.append(r)
        if marker in r[2]:
            cuts.append(cur)
            cur = []
    sig = collections.Counter(tuple((n, w) for _, _, n, w, _ in c) for c in cuts)
    if not sig:
        print("no step marker in the trace")
        return
    best, cnt = sig.most_common(1)[0]
    sel = [c for c in cuts if tuple((n, w) for _, _, n, w, _ in c) == best]
    # the launches after the marker (KV commit, record copy) belong to the NEXT cut's head: show them where they run
    print(f"steady step: {cnt} of {len(cuts)} cuts share the sequence of {len(best)} launches (cut at {marker}); medians over them")
    run = 0.0
    per_name = collections.defaultdict(float)
    for i, (name, n_wg) in enumerate(best):
        d = statistics.median((c[i][1] - c[i][0]) / 1e3 for c in sel)
        gap = statistics.median(((c[i][0] - c[i - 1][1]) / 1e3) for c in sel) if i else 0.0
        run += d + max(gap, 0.0)
        per_name[name] += d
        print(f"  {i:4d} {name:64s} wgs {n_wg:6d}  {d:8.2f} us  gap {gap:6.2f}  sum {run:9.2f}")
    span = statistics.median((c[-1][1] - c[0][0]) / 1e3 for c in sel)
    period = statistics.median((b[0][0] - a[0][0]) / 1e3 for a, b in zip(sel, sel[1:]) if b[0][0] > a[0][0]) if len(sel) > 1 else float("nan")
    print(f"steady step: first launch start -> marker end {span:.1f} us (median); start-to-start of consecutive steady steps {period:.1f} us")
    print("per kernel name inside one steady step (sum of position medians):")
    for name, t in sorted(per_name.items(), key=lambda kv: -kv[1]):
        print(f"  {name:64s} {t:9.2f} us  {100 * t / span:5.1f}%")


def main(argv):
    path = argv[0]
    top = int(argv[argv.index("--top") + 1]) if "--top" in argv else 0
    rows = load(path)
    table(rows, top)
    if "--steps" in argv:
        print()
        steps(rows)


if __name__ == "__main__":
    main(sys.argv[1:])
