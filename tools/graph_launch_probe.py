"""How long a hipGraph launch takes to reach the GPU as a function of the graph's node count (the steady step is one graph of ~300 kernel
nodes; tools/host_gap_probe.py: the replay call costs the host ~110 us, the GPU idles ~40 us between two steps).  wall(replay + sync) of
graphs of n small kernels: the intercept over n is the launch's fixed latency, the slope the per-node time; and the same for a graph
cut in two (a head of h nodes replayed first, the tail behind it): does the tail's launch hide behind the head's execution?"""
import statistics
import time

import torch


def graph_of(n, x):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            x.add_(1.0)
    return g


def wall(fn, reps=200):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter_ns()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter_ns() - t0) / 1e3)
    return statistics.median(ts)


def main():
    x = torch.zeros(1 << 23, device="cuda")          # 32 MB: a kernel of ~13 us, the step's average node
    for _ in range(3):
        x.add_(1.0)
    torch.cuda.synchronize()
    print("eager kernel + sync: %.1f us" % wall(lambda: x.add_(1.0)))
    gs = {n: graph_of(n, x) for n in (1, 4, 16, 64, 150, 300, 600)}
    w = {}
    for n, g in gs.items():
        for _ in range(3):
            g.replay()
        w[n] = wall(g.replay, 100)
    tk = (w[600] - w[300]) / 300.0
    print("per node (600 vs 300 nodes): %.2f us" % tk)
    for n in gs:
        print("graph of %3d nodes: replay + sync %.1f us = %d nodes x %.2f + %.1f us" % (n, w[n], n, tk, w[n] - n * tk))
    for h in (1, 4, 16):
        head, tail = graph_of(h, x), graph_of(300 - h, x)

        def two():
            head.replay()
            tail.replay()
        for _ in range(3):
            two()
        t = wall(two, 100)
        print("graph of 300 nodes cut into %d + %d: replay + replay + sync %.1f us = 300 nodes x %.2f + %.1f us" % (h, 300 - h, t, tk, t - 300 * tk))


if __name__ == "__main__":
    main()
