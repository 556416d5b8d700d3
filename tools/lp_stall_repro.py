"""Hunts the start-up stall `tests/test_gpu_bench_pins.py` once saw (1 in 11: a rank of the 2-process gloo run stalled at start-up right
after another process had torn down an RCCL communicator on the same GPU; VERDICT r3 "weak" item 1).  Repeats, N times: (a) the 1-rank
RCCL process of `test_config_lade_dist_workers_generate_one_rccl_rank`, then immediately (b) the 2-rank gloo run sharing the GPU - with
NCCL_DEBUG=INFO and a 45 s faulthandler watchdog in every worker - and records per run: wall seconds of each process, where a stalled
rank was (the faulthandler dump), and the last RCCL / gloo lines before it.   python tools/lp_stall_repro.py [N] > gpurun_out/lp_stall.txt

Result (round 4, profiles/r4_lp_stall.txt): 0 of 14, then 1 of 36 - and the one caught names the cause: rank 0's TCPStore failed to bind the
rendezvous port (EADDRINUSE), rank 1 waited for the store.  The port had come from bind(0), i.e. from the kernel's ephemeral range, and was
taken by another connection before rank 0 bound it.  Nothing of RCCL or the GPU; `tests/conftest.free_port` now picks below that range
(this tool uses the test module's picker, so it now runs with the fix)."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import test_gpu_bench_pins as T

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
tmp = "/tmp/lp_stall"
os.makedirs(tmp, exist_ok=True)
worker = T._WORKER.format(root=ROOT).replace("dump_traceback_later(90, exit=True)", "dump_traceback_later(45, exit=True)")
# stage markers so that a stall can be placed: printed to stderr with a timestamp
worker = worker.replace("import lade\n", "import lade, time as _t\n_t0 = _t.time()\ndef _mark(s): print(f'[mark {_t.time() - _t0:6.2f}s] {s}', file=sys.stderr, flush=True)\n_mark('imports done')\n", 1)
worker = worker.replace("lade.augment_all()", "_mark('model on the GPU, plain generate done'); lade.augment_all()")
worker = worker.replace('os.environ["USE_LADE"] = "1"', '_mark("config_lade done (group joined)"); os.environ["USE_LADE"] = "1"')
worker = worker.replace("log = lade.decoding.CONFIG_MAP.get", "_mark('lookahead generate done'); log = lade.decoding.CONFIG_MAP.get")
open(os.path.join(tmp, "w.py"), "w").write(worker)
one = worker.replace(
    'lade.config_lade(LEVEL=4, WINDOW_SIZE=5, GUESS_SET_SIZE=5, DEBUG=1, DIST_WORKERS=int(os.environ["WORLD_SIZE"]), POOL_FROM_PROMPT=1, backend=backend)',
    'from lookaheaddecoding_amd import utils as U\n'
    '    lade.config_lade(LEVEL=4, WINDOW_SIZE=5, GUESS_SET_SIZE=5, DEBUG=1, POOL_FROM_PROMPT=1)\n'
    '    U._join_lookahead_parallel_group(1, "nccl")\n'
    '    lade.decoding.CONFIG_MAP["FORCE_LP"] = 1')
open(os.path.join(tmp, "one.py"), "w").write(one)


def run(script, world, backend, share):
    port = T._free_port()
    procs, t0 = [], time.time()
    for r in range(world):
        env = dict(os.environ, LOCAL_RANK=str(r), RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LADE_TEST_BACKEND=backend,
                   LADE_TEST_SHARE_GPU="1" if share else "0", HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="INFO", TORCH_DISTRIBUTED_DEBUG="DETAIL")
        procs.append(subprocess.Popen([sys.executable, os.path.join(tmp, script)], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    out = []
    for r, p in enumerate(procs):
        try:
            so, se = p.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            p.kill()
            so, se = p.communicate()
        marks = [l for l in se.splitlines() if l.startswith("[mark")]
        stalled = "Timeout (0:00:45)!" in se
        out.append(dict(rank=r, rc=p.returncode, wall=round(time.time() - t0, 1), stalled=stalled, marks=marks, tail=se[-2500:] if (stalled or p.returncode != 0) else ""))
    return out


stalls = 0
for i in range(N):
    a = run("one.py", 1, "nccl", False)
    b = run("w.py", 2, "gloo", True)
    bad = [x for x in a + b if x["stalled"] or x["rc"] != 0]
    stalls += bool(bad)
    print(f"iteration {i}: rccl-1-rank {a[0]['wall']} s rc {a[0]['rc']} | gloo ranks {[x['wall'] for x in b]} s rc {[x['rc'] for x in b]} | "
          f"last marks {[x['marks'][-1] if x['marks'] else None for x in b]}", flush=True)
    for x in bad:
        print(f"  ---- rank {x['rank']} rc {x['rc']} stalled {x['stalled']}; marks {x['marks']}\n{x['tail']}\n  ----", flush=True)
print(f"{stalls} of {N} iterations had a stalled or failed process", flush=True)
