import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def free_port():
    """A rendezvous port OUTSIDE the kernel's ephemeral range.  A port obtained from bind(("127.0.0.1", 0)) lies INSIDE that range, and the
    kernel may hand it to any outgoing connection of the box between this probe and the TCPStore's bind seconds later (the workers load a
    model first): rank 0 then dies with EADDRINUSE and the other ranks wait for a store that never comes - the 1-in-50 "start-up stall" of
    rounds 3 and 4 (`profiles/r4_lp_stall.txt`: iteration 8 of the second hunt).  Ports below the range are only taken by explicit binds."""
    import random
    import socket
    import time
    try:
        with open("/proc/sys/net/ipv4/ip_local_port_range") as f:
            lo = int(f.read().split()[0])
    except (OSError, ValueError, IndexError):
        lo = 32768
    hi = max(lo, 14000)
    rnd = random.Random(os.getpid() ^ time.time_ns())
    for _ in range(128):
        port = rnd.randrange(max(10000, hi - 16000), hi)
        s = socket.socket()
        try:
            s.bind(("127.0.0.1", port))
            return port
        except OSError:
            continue
        finally:
            s.close()
    raise RuntimeError("no free port below the ephemeral range")
