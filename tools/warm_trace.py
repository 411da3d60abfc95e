#!/usr/bin/env python3
"""What m6a_create's background set-up launches (run under `rocprofv3 --kernel-trace --stats`): a context is created, the
first entry point waits for the set-up, one small ragged job with the default parameters follows."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m6anet_amd import synthetic  # noqa: E402
from m6anet_amd.engine import M6ANetEngine  # noqa: E402

t0 = time.perf_counter()
e = M6ANetEngine()
t1 = time.perf_counter()
e.sync()
t2 = time.perf_counter()
print("m6a_create returned after %.2f ms; the first entry point waited another %.2f ms for the background set-up" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
d = synthetic.make_sites(2000, (20, 90), seed=1)
e.infer(d["X"], d["site_kmers"], d["off"], 1000)
print("pooling kernel of the first call:", e.last_pool_variant)
e.close()
