#!/usr/bin/env python3
"""PCIe-inclusive rate of the bench workload through the host-pointer path (pageable numpy arrays in and out),
for several staging-slot sizes and copy-thread counts (M6A_STAGE_MB / M6A_COPY_THREADS are read when the ring is
first set up, so every setting runs in its own process)."""
import json
import os
import subprocess
import sys

CHILD = r'''
import json, sys, time, numpy as np
sys.path.insert(0, %r)
from m6anet_amd import synthetic
from m6anet_amd.engine import M6ANetEngine, load_weights
eng = M6ANetEngine(weights=load_weights())
d = synthetic.make_sites(1_000_000, 20, seed=2)
t0 = time.perf_counter(); eng.prepare_host_io(); t_prep = time.perf_counter() - t0
t0 = time.perf_counter(); eng.infer(d["X"], d["site_kmers"], d["off"], 1000); t_cold = time.perf_counter() - t0
ts = []
for _ in range(5):
    t0 = time.perf_counter(); eng.infer(d["X"], d["site_kmers"], d["off"], 1000); ts.append(time.perf_counter() - t0)
out = (np.empty(20_000_000, np.float32), np.empty(1_000_000, np.float32), np.empty(1_000_000, np.float64))
tr = []
for _ in range(5):
    t0 = time.perf_counter(); eng.infer(d["X"], d["site_kmers"], d["off"], 1000, out=out); tr.append(time.perf_counter() - t0)
print(json.dumps({"prepare_ms": t_prep * 1e3, "cold_ms": t_cold * 1e3, "fresh_outputs_sites_per_s": 1e6 / min(ts),
                  "fresh_outputs_median_sites_per_s": 1e6 / sorted(ts)[2], "reused_outputs_sites_per_s": 1e6 / min(tr)}))
'''


def main():
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mb, th in ((24, 16), (8, 16), (64, 16), (24, 4), (24, 8), (24, 32), (48, 24)):
        env = dict(os.environ, M6A_STAGE_MB=str(mb), M6A_COPY_THREADS=str(th))
        out = subprocess.run([sys.executable, "-c", CHILD % repo], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        res["stage_mb=%d threads=%d" % (mb, th)] = json.loads(line[-1]) if line else {"error": out.stderr[-300:]}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
