#!/usr/bin/env python3
"""PCIe-inclusive rate of m6a_infer with HOST pointers: pageable numpy arrays vs page-locked (pinned) ones."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from m6anet_amd import synthetic
from m6anet_amd.engine import M6ANetEngine, load_weights
eng = M6ANetEngine(weights=load_weights())
d = synthetic.make_sites(1_000_000, 20, seed=2)
out = {}
def rate(X, km, off, outs):
    eng.infer(X, km, off, 1000, out=outs)
    t0 = time.perf_counter()
    for _ in range(3): eng.infer(X, km, off, 1000, out=outs)
    return 3e6 / (time.perf_counter() - t0)
R, S = len(d["X"]), len(d["off"]) - 1
outs = (np.empty(R, np.float32), np.empty(S, np.float32), np.empty(S, np.float64))
out["pageable_sites_per_s"] = rate(d["X"], d["site_kmers"], d["off"], outs)
pin = lambda a: torch.from_numpy(a).pin_memory().numpy()
pouts = tuple(pin(o) for o in outs)
out["pinned_sites_per_s"] = rate(pin(d["X"]), pin(d["site_kmers"]), pin(d["off"]), pouts)
print(json.dumps(out))
