#!/usr/bin/env python3
"""GPU box: per-call wall time of m6a_infer on HOST pointers for the bench workload (1 M sites x 20 reads, T = 1000) -- pageable
NumPy arrays against page-locked tensors, inputs and outputs separately (the pinned ones skip the staging copy).
    python tools/h2d_probe.py [calls]"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import numpy as np
    import torch
    from m6anet_amd import synthetic
    from m6anet_amd.constants import DEFAULT_READ_THRESHOLD
    from m6anet_amd.engine import M6ANetEngine, load_weights
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    eng = M6ANetEngine(weights=load_weights("HCT116_RNA002"))
    eng.prepare_host_io()
    d = synthetic.make_sites(1_000_000, 20, seed=20250328)
    thr = np.float32(DEFAULT_READ_THRESHOLD)
    page_in = (d["X"], d["site_kmers"], d["off"])
    pin_in = tuple(torch.from_numpy(a).pin_memory() for a in page_in)
    page_out = (np.empty(20_000_000, np.float32), np.empty(1_000_000, np.float32), np.empty(1_000_000, np.float64))
    pin_out = tuple(torch.empty(o.shape, dtype=getattr(torch, str(o.dtype))).pin_memory() for o in page_out)
    res = {}
    for rep in range(2):
        for name, ins, outs in (("pageable in, pageable out", page_in, page_out), ("pinned in, pageable out", pin_in, page_out),
                                ("pageable in, pinned out", page_in, pin_out), ("pinned in, pinned out", pin_in, pin_out)):
            eng.infer(*ins, 1000, 20, thr, 0, 16, 2, out=outs)
            ts = []
            for _ in range(n):
                t0 = time.perf_counter()
                eng.infer(*ins, 1000, 20, thr, 0, 16, 2, out=outs)
                ts.append(round((time.perf_counter() - t0) * 1e3, 2))
            res.setdefault(name, []).append(ts)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
