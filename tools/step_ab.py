#!/usr/bin/env python3
"""GPU box: wall time per step of the bench workload (1 M sites x 20 reads, T = 1000, inputs resident) for the encoder variants,
in alternating legs of N steps -- are the 10-step legs of bench.py's `reference_order_encoder` representative?
    python tools/step_ab.py [steps_per_leg] [legs]"""
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
    legs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    eng = M6ANetEngine(weights=load_weights("HCT116_RNA002"))
    eng.use_torch_stream()
    d = synthetic.make_sites(1_000_000, 20, seed=20250328)
    X, km, off = (torch.from_numpy(d[k]).cuda() for k in ("X", "site_kmers", "off"))
    outs = (torch.empty(20_000_000, dtype=torch.float32, device="cuda"), torch.empty(1_000_000, dtype=torch.float32, device="cuda"),
            torch.empty(1_000_000, dtype=torch.float64, device="cuda"))
    thr = np.float32(DEFAULT_READ_THRESHOLD)
    off_h = np.ascontiguousarray(d["off"])

    def step():
        eng.set_host_offsets(off_h)
        eng.infer(X, km, off, 1000, 20, thr, 0, 16, 2, out=outs)
    res = {}
    for leg in range(legs):
        for mode in (2, 1, 3):
            eng.set_encoder_variant(mode)
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                step()
            torch.cuda.synchronize()
            res.setdefault(eng.last_encoder_kernel, []).append(round((time.perf_counter() - t0) / n * 1e3, 4))
    print(json.dumps({"steps_per_leg": n, "ms_per_step_by_leg": res}))


if __name__ == "__main__":
    main()
