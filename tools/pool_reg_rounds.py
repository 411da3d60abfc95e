#!/usr/bin/env python3
"""pool_reg_kernel's time against the number of resident-wave rounds: a wavefront = 256 sites, the chip holds 2 048 such
waves (2 per SIMD); 1 M sites = 3 936 waves = 1.92 rounds.  Times the pooling alone (T = 1000, 20-read bags) at site
counts that give 0.5, 1, 1.92 (the bench), 2, 3, 4 rounds:   python tools/pool_reg_rounds.py  (GPU box)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m6anet_amd.engine import M6ANetEngine, load_weights   # noqa: E402

T = 1000
eng = M6ANetEngine(weights=load_weights("HCT116_RNA002"))
rows = []
for S in (262144, 524288, 1000000, 1048576, 1572864, 2097152):
    g = torch.Generator(device="cuda").manual_seed(1)
    p = torch.rand(S * 20, device="cuda", generator=g) ** 4
    off = torch.arange(0, S * 20 + 1, 20, device="cuda", dtype=torch.int64)
    for _ in range(3):
        eng.calculate_site_proba(p, off, T)
    eng.sync()
    eng.profile("pooling")
    for _ in range(20):
        eng.calculate_site_proba(p, off, T)
    ms, n = eng.profile_read(1)
    eng.profile(False)
    groups = (S + 31) // 32
    waves = 32 * ((groups + 255) // 256)
    rows.append({"sites": S, "waves": waves, "rounds": waves / 2048.0, "pool_ms": ms / n, "T_draws_per_s": S * T * 20 / (ms / n * 1e-3) / 1e12,
                 "ms_per_round": ms / n / (waves / 2048.0)})
    print(rows[-1], file=sys.stderr)
print(json.dumps(rows, indent=1))
