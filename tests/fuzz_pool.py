#!/usr/bin/env python3
"""Fuzz of the pooling kernels against the CPU oracle (test infrastructure, like tests/): random jobs --
bag sizes, site counts, iteration counts, sample counts, batch geometries, seeds -- through every kernel
that applies (uniform bags: LDS-table and register kernel; ragged bags: both scan drivers and the
index-table kernel), all of which must reproduce the oracle's site probabilities and mod_ratio BIT FOR BIT.

    python tests/fuzz_pool.py [seconds] [seed]      -> one JSON line with the case counts
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m6anet_amd.engine import M6ANetEngine, load_weights  # noqa: E402
from oracle import m6a_oracle as orc  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    g = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 1))
    orc.build()
    eng = M6ANetEngine(weights=load_weights())
    thr = np.float32(0.033379376)
    t_end = time.time() + budget
    n_cases = n_runs = 0
    by_kernel = {}
    while time.time() < t_end:
        kind = int(g.integers(0, 6))
        S = int(g.integers(1, 500))
        if kind == 0:
            bags = np.full(S, int(g.integers(1, 33)))
        elif kind == 1:
            bags = g.integers(1, 40, size=S)
        elif kind == 2:
            bags = g.integers(20, 700, size=S)
        elif kind == 3:
            bags = g.choice([2, 3, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 4095, 4096], size=S)
        elif kind == 4:
            bags = np.where(g.random(S) < 0.05, g.integers(1025, 4500, size=S), g.integers(1, 200, size=S))
        else:
            bags = np.full(S, int(g.integers(33, 300)))
        T = int(g.choice([1, 2, 3, 7, 8, 9, 17, 33, 64, 100, 127, 129, 250, 257, 500, 1000, 1024, 1500, 3001]))
        K = 20 if g.random() < 0.8 else int(g.integers(1, 65))
        bs, spb = int(g.choice([1, 2, 5, 16, 32, 50])), int(g.choice([1, 2, 3]))
        seed = int(g.integers(0, 2 ** 32))
        if g.random() < 0.15:
            bags = np.where(g.random(S) < 0.2, 0, bags)              # empty sites: NaN outputs, no words drawn
        if bags.max() == 0 or T * K * bags.max() > 4e7 or T * K * S > 4e7:
            continue
        off = np.concatenate([[0], np.cumsum(bags)]).astype(np.int64)
        p = (g.random(int(off[-1]), dtype=np.float32) ** 4).astype(np.float32)
        want_site, want_mod = orc.site_pool(p, off, T, thr, seed=seed, batch_size=bs, save_per_batch=spb, n_samples=K, n_threads=8)
        uniform = bags.min() == bags.max() and bags.max() <= 32 and K == 20
        runs = [("table", 1, 0), ("table", 2, 0)] if uniform else [("scan", 0, 1), ("scan", 0, 2)]
        if not uniform and bags.max() <= 4096:
            runs.append(("rtab", 0, 3))
        runs.append(("auto", 0, 0))
        n_cases += 1
        for name, tv, sd in runs:
            eng.set_table_variant(tv)
            eng.set_scan_driver(sd)
            site, mod = eng.calculate_site_proba(p, off, T, K, thr, seed=seed, batch_size=bs, save_per_batch=spb)
            n_runs += 1
            by_kernel[eng.last_pool_variant] = by_kernel.get(eng.last_pool_variant, 0) + 1
            if not (np.array_equal(site, want_site, equal_nan=True) and np.array_equal(mod, want_mod, equal_nan=True)):
                bad = np.flatnonzero(~((site == want_site) | (np.isnan(site) & np.isnan(want_site))))
                print(json.dumps({"FAIL": name, "kernel": eng.last_pool_variant, "kind": kind, "S": S, "T": T, "K": K, "bs": bs, "spb": spb,
                                  "seed": seed, "bags_head": [int(x) for x in bags[:12]], "first_bad_sites": [int(x) for x in bad[:8]],
                                  "max_abs": float(np.nanmax(np.abs(site - want_site)))}))
                sys.exit(1)
        eng.set_table_variant(0)
        eng.set_scan_driver(0)
    print(json.dumps({"cases": n_cases, "kernel_runs": n_runs, "by_kernel": by_kernel, "seconds": budget, "result": "all bit-identical to the oracle"}))


if __name__ == "__main__":
    main()
