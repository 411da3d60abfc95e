#!/usr/bin/env python3
"""Fuzz of the streaming job API (m6a_job_begin / feed / end; test infrastructure, like tests/): random jobs -- site counts,
bag mixes (uniform, ragged, with tiny and empty bags), batch sizes from one site to the whole job, host and device batches
mixed, size hints or none, job offsets -- fed batch by batch must equal ONE m6a_infer over the same job bit for bit (the
encoder kernel is pinned when a job holds bags below 16 reads: the choice is per chunk in one and per job in the other).

    python tests/fuzz_stream.py [seconds] [seed]      -> one JSON line with the case counts
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from m6anet_amd import synthetic  # noqa: E402
from m6anet_amd.engine import M6ANetEngine, flush_groups, load_weights  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    g = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 1))
    eng = M6ANetEngine(weights=load_weights())
    thr = np.float32(0.033379376)
    t_end = time.time() + budget
    n_cases = n_feeds = n_dev = 0
    while time.time() < t_end:
        S = int(g.choice([1, 2, 17, 300, 4095, 4096, 4097, 9000, 30000, 120000], p=[.05, .05, .1, .2, .1, .1, .1, .15, .1, .05]))
        kind = int(g.integers(0, 4))
        if kind == 0:
            bags = np.full(S, 20)
        elif kind == 1:
            bags = g.integers(16, 200, size=S)
        elif kind == 2:
            bags = g.integers(0, 40, size=S)
        else:
            bags = g.choice([0, 1, 15, 16, 20, 64, 700, 3000], size=S, p=[.05, .05, .1, .2, .3, .2, .08, .02])
        if S > 20000:
            bags = np.minimum(bags, 64)
        d = synthetic.make_sites(S, seed=int(g.integers(1 << 30)), n_reads=bags.astype(np.int64))
        off = d["off"]
        T = int(g.choice([1, 5, 33, 100]))
        bs, spb = int(g.choice([16, 7, 1, 64])), int(g.choice([2, 3, 1]))
        seed = int(g.integers(0, 1 << 32))
        pin = bags.min() < 16
        eng.set_encoder_variant(1 if pin else 0)
        groups = flush_groups(S, bs, spb)
        base = int(groups[int(g.integers(0, len(groups) - 1))]) if g.random() < 0.3 and len(groups) > 2 else 0
        # the job = sites [base, S) of a larger job whose first `base` sites somebody else handles
        sl_off = off[base:] - off[base]
        X, km = d["X"][off[base]:], d["site_kmers"][base:]
        eng.set_job_offset(base)
        want = eng.infer(X, km, sl_off, T, 20, thr, seed, bs, spb)
        hints = dict(expect_sites=S - base, expect_reads=int(sl_off[-1])) if g.random() < 0.3 else {}
        eng.job_begin(T, 20, thr, seed, bs, spb, **hints)
        n = S - base
        batch = int(g.choice([1, 3, 16, 100, 1000, 5000, max(n, 1)]))
        if n > 20000 and batch < 16:
            batch = 16
        dev_X = dev_k = None
        s0 = 0
        any_dev = False
        while s0 < n:
            s1 = min(n, s0 + batch)
            bo = np.ascontiguousarray(sl_off[s0:s1 + 1] - sl_off[s0])
            if g.random() < 0.15:
                if dev_X is None:
                    dev_X, dev_k = torch.from_numpy(X).cuda(), torch.from_numpy(km).cuda()
                eng.job_feed(dev_X[sl_off[s0]:sl_off[s1]], dev_k[s0:s1], bo)
                any_dev = True
            else:
                eng.job_feed(X[sl_off[s0]:sl_off[s1]], km[s0:s1], bo)
            n_feeds += 1
            s0 = s1
        got = eng.job_end(device_outputs=False)
        eng.set_job_offset(0)
        n_dev += any_dev
        for a, b, name in zip(got, want, ("read_prob", "site_prob", "mod_ratio")):
            if not np.array_equal(a, b, equal_nan=True):
                print(json.dumps({"FAILED": name, "S": S, "kind": kind, "T": T, "bs": bs, "spb": spb, "seed": seed, "base": base, "batch": batch,
                                  "pinned": bool(pin), "first_bad": int(np.flatnonzero(~((a == b) | (np.isnan(a) & np.isnan(b))))[0])}))
                sys.exit(1)
        n_cases += 1
    print(json.dumps({"cases": n_cases, "feed_calls": n_feeds, "cases_with_device_batches": n_dev, "seconds": budget,
                      "result": "every streamed job bit-identical to one m6a_infer"}))


if __name__ == "__main__":
    main()
