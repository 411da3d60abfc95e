#!/usr/bin/env python3
"""Feasibility probe: does the MFMA-bound encoder overlap with the LDS-bound table pooling when
they run on two streams?  Two contexts, independent data.  Prints alone / back-to-back /
concurrent times."""
import os
import sys
import time


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from m6anet_amd import synthetic  # noqa: E402
from m6anet_amd.engine import M6ANetEngine, load_weights  # noqa: E402

dev = torch.device("cuda:0")
S = 1_000_000
d = synthetic.make_sites(S, 20, seed=3)
X, km, off = (torch.from_numpy(d[k]).to(dev) for k in ("X", "site_kmers", "off"))
w = load_weights()
A, B = M6ANetEngine(weights=w), M6ANetEngine(weights=w)
rp = torch.empty(S * 20, dtype=torch.float32, device=dev)
A.get_read_probability(X, km, off, out=rp)
A.sync()
rp2 = rp.clone()


def enc():
    A.get_read_probability(X, km, off, out=rp2)


def pool():
    return B.calculate_site_proba(rp, off, 1000)


def t(fn, n=10):
    fn(); A.sync(); B.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    A.sync(); B.sync()
    return (time.perf_counter() - t0) / n * 1e3


def both():
    enc(); pool()


def serial():
    enc(); A.sync(); pool(); B.sync()


print("blocks/CU env:", os.environ.get("M6A_ENC_BLOCKS_PER_CU", "2"))
print("enc alone   %.3f ms" % t(enc))
print("pool alone  %.3f ms" % t(pool))
print("serial      %.3f ms" % t(serial))
print("two streams %.3f ms" % t(both))
