#!/usr/bin/env python3
"""GPU box: both HIP encoder kernels (and the oracle) against the REFERENCE's read probabilities on ALL 20 000 000 reads of
BASELINE.json configs[2] -- with --ragged: all 34 357 966 reads of configs[4]'s per-GPU shape --, four checkpoints (tests/golden/_big/*.npy from tests/golden/make_full_size_reference.py; 80 MB each,
not committed; .gpurunignore lists that directory -- take the line out for this call).  Bar: rtol 1e-5 / atol 1e-8 (m6anet/tests/test_inference.py:32).
    python tests/report_full_size_vs_reference.py > gpurun_out/r04_full_size_vs_reference.json"""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from m6anet_amd import synthetic                                     # noqa: E402
from m6anet_amd.constants import asset_path                          # noqa: E402
from m6anet_amd.engine import M6ANetEngine                           # noqa: E402
from oracle import m6a_oracle as orc                                 # noqa: E402  (tests/: the checker beside the kernels)


def use(got, want):
    want = want.astype(np.float64)
    return np.abs(got.astype(np.float64) - want) / (1e-8 + 1e-5 * np.abs(want))


def main():
    tag = "configs4" if "--ragged" in sys.argv else "configs2"
    d = synthetic.make_sites(*{"configs2": (1_000_000, 20), "configs4": (125_000, (50, 500))}[tag], seed=20250328)
    out = {"shape": tag, "sites": int(d["off"].size - 1), "reads": int(d["off"][-1]),
           "bar": "rtol 1e-5, atol 1e-8; use = |got - ref| / (atol + rtol |ref|)", "checkpoints": {}}
    for name in ("hct116", "arabidopsis", "hek293t_glori", "hek293t_m6ace"):
        path = os.path.join(REPO, "tests", "golden", "_big", "%s_%s.npy" % (tag, name))
        if not os.path.exists(path):                       # a snapshot carries at most 512 MiB: the ragged shape goes in two calls
            continue
        ref = np.load(path)
        w = np.fromfile(asset_path("weights_%s.bin" % name), np.float32)
        e = M6ANetEngine(weights=w)
        row = {}
        for mode, label in ((0, "auto (general16)"), (2, "csite12 (opt-in)")):       # 0 = nothing set: what the product runs
            e.set_encoder_variant(mode)
            u = use(e.get_read_probability(d["X"], d["site_kmers"], d["off"]), ref)
            row[label] = {"kernel": e.last_encoder_kernel, "worst_use": float(u.max()), "reads_beyond_bar": int((u > 1).sum()), "p99.9999_use": float(np.quantile(u, 0.999999)),
                          "reads_not_bit_identical": int((u > 0).sum())}
        u = use(orc.encode_reads(w, d["X"], d["site_kmers"], d["off"], n_threads=16), ref)
        row["oracle"] = {"worst_use": float(u.max()), "reads_beyond_bar": int((u > 1).sum())}
        exact = os.path.join(REPO, "tests", "golden", "_big", "%s_%s_f64.npy" % (tag, name))
        if os.path.exists(exact):
            # the same model evaluated in float64 (MILModel.double(), same script): how far is each float32 evaluation -- the
            # reference's own included -- from the value they all round towards?
            ex = np.load(exact)
            u = use(ref, ex)
            row["reference_vs_float64"] = {"worst_use": float(u.max()), "reads_beyond_bar": int((u > 1).sum()), "rms_use": float(np.sqrt((u * u).mean()))}
            for mode, label in ((0, "auto (general16)"), (2, "csite12 (opt-in)")):
                e.set_encoder_variant(mode)
                u = use(e.get_read_probability(d["X"], d["site_kmers"], d["off"]), ex)
                row[label + "_vs_float64"] = {"worst_use": float(u.max()), "reads_beyond_bar": int((u > 1).sum()), "rms_use": float(np.sqrt((u * u).mean()))}
        out["checkpoints"][name] = row
        e.close()
        print(name, row, file=sys.stderr, flush=True)
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
