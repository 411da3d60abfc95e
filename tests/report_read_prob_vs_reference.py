#!/usr/bin/env python3
"""Worst use of the reference's read-probability bar (rtol 1e-5 / atol 1e-8, m6anet/tests/test_inference.py:32) by
each encoder kernel and by the oracle, measured DIRECTLY against the reference's values at scale
(tests/golden/reference_at_scale.npz: 1.0 M reads of configs[2], 1.1 M reads of configs[4], four checkpoints), plus the
end-to-end site-probability distance at T = 1000.  Runs on the GPU box:

    python tests/report_read_prob_vs_reference.py > gpurun_out/r04_read_prob_vs_reference.json
"""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from m6anet_amd import synthetic                                     # noqa: E402
from m6anet_amd.constants import DEFAULT_READ_THRESHOLD, asset_path  # noqa: E402
from m6anet_amd.engine import M6ANetEngine                           # noqa: E402

SHAPES = {"uniform": (1_000_000, 20, 50_000, 10_000), "ragged": (1_000_000, (50, 500), 4_000, 10_000)}
MODELS = ("hct116", "arabidopsis", "hek293t_glori", "hek293t_m6ace")


def use(got, want):
    want = want.astype(np.float64)
    return np.abs(got.astype(np.float64) - want) / (1e-8 + 1e-5 * np.abs(want))


def main():
    from oracle import m6a_oracle as orc      # the checker, reported beside the kernels (tests/: the oracle is test infrastructure)
    G = np.load(os.path.join(REPO, "tests", "golden", "reference_at_scale.npz"))
    out = {"bar": "rtol 1e-5, atol 1e-8 (m6anet/tests/test_inference.py:32); use = |got-ref| / (atol + rtol*|ref|)",
           "reference": "tests/golden/make_golden.py --only-scale: encoder per 16-site batch (inference_utils.py:33-37)",
           "shapes": {}}
    for tag, (n, bag, keep_reads, keep_sites) in SHAPES.items():
        d = synthetic.make_sites(n, bag, seed=20250328, prefix_sites=max(keep_reads, keep_sites))
        R = int(d["off"][keep_reads])
        Rs = int(d["off"][keep_sites])
        row = {"reads": R, "sites_with_site_prob": keep_sites, "checkpoints": {}}
        for name in MODELS:
            w = np.fromfile(asset_path("weights_%s.bin" % name), np.float32)
            e = M6ANetEngine(weights=w)
            ref = G["%s_%s_readprob" % (tag, name)]
            r = {}
            for mode, label in ((1, "general16"), (2, "csite12")):
                e.set_encoder_variant(mode)
                got = e.get_read_probability(d["X"][:R], d["site_kmers"][:keep_reads], d["off"][:keep_reads + 1])
                u = use(got, ref)
                r[label] = {"worst_use": float(u.max()), "reads_beyond_bar": int((u > 1).sum()), "p99.99_use": float(np.quantile(u, 0.9999)),
                            "ref_p_at_worst": float(ref[int(u.argmax())]),
                            "bit_identical_fraction": float((got.view(np.uint32) == ref.view(np.uint32)).mean())}
            e.set_encoder_variant(0)
            po = orc.encode_reads(w, d["X"][:R], d["site_kmers"][:keep_reads], d["off"][:keep_reads + 1], n_threads=8)
            u = use(po, ref)
            r["oracle"] = {"worst_use": float(u.max()), "reads_beyond_bar": int((u > 1).sum()), "p99.99_use": float(np.quantile(u, 0.9999)),
                           "bit_identical_fraction": float((po.view(np.uint32) == ref.view(np.uint32)).mean())}
            rp, site, mod = e.infer(d["X"][:Rs], d["site_kmers"][:keep_sites], d["off"][:keep_sites + 1], 1000)
            r["site_prob_T1000_max_abs_diff"] = float(np.abs(site.astype(np.float64) - G["%s_%s_site_T1000" % (tag, name)]).max())
            r["site_prob_encoder"] = e.last_encoder_variant
            r["mod_ratio_sites_differing"] = int((mod != G["%s_%s_mod" % (tag, name)]).sum())
            row["checkpoints"][name] = r
            e.close()
        out["shapes"][tag] = row
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
