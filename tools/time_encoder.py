#!/usr/bin/env python3
"""GPU box: HIP-event time of each encoder kernel on the bench shapes (inputs resident), for A/B of builds via M6A_HIP_LIB.
    python tools/time_encoder.py [lib.so ...]     # no argument: the in-tree library"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def one():
    import torch
    from m6anet_amd import synthetic
    from m6anet_amd.engine import M6ANetEngine, load_weights
    eng = M6ANetEngine(weights=load_weights("HCT116_RNA002"))
    out = {}
    for tag, bag, S in (("uniform20", 20, 1_000_000), ("ragged50_500", (50, 500), 141_000)):
        d = synthetic.make_sites(S, bag, seed=20250328)
        X, km, off = (torch.from_numpy(d[k]).cuda() for k in ("X", "site_kmers", "off"))
        rp = torch.empty(int(d["off"][-1]), dtype=torch.float32, device="cuda")
        for mode, label in ((1, "general16"), (2, "csite12")):
            eng.set_encoder_variant(mode)
            for _ in range(3):
                eng.get_read_probability(X, km, off, out=rp)
            eng.sync()
            eng.profile("encoder")
            for _ in range(30):
                eng.get_read_probability(X, km, off, out=rp)
            ms, n = eng.profile_read(0)
            eng.profile(False)
            out["%s_%s_ms" % (tag, label)] = round(ms / n, 4)
        del X, km, off, rp
    print(json.dumps(out))


if __name__ == "__main__":
    if "--one" in sys.argv:
        one()
    else:
        libs = sys.argv[1:] or [""]
        for lib in libs:
            env = dict(os.environ)
            if lib:
                env["M6A_HIP_LIB"] = os.path.join(REPO, lib)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, capture_output=True, text=True, timeout=900)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            print(lib or "in-tree", line[-1] if line else r.stderr[-400:], flush=True)
