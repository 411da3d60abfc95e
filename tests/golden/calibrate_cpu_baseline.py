#!/usr/bin/env python3
"""BASELINE.md section 3 / SURVEY.md section 8(d): the CPU baseline that travels to the GPU box is the oracle (a C port);
the reference's own Python cannot.  This script -- BUILD CONTAINER ONLY, it imports the reference like
tests/golden/make_golden.py -- times both on the SAME inputs, one thread each, and writes the ratio:

    reference  m6anet.utils.inference_utils._calculate_site_proba (:74-87) per site, after np.random.seed;
               get_read_representation + probability_layer per 16-site batch (:33-37), torch on one thread
    oracle     oracle/m6a_oracle.c through oracle/m6a_oracle.py, n_threads = 1

    python tests/golden/calibrate_cpu_baseline.py        # -> profiles/r06_cpu_calibration.json

bench.py's cpu_baseline carries the factor (`calibration`, `reference_equivalent_value` = oracle sites/s / factor).
"""
import json
import os
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
_shim = tempfile.mkdtemp(prefix="m6a_shims_")
with open(os.path.join(_shim, "toml.py"), "w") as f:
    f.write("import tomli\ndef load(p):\n    with open(p,'rb') as f:\n        return tomli.load(f)\n")
with open(os.path.join(_shim, "ujson.py"), "w") as f:
    f.write("from json import *\n")
sys.dont_write_bytecode = True
sys.path[:0] = [_shim, REF, REPO]

import numpy as np  # noqa: E402
import torch  # noqa: E402
import toml  # noqa: E402
from m6anet.model.model import MILModel  # noqa: E402
from m6anet.utils import constants as C  # noqa: E402
from m6anet.utils.inference_utils import _calculate_site_proba  # noqa: E402

from m6anet_amd import synthetic  # noqa: E402
from m6anet_amd.engine import load_weights  # noqa: E402
from oracle import m6a_oracle as orc  # noqa: E402

torch.set_num_threads(1)
T = 1000


def best_of(f, n=3):
    best = None
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best


def main():
    sys.path.insert(0, REPO)
    import bench
    thr = np.float32(C.DEFAULT_READ_THRESHOLD)
    out = {"host": bench.host_cpu_facts(), "num_iterations": T, "shapes": {}}
    tot_ref = tot_orc = 0.0
    for tag, bag, S, model_path, mname in (("uniform (configs[2] shape)", 20, 1024, C.DEFAULT_MODEL_WEIGHTS, "HCT116_RNA002"),
                                           ("ragged (configs[4] shape)", (50, 500), 256, C.HEK293TRNA004_GLORI_MODEL_WEIGHTS, "HEK293T_RNA004")):
        d = synthetic.make_sites(S, bag, seed=20250328)
        X, sk, off = d["X"], d["site_kmers"], d["off"]
        nr = np.diff(off)
        m = MILModel(toml.load(C.DEFAULT_MODEL_CONFIG))
        m.load_state_dict(torch.load(model_path, map_location="cpu"))
        m.eval()
        w = load_weights(mname)
        probs = {}

        def ref_encoder():
            p = np.empty(int(off[-1]), np.float32)
            with torch.no_grad():
                for a in range(0, S, 16):
                    b = min(S, a + 16)
                    lo, hi = int(off[a]), int(off[b])
                    kpr = np.repeat(sk[a:b].astype(np.int64), nr[a:b], axis=0)
                    feat = m.get_read_representation({"X": torch.from_numpy(X[lo:hi]), "kmer": torch.from_numpy(kpr)})
                    p[lo:hi] = m.pooling_filter.probability_layer(feat).flatten().numpy()
            probs["p"] = p

        def orc_encoder():
            orc.encode_reads(w, X, sk, off, n_threads=1)

        t_ref_enc, t_orc_enc = best_of(ref_encoder), best_of(orc_encoder)
        p = probs["p"]

        def ref_sampling():
            for g in range(0, S, 32):
                np.random.seed(0)
                for s in range(g, min(S, g + 32)):
                    _calculate_site_proba((p[off[s]:off[s + 1]], T, 20))

        def orc_sampling():
            orc.site_pool(p, off, T, thr, n_threads=1, batch_size=32, save_per_batch=1)

        t_ref_s, t_orc_s = best_of(ref_sampling), best_of(orc_sampling)
        out["shapes"][tag] = {
            "sites": S, "reads": int(off[-1]),
            "reference": {"encoder_reads_per_s": off[-1] / t_ref_enc, "sampling_sites_per_s": S / t_ref_s, "whole_path_sites_per_s": S / (t_ref_enc + t_ref_s)},
            "oracle": {"encoder_reads_per_s": off[-1] / t_orc_enc, "sampling_sites_per_s": S / t_orc_s, "whole_path_sites_per_s": S / (t_orc_enc + t_orc_s)},
            "oracle_over_reference": {"encoder": t_ref_enc / t_orc_enc, "sampling": t_ref_s / t_orc_s, "whole_path": (t_ref_enc + t_ref_s) / (t_orc_enc + t_orc_s)},
        }
        tot_ref += t_ref_enc + t_ref_s
        tot_orc += t_orc_enc + t_orc_s
        print(tag, json.dumps(out["shapes"][tag]["oracle_over_reference"]), file=sys.stderr)
    u = out["shapes"]["uniform (configs[2] shape)"]["oracle_over_reference"]
    out["oracle_over_reference"] = u      # the headline workload's factor is the one bench.py applies
    within = abs(u["whole_path"] - 1.0) <= 0.10
    out["within_10_percent"] = within
    out["note"] = ("one thread each, best of 3, same inputs, in the build container (the reference cannot travel to the GPU box); "
                   "oracle_over_reference > 1 means the C port is FASTER than the reference's NumPy/torch code, i.e. the cpu_baseline "
                   "in the bench line flatters the CPU by that factor; reference_equivalent_value = value / whole_path")
    with open(os.path.join(REPO, "profiles", "r06_cpu_calibration.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps(out["oracle_over_reference"]), "within 10%:", within)


if __name__ == "__main__":
    main()
