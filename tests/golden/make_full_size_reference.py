#!/usr/bin/env python3
"""One-off, BUILD CONTAINER ONLY (imports the reference like make_golden.py): the reference's read probabilities for ALL
20 000 000 reads of BASELINE.json configs[2] -- or, with --ragged, all 34 357 966 reads of configs[4]'s per-GPU shape (125 000
sites x 50..500 reads, what `bench.py --workload ragged` runs) -- from this repository's generator, four checkpoints, encoder per 16-site batch
(m6anet/utils/inference_utils.py:33-37) -> tests/golden/_big/configs2_<model>.npy (80 MB each: git-ignored, and listed in
.gpurunignore so that ordinary snapshots stay small -- take that line out for the one call that runs the report; a snapshot
carries at most 512 MiB, so the ragged shape goes three checkpoints at a time).  tests/report_full_size_vs_reference.py compares both HIP encoder kernels with them there
and writes the summary that IS committed (profiles/r04_full_size_vs_reference.json)."""
import os
import sys
import tempfile

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
from m6anet_amd import synthetic  # noqa: E402

MODELS = {"hct116": C.DEFAULT_MODEL_WEIGHTS, "arabidopsis": C.ARABIDOPSIS_MODEL_WEIGHTS,
          "hek293t_glori": C.HEK293TRNA004_GLORI_MODEL_WEIGHTS, "hek293t_m6ace": C.HEK293TRNA004_M6ACE_MODEL_WEIGHTS}


SHAPES = {"configs2": (1_000_000, 20), "configs4": (125_000, (50, 500))}   # configs4: the per-GPU shape bench.py --workload ragged runs


def main():
    torch.set_num_threads(8)
    out = os.path.join(HERE, "_big")
    os.makedirs(out, exist_ok=True)
    tag = "configs4" if "--ragged" in sys.argv else "configs2"
    S, bag = SHAPES[tag]
    d = synthetic.make_sites(S, bag, seed=20250328)
    X, sk, off = d["X"], d["site_kmers"], d["off"]
    nr = np.diff(off)
    for name, path in MODELS.items():
        m = MILModel(toml.load(C.DEFAULT_MODEL_CONFIG))
        m.load_state_dict(torch.load(path, map_location="cpu"))
        m.eval()
        p = np.empty(int(off[-1]), np.float32)
        with torch.no_grad():
            for a in range(0, S, 16):
                b = min(S, a + 16)
                lo, hi = int(off[a]), int(off[b])
                kpr = torch.from_numpy(np.repeat(sk[a:b].astype(np.int64), nr[a:b], axis=0))
                f = m.get_read_representation({"X": torch.from_numpy(X[lo:hi]), "kmer": kpr})
                p[lo:hi] = m.pooling_filter.probability_layer(f).flatten().numpy()
        np.save(os.path.join(out, "%s_%s.npy" % (tag, name)), p)
        print(name, p.size, float(p.min()), float(p.max()), flush=True)
        if name in sys.argv[1:]:                       # e.g. `make_full_size_reference.py hek293t_glori`: also the model in float64
            m = m.double()
            q = np.empty(int(off[-1]), np.float64)
            with torch.no_grad():
                for a in range(0, S, 4096):
                    b = min(S, a + 4096)
                    lo, hi = int(off[a]), int(off[b])
                    kpr = torch.from_numpy(np.repeat(sk[a:b].astype(np.int64), nr[a:b], axis=0))
                    f = m.get_read_representation({"X": torch.from_numpy(X[lo:hi]).double(), "kmer": kpr})
                    q[lo:hi] = m.pooling_filter.probability_layer(f).flatten().numpy()
            np.save(os.path.join(out, "%s_%s_f64.npy" % (tag, name)), q)


if __name__ == "__main__":
    main()
