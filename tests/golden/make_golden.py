#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Runs only in the build container, where the read-only reference tree is mounted at
/root/reference.  Nothing from the reference's source is copied: this script calls the
reference's public functions on fixed inputs and stores inputs/outputs as data.  The
GPU box never sees the reference; tests there read only the files written here.

    python tests/golden/make_golden.py            # regenerate everything (~2 min)

What each file pins (reference file:line):
  weights_*.bin (written to m6anet_amd/assets/)  m6anet/model/model_states/*.pt, loaded like
                                   m6anet/scripts/inference.py:88-90
  norm_*.npz   (m6anet_amd/assets/)              m6anet/model/norm_factors/*.joblib
                                   (m6anet/utils/data_utils.py:233-248)
  bundled_inputs.npz               NanopolishDS.__getitem__ + inference_collate
                                   (m6anet/utils/data_utils.py:192-231,498-506)
  bundled_readprob.npz             get_read_representation + probability_layer
                                   (m6anet/utils/inference_utils.py:35-37)
  bundled_site.npz                 full `inference.main` runs, n_processes=1
                                   (m6anet/scripts/inference.py:70-106,
                                    m6anet/utils/inference_utils.py:14-104)
  rng_known.npz                    np.random.seed / RandomState.choice stream
                                   (m6anet/scripts/inference.py:86, inference_utils.py:85)
  synthetic_small.npz              same functions on this repo's synthetic generator
  bag_forward.npz                  MILModel.forward on fixed bags (m6anet/model/model.py:155-164,
                                   m6anet/model/model_blocks/pooling_blocks.py:127-129)
  config1_*.csv(.gz)               exact CSV bytes (inference_utils.py:62-67)
  ref_tests_data/                  the reference's own test fixtures (data files only):
                                   m6anet/tests/data/{data.info,data.json,
                                   data.site_proba.csv.gz,data.indiv_proba.csv.gz,
                                   eventalign.txt(.gz here),eventalign.index}
  dataprep_ref_run/                parallel_index + parallel_preprocess_tx at n_processes=1
                                   (m6anet/utils/dataprep_utils.py:210-266,328-488)
  validate.npz                     `validate` over a 'Val'-mode NanopolishDS, DataLoader num_workers=0
                                   (m6anet/utils/training_utils.py:213-268; the 20-read sampler without
                                   replacement at m6anet/utils/data_utils.py:213-214)
"""
import gzip
import io
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
ASSETS = os.path.join(REPO, "m6anet_amd", "assets")

# --- two import shims for modules the image lacks (toml -> tomli, ujson -> json) -----------
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
import joblib  # noqa: E402
from m6anet.model.model import MILModel  # noqa: E402
from m6anet.scripts import inference as ref_inference  # noqa: E402
from m6anet.utils import constants as C  # noqa: E402
from m6anet.utils.data_utils import NanopolishDS, inference_collate  # noqa: E402
from m6anet.utils.inference_utils import _calculate_site_proba  # noqa: E402

torch.set_num_threads(1)

MODELS = {
    "hct116": C.DEFAULT_MODEL_WEIGHTS,
    "arabidopsis": C.ARABIDOPSIS_MODEL_WEIGHTS,
    "hek293t_glori": C.HEK293TRNA004_GLORI_MODEL_WEIGHTS,
    "hek293t_m6ace": C.HEK293TRNA004_M6ACE_MODEL_WEIGHTS,
}
NORMS = {"hct116": C.DEFAULT_NORM_PATH, "arabidopsis": C.ARABIDOPSIS_NORM_PATH}
DATA = os.path.join(REF, "m6anet", "tests", "data")


def load_model(path):
    m = MILModel(toml.load(C.DEFAULT_MODEL_CONFIG))
    m.load_state_dict(torch.load(path, map_location="cpu"))
    m.eval()
    return m


def flat_weights(sd):
    """Flat f32 blob in the order include/m6a.h documents (7997 floats)."""
    order = [
        "read_level_encoder.1.embedding_layer.weight",   # E   [66,2]
        "read_level_encoder.3.layers.0.weight",          # W1  [150,15]
        "read_level_encoder.3.layers.0.bias",            # b1  [150]
        "read_level_encoder.3.layers.1.weight",          # gamma
        "read_level_encoder.3.layers.1.bias",            # beta
        "read_level_encoder.3.layers.1.running_mean",    # mu
        "read_level_encoder.3.layers.1.running_var",     # var
        "read_level_encoder.4.layers.0.weight",          # W2  [32,150]
        "read_level_encoder.4.layers.0.bias",            # b2  [32]
        "pooling_filter.probability_layer.0.weight",     # W3  [1,32]
        "pooling_filter.probability_layer.0.bias",       # b3  [1]
    ]
    out = np.concatenate([sd[k].detach().cpu().numpy().astype(np.float32).ravel() for k in order])
    assert out.size == 7997, out.size
    return out


def read_probs(model, X, kmers_per_read):
    with torch.no_grad():
        feat = model.get_read_representation({"X": torch.from_numpy(X), "kmer": torch.from_numpy(kmers_per_read)})
        return model.pooling_filter.probability_layer(feat).flatten().numpy().astype(np.float32)


def flush_groups(n_sites, batch_size, save_per_batch):
    """Site ranges of the reference's flush groups (inference_utils.py:47 -- the inverted
    modulo), plus the never-flushed tail as a final group (flushed=False)."""
    n_batches = (n_sites + batch_size - 1) // batch_size
    groups, start_b = [], 0
    for it in range(n_batches):
        if (it + 1) % save_per_batch:
            groups.append((start_b * batch_size, min((it + 1) * batch_size, n_sites), True))
            start_b = it + 1
    if start_b < n_batches:
        groups.append((start_b * batch_size, n_sites, False))
    return groups


def emulate_sites(p, off, T, seed, batch_size, save_per_batch, thr):
    """Reference semantics at n_processes=1: every flush group's Pool worker starts from the
    parent's never-advanced state, i.e. reseed per group, sites sequential inside it."""
    S = len(off) - 1
    site = np.zeros(S, np.float32)
    written = np.zeros(S, bool)
    for (a, b, flushed) in flush_groups(S, batch_size, save_per_batch):
        np.random.seed(seed)
        for s in range(a, b):
            site[s] = _calculate_site_proba((p[off[s]:off[s + 1]], T, 20))
        written[a:b] = flushed
    mod = np.array([np.mean(p[off[s]:off[s + 1]] >= thr) for s in range(S)], np.float64)
    return site, mod, written


class Args:
    pass


def run_main(input_dirs, out_dir, T, batch_size, save_per_batch, seed=0, pretrained="HCT116_RNA002"):
    a = Args()
    a.input_dir = list(input_dirs)
    a.out_dir = out_dir
    a.model_config = C.DEFAULT_MODEL_CONFIG
    a.model_state_dict = None
    a.pretrained_model = pretrained
    a.norm_path = C.DEFAULT_NORM_PATH
    a.batch_size = batch_size
    a.save_per_batch = save_per_batch
    a.n_processes = 1
    a.num_iterations = T
    a.device = "cpu"
    a.seed = seed
    a.read_proba_threshold = C.DEFAULT_READ_THRESHOLD
    ref_inference.main(a)
    with open(os.path.join(out_dir, "data.site_proba.csv"), "rb") as f:
        site_csv = f.read()
    with open(os.path.join(out_dir, "data.indiv_proba.csv"), "rb") as f:
        indiv_csv = f.read()
    return site_csv, indiv_csv


def parse_site_csv(b):
    import pandas as pd
    df = pd.read_csv(io.BytesIO(b), float_precision="round_trip")
    return df


def dataprep_goldens():
    """eventalign.txt -> index / data.json / data.info through the reference at n_processes=1
    (m6anet/scripts/dataprep.py:54-70), plus the reference's own fixtures (data files only)."""
    from m6anet.utils.dataprep_utils import parallel_index, parallel_preprocess_tx
    rdir = os.path.join(HERE, "ref_tests_data")
    os.makedirs(rdir, exist_ok=True)
    with open(os.path.join(DATA, "eventalign.txt"), "rb") as f, \
            gzip.GzipFile(os.path.join(rdir, "eventalign.txt.gz"), "wb", mtime=0) as g:
        g.write(f.read())
    shutil.copyfile(os.path.join(DATA, "eventalign.index"), os.path.join(rdir, "eventalign.index"))
    os.chmod(os.path.join(rdir, "eventalign.index"), 0o644)
    out = os.path.join(HERE, "dataprep_ref_run")
    os.makedirs(out, exist_ok=True)
    for tag, kw in (("msc1", dict(min_segment_count=1, compress=False)),
                    ("msc20_compress", dict(min_segment_count=20, compress=True))):
        tmp = tempfile.mkdtemp(prefix="m6a_dp_")
        parallel_index(os.path.join(DATA, "eventalign.txt"), 1000000, tmp, 1)
        assert open(os.path.join(tmp, "eventalign.index")).read() == open(os.path.join(DATA, "eventalign.index")).read()
        parallel_preprocess_tx(os.path.join(DATA, "eventalign.txt"), tmp, 1, 1, 1000, 1, kw["min_segment_count"], kw["compress"])
        with open(os.path.join(tmp, "data.json"), "rb") as f, gzip.GzipFile(os.path.join(out, tag + ".data.json.gz"), "wb", mtime=0) as g:
            g.write(f.read())
        shutil.copyfile(os.path.join(tmp, "data.info"), os.path.join(out, tag + ".data.info"))
        shutil.rmtree(tmp)
        print("dataprep golden", tag)


def validate_goldens():
    """The reference's validation pass on the bundled data: a labelled copy of data.info (every site in
    the 'Val' split), `validate()` at num_workers=0 -- the only setting in which its sampler is
    reproducible -- for a few (seed, n_iterations, batch_size).  Also the sampler alone:
    np.random.choice(n, 20, replace=False) in the same order, as index arrays."""
    import pandas as pd
    from torch.utils.data import DataLoader
    from m6anet.utils.data_utils import train_collate
    from m6anet.utils.training_utils import validate
    tmp = tempfile.mkdtemp(prefix="m6a_val_")
    shutil.copy(os.path.join(DATA, "data.json"), tmp)
    info = pd.read_csv(os.path.join(DATA, "data.info"))
    info["set_type"] = "Val"
    info["modification_status"] = (np.arange(len(info)) % 3 == 0).astype(int)
    info.to_csv(os.path.join(tmp, "data.info.labelled"), index=False)
    ds = NanopolishDS(tmp, min_reads=20, norm_path=NORMS["hct116"], mode="Val")
    model = load_model(MODELS["hct116"])
    out = {"n_reads": ds.data_info["n_reads"].values.astype(np.int64), "y_true": ds.labels.astype(np.float32)}
    for seed, T, bs in ((0, 5, 16), (7, 3, 101), (1, 12, 1)):
        dl = DataLoader(ds, batch_size=bs, shuffle=False, num_workers=0, collate_fn=train_collate)
        np.random.seed(seed)
        res = validate(model, dl, "cpu", torch.nn.BCELoss(), n_iterations=T)
        key = "seed%d_T%d_bs%d" % (seed, T, bs)
        out[key + "_y_pred"] = np.asarray(res["y_pred"], dtype=np.float32)               # [T][S]
        out[key + "_y_pred_avg"] = np.mean(res["y_pred"], axis=0)                        # as validate() averages
        out[key + "_avg_loss"] = np.float64(res["avg_loss"])
        out[key + "_roc_auc"] = np.float64(res["roc_auc"])
        out[key + "_pr_auc"] = np.float64(res["pr_auc"])
        np.random.seed(seed)
        out[key + "_idx"] = np.stack([np.stack([np.random.choice(int(n), 20, replace=False) for n in out["n_reads"]])
                                      for _ in range(T)]).astype(np.int32)             # [T][S][20]
        print(key, out[key + "_y_pred"].shape, out[key + "_y_pred_avg"].dtype, float(out[key + "_avg_loss"]))
    np.savez_compressed(os.path.join(HERE, "validate.npz"), **out)
    shutil.rmtree(tmp)


def main():
    if "--only-validate" in sys.argv:
        validate_goldens()
        return
    if "--only-dataprep" in sys.argv:
        dataprep_goldens()
        return
    dataprep_goldens()
    validate_goldens()
    os.makedirs(ASSETS, exist_ok=True)
    np.set_printoptions(precision=9)

    # ---------------- weights + vocab + norm factors (assets the product ships) -------------
    models = {}
    for name, path in MODELS.items():
        m = load_model(path)
        models[name] = m
        flat_weights(m.state_dict()).tofile(os.path.join(ASSETS, f"weights_{name}.bin"))
    with open(os.path.join(HERE, "vocab66.txt"), "w") as f:
        f.write("\n".join(C.ALL_KMERS.tolist()) + "\n")
    for name, path in NORMS.items():
        d = joblib.load(path)
        kmers = np.array(sorted(d.keys()))
        mean = np.stack([np.asarray(d[k][0], np.float64) for k in kmers])
        std = np.stack([np.asarray(d[k][1], np.float64) for k in kmers])
        np.savez_compressed(os.path.join(ASSETS, f"norm_{name}.npz"), kmers=kmers, mean=mean, std=std)

    # ---------------- reference test data (data files only) ----------------------------------
    rdir = os.path.join(HERE, "ref_tests_data")
    os.makedirs(rdir, exist_ok=True)
    for fn in ("data.info", "data.json", "data.site_proba.csv.gz", "data.indiv_proba.csv.gz"):
        shutil.copyfile(os.path.join(DATA, fn), os.path.join(rdir, fn))
        os.chmod(os.path.join(rdir, fn), 0o644)

    # ---------------- bundled inputs via the reference's own dataset --------------------------
    ds = NanopolishDS(DATA, C.DEFAULT_MIN_READS, C.DEFAULT_NORM_PATH, mode="Inference")
    items = [ds[i] for i in range(len(ds))]
    feats, kmers, n_reads, tx_ids, tx_pos, read_ids = inference_collate(items)
    X = feats.numpy().astype(np.float32)
    kmers_pr = kmers.numpy()
    n_reads = n_reads.numpy()
    off = np.concatenate([[0], np.cumsum(n_reads)]).astype(np.int64)
    site_kmers = kmers_pr[off[:-1]].astype(np.uint8)
    raw = np.concatenate([ds.load_data(i)[3] for i in range(len(ds))]).astype(np.float64)
    kmer7 = np.array([ds.load_data(i)[4] for i in range(len(ds))])
    np.savez_compressed(os.path.join(HERE, "bundled_inputs.npz"), X=X, site_kmers=site_kmers, off=off,
                        read_ids=np.asarray(read_ids, np.float64), tx_ids=np.asarray(tx_ids[off[:-1]]),
                        tx_pos=np.asarray(tx_pos[off[:-1]], np.int64), raw_features=raw, kmer7=kmer7)
    print("bundled:", X.shape, len(off) - 1, "sites")

    rp = {name: read_probs(m, X, kmers_pr) for name, m in models.items()}
    np.savez_compressed(os.path.join(HERE, "bundled_readprob.npz"), **rp)

    # ---------------- full reference runs (n_processes=1 => deterministic) -------------------
    thr32 = np.float32(C.DEFAULT_READ_THRESHOLD)
    site_out = {}
    for (T, bs, spb, seed) in [(5, 16, 2, 0), (100, 16, 2, 0), (1000, 16, 2, 0), (50, 8, 3, 0),
                               (20, 13, 2, 7), (30, 51, 2, 0)]:
        tmp = tempfile.mkdtemp(prefix="m6a_run_")
        site_csv, indiv_csv = run_main([DATA], tmp, T, bs, spb, seed)
        df = parse_site_csv(site_csv)
        emu_site, emu_mod, written = emulate_sites(rp["hct116"], off, T, seed, bs, spb, thr32)
        # the emulation (reseed per flush group, sequential sites) must equal the real run
        # on every site the reference wrote
        assert len(df) == int(written.sum()), (len(df), written.sum())
        got = df["probability_modified"].to_numpy()
        # bitwise when the reference's per-batch sgemm rounds like the one-batch encode above
        # (it does at batch 16); other batch shapes move read probs by ~1e-8, hence the bound
        dmax = float(np.abs(got - emu_site[written].astype(np.float64)).max())
        print("   max |reference run - emulation| =", dmax)
        assert dmax < 5e-7, (T, bs, spb, dmax)
        # mod_ratio goes through "%.16f" (inference_utils.py:62): equal to 16 decimals, not bitwise
        assert np.abs(df["mod_ratio"].to_numpy() - emu_mod[written]).max() < 5e-16
        key = f"T{T}_bs{bs}_spb{spb}_seed{seed}"
        site_out[key + "_site"] = emu_site
        site_out[key + "_written"] = written
        site_out[key + "_mod"] = emu_mod
        print(key, "written", int(written.sum()), "of", len(written), "emulation == reference run")
        if (T, bs, spb, seed) == (5, 16, 2, 0):
            with open(os.path.join(HERE, "config1_site_proba.csv"), "wb") as f:
                f.write(site_csv)
            with gzip.GzipFile(os.path.join(HERE, "config1_indiv_proba.csv.gz"), "wb", mtime=0) as f:
                f.write(indiv_csv)
        shutil.rmtree(tmp)
    np.savez_compressed(os.path.join(HERE, "bundled_site.npz"), **site_out)

    # replicate run (two copies of the bundled dir): pins the `_0/_1` read ids + pooled bags
    tmp = tempfile.mkdtemp(prefix="m6a_rep_")
    rep = os.path.join(tmp, "rep1")
    os.makedirs(rep)
    for fn in ("data.info", "data.json"):
        shutil.copyfile(os.path.join(DATA, fn), os.path.join(rep, fn))
    site_csv, indiv_csv = run_main([DATA, rep], os.path.join(tmp, "out"), 5, 16, 2, 0)
    with open(os.path.join(HERE, "replicate_site_proba.csv"), "wb") as f:
        f.write(site_csv)
    with gzip.GzipFile(os.path.join(HERE, "replicate_indiv_proba.csv.gz"), "wb", mtime=0) as f:
        f.write(indiv_csv)
    shutil.rmtree(tmp)

    # ---------------- RNG known answers ----------------------------------------------------------
    rng = {}
    for seed in (0, 1, 42, 20250328, 4294967295):
        rs = np.random.RandomState(seed)
        # raw 32-bit outputs: randint over the full uint32 range draws next_uint32 directly
        rng[f"raw_seed{seed}"] = rs.randint(0, 2**32, size=4096, dtype=np.uint64).astype(np.uint32)
    for n in (1, 2, 3, 20, 23, 32, 33, 64, 65, 500, 662, 1000, 1024, 1025, 70000):
        np.random.seed(0)
        rng[f"choice_seed0_n{n}"] = np.random.choice(np.arange(n), 4096, replace=True).astype(np.int32)
    # sequential consumption across bags of different size from one stream
    np.random.seed(42)
    seq = [np.random.choice(np.arange(n), 100, replace=True) for n in (20, 33, 1, 64, 21)]
    rng["choice_seed42_seq_20_33_1_64_21"] = np.concatenate(seq).astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "rng_known.npz"), **rng)

    # ---------------- synthetic small (this repo's generator, reference's arithmetic) ------
    from m6anet_amd import synthetic
    syn = {}
    for tag, mname, kw in [("uniform20", "hct116", dict(n_sites=1000, bag=20)),
                           ("ragged", "hek293t_glori", dict(n_sites=200, bag=(50, 500)))]:
        d = synthetic.make_sites(seed=20250328, **kw)
        Xs, sk, offs = d["X"], d["site_kmers"], d["off"]
        nr = np.diff(offs)
        kpr = np.repeat(sk.astype(np.int64), nr, axis=0)
        p = read_probs(models[mname], Xs, kpr)
        syn[f"{tag}_readprob"] = p
        syn[f"{tag}_off"] = offs
        syn[f"{tag}_xsum"] = np.array([np.float64(Xs.astype(np.float64).sum()), float(sk.sum())])
        for T in (100, 1000):
            s, mr, _ = emulate_sites(p, offs, T, 0, 16, 2, thr32)
            syn[f"{tag}_site_T{T}"] = s
            syn[f"{tag}_mod"] = mr
        print("synthetic", tag, Xs.shape, "reads; site[0:3] =", syn[f"{tag}_site_T1000"][:3])
    np.savez_compressed(os.path.join(HERE, "synthetic_small.npz"), **syn)

    # ---------------- MILModel.forward on fixed bags (a8) -------------------------------------
    g = np.random.Generator(np.random.PCG64(7))
    B = 64
    Xb = np.clip(g.standard_normal((B, 20, 9)), -6, 6).astype(np.float32)
    kb = np.repeat(g.integers(0, 66, size=(B, 1, 3)), 20, axis=1).astype(np.int64)
    with torch.no_grad():
        yb = models["hct116"]({"X": torch.from_numpy(Xb), "kmer": torch.from_numpy(kb)}).numpy()
    np.savez_compressed(os.path.join(HERE, "bag_forward.npz"), X=Xb, kmer=kb[:, 0, :].astype(np.uint8),
                        site_prob=yb.astype(np.float32))
    print("done")


if __name__ == "__main__":
    main()
