"""Loader for dataprep output (`data.info` + `data.json`), inference mode only.

Host-side counterpart of the reference's NanopolishDS / NanopolishReplicateDS / inference_collate
(m6anet/utils/data_utils.py:20-291, 293-427, 498-506) -- same filtering, feature selection and
normalisation -- but it produces the flat layout the C ABI takes (include/m6a.h): X f32 [R,9],
site_kmers u8 [S,3] (once per site), off i64 [S+1], plus ids for the CSV writers.

  * sites with n_reads >= min_reads, in data.info order          (data_utils.py:118-129)
  * one JSON record per site at byte range [start, end)           (data_utils.py:169-190)
  * features = columns `indices` of the record, last column = read id (data_utils.py:105-116,166)
  * z-normalisation in float64, then float32                       (data_utils.py:216-218,233-248)
  * 7-mer -> three 5-mer vocabulary ids                            (data_utils.py:195-196,223)
  * replicates: union of sites over the directories, reads concatenated in directory order,
    read ids "<int id>_<replicate>"                                (data_utils.py:341-427)
"""
import csv
import json
import os

import numpy as np

from .constants import DEFAULT_MIN_READS, KMER_TO_INT, NUM_NEIGHBORING_FEATURES, asset_path


def load_norm_factors(path):
    """dict kmer -> (mean f64[3], std f64[3]).  Accepts this repo's .npz or the reference's .joblib."""
    if path is None:
        return None
    if not os.path.exists(path) and os.path.exists(asset_path(path)):
        path = asset_path(path)
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            kmers, mean, std = z["kmers"], z["mean"], z["std"]     # one read each (an NpzFile re-reads a member per access)
        return {str(k): (mean[i], std[i]) for i, k in enumerate(kmers)}
    import joblib
    d = joblib.load(path)
    return {k: (np.asarray(v[0], np.float64), np.asarray(v[1], np.float64)) for k, v in d.items()}


def _read_info(root_dir):
    rows = []
    with open(os.path.join(root_dir, "data.info"), newline="") as f:
        for r in csv.DictReader(f):
            rows.append((r["transcript_id"], int(r["transcript_position"]), int(r["start"]), int(r["end"]),
                         int(r["n_reads"])))
    return rows


class SiteBatch:
    """Everything one job needs, flat.  `native` is the libm6a_io handle when the batch came from
    the native loader (ids then stay on the C side and the CSVs are written natively)."""

    def __init__(self, X, site_kmers, off, tx_ids, tx_pos, read_ids, kmer5, native=None):
        self.X, self.site_kmers, self.off = X, site_kmers, off
        self._tx_ids, self.tx_pos, self.read_ids, self._kmer5 = tx_ids, tx_pos, read_ids, kmer5
        self.native = native

    # id strings stay on the C side for native batches until somebody asks for them
    @property
    def tx_ids(self):
        if self._tx_ids is None:
            self._tx_ids = [self.native.tx_id(i) for i in range(self.n_sites)]
        return self._tx_ids

    @property
    def kmer5(self):
        if self._kmer5 is None:
            self._kmer5 = [self.native.kmer5(i) for i in range(self.n_sites)]
        return self._kmer5

    @property
    def n_sites(self):
        return len(self.off) - 1

    @property
    def n_reads(self):
        return np.diff(self.off)


def load_sites_native(input_dirs, min_reads=DEFAULT_MIN_READS, norm_path=None, n_threads=0):
    """Same result as load_sites (bit-identical arrays, tests/test_host_io.py) through the native
    multi-threaded loader of libm6a_io.so."""
    from . import _io
    if isinstance(input_dirs, str):
        input_dirs = [input_dirs]
    nat = _io.NativeSites(list(input_dirs), min_reads, load_norm_factors(norm_path), n_threads)
    return SiteBatch(nat.X, nat.site_kmers, nat.off, None, nat.tx_pos, None, None, native=nat)


STORE_SUFFIX = ".m6astore"


def norm_digest(norm):
    """Content hash of a set of normalisation factors (dict kmer -> (mean[3], std[3]), or None): what a binary site
    store's header records, so that two different files with the same name cannot be confused."""
    import hashlib
    if not norm:
        return "none"
    h = hashlib.sha256()
    for k in sorted(norm):
        h.update(k.encode())
        h.update(np.ascontiguousarray(norm[k][0], np.float64).tobytes())
        h.update(np.ascontiguousarray(norm[k][1], np.float64).tobytes())
    return h.hexdigest()[:24]


def store_tag(norm_path, min_reads=DEFAULT_MIN_READS):
    """What a binary site store was built with (it holds NORMALISED features of the sites that passed the
    read-count filter): the content hash of the normalisation factors -- "none" for un-normalised features, which is
    what the reference feeds when --norm_path is absent -- and the filter.  Checked whenever the store is opened."""
    return "norm=%s min_reads=%d" % (norm_digest(load_norm_factors(norm_path)), min_reads)


def pack_sites(input_dirs, out_path, min_reads=DEFAULT_MIN_READS, norm_path=None, n_threads=0):
    """data.info + data.json (one directory, or several replicates) -> one binary site store: the dataset is parsed
    and normalised once, later runs map the file."""
    from . import _io
    if isinstance(input_dirs, str):
        input_dirs = [input_dirs]
    nat = _io.NativeSites(list(input_dirs), min_reads, load_norm_factors(norm_path), n_threads)
    nat.save_store(out_path, store_tag(norm_path, min_reads))
    return nat


def open_store(path, norm_path=None, min_reads=DEFAULT_MIN_READS):
    """Maps a binary site store (zero-copy views into the page cache).  Refuses a store that was normalised with
    other factors / another read-count filter than this run's: norm_path=None means the run expects UN-normalised
    features (the reference's behaviour without --norm_path), so a normalised store is refused then too."""
    from . import _io
    nat = _io.NativeSites(store=path)
    want = store_tag(norm_path, min_reads)
    if nat.tag != want:
        nat.close()
        raise ValueError("%s was packed with '%s', this run needs '%s' (normalisation factors %s): re-run `m6anet_amd pack`"
                         % (path, nat.tag, want, norm_path))
    return SiteBatch(nat.X, nat.site_kmers, nat.off, None, nat.tx_pos, None, None, native=nat)


def load_sites(input_dirs, min_reads=DEFAULT_MIN_READS, norm_path=None,
               num_neighboring_features=NUM_NEIGHBORING_FEATURES):
    """input_dirs: one directory (NanopolishDS) or several (NanopolishReplicateDS).  Pure-Python
    reference implementation of the loader; load_sites_native is the fast path."""
    if isinstance(input_dirs, str):
        input_dirs = [input_dirs]
    replicate = len(input_dirs) > 1
    norm = load_norm_factors(norm_path)
    blobs = []
    for d in input_dirs:
        with open(os.path.join(d, "data.json"), "rb") as f:
            blobs.append(f.read())

    # site list: single directory keeps data.info order; replicates take the union in order of
    # first appearance with summed read counts
    sites, index = [], {}
    for rep, d in enumerate(input_dirs):
        for tx, pos, start, end, n in _read_info(d):
            key = (tx, pos)
            if key not in index:
                index[key] = len(sites)
                sites.append([tx, pos, 0, []])
            s = sites[index[key]]
            s[2] += n
            s[3].append((rep, start, end))
    sites = [s for s in sites if s[2] >= min_reads]
    if not sites:
        raise ValueError("no site with at least %d reads in %s" % (min_reads, input_dirs))

    def record(tx, pos, rep, start, end):
        rec = json.loads(blobs[rep][start:end])[tx][str(pos)]
        if len(rec) != 1:
            raise ValueError("site %s:%d has %d sequence keys" % (tx, pos, len(rec)))
        (kmer, feats), = rec.items()
        return kmer, np.array(feats, dtype=np.float64)

    # feature columns (data_utils.py:105-116): inferred from the first site's sequence length
    k0, _ = record(sites[0][0], sites[0][1], *sites[0][3][0])
    total_nb = (len(k0) - 5) // 2
    nb = num_neighboring_features
    if total_nb != nb:
        # the reference's slice for this case (data_utils.py:276-277) yields a 4-mer and fails too
        raise ValueError("data.json was prepared with n_neighbors=%d; only %d is supported" % (total_nb, nb))
    idx = np.array([(total_nb - nb + j) * 3 + i for j in range(nb) for i in range(3)] +
                   [total_nb * 3 + i for i in range(3)] +
                   [(total_nb + j) * 3 + i for j in range(1, nb + 1) for i in range(3)])

    Xs, kms, rids, n_reads, tx_ids, tx_pos, kmer5 = [], [], [], [], [], [], []
    for tx, pos, _, parts in sites:
        feats, ids, kmer = [], [], None
        for rep, start, end in parts:
            k, a = record(tx, pos, rep, start, end)
            if kmer is None:
                kmer = k
            elif kmer != k:
                raise ValueError("replicates disagree on the sequence of %s:%d" % (tx, pos))
            feats.append(a[:, idx])
            if replicate:
                ids.extend("%d_%d" % (int(r), rep) for r in a[:, -1])
            else:
                ids.extend(a[:, -1])
        feats = np.concatenate(feats)
        k5 = [kmer[i:i + 5] for i in range(2 * nb + 1)]
        if norm is not None:
            mean = np.concatenate([norm[k][0] for k in k5])
            std = np.concatenate([norm[k][1] for k in k5])
            feats = (feats - mean) / std
        Xs.append(feats.astype(np.float32))
        kms.append([KMER_TO_INT[k] for k in k5])
        rids.append(ids)
        n_reads.append(len(feats))
        tx_ids.append(tx)
        tx_pos.append(pos)
        kmer5.append(k5[nb])                     # centre 5-mer, the CSV's kmer column (inference_utils.py:39)
    off = np.zeros(len(sites) + 1, np.int64)
    np.cumsum(n_reads, out=off[1:])
    return SiteBatch(np.ascontiguousarray(np.concatenate(Xs)), np.array(kms, np.uint8), off,
                     tx_ids, np.array(tx_pos, np.int64), rids, kmer5)
