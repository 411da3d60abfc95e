"""Deterministic synthetic DRACH site bags.

Generator: numpy.random.Generator(PCG64(seed)); per site one of the 288 N-DRACH-N 7-mers
uniformly -> three vocabulary ids (u8); already-normalised signal features
X ~ N(0,1) clipped to +-6, float32, [R, 9] (they are z-scores after
m6anet/utils/data_utils.py:216-218); bag size fixed (20) or integers(lo, hi+1); CSR `off`.
"""
import numpy as np

from .constants import ALL_7MERS, kmer7_to_ids

_IDS_288 = np.array([kmer7_to_ids(k) for k in ALL_7MERS], dtype=np.uint8)   # [288,3]


def bag_sizes(n_sites, bag=20, seed=20250328):
    """int64 [n_sites] reads per site: fixed, or integers(lo, hi+1) for bag = (lo, hi)."""
    if isinstance(bag, (tuple, list)):
        g = np.random.Generator(np.random.PCG64([seed, 7]))
        return g.integers(bag[0], bag[1] + 1, size=n_sites).astype(np.int64)
    return np.full(n_sites, int(bag), np.int64)


def make_sites(n_sites, bag=20, seed=20250328, chunk_reads=1 << 22, n_reads=None, prefix_sites=None):
    """Returns dict(X f32 [R,9], site_kmers u8 [S,3], off i64 [S+1]).  `n_reads` (int64 [n_sites])
    fixes the bag sizes explicitly -- a rank's slice of a job-wide bag_sizes() array.
    `prefix_sites` = P returns sites [0, P) of the n_sites-job, bit for bit, without generating the
    features of the rest (the feature stream is drawn last and in read order, so a prefix of the job is a
    prefix of the stream): how tests/golden/reference_at_scale.npz addresses "the first P sites of
    BASELINE.json configs[2] / configs[4]" cheaply."""
    g = np.random.Generator(np.random.PCG64(seed))
    site_kmers = _IDS_288[g.integers(0, len(ALL_7MERS), size=n_sites)]
    if n_reads is not None:
        n_reads = np.ascontiguousarray(n_reads, np.int64)
        assert n_reads.shape == (n_sites,)
    elif isinstance(bag, (tuple, list)):
        n_reads = g.integers(bag[0], bag[1] + 1, size=n_sites).astype(np.int64)
    else:
        n_reads = np.full(n_sites, int(bag), np.int64)
    off = np.zeros(n_sites + 1, np.int64)
    np.cumsum(n_reads, out=off[1:])
    if prefix_sites is not None:
        P = int(prefix_sites)
        assert 0 <= P <= n_sites
        off, site_kmers = off[:P + 1].copy(), site_kmers[:P]
    R = int(off[-1])
    X = np.empty((R, 9), np.float32)
    for a in range(0, R, chunk_reads):
        b = min(R, a + chunk_reads)
        blk = g.standard_normal((b - a, 9), dtype=np.float32)
        np.clip(blk, -6.0, 6.0, out=X[a:b])
    return {"X": X, "site_kmers": np.ascontiguousarray(site_kmers), "off": off}
