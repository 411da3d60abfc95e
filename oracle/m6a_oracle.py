"""ctypes loader for the CPU oracle (oracle/m6a_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this
module; the product (m6anet_amd/) never does.  Parity status: pinned -- see
tests/test_oracle_golden.py and tests/test_reference_at_scale.py (read and site probabilities of the 20-read-bag capture
reproduced bit for bit).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("M6A_ORACLE_LIB") or os.path.join(_HERE, "_build", "libm6a_oracle.so")   # M6A_ORACLE_LIB: the sanitizer build
_lib = None


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("m6a_oracle.c", "m6a_oracle.h", "Makefile")]
    if os.environ.get("M6A_ORACLE_LIB"):
        return _SO
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-s", "-C", _HERE], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()                                   # no-op when the library is newer than its sources
        L = C.CDLL(_SO)
        f32p, i64p, u8p, i32p, u32p, f64p = (C.POINTER(t) for t in
                                             (C.c_float, C.c_int64, C.c_uint8, C.c_int32, C.c_uint32, C.c_double))
        L.m6a_or_mt_fill.argtypes = [C.c_uint32, C.c_int64, u32p]
        L.m6a_or_mt_seed.argtypes = [C.c_void_p, C.c_uint32]
        L.m6a_or_choice.argtypes = [C.c_void_p, C.c_int64, C.c_int64, i32p]
        L.m6a_or_pairwise_sum_f32.argtypes = [f32p, C.c_int64]
        L.m6a_or_pairwise_sum_f32.restype = C.c_float
        L.m6a_or_encode_reads_mt.argtypes = [f32p, f32p, u8p, i64p, C.c_int64, C.c_int, f32p]
        L.m6a_or_encode_layers.argtypes = [f32p, f32p, u8p, i64p, C.c_int64, f32p, f32p, f32p]
        L.m6a_or_flush_groups.argtypes = [C.c_int64, C.c_int64, C.c_int64, i64p]
        L.m6a_or_flush_groups.restype = C.c_int64
        L.m6a_or_site_pool.argtypes = [f32p, i64p, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_uint32,
                                       C.c_int64, C.c_int64, C.c_int, f32p, f64p]
        L.m6a_or_site_pool.restype = C.c_int
        L.m6a_or_site_pool_at.argtypes = [f32p, i64p, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_uint32,
                                          C.c_int64, C.c_int64, C.c_int64, C.c_int, f32p, f64p]
        L.m6a_or_site_pool_at.restype = C.c_int
        L.m6a_or_bag_noisy_or.argtypes = [f32p, C.c_int64, C.c_int, f32p]
        L.m6a_or_validation_indices.argtypes = [C.c_uint32, i64p, C.c_int64, C.c_int, C.c_int, i32p]
        L.m6a_or_validate.argtypes = [f32p, i64p, C.c_int64, C.c_int, C.c_int, C.c_uint32, f32p, f32p]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class MT(C.Structure):
    _fields_ = [("mt", C.c_uint32 * 624), ("pos", C.c_int)]


def mt_raw(seed, count):
    out = np.empty(count, np.uint32)
    lib().m6a_or_mt_fill(seed, count, _p(out, C.c_uint32))
    return out


def choice_stream(seed, ns_counts):
    """Sequential choice(n, count) calls on one stream seeded once; returns list of index arrays."""
    st = MT()
    lib().m6a_or_mt_seed(C.byref(st), seed)
    outs = []
    for n, count in ns_counts:
        o = np.empty(count, np.int32)
        lib().m6a_or_choice(C.byref(st), n, count, _p(o, C.c_int32))
        outs.append(o)
    return outs


def pairwise_sum(a):
    a = np.ascontiguousarray(a, np.float32)
    return np.float32(lib().m6a_or_pairwise_sum_f32(_p(a, C.c_float), a.size))


def encode_reads(weights, X, site_kmers, off, n_threads=1):
    weights = np.ascontiguousarray(weights, np.float32)
    X = np.ascontiguousarray(X, np.float32)
    site_kmers = np.ascontiguousarray(site_kmers, np.uint8)
    off = np.ascontiguousarray(off, np.int64)
    assert weights.size == 7997 and X.shape[0] == off[-1]
    out = np.empty(int(off[-1]), np.float32)
    lib().m6a_or_encode_reads_mt(_p(weights, C.c_float), _p(X, C.c_float), _p(site_kmers, C.c_uint8),
                                 _p(off, C.c_int64), len(off) - 1, n_threads, _p(out, C.c_float))
    return out


def encode_layers(weights, X, site_kmers, off):
    """(read_prob [R], hidden [R][32] = the read representation, logit [R]) -- one thread."""
    weights = np.ascontiguousarray(weights, np.float32)
    X = np.ascontiguousarray(X, np.float32)
    site_kmers = np.ascontiguousarray(site_kmers, np.uint8)
    off = np.ascontiguousarray(off, np.int64)
    assert weights.size == 7997 and X.shape[0] == off[-1]
    R = int(off[-1])
    p, h, z = np.empty(R, np.float32), np.empty((R, 32), np.float32), np.empty(R, np.float32)
    lib().m6a_or_encode_layers(_p(weights, C.c_float), _p(X, C.c_float), _p(site_kmers, C.c_uint8), _p(off, C.c_int64),
                               len(off) - 1, _p(p, C.c_float), _p(h, C.c_float), _p(z, C.c_float))
    return p, h, z


def flush_groups(n_sites, batch_size, save_per_batch):
    nb = (n_sites + batch_size - 1) // batch_size
    g = np.zeros(nb + 2, np.int64)
    G = lib().m6a_or_flush_groups(n_sites, batch_size, save_per_batch, _p(g, C.c_int64))
    return g[:G + 1].copy()


def site_pool(read_prob, off, n_iters, thr, seed=0, batch_size=16, save_per_batch=2, n_samples=20,
              n_threads=1, first_site=0):
    read_prob = np.ascontiguousarray(read_prob, np.float32)
    off = np.ascontiguousarray(off, np.int64)
    S = len(off) - 1
    site = np.empty(S, np.float32)
    mod = np.empty(S, np.float64)
    rc = lib().m6a_or_site_pool_at(_p(read_prob, C.c_float), _p(off, C.c_int64), S, n_iters, n_samples,
                                   np.float32(thr), seed, batch_size, save_per_batch, first_site, n_threads,
                                   _p(site, C.c_float), _p(mod, C.c_double))
    assert rc == 0
    return site, mod


def bag_noisy_or(read_prob, bag=20):
    read_prob = np.ascontiguousarray(read_prob, np.float32)
    nb = read_prob.size // bag
    out = np.empty(nb, np.float32)
    lib().m6a_or_bag_noisy_or(_p(read_prob, C.c_float), nb, bag, _p(out, C.c_float))
    return out


def validation_indices(seed, off, n_iters, k=20):
    """[n_iters][S][k] read indices of the training-mode sampler (choice without replacement)."""
    off = np.ascontiguousarray(off, np.int64)
    S = len(off) - 1
    idx = np.empty((n_iters, S, k), np.int32)
    rc = lib().m6a_or_validation_indices(seed, _p(off, C.c_int64), S, n_iters, k, _p(idx, C.c_int32))
    if rc:
        raise ValueError("a bag holds fewer than %d reads" % k)
    return idx


def validate(read_prob, off, n_iters, seed=0, k=20):
    """(y_pred [n_iters][S], y_pred_avg [S]) of validate() at num_workers=0."""
    read_prob = np.ascontiguousarray(read_prob, np.float32)
    off = np.ascontiguousarray(off, np.int64)
    S = len(off) - 1
    y = np.empty((n_iters, S), np.float32)
    avg = np.empty(S, np.float32)
    rc = lib().m6a_or_validate(_p(read_prob, C.c_float), _p(off, C.c_int64), S, n_iters, k, seed,
                               _p(y, C.c_float), _p(avg, C.c_float))
    if rc:
        raise ValueError("a bag holds fewer than %d reads" % k)
    return y, avg
