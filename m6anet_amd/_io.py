"""ctypes binding of libm6a_io.so (include/m6a_io.h): native loader + CSV writers."""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("M6A_IO_LIB") or os.path.join(_PKG, "libm6a_io.so")   # M6A_IO_LIB: e.g. the sanitizer build (tests/sanitize.sh)
SYMBOLS = ["m6a_io_last_error", "m6a_io_load_sites", "m6a_io_free", "m6a_io_n_sites", "m6a_io_n_reads",
           "m6a_io_n_replicates", "m6a_io_X", "m6a_io_site_kmers", "m6a_io_off", "m6a_io_tx_pos",
           "m6a_io_read_ids", "m6a_io_read_rep", "m6a_io_tx_id", "m6a_io_kmer5", "m6a_io_write_csv", "m6a_io_write_csv_n", "m6a_io_csv_shard_size", "m6a_io_csv_shard_write", "m6a_io_csv_header_bytes", "m6a_io_format_f16", "m6a_io_py_repr", "m6a_io_repr_rounded",
           "m6a_io_save_store", "m6a_io_open_store", "m6a_io_store_tag", "m6a_io_dataprep"]
_lib = None


class M6AIOError(RuntimeError):
    """`code` = the M6A_IO_E* value the call returned (include/m6a_io.h: -1 EINVAL, -2 ENOMEM, -3 EIO, -4 EFORMAT)."""

    def __init__(self, msg, code=None):
        super().__init__(msg)
        self.code = code


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libm6a_io.so is not built: run `python -m m6anet_amd.build`")
    L = C.CDLL(LIB_PATH)
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
    L.m6a_io_last_error.restype = C.c_char_p
    L.m6a_io_load_sites.argtypes = [C.POINTER(C.c_char_p), i32, i32, C.c_char_p, vp, vp, i32, i32, C.POINTER(vp)]
    L.m6a_io_free.argtypes = [vp]
    L.m6a_io_free.restype = None
    for name, rt in (("m6a_io_n_sites", i64), ("m6a_io_n_reads", i64), ("m6a_io_n_replicates", i32)):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = rt
    for name in ("m6a_io_X", "m6a_io_site_kmers", "m6a_io_off", "m6a_io_tx_pos", "m6a_io_read_ids", "m6a_io_read_rep"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = vp
    for name in ("m6a_io_tx_id", "m6a_io_kmer5"):
        getattr(L, name).argtypes = [vp, i64]
        getattr(L, name).restype = C.c_char_p
    L.m6a_io_write_csv.argtypes = [vp, C.c_char_p, vp, vp, vp, i32, i32]
    L.m6a_io_write_csv_n.argtypes = [vp, C.c_char_p, vp, vp, vp, i32, i32, i64]
    L.m6a_io_csv_shard_size.argtypes = [vp, vp, vp, vp, i64, i64, i32, C.POINTER(i64), C.POINTER(i64)]
    L.m6a_io_csv_shard_write.argtypes = [vp, C.c_char_p, vp, vp, vp, i64, i64, i32, i64, i64, i32, i64, i64]
    L.m6a_io_csv_header_bytes.argtypes = [i32]
    L.m6a_io_csv_header_bytes.restype = i64
    L.m6a_io_format_f16.argtypes = [C.c_double, C.c_char_p]
    L.m6a_io_format_f16.restype = i32
    L.m6a_io_py_repr.argtypes = [C.c_double, C.c_char_p]
    L.m6a_io_py_repr.restype = i32
    L.m6a_io_repr_rounded.argtypes = [C.c_double, i32, C.c_char_p]
    L.m6a_io_repr_rounded.restype = i32
    L.m6a_io_save_store.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.m6a_io_open_store.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.m6a_io_store_tag.argtypes = [vp]
    L.m6a_io_store_tag.restype = C.c_char_p
    L.m6a_io_dataprep.argtypes = [C.c_char_p, C.c_char_p, i32, i32, i32, i32, i32, i32, i32]
    _lib = L
    return L


def usable_cpus():
    """CPUs this process may really use: the affinity mask cut by the cgroup quota (the GPU boxes lease 16 of 256)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.999)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _chk(rc):
    if rc != 0:
        raise M6AIOError("m6a_io error %d: %s" % (rc, load().m6a_io_last_error().decode()), rc)


def dataprep(eventalign, out_dir, n_threads=0, readcount_min=1, readcount_max=1000, min_segment_count=20,
             n_neighbors=1, compress=False, skip_index=False):
    """Native `m6anet dataprep` (m6anet/scripts/dataprep.py:54-70)."""
    os.makedirs(out_dir, exist_ok=True)
    _chk(load().m6a_io_dataprep(os.fsencode(eventalign), os.fsencode(out_dir), int(n_threads), int(readcount_min),
                                int(readcount_max), int(min_segment_count), int(n_neighbors), 1 if compress else 0,
                                1 if skip_index else 0))


def _weakrefable(x):
    try:
        import weakref
        weakref.ref(x)
        return True
    except TypeError:
        return False


class NativeSites:
    """Owns an m6a_sites handle; exposes its arrays as zero-copy numpy views."""

    def __init__(self, input_dirs=None, min_reads=20, norm=None, n_threads=0, store=None):
        """Either parses `input_dirs` (data.info + data.json per directory) or maps a binary site `store` file."""
        L = load()
        h = C.c_void_p()
        if store is not None:
            _chk(L.m6a_io_open_store(os.fsencode(store), C.byref(h)))
        else:
            dirs = (C.c_char_p * len(input_dirs))(*[os.fsencode(d) for d in input_dirs])
            if norm:
                kmers = sorted(norm)
                blob = "".join(kmers).encode()
                mean = np.ascontiguousarray([norm[k][0] for k in kmers], np.float64)
                std = np.ascontiguousarray([norm[k][1] for k in kmers], np.float64)
                args = (blob, mean.ctypes.data, std.ctypes.data, len(kmers))
            else:
                args = (None, None, None, 0)
            _chk(L.m6a_io_load_sites(dirs, len(input_dirs), int(min_reads), *args, int(n_threads), C.byref(h)))
        self._h, self._L = h, L
        self.tag = L.m6a_io_store_tag(h).decode()
        S, R = L.m6a_io_n_sites(h), L.m6a_io_n_reads(h)
        self.n_replicates = L.m6a_io_n_replicates(h)

        def view(fn, ctype, shape):
            n = int(np.prod(shape))
            arr = np.ctypeslib.as_array(C.cast(fn(h), C.POINTER(ctype)), shape=(n,)).reshape(shape)
            arr.flags.writeable = False
            return arr
        self.X = view(L.m6a_io_X, C.c_float, (R, 9))
        self.site_kmers = view(L.m6a_io_site_kmers, C.c_uint8, (S, 3))
        self.off = view(L.m6a_io_off, C.c_int64, (S + 1,))
        self.tx_pos = view(L.m6a_io_tx_pos, C.c_int64, (S,))
        self.read_id_values = view(L.m6a_io_read_ids, C.c_double, (R,))
        self.read_rep = view(L.m6a_io_read_rep, C.c_int32, (R,))

    def tx_id(self, i):
        return self._L.m6a_io_tx_id(self._h, i).decode()

    def kmer5(self, i):
        return self._L.m6a_io_kmer5(self._h, i).decode()

    def save_store(self, path, tag=""):
        """Writes everything this handle holds as one binary site store (include/m6a_io.h)."""
        _chk(self._L.m6a_io_save_store(self._h, os.fsencode(path), tag.encode()[:63]))

    def write_csv(self, out_dir, read_prob, site_prob, mod_ratio, write_header=False, n_threads=0, n_sites=None):
        self._drop_shard_kept()
        rp = np.ascontiguousarray(read_prob, np.float32)
        sp = np.ascontiguousarray(site_prob, np.float32)
        mr = np.ascontiguousarray(mod_ratio, np.float64)
        assert rp.size == self.X.shape[0] and sp.size == self.tx_pos.size == mr.size
        _chk(self._L.m6a_io_write_csv_n(self._h, os.fsencode(out_dir), rp.ctypes.data, sp.ctypes.data, mr.ctypes.data,
                                        1 if write_header else 0, int(n_threads), -1 if n_sites is None else int(n_sites)))

    def _shard_args(self, a, b, read_prob, site_prob, mod_ratio):
        rp = np.ascontiguousarray(read_prob, np.float32)
        sp = np.ascontiguousarray(site_prob, np.float32)
        mr = np.ascontiguousarray(mod_ratio, np.float64)
        assert sp.size == mr.size == b - a and rp.size == int(self.off[b] - self.off[a])
        return rp, sp, mr

    def _drop_shard_kept(self):
        self._shard_kept = None

    def csv_shard_size(self, a, b, read_prob, site_prob, mod_ratio, n_threads=0):
        """Bytes the rows of sites [a, b) take in (data.site_proba.csv, data.indiv_proba.csv); the arrays hold that range only."""
        import weakref
        self._drop_shard_kept()                      # an earlier size call that was never followed by its write
        rp, sp, mr = self._shard_args(a, b, read_prob, site_prob, mod_ratio)
        ns, ni = C.c_int64(), C.c_int64()
        _chk(self._L.m6a_io_csv_shard_size(self._h, rp.ctypes.data, sp.ctypes.data, mr.ctypes.data, int(a), int(b), int(n_threads),
                                           C.byref(ns), C.byref(ni)))        # raises: nothing is kept
        # The library keeps the text it formatted for the csv_shard_write that follows, keyed on the arrays' addresses (and a
        # checksum).  Kept here until then: CONVERTED temporaries (strongly -- nobody else holds them, and the write must pass the
        # same addresses), and only weak references to the caller's own objects (read_prob is 4 bytes per read and
        # can reach GBs; a size call that is never followed by its write must not pin it for the life of this handle).  If a
        # caller array is gone by then, or another csv_* call came between, the write formats again -- correct, just slower.
        def keep(conv, orig):
            return (weakref.ref(orig), None) if conv is orig else (weakref.ref(orig) if _weakrefable(orig) else None, conv)
        self._shard_kept = ((int(a), int(b)), tuple(keep(c_, o_) for c_, o_ in zip((rp, sp, mr), (read_prob, site_prob, mod_ratio))))
        return ns.value, ni.value

    def csv_shard_write(self, out_dir, a, b, read_prob, site_prob, mod_ratio, site_offset, indiv_offset, header_and_totals=None, n_threads=0):
        """pwrite()s the rows of sites [a, b) at the given byte offsets; `header_and_totals` = (site_total, indiv_total) on the
        one rank that also writes the header lines and sets the files' final sizes."""
        kept, self._shard_kept = getattr(self, "_shard_kept", None), None
        arrs = None
        if kept is not None and kept[0] == (int(a), int(b)):
            same = [r is not None and r() is o for (r, _), o in zip(kept[1], (read_prob, site_prob, mod_ratio))]
            if all(same):                            # the very arrays csv_shard_size formatted: the kept text is reused
                arrs = tuple(conv if conv is not None else o for (_, conv), o in zip(kept[1], (read_prob, site_prob, mod_ratio)))
        del kept
        rp, sp, mr = arrs if arrs is not None else self._shard_args(a, b, read_prob, site_prob, mod_ratio)
        st, it = header_and_totals if header_and_totals is not None else (-1, -1)
        _chk(self._L.m6a_io_csv_shard_write(self._h, os.fsencode(out_dir), rp.ctypes.data, sp.ctypes.data, mr.ctypes.data, int(a), int(b),
                                            int(n_threads), int(site_offset), int(indiv_offset), 1 if header_and_totals is not None else 0,
                                            int(st), int(it)))

    def csv_header_bytes(self):
        return int(self._L.m6a_io_csv_header_bytes(0)), int(self._L.m6a_io_csv_header_bytes(1))

    def close(self):
        self._shard_kept = None
        if getattr(self, "_h", None):
            for name in ("X", "site_kmers", "off", "tx_pos", "read_id_values", "read_rep"):
                setattr(self, name, None)
            self._L.m6a_io_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
