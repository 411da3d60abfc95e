"""Host-side mirror of the reference's operator interface for the inference hot path, on top of
the C ABI (include/m6a.h).  Names follow the reference:

  M6ANetEngine.get_read_probability(X, site_kmers, off)
      == model.get_read_representation({'X','kmer'}) + model.pooling_filter.probability_layer(.)
         (m6anet/utils/inference_utils.py:35-37)
  M6ANetEngine.calculate_site_proba(read_probs, off, n_iters, n_samples, ...)
      == calculate_site_proba(read_probs, n_iters, n_samples, n_processes=1) + mod_ratio
         (m6anet/utils/inference_utils.py:53-54,90-104)
  M6ANetEngine.infer(...)     == one whole run_inference job minus text I/O (inference_utils.py:14-71)
  M6ANetEngine.forward(X, kmer)  == MILModel.forward on fixed bags (m6anet/model/model.py:155-164)

Arrays may be numpy arrays (host pointers: the library stages them and the call is synchronous)
or torch tensors on the context's GPU (device pointers: zero-copy, stream-ordered).  PyTorch is
only plumbing here (device memory + streams); nothing in this file computes.
"""
import ctypes as C

import numpy as np

from . import _lib
from .constants import (DEFAULT_PRETRAINED_MODEL, DEFAULT_READ_THRESHOLD, N_SAMPLES, N_WEIGHT_FLOATS,
                        PRETRAINED_CONFIGS, asset_path)


def load_weights(pretrained_model=DEFAULT_PRETRAINED_MODEL):
    """Flat float32 weight blob (layout in include/m6a.h) of a bundled pretrained model."""
    if pretrained_model not in PRETRAINED_CONFIGS:
        raise ValueError("Invalid pretrained model {}, must be one of {}".format(
            pretrained_model, list(PRETRAINED_CONFIGS)))
    w = np.fromfile(asset_path(PRETRAINED_CONFIGS[pretrained_model][0]), np.float32)
    assert w.size == N_WEIGHT_FLOATS
    return w


def weights_from_state_dict(sd):
    """Flat blob from a reference checkpoint's state_dict (m6anet/model/model_states/*.pt,
    keys as built by m6anet/model/model.py:40-69)."""
    order = ["read_level_encoder.1.embedding_layer.weight", "read_level_encoder.3.layers.0.weight",
             "read_level_encoder.3.layers.0.bias", "read_level_encoder.3.layers.1.weight",
             "read_level_encoder.3.layers.1.bias", "read_level_encoder.3.layers.1.running_mean",
             "read_level_encoder.3.layers.1.running_var", "read_level_encoder.4.layers.0.weight",
             "read_level_encoder.4.layers.0.bias", "pooling_filter.probability_layer.0.weight",
             "pooling_filter.probability_layer.0.bias"]
    w = np.concatenate([np.asarray(sd[k].detach().cpu().numpy() if hasattr(sd[k], "detach") else sd[k],
                                   np.float32).ravel() for k in order])
    if w.size != N_WEIGHT_FLOATS:
        raise ValueError("state_dict does not have the m6anet.toml topology (%d floats)" % w.size)
    return w


_libc = None


def _hint_huge_pages(a):
    """A fresh output array is first touched by the library's copy threads; with 4 KB pages that is tens of
    thousands of page faults (80 MB of read probabilities per million sites).  Ask for transparent huge pages on
    its 2 MB-aligned interior -- best effort, any failure is ignored."""
    global _libc
    if a.nbytes < (8 << 20):
        return
    try:
        if _libc is None:
            _libc = C.CDLL(None, use_errno=True)
            _libc.madvise.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
        lo = (a.ctypes.data + (1 << 21) - 1) & ~((1 << 21) - 1)
        hi = (a.ctypes.data + a.nbytes) & ~((1 << 21) - 1)
        if hi > lo:
            _libc.madvise(lo, hi - lo, 14)          # MADV_HUGEPAGE
    except Exception:
        pass


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def pinned_empty(shape, dtype=np.float32):
    """A NumPy array in page-locked host memory (m6a_host_alloc): what the host-pointer calls DMA in place instead of copying
    through the staging ring -- for callers without torch.  Freed when the array (and every view of it) is gone."""
    import weakref
    L = _lib.load()
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    p = C.c_void_p()
    rc = L.m6a_host_alloc(max(n, 1), C.byref(p))
    if rc != 0:
        raise _lib.M6AError(rc, "m6a_host_alloc(%d bytes)" % n)
    buf = (C.c_char * max(n, 1)).from_address(p.value)
    weakref.finalize(buf, L.m6a_host_free, C.c_void_p(p.value))
    return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)


class _Arg:
    """Pointer + keep-alive for one array argument."""

    def __init__(self, x, dtype, torch_dtype_name):
        self.is_dev = False
        if _is_torch(x):
            import torch
            want = getattr(torch, torch_dtype_name)
            if x.dtype != want or not x.is_contiguous():
                raise TypeError("tensor must be contiguous %s" % torch_dtype_name)
            self.keep = x
            self.ptr = x.data_ptr()
            self.is_dev = x.is_cuda
            self.size = x.numel()
        else:
            a = np.ascontiguousarray(x, dtype)
            self.keep = a
            self.ptr = a.ctypes.data
            self.size = a.size


class M6ANetEngine:
    def __init__(self, weights=None, pretrained_model=DEFAULT_PRETRAINED_MODEL, device=0):
        self._L = _lib.load()
        if weights is None:
            weights = load_weights(pretrained_model)
        w = np.ascontiguousarray(weights, np.float32)
        if w.size != N_WEIGHT_FLOATS:
            raise ValueError("weights must have %d floats" % N_WEIGHT_FLOATS)
        if isinstance(device, str):
            device = int(device.split(":")[1]) if ":" in device else 0
        self.device = int(device)
        h = C.c_void_p()
        rc = self._L.m6a_create(C.byref(h), w.ctypes.data, w.size, self.device)
        if rc != 0:
            raise _lib.M6AError(rc, self._L.m6a_last_error(None).decode())
        self._h = h

    # -- plumbing ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.m6a_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise _lib.M6AError(rc, self._L.m6a_last_error(self._h).decode())

    def set_stream(self, stream_handle):
        self._chk(self._L.m6a_set_stream(self._h, C.c_void_p(stream_handle or 0)))
        self._torch_stream = None

    def use_torch_stream(self):
        import torch
        st = torch.cuda.current_stream(self.device)
        self.set_stream(st.cuda_stream)
        self._torch_stream = st.cuda_stream

    def set_job_offset(self, first_site):
        """The sites given to calculate_site_proba/infer are sites [first_site, ...) of a larger
        job (multi-GPU shards): flush groups and RNG restarts follow the job's batch indices."""
        self._chk(self._L.m6a_set_job_offset(self._h, int(first_site)))

    def set_encoder_variant(self, mode):
        """0 auto = the 16-slot kernels (the reference's operation order, its bits), 1 the same said explicitly, 2 the 12-slot
        kernel (opt-in; needs every bag >= 16 reads), 3 the 16-slot arithmetic behind the per-lane walk of off[] even when the
        scalar site chain would do (A/B, tests), 4 fast: the 12-slot kernel where every bag has >= 16 reads, 16-slot elsewhere."""
        self._chk(self._L.m6a_set_encoder_variant(self._h, int(mode)))

    @property
    def last_encoder_variant(self):
        return self._L.m6a_last_encoder_variant(self._h).decode()

    @property
    def last_encoder_kernel(self):
        """Name of the __global__ function the last encode launched (enc_kernel | enc_site16_kernel | enc_csite_kernel)."""
        return self._L.m6a_last_encoder_kernel(self._h).decode()

    def set_scan_driver(self, mode):
        """Ragged bags: 0 auto, 1 one wavefront per flush group, 2 counting pass + one wavefront per site, 3 index tables."""
        self._chk(self._L.m6a_set_scan_driver(self._h, int(mode)))

    def set_table_variant(self, mode):
        """Uniform bags: 0 auto, 1 LDS-gather kernel, 2 register kernel."""
        self._chk(self._L.m6a_set_table_variant(self._h, int(mode)))

    def sync(self):
        self._chk(self._L.m6a_sync(self._h))

    def set_host_offsets(self, off_host):
        """For the NEXT call with device tensors: the host copy (numpy int64 [S+1]) of its CSR offsets.  The call then
        takes the bag statistics from it, does not block on the stream, and only checks the device array against it
        (a mismatch is reported by the next `sync()`).  include/m6a.h: m6a_set_host_offsets."""
        if off_host is None:
            self._chk(self._L.m6a_set_host_offsets(self._h, None))
            return
        a = np.ascontiguousarray(off_host, dtype=np.int64)
        self._host_off_keepalive = a                      # read inside the next call
        self._chk(self._L.m6a_set_host_offsets(self._h, a.ctypes.data))

    # -- the multi-GPU exchange on the library's own RCCL communicator (include/m6a.h: m6a_gather) --------
    def comm_init(self, unique_id, rank, world_size):
        """unique_id: the 128 bytes rank 0 got from `comm_unique_id()`, handed to every rank by the launcher."""
        buf = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
        self._chk(self._L.m6a_comm_init(self._h, buf, int(rank), int(world_size)))
        self._comm = (int(rank), int(world_size))

    def comm_info(self):
        """What the communicator itself reports: {'ranks_seen': ncclCommCount, 'rank', 'device', 'rccl_version'}."""
        n, r, d, v = C.c_int(0), C.c_int(-1), C.c_int(-1), C.c_int(0)
        self._chk(self._L.m6a_comm_count(self._h, C.byref(n)))
        self._chk(self._L.m6a_comm_info(self._h, C.byref(r), C.byref(d), C.byref(v)))
        return {"ranks_seen": n.value, "rank": r.value, "device": d.value, "rccl_version": v.value}

    def comm_destroy(self):
        self._chk(self._L.m6a_comm_destroy(self._h))

    def gather(self, site, mod, cuts, dst=0, out=None):
        """site_prob / mod_ratio of this rank's shard to rank `dst` in ONE grouped RCCL exchange on the context's
        stream; returns (site_all, mod_all) on dst, (None, None) elsewhere.  torch tensors on the context's GPU
        (stream-ordered: `sync()` before reading) or numpy arrays (staged by the library, synchronous)."""
        rank, world = self._comm
        cuts = np.ascontiguousarray(cuts, np.int64)
        assert cuts.size == world + 1
        aS, aM = _Arg(site, np.float32, "float32"), _Arg(mod, np.float64, "float64")
        total = int(cuts[-1] - cuts[0])
        sa = ma = None
        if rank == dst:
            sa, ma = out if out is not None else (self._out(aS.is_dev, total, np.float32, "float32"),
                                                  self._out(aS.is_dev, total, np.float64, "float64"))
        oS = _Arg(sa, np.float32, "float32") if sa is not None else None
        oM = _Arg(ma, np.float64, "float64") if ma is not None else None
        self._chk(self._L.m6a_gather(self._h, aS.ptr, aM.ptr, cuts.ctypes.data, int(dst),
                                     oS.ptr if oS else None, oM.ptr if oM else None))
        return sa, ma

    def gather_reads(self, read_prob, read_cuts, dst=0, out=None):
        """The per-read output of this rank's shard to rank `dst` (include/m6a.h: m6a_gather_reads); read_cuts [world+1]
        = off[site cuts] of the job's CSR offsets.  Returns read_all on dst, None elsewhere."""
        rank, world = self._comm
        cuts = np.ascontiguousarray(read_cuts, np.int64)
        assert cuts.size == world + 1
        aP = _Arg(read_prob, np.float32, "float32")
        ra = None
        if rank == dst:
            ra = out if out is not None else self._out(aP.is_dev, int(cuts[-1] - cuts[0]), np.float32, "float32")
        oP = _Arg(ra, np.float32, "float32") if ra is not None else None
        self._chk(self._L.m6a_gather_reads(self._h, aP.ptr, cuts.ctypes.data, int(dst), oP.ptr if oP else None))
        return ra

    def random_stream(self, seed, n_words):
        """First n_words uint32 outputs of NumPy's legacy generator after np.random.seed(seed), generated on the GPU
        (include/m6a.h: m6a_random_stream): np.frombuffer(np.random.RandomState(seed).bytes(4 * n), np.uint32)."""
        out = np.empty(int(n_words), np.uint32)
        self._chk(self._L.m6a_random_stream(self._h, int(seed) & 0xffffffff, int(n_words), out.ctypes.data))
        return out

    def prepare_host_io(self):
        """Pin the staging ring of the host-pointer path now (otherwise the first numpy-array call does it)."""
        self._chk(self._L.m6a_prepare_host_io(self._h))

    def profile(self, on=True):
        """HIP-event timing of the launches: False off, True both kernels, 'encoder' / 'pooling' one kind."""
        self._chk(self._L.m6a_profile_enable(self._h, {False: 0, True: 1, 'encoder': 2, 'pooling': 3}[on]))

    def profile_read(self, kind):
        ms, n = C.c_double(), C.c_int64()
        self._chk(self._L.m6a_profile_read(self._h, kind, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_clock(self, kind):
        """The shader clock the last PROFILED launch of `kind` (0 encoder, 1 pooling) ran at, from in-kernel stamps of up to 64
        waves (m6a_profile_clock): {'ghz': median, 'ghz_min', 'ghz_max', 'span_ms', 'waves'}; None if nothing was stamped."""
        g, lo, hi, span, n = C.c_double(), C.c_double(), C.c_double(), C.c_double(), C.c_int32()
        self._chk(self._L.m6a_profile_clock(self._h, kind, C.byref(g), C.byref(lo), C.byref(hi), C.byref(span), C.byref(n)))
        if n.value == 0:
            return None
        return {"ghz": g.value, "ghz_min": lo.value, "ghz_max": hi.value, "span_ms": span.value, "waves": n.value}

    @property
    def last_pool_variant(self):
        return self._L.m6a_last_pool_variant(self._h).decode()

    def _out(self, like_dev, n, np_dtype, torch_name):
        if like_dev:
            import torch
            return torch.empty(n, dtype=getattr(torch, torch_name), device="cuda:%d" % self.device)
        a = np.empty(n, np_dtype)
        _hint_huge_pages(a)
        return a

    # -- the path ---------------------------------------------------------------------------
    def get_read_probability(self, X, site_kmers, off, out=None):
        aX, aK, aO = _Arg(X, np.float32, "float32"), _Arg(site_kmers, np.uint8, "uint8"), _Arg(off, np.int64, "int64")
        S = aO.size - 1
        R = aX.size // 9
        if aK.size != 3 * S:
            raise ValueError("site_kmers must be [n_sites, 3]")
        rp = out if out is not None else self._out(aX.is_dev, R, np.float32, "float32")
        aP = _Arg(rp, np.float32, "float32")
        if aP.size != R:
            raise ValueError("read_prob must have one float per read")
        self._chk(self._L.m6a_encode_reads(self._h, aX.ptr, aK.ptr, aO.ptr, S, aP.ptr))
        return rp

    def calculate_site_proba(self, read_probs, off, n_iters, n_samples=N_SAMPLES,
                             read_proba_threshold=DEFAULT_READ_THRESHOLD, seed=0, batch_size=16,
                             save_per_batch=2):
        """Returns (site_probs float32 [S], mod_ratios float64 [S])."""
        aP, aO = _Arg(read_probs, np.float32, "float32"), _Arg(off, np.int64, "int64")
        S = aO.size - 1
        site = self._out(aP.is_dev, S, np.float32, "float32")
        mod = self._out(aP.is_dev, S, np.float64, "float64")
        aS, aM = _Arg(site, np.float32, "float32"), _Arg(mod, np.float64, "float64")
        self._chk(self._L.m6a_site_pool(self._h, aP.ptr, aO.ptr, S, int(n_iters), int(n_samples),
                                        float(np.float32(read_proba_threshold)), int(seed) & 0xffffffff,
                                        _lib.RNG_NUMPY, int(batch_size), int(save_per_batch), aS.ptr, aM.ptr))
        return site, mod

    def infer(self, X, site_kmers, off, n_iters, n_samples=N_SAMPLES,
              read_proba_threshold=DEFAULT_READ_THRESHOLD, seed=0, batch_size=16, save_per_batch=2,
              want_read_probs=True, out=None):
        """Returns (read_probs or None, site_probs, mod_ratios).  `out` = preallocated
        (read_probs|None, site_probs, mod_ratios) to reuse buffers across calls."""
        aX, aK, aO = _Arg(X, np.float32, "float32"), _Arg(site_kmers, np.uint8, "uint8"), _Arg(off, np.int64, "int64")
        S = aO.size - 1
        R = aX.size // 9
        if out is not None:
            rp, site, mod = out
        else:
            rp = self._out(aX.is_dev, R, np.float32, "float32") if want_read_probs else None
            site = self._out(aX.is_dev, S, np.float32, "float32")
            mod = self._out(aX.is_dev, S, np.float64, "float64")
        aP = _Arg(rp, np.float32, "float32") if rp is not None else None
        aS, aM = _Arg(site, np.float32, "float32"), _Arg(mod, np.float64, "float64")
        self._chk(self._L.m6a_infer(self._h, aX.ptr, aK.ptr, aO.ptr, S, int(n_iters), int(n_samples),
                                    float(np.float32(read_proba_threshold)), int(seed) & 0xffffffff,
                                    _lib.RNG_NUMPY, int(batch_size), int(save_per_batch),
                                    aP.ptr if aP else None, aS.ptr, aM.ptr))
        return rp, site, mod

    # -- the same job, streamed: run_inference's batch loop (inference_utils.py:33-54) feeds as the loader produces --------
    def job_begin(self, n_iters, n_samples=N_SAMPLES, read_proba_threshold=DEFAULT_READ_THRESHOLD, seed=0, batch_size=16,
                  save_per_batch=2, expect_sites=0, expect_reads=0):
        """Opens a streaming job (include/m6a.h: m6a_job_begin); feed batches with `job_feed`, finish with `job_end`."""
        self._chk(self._L.m6a_job_begin(self._h, int(n_iters), int(n_samples), float(np.float32(read_proba_threshold)),
                                        int(seed) & 0xffffffff, _lib.RNG_NUMPY, int(batch_size), int(save_per_batch),
                                        int(expect_sites), int(expect_reads)))
        self._job_dev = False
        self._job_keep = []                                  # device tensors fed to the open job (released by job_end / job_abort)

    def job_feed(self, X, site_kmers, off):
        """One batch: X [r,9], site_kmers [n,3] (numpy or torch-on-GPU), off [n+1] batch-local CSR offsets (host).
        Asynchronous.  HOST arrays are copied into the pinned ring: they are the caller's again on return.  DEVICE tensors
        are read in place by the encoder queued on the context's stream, possibly after this call has returned
        (include/m6a.h): the engine keeps them referenced until job_end / job_abort, so the caching allocator cannot hand
        their memory to the next batch, and -- unless the context runs on the very stream torch is producing on
        (use_torch_stream) -- waits for torch's current stream first, so whatever wrote them is done before the encoder
        reads."""
        aX, aK = _Arg(X, np.float32, "float32"), _Arg(site_kmers, np.uint8, "uint8")
        if aX.is_dev or aK.is_dev:
            import torch
            cur = torch.cuda.current_stream(self.device)
            if getattr(self, "_torch_stream", None) != cur.cuda_stream:
                cur.synchronize()
            self._job_keep.append((aX.keep, aK.keep))
        o = off if (isinstance(off, np.ndarray) and off.dtype == np.int64 and off.flags.c_contiguous) else \
            np.ascontiguousarray(off.cpu().numpy() if _is_torch(off) else off, dtype=np.int64)
        n = o.size - 1
        if aK.size != 3 * n or aX.size != 9 * int(o[-1]):
            raise ValueError("batch shapes disagree: X [off[-1], 9], site_kmers [n_sites, 3], off [n_sites + 1]")
        self._job_dev = self._job_dev or aX.is_dev
        self._chk(self._L.m6a_job_feed(self._h, aX.ptr, aK.ptr, o.ctypes.data, n))

    def job_feed_collated(self, features, kmers, n_reads):
        """One batch exactly as the reference's inference_collate builds it (m6anet/utils/data_utils.py:498-506): features
        [r,9] float32, kmers [r,3] int64 (per READ), n_reads [n] int64 -- host tensors or arrays; nothing is converted here."""
        aX, aK, aN = _Arg(features, np.float32, "float32"), _Arg(kmers, np.int64, "int64"), _Arg(n_reads, np.int64, "int64")
        if aX.is_dev or aK.is_dev or aN.is_dev:
            raise TypeError("job_feed_collated takes the collate's host tensors")
        if aK.size * 3 != aX.size:
            raise ValueError("batch shapes disagree: features [r, 9], kmers [r, 3]")
        self._chk(self._L.m6a_job_feed_collated(self._h, aX.ptr, aK.ptr, aN.ptr, aN.size))

    def job_size(self):
        S, R = C.c_int64(), C.c_int64()
        self._chk(self._L.m6a_job_size(self._h, C.byref(S), C.byref(R)))
        return S.value, R.value

    def job_end(self, want_read_probs=True, device_outputs=None):
        """Pools everything fed and returns (read_probs or None, site_probs, mod_ratios) -- numpy arrays, or torch
        tensors on the GPU when the job was fed device tensors (or device_outputs=True)."""
        S, R = self.job_size()
        dev = self._job_dev if device_outputs is None else bool(device_outputs)
        rp = self._out(dev, R, np.float32, "float32") if want_read_probs else None
        site = self._out(dev, S, np.float32, "float32")
        mod = self._out(dev, S, np.float64, "float64")
        aP = _Arg(rp, np.float32, "float32") if rp is not None else None
        aS, aM = _Arg(site, np.float32, "float32"), _Arg(mod, np.float64, "float64")
        try:
            self._chk(self._L.m6a_job_end(self._h, aP.ptr if aP else None, aS.ptr, aM.ptr))     # synchronises: every encoder has read its rows
        finally:
            self._job_keep = []
        return rp, site, mod

    def job_abort(self):
        try:
            self._chk(self._L.m6a_job_abort(self._h))
            self.sync()                                      # queued encoders may still be reading device batches
        finally:
            self._job_keep = []

    def forward(self, X, kmer, bag=N_SAMPLES):
        """Site probability of fixed-size bags: X [B, bag, 9], kmer [B, 3] -> [B]."""
        aX, aK = _Arg(X, np.float32, "float32"), _Arg(kmer, np.uint8, "uint8")
        B = aK.size // 3
        if aX.size != B * bag * 9:
            raise ValueError("X must be [B, bag, 9]")
        site = self._out(aX.is_dev, B, np.float32, "float32")
        aS = _Arg(site, np.float32, "float32")
        self._chk(self._L.m6a_bag_forward(self._h, aX.ptr, aK.ptr, B, int(bag), aS.ptr))
        return site


    def validate_pool(self, read_probs, off, n_iterations=1, n_samples=N_SAMPLES, seed=0):
        """The prediction part of the reference's `validate` (training_utils.py:233-253) from read
        probabilities: each pass samples n_samples reads per site without replacement
        (data_utils.py:213-214).  Returns (y_pred float32 [n_iterations, S], y_pred_avg float32 [S])."""
        aP, aO = _Arg(read_probs, np.float32, "float32"), _Arg(off, np.int64, "int64")
        S = aO.size - 1
        y = self._out(aP.is_dev, int(n_iterations) * S, np.float32, "float32")
        avg = self._out(aP.is_dev, S, np.float32, "float32")
        aY, aA = _Arg(y, np.float32, "float32"), _Arg(avg, np.float32, "float32")
        self._chk(self._L.m6a_validate_pool(self._h, aP.ptr, aO.ptr, S, int(n_iterations), int(n_samples),
                                            int(seed) & 0xffffffff, aY.ptr, aA.ptr))
        return y.reshape(int(n_iterations), S), avg

    def validate_forward(self, X, site_kmers, off, n_iterations=1, n_samples=N_SAMPLES, seed=0, want_read_probs=False):
        """Encoder + validate_pool in one call.  Returns (y_pred, y_pred_avg[, read_probs])."""
        aX, aK, aO = _Arg(X, np.float32, "float32"), _Arg(site_kmers, np.uint8, "uint8"), _Arg(off, np.int64, "int64")
        S = aO.size - 1
        R = aX.size // 9
        y = self._out(aX.is_dev, int(n_iterations) * S, np.float32, "float32")
        avg = self._out(aX.is_dev, S, np.float32, "float32")
        rp = self._out(aX.is_dev, R, np.float32, "float32") if want_read_probs else None
        aY, aA = _Arg(y, np.float32, "float32"), _Arg(avg, np.float32, "float32")
        aP = _Arg(rp, np.float32, "float32") if rp is not None else None
        self._chk(self._L.m6a_validate(self._h, aX.ptr, aK.ptr, aO.ptr, S, int(n_iterations), int(n_samples),
                                       int(seed) & 0xffffffff, aP.ptr if aP else None, aY.ptr, aA.ptr))
        y = y.reshape(int(n_iterations), S)
        return (y, avg, rp) if want_read_probs else (y, avg)


def comm_unique_id():
    """128-byte RCCL id for `M6ANetEngine.comm_init` (call on ONE rank, broadcast to the others)."""
    buf = (C.c_char * 128)()
    L = _lib.load()
    rc = L.m6a_comm_unique_id(buf)
    if rc != 0:
        raise _lib.M6AError(rc, L.m6a_last_error(None).decode())
    return bytes(buf.raw)


LINK_TYPES = {0: "hypertransport", 1: "qpi", 2: "pcie", 3: "infiniband", 4: "xgmi"}


def device_link(dev_a, dev_b):
    """{'link': 'xgmi' | 'pcie' | ..., 'hops', 'peer_access'} between two visible HIP devices (m6a_device_link)."""
    L = _lib.load()
    lt, hc, pa = C.c_int(-1), C.c_int(-1), C.c_int(0)
    rc = L.m6a_device_link(int(dev_a), int(dev_b), C.byref(lt), C.byref(hc), C.byref(pa))
    if rc != 0:
        raise _lib.M6AError(rc, "m6a_device_link(%d, %d)" % (dev_a, dev_b))
    return {"link": "self" if dev_a == dev_b else LINK_TYPES.get(lt.value, "type %d" % lt.value), "hops": hc.value, "peer_access": bool(pa.value)}


def device_count():
    """HIP devices visible to this process (include/m6a.h: m6a_device_count)."""
    return int(_lib.load().m6a_device_count())


def flush_groups(n_sites, batch_size=16, save_per_batch=2):
    """Site offsets of the reference's flush groups (inference_utils.py:33,47)."""
    L = _lib.load()
    nb = (n_sites + batch_size - 1) // batch_size
    g = np.zeros(nb + 2, np.int64)
    G = L.m6a_flush_groups(n_sites, batch_size, save_per_batch, g.ctypes.data, g.size)
    if G < 0:
        raise _lib.M6AError(int(G), "m6a_flush_groups")
    return g[:G + 1].copy()


def reference_written_sites(n_sites, batch_size=16, save_per_batch=2):
    """How many (leading) sites the reference writes for this geometry: its inverted flush test
    (inference_utils.py:47) never writes the batches after the last flush."""
    n = _lib.load().m6a_reference_written_sites(int(n_sites), int(batch_size), int(save_per_batch))
    if n < 0:
        raise _lib.M6AError(int(n), "m6a_reference_written_sites")
    return int(n)


def shard_plan(off, n_shards, batch_size=16, save_per_batch=2):
    """Flush-group-aligned contiguous site shards balanced by read count -> int64 [n_shards+1]."""
    L = _lib.load()
    off = np.ascontiguousarray(off, np.int64)
    out = np.zeros(n_shards + 1, np.int64)
    rc = L.m6a_shard_plan(off.ctypes.data, off.size - 1, batch_size, save_per_batch, n_shards, out.ctypes.data)
    if rc != 0:
        raise _lib.M6AError(rc, "m6a_shard_plan")
    return out
