"""ctypes binding of libm6a_hip.so (include/m6a.h).  No CPU fallback: if the HIP library is
missing or a call fails, this raises."""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("M6A_HIP_LIB") or os.path.join(_PKG, "libm6a_hip.so")   # M6A_HIP_LIB: an experimental build

M6A_OK = 0
ERRORS = {-1: "M6A_EINVAL", -2: "M6A_ENOMEM", -3: "M6A_EHIP", -4: "M6A_ESTREAM", -5: "M6A_ENODEV",
          -6: "M6A_EUNSUPPORTED"}
RNG_NUMPY = 0

# every symbol include/m6a.h declares (tests check the .so exports exactly these)
SYMBOLS = ["m6a_create", "m6a_destroy", "m6a_last_error", "m6a_set_stream", "m6a_set_job_offset", "m6a_set_scan_driver", "m6a_set_table_variant", "m6a_set_encoder_variant", "m6a_last_encoder_variant", "m6a_last_encoder_kernel", "m6a_sync", "m6a_set_host_offsets", "m6a_prepare_host_io", "m6a_host_alloc", "m6a_host_free", "m6a_host_is_pinned",
           "m6a_encode_reads", "m6a_site_pool", "m6a_infer", "m6a_job_begin", "m6a_job_feed", "m6a_job_feed_collated", "m6a_job_size", "m6a_job_end", "m6a_job_abort", "m6a_bag_forward", "m6a_validate_pool", "m6a_validate", "m6a_flush_groups",
           "m6a_reference_written_sites",
           "m6a_shard_plan", "m6a_comm_unique_id", "m6a_comm_init", "m6a_gather", "m6a_gather_reads", "m6a_device_count", "m6a_random_stream", "m6a_comm_destroy", "m6a_comm_count", "m6a_comm_info", "m6a_device_link", "m6a_profile_enable", "m6a_profile_read", "m6a_profile_clock", "m6a_last_pool_variant",
           "m6a_version"]

_lib = None


class M6AError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERRORS.get(code, "M6A_E?"), code, msg))
        self.code = code


def _preload_hip_runtime():
    """One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so.7 and refuses to
    see the GPU if another copy (e.g. /opt/rocm's) was mapped first.  When torch is installed,
    map ITS runtime before libm6a_hip.so so both bind to the same one, whichever is imported
    first.  (torch itself is not imported here.)"""
    import importlib.util
    # (HSA_ENABLE_IPC_MODE_LEGACY=0, which multi-process RCCL work needs on these boxes, is NOT set here any more: the library
    # binding must not edit the environment of whatever application imports it.  The package's own multi-process
    # entry points set it for their processes -- `python -m m6anet_amd` (m6anet_amd/_early.py) and bench.py; a host application
    # that drives m6a_comm_* itself exports it, as INTEGRATION.md says.)
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.submodule_search_locations:
        libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
        rt = os.path.join(libdir, "libamdhip64.so")
        if os.path.exists(rt):
            C.CDLL(rt, mode=C.RTLD_GLOBAL)
        # the same goes for RCCL (m6a_gather binds it at run time): PyTorch's copy is linked to PyTorch's HIP runtime
        rccl = os.path.join(libdir, "librccl.so")
        if os.path.exists(rccl):
            os.environ.setdefault("M6A_RCCL_LIB", rccl)


def load():
    """Loads the in-tree HIP library.  Raises if it has not been built (python -m m6anet_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libm6a_hip.so is not built: run `python -m m6anet_amd.build` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
    _preload_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, i64, i32, u32, f32, sz = C.c_void_p, C.c_int64, C.c_int, C.c_uint32, C.c_float, C.c_size_t
    L.m6a_create.argtypes = [C.POINTER(vp), vp, sz, i32]
    L.m6a_destroy.argtypes = [vp]
    L.m6a_destroy.restype = None
    L.m6a_last_error.argtypes = [vp]
    L.m6a_last_error.restype = C.c_char_p
    L.m6a_set_stream.argtypes = [vp, vp]
    L.m6a_sync.argtypes = [vp]
    L.m6a_set_host_offsets.argtypes = [vp, vp]
    L.m6a_prepare_host_io.argtypes = [vp]
    L.m6a_set_job_offset.argtypes = [vp, i64]
    L.m6a_set_scan_driver.argtypes = [vp, i32]
    L.m6a_set_table_variant.argtypes = [vp, i32]
    L.m6a_set_encoder_variant.argtypes = [vp, i32]
    L.m6a_last_encoder_variant.argtypes = [vp]
    L.m6a_last_encoder_variant.restype = C.c_char_p
    L.m6a_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.m6a_host_free.argtypes = [vp]
    L.m6a_host_is_pinned.argtypes = [vp, C.c_size_t]
    L.m6a_last_encoder_kernel.argtypes = [vp]
    L.m6a_last_encoder_kernel.restype = C.c_char_p
    L.m6a_encode_reads.argtypes = [vp, vp, vp, vp, i64, vp]
    L.m6a_site_pool.argtypes = [vp, vp, vp, i64, i32, i32, f32, u32, i32, i64, i64, vp, vp]
    L.m6a_infer.argtypes = [vp, vp, vp, vp, i64, i32, i32, f32, u32, i32, i64, i64, vp, vp, vp]
    L.m6a_job_begin.argtypes = [vp, i32, i32, f32, u32, i32, i64, i64, i64, i64]
    L.m6a_job_feed.argtypes = [vp, vp, vp, vp, i64]
    L.m6a_job_feed_collated.argtypes = [vp, vp, vp, vp, i64]
    L.m6a_job_size.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    L.m6a_job_end.argtypes = [vp, vp, vp, vp]
    L.m6a_job_abort.argtypes = [vp]
    L.m6a_bag_forward.argtypes = [vp, vp, vp, i64, i32, vp]
    L.m6a_validate_pool.argtypes = [vp, vp, vp, i64, i32, i32, C.c_uint32, vp, vp]
    L.m6a_validate.argtypes = [vp, vp, vp, vp, i64, i32, i32, C.c_uint32, vp, vp, vp]
    L.m6a_flush_groups.argtypes = [i64, i64, i64, vp, i64]
    L.m6a_flush_groups.restype = i64
    L.m6a_reference_written_sites.argtypes = [i64, i64, i64]
    L.m6a_reference_written_sites.restype = i64
    L.m6a_shard_plan.argtypes = [vp, i64, i64, i64, i32, vp]
    L.m6a_comm_unique_id.argtypes = [vp]
    L.m6a_comm_init.argtypes = [vp, vp, i32, i32]
    L.m6a_gather.argtypes = [vp, vp, vp, vp, i32, vp, vp]
    L.m6a_gather_reads.argtypes = [vp, vp, vp, i32, vp]
    L.m6a_device_count.argtypes = []
    L.m6a_random_stream.argtypes = [vp, u32, i64, vp]
    L.m6a_comm_destroy.argtypes = [vp]
    pi = C.POINTER(i32)
    L.m6a_comm_count.argtypes = [vp, pi]
    L.m6a_comm_info.argtypes = [vp, pi, pi, pi]
    L.m6a_device_link.argtypes = [i32, i32, pi, pi, pi]
    L.m6a_profile_enable.argtypes = [vp, i32]
    L.m6a_profile_read.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(i64)]
    L.m6a_profile_clock.argtypes = [vp, i32] + [C.POINTER(C.c_double)] * 4 + [C.POINTER(i32)]
    L.m6a_last_pool_variant.argtypes = [vp]
    L.m6a_last_pool_variant.restype = C.c_char_p
    L.m6a_version.restype = C.c_char_p
    for name in SYMBOLS:
        fn = getattr(L, name)
        if fn.restype is C.c_int and name not in ("m6a_flush_groups", "m6a_reference_written_sites"):
            fn.restype = C.c_int
    _lib = L
    return L
