"""`inference --gpus N`: ranks 1..N-1 are started BEFORE this process imports anything heavy.

A rank's critical path is its own start-up (the interpreter, NumPy, the HIP runtime: ~0.5 s).  Spawned from
multi_gpu.launch() the other ranks began ~0.1 s into rank 0's life, after its imports and argument handling; spawned from
here -- the first thing `python -m m6anet_amd` runs, standard library only -- they begin with it.  multi_gpu.launch() adopts
what was started here (or starts the ranks itself when this module did not run: the CLI called as a function, odd argv).
Whatever happens afterwards -- an argument error, an exception before launch() -- the ranks are killed and the exchange
directory removed at exit.
"""
import atexit
import os
import shutil
import subprocess
import sys
import tempfile

STORE_SUFFIX = ".m6astore"          # = data_utils.STORE_SUFFIX (not imported: that module pulls NumPy in)
state = None                        # {"world", "xdir", "store", "given_store", "procs", "argv"} once ranks were started here


def store_size_estimate(input_dirs):
    """Upper estimate of the packed store: the normalised features (36 B per read) and the ids are about 0.45 of the JSON
    text they were parsed from (913 MB of data.json -> 384 MB); 0.6 + 64 MB leaves room."""
    total = 0
    for d in input_dirs:
        for fn in ("data.json", "data.info"):
            try:
                total += os.path.getsize(os.path.join(str(d), fn))
            except OSError:
                pass
    return int(0.6 * total) + (64 << 20)


def exchange_base(need_bytes, out_dir):
    """Where the exchange directory (RCCL id, byte counts, and the packed store unless the input already is one) goes:
    M6A_XDIR_BASE if set; /dev/shm when it is writable AND has room for the store with a margin (Docker's default /dev/shm
    is 64 MB, and a tmpfs store is RAM next to every rank's page-cache mapping of it); else the system's temporary
    directory if IT has room; else the output directory."""
    forced = os.environ.get("M6A_XDIR_BASE")
    if forced:
        return forced
    for c in ("/dev/shm", tempfile.gettempdir(), os.path.abspath(str(out_dir))):
        try:
            if os.path.isdir(c) and os.access(c, os.W_OK) and shutil.disk_usage(c).free >= 1.25 * need_bytes + (16 << 20):
                return c
        except OSError:
            continue
    return None                                              # tempfile's default; pack_sites will say what went wrong


def _values(argv, flag):
    """Values of an argparse-style option: `--flag a b`, `--flag=a`; None if absent."""
    for i, a in enumerate(argv):
        if a == flag:
            out = []
            for b in argv[i + 1:]:
                if b.startswith("--"):
                    break
                out.append(b)
            return out
        if a.startswith(flag + "="):
            return [a[len(flag) + 1:]]
    return None


def start_ranks(world, input_dirs, out_dir, argv):
    """Exchange directory + ranks 1..world-1 (the same command line, M6A_RANK / M6A_WORLD / M6A_XDIR / M6A_STORE in the
    environment).  Returns the state dict."""
    global state
    given_store = len(input_dirs) == 1 and str(input_dirs[0]).endswith(STORE_SUFFIX)
    os.makedirs(str(out_dir), exist_ok=True)
    xdir = tempfile.mkdtemp(prefix="m6a_gpus_", dir=exchange_base(0 if given_store else store_size_estimate(input_dirs), out_dir))
    store = os.path.abspath(str(input_dirs[0])) if given_store else os.path.join(xdir, "job" + STORE_SUFFIX)
    st = {"world": world, "xdir": xdir, "store": store, "given_store": given_store, "procs": [], "argv": list(argv),
          "input_dirs": [str(d) for d in input_dirs], "out_dir": str(out_dir)}
    state = st
    atexit.register(cleanup)
    env = dict(os.environ, M6A_WORLD=str(world), M6A_XDIR=xdir, M6A_STORE=store)
    for r in range(1, world):
        st["procs"].append(subprocess.Popen([sys.executable, "-m", "m6anet_amd", "inference"] + list(argv), env=dict(env, M6A_RANK=str(r))))
    return st


def ranks_may_share_a_gpu():
    return os.environ.get("M6A_SHARE_GPU") == "1" or os.environ.get("M6A_EXCHANGE") == "host"


def rccl_is_a_standin():
    """M6A_RCCL_STANDIN=1: the library named by M6A_RCCL_LIB is not RCCL but a stand-in transport that accepts several ranks on
    one device (tests/stub_rccl: how a one-GPU box executes the M6A_EXCHANGE=rccl leg).  A real RCCL refuses two ranks on a device."""
    return os.environ.get("M6A_RCCL_STANDIN") == "1" and bool(os.environ.get("M6A_RCCL_LIB"))


def strip_command(argv):
    """argv without the sub-command word: what start_ranks was given by maybe_start."""
    return list(argv[1:]) if argv and argv[0] == "inference" else list(argv)


def visible_devices_estimate():
    """Devices this process will see, without loading the HIP runtime: the render nodes of the container, cut down by
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when they are set (a list of indices or UUIDs: its
    length; empty = none).  An estimate that errs low only delays the ranks' start (launch() starts them), one that errs
    high starts ranks that exchange_mode() then ends with its message."""
    try:
        n = sum(1 for f in os.listdir("/dev/dri") if f.startswith("renderD"))
    except OSError:
        n = 0
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def cleanup():
    global state
    st, state = state, None
    if not st:
        return
    for p in st["procs"]:
        if p.poll() is None:
            p.kill()
    shutil.rmtree(st["xdir"], ignore_errors=True)


def maybe_start(argv):
    """Called by `python -m m6anet_amd` before any other import: `inference ... --gpus N` with N > 1, not itself a rank."""
    # The CLI's own process (launcher, and the ranks it starts: they inherit it): multi-process GPU work -- RCCL's P2P set-up --
    # exchanges memory handles, and the host driver of these boxes supports dmabuf IPC only; without this hipIpcGetMemHandle fails
    # with "invalid argument".  Set HERE, for this command's processes, before any HIP runtime is mapped -- not in the library
    # binding, which must not edit the environment of a host application that imports it.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        if not argv or argv[0] != "inference" or "M6A_RANK" in os.environ or "-h" in argv or "--help" in argv:
            return
        g = _values(argv[1:], "--gpus")
        if not g or len(g) != 1 or int(g[0]) < 2 or int(g[0]) > 64:
            return
        # more ranks than GPUs is refused later, with a message, unless the debugging transport lets ranks share one: do not
        # start what would only be killed (the render nodes the container sees are a cheap stand-in for hipGetDeviceCount)
        if not ranks_may_share_a_gpu() and int(g[0]) > visible_devices_estimate():
            return
        dirs, out = _values(argv[1:], "--input_dir"), _values(argv[1:], "--out_dir")
        if not dirs or not out or len(out) != 1:
            return
        start_ranks(int(g[0]), dirs, out[0], argv[1:])
    except (ValueError, OSError):
        cleanup()                                            # the regular path will start the ranks, or report what is wrong
