"""`m6anet_amd inference --gpus N`: the job's sites split over N GPUs of one node.

The reference has no multi-device path; this is north_star's: candidate sites are independent, so the job is cut
into contiguous, flush-group-aligned shards balanced by read count (`m6a_shard_plan`), every rank -- one process per
GPU -- runs the whole hot path on its shard with `m6a_set_job_offset(first_site)` (flush groups and RNG restarts are
those of the whole job, so the CSVs do not depend on N).
The OUTPUT is sharded like the compute: every rank maps the store, so it formats the rows of its own sites and
pwrite()s them at its offset into data.site_proba.csv / data.indiv_proba.csv (`m6a_io_csv_shard_size / _write`; the
byte counts meet in the exchange directory, 16 bytes per rank) -- neither the per-read probabilities (4 B per read, the
bulk of the output) nor the site results leave the rank that computed them, and the 8-GPU job is not one host thread's
write().  So the CSVs need NO device exchange, and by default the command performs none (round 4 always ran north_star's
gather and then used it only to re-check rank 0's own shard: an RCCL bring-up the result did not depend on, and one more
way for the job to fail -- ADVICE r4).  The gather stays available for callers that want the site results in ONE place:
M6A_EXCHANGE=rccl brings site_prob + mod_ratio (`m6a_gather`, 12 B per site) to rank 0 over RCCL/xGMI -- the library's own
communicator, no torch.distributed -- and rank 0 checks them against the rows it can see; `bench.py --gpus N` measures
that gather every step.

    the process the user started = rank 0
      |- makes an exchange directory, starts ranks 1..N-1: the same command line with M6A_RANK / M6A_WORLD / M6A_XDIR /
      |  M6A_STORE in the environment
      |- packs data.json -> one binary site store there unless the input already is one (every rank maps the same file:
      |  nothing is parsed N times, a rank touches only its shard's pages)
      |- runs its own shard, joins the exchange, writes its rows
      '- waits for the others; a rank that dies ends the job (exact pids) with its exit code

M6A_EXCHANGE = none (default) | rccl | host.  rccl: the 128-byte RCCL id travels through the exchange directory (rank 0
writes it, the others wait for it).  host is a debugging aid like bench.py's M6A_BENCH_BACKEND=gloo: the gather goes
through files in the exchange directory instead of RCCL.  M6A_SHARE_GPU=1 (implied by host) lets ranks share a GPU
(rank % devices; RCCL refuses two ranks on one device, so rccl + M6A_SHARE_GPU is accepted only with M6A_RCCL_STANDIN=1 and the
stand-in transport named in M6A_RCCL_LIB: tests/test_gpu_comm_stub.py), which is how the one-GPU test box checks that N = 2..8 ranks give
the CSV bytes of one.
"""
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

from . import _early
from ._early import exchange_base, store_size_estimate  # noqa: F401  (re-exported: tests and docs name them here)
from .constants import DEFAULT_MIN_READS, N_SAMPLES
from .data_utils import STORE_SUFFIX, open_store, pack_sites
from .engine import M6ANetEngine, comm_unique_id, device_count, reference_written_sites, shard_plan


def _timeout():
    return float(os.environ.get("M6A_EXCHANGE_TIMEOUT", "900"))


def _wait_for(path, what, parent, busy=None):
    """Blocks until `path` exists (written atomically by its producer).  Gives up when the launcher has gone away or
    after M6A_EXCHANGE_TIMEOUT seconds -- a rank must never wait for ever on a peer that died.  While `busy` exists
    (the launcher's "still packing" marker) the clock does not run: a very large data.json may take longer to pack than
    any sensible exchange timeout."""
    t0, nap = time.time(), 0.0002
    while not os.path.exists(path):
        if os.getppid() != parent:
            raise RuntimeError("launcher has gone away while waiting for %s" % what)
        if busy is not None and os.path.exists(busy):
            t0 = time.time()
        if time.time() - t0 > _timeout():
            raise TimeoutError("timed out after %.0f s waiting for %s (%s)" % (_timeout(), what, path))
        time.sleep(nap)
        nap = min(nap * 1.5, 0.004)                # a rank waits a few times per job; 4 ms naps cost nothing and add nothing


def _publish(path, data):
    tmp = "%s.tmp%d" % (path, os.getpid())
    with open(tmp, "wb") as f:
        f.write(data)
    os.rename(tmp, path)


def exchange_mode(world):
    """(mode, share): mode = what brings the site results to rank 0 (none: nothing, the default -- the CSVs are written
    sharded), share = ranks may sit on the same GPU (debugging)."""
    mode = os.environ.get("M6A_EXCHANGE", "none")
    if mode not in ("none", "rccl", "host"):
        raise ValueError("M6A_EXCHANGE must be 'none', 'rccl' or 'host'")
    share = _early.ranks_may_share_a_gpu()
    if mode == "rccl" and share and not _early.rccl_is_a_standin():
        # a real RCCL refuses two ranks on one device; M6A_RCCL_STANDIN=1 + M6A_RCCL_LIB name a stand-in transport that does
        # not -- tests/test_gpu_comm_stub.py runs this leg that way on a one-GPU box
        raise ValueError("M6A_EXCHANGE=rccl needs one GPU per rank (M6A_SHARE_GPU is set)")
    if not share and device_count() < world:
        raise RuntimeError("--gpus %d but only %d HIP device(s) visible (M6A_SHARE_GPU=1 lets ranks share a GPU for debugging)"
                           % (world, device_count()))
    return mode, share


# ---------------------------------------------------------------------------------------------------------------
# launcher
# ---------------------------------------------------------------------------------------------------------------
def rank_argv(args):
    """The command line of a rank: the launcher's own options (the pretrained-model defaults are resolved again there)."""
    argv = ["--input_dir"] + [str(d) for d in args.input_dir] + ["--out_dir", str(args.out_dir)]
    for name in ("pretrained_model", "model_config", "model_state_dict", "norm_path", "batch_size", "save_per_batch", "n_processes",
                 "num_iterations", "device", "seed", "read_proba_threshold", "gpus", "encoder"):
        v = getattr(args, name)
        if v is not None:
            argv += ["--" + name, str(v)]
    if args.drop_unflushed_tail:
        argv.append("--drop_unflushed_tail")
    return argv


def launch(args, weights):
    """The process the user started IS rank 0: it starts ranks 1..N-1 (the same command line with M6A_RANK / M6A_WORLD / M6A_XDIR /
    M6A_STORE in the environment), packs the store if the input is not one, runs its own shard, and waits for the others --
    no interpreter start-up sits between the command and rank 0's work.  Returns the exit code."""
    world = int(args.gpus)
    # ranks 1..N-1 were normally started by m6anet_amd/_early.py, before this process imported NumPy; if not (the CLI called
    # as a function), they are started here
    st = _early.state
    dirs = [str(d) for d in args.input_dir]
    if st is not None and (st["world"] != world or st["input_dirs"] != dirs or st["out_dir"] != str(args.out_dir)
                           or st["argv"] != _early.strip_command(sys.argv[1:])):
        # the hand-parsed command line of the early start is not what argparse made of it (an option given twice: argparse keeps
        # the last, the hand parser saw the first; the CLI called as a function with another argv): those ranks wait for a store
        # nobody will write, or run another job -- end them, start again from the parsed arguments
        _early.cleanup()
        st = None
    if st is None:
        exchange_mode(world)                               # fails before anything is started, with the reason
        st = _early.start_ranks(world, dirs, args.out_dir, rank_argv(args))
    xdir, store, given_store, procs = st["xdir"], st["store"], st["given_store"], st["procs"]
    exchange_mode(world)                                   # more ranks than devices etc.: the `finally` below ends the early ranks
    # a terminated launcher still takes its ranks down and removes the exchange directory (a packed store can be hundreds of MB
    # of /dev/shm): SIGTERM becomes an exception, so the `finally` below runs
    import signal
    import threading

    def on_term(signum, frame):
        raise SystemExit(128 + signum)
    try:
        signal.signal(signal.SIGTERM, on_term)
    except ValueError:                                       # not the main thread (the CLI called from a library): keep the default
        pass
    stop = threading.Event()

    def watch():
        # rank 0 may be blocked in the exchange, waiting for a rank that has died: a failed child ends the job with its code
        while not stop.wait(0.005):
            bad = [p.returncode for p in procs if p.poll() not in (None, 0)]
            if bad:
                for p in procs:
                    if p.poll() is None:
                        p.kill()
                shutil.rmtree(xdir, ignore_errors=True)
                sys.stderr.flush()
                os._exit(bad[0] if 0 < bad[0] < 256 else 1)
    try:
        rank_env = dict(M6A_WORLD=str(world), M6A_XDIR=xdir, M6A_STORE=store)
        threading.Thread(target=watch, daemon=True).start()
        # rank 0's GPU context comes up on a thread while the store is packed, like the other ranks' in their processes
        made = {}

        def make_engine():
            try:
                from .scripts.inference import make_engine_for
                made["engine"] = make_engine_for(args, weights, 0)
            except BaseException as exc:                    # noqa: BLE001 -- re-raised on the main thread
                made["error"] = exc
        starter = threading.Thread(target=make_engine)
        starter.start()
        if not given_store:
            # parse + normalise ONCE, while the ranks bring their HIP runtimes up; they map the result.  The marker stops the
            # ranks' timeout clocks for as long as the packing takes.
            marker = os.path.join(xdir, "packing")
            _publish(marker, b"1")
            try:
                pack_sites(args.input_dir, store, DEFAULT_MIN_READS, args.norm_path, n_threads=args.n_processes)
            finally:
                os.remove(marker)
        starter.join()
        if "error" in made:
            raise made["error"]
        os.environ.update(rank_env, M6A_RANK="0")
        try:
            run_rank(args, weights, engine=made["engine"])
        finally:
            for k in ("M6A_RANK", "M6A_WORLD", "M6A_XDIR", "M6A_STORE"):
                os.environ.pop(k, None)
        stop.set()
        return _wait_all(procs)
    finally:
        stop.set()
        _early.cleanup()                                     # kills what is still running, removes the exchange directory


def _wait_all(procs):
    """0 when every rank exits 0.  The first rank that fails ends the job: the others are terminated (they would wait
    for it in the exchange) and its code is returned."""
    t0 = time.time()
    limit = float(os.environ.get("M6A_JOB_TIMEOUT", "0"))      # 0: no limit
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad or (limit and time.time() - t0 > limit):
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            if not bad:
                print("m6anet_amd: job exceeded M6A_JOB_TIMEOUT=%.0f s" % limit, file=sys.stderr)
            return bad[0] if bad else 124
        if all(c == 0 for c in codes):
            return 0
        time.sleep(0.002)


# ---------------------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------------------
def write_rows_sharded(native, out_dir, xdir, rank, world, a, b, read_prob, site_prob, mod_ratio, n_write, parent):
    """Every rank writes the rows of ITS sites [a, b) (clipped to the first n_write sites of the job: the reference's row set may
    end inside or before this shard): they are formatted once to learn their size, the sizes meet in the exchange directory
    (16 bytes per rank), and each rank pwrite()s its rows at its offset into the two CSVs -- N hosts' worth of formatting and
    write() instead of rank 0's alone, and nobody holds the job's 4 B per read.  `native` = the mapped sites (_io.NativeSites);
    the arrays hold the shard's values.  Rank 0 returns when every rank's rows are in the files."""
    from ._io import usable_cpus
    off = native.off
    wa, wb = min(a, n_write), min(b, n_write)
    nr = int(off[wb] - off[wa])
    threads = max(1, usable_cpus() // world)                 # the ranks share one host: each formats on its share of the CPUs
    rp, sp, mr = read_prob[:nr], site_prob[:wb - wa], mod_ratio[:wb - wa]
    sizes = native.csv_shard_size(wa, wb, rp, sp, mr, n_threads=threads)
    _publish(os.path.join(xdir, "csv_size%d" % rank), np.array(sizes, np.int64).tobytes())
    all_sizes = np.zeros((world, 2), np.int64)
    for r in range(world):
        p = os.path.join(xdir, "csv_size%d" % r)
        _wait_for(p, "rank %d's CSV sizes" % r, parent)
        all_sizes[r] = np.fromfile(p, np.int64, 2)
    head = np.array(native.csv_header_bytes(), np.int64)
    start = head + all_sizes[:rank].sum(axis=0)
    totals = head + all_sizes.sum(axis=0)
    native.csv_shard_write(out_dir, wa, wb, rp, sp, mr, int(start[0]), int(start[1]),
                           header_and_totals=(int(totals[0]), int(totals[1])) if rank == 0 else None, n_threads=threads)
    _publish(os.path.join(xdir, "csv_done%d" % rank), b"1")
    if rank == 0:
        for r in range(world):
            _wait_for(os.path.join(xdir, "csv_done%d" % r), "rank %d's rows" % r, parent)


def run_rank(args, weights, engine=None):
    rank, world = int(os.environ["M6A_RANK"]), int(os.environ["M6A_WORLD"])
    xdir, store = os.environ["M6A_XDIR"], os.environ["M6A_STORE"]
    parent = os.getppid()                                    # ranks 1..: the launcher (= rank 0); rank 0: whoever started the job
    mode, share = exchange_mode(world)
    device = rank % max(device_count(), 1) if share else rank

    # the GPU context comes up while the launcher may still be packing the store
    if engine is None:
        from .scripts.inference import make_engine_for
        engine = make_engine_for(args, weights, device)
    _wait_for(store, "the launcher's site store", parent, busy=os.path.join(xdir, "packing"))
    batch = open_store(store, args.norm_path, DEFAULT_MIN_READS)
    off = batch.off
    cuts = shard_plan(off, world, args.batch_size, args.save_per_batch)
    a, b = int(cuts[rank]), int(cuts[rank + 1])
    r0, r1 = int(off[a]), int(off[b])
    if r1 - r0 >= (1 << 22):
        engine.prepare_host_io()                         # pinned ring: the shard's features stream from the mapping
    engine.set_job_offset(a)
    read_prob, site_prob, mod_ratio = engine.infer(
        batch.X[r0:r1], batch.site_kmers[a:b], np.ascontiguousarray(off[a:b + 1] - r0), args.num_iterations, N_SAMPLES,
        args.read_proba_threshold, args.seed, args.batch_size, args.save_per_batch)

    # ---- opt-in (M6A_EXCHANGE): site_prob + mod_ratio (12 B per site) to rank 0 -- over RCCL/xGMI, or through the exchange
    # directory in the debugging mode.  The CSVs below do not need it (every rank writes its own rows); the per-read
    # probabilities (4 B per read, the bulk of the output) never travel.
    site_all = mod_all = None
    if mode == "rccl":
        ident_path = os.path.join(xdir, "rccl_id")
        if rank == 0:
            _publish(ident_path, comm_unique_id())
        _wait_for(ident_path, "rank 0's RCCL id", parent)
        with open(ident_path, "rb") as f:
            engine.comm_init(f.read(), rank, world)
        seen = engine.comm_info()["ranks_seen"]
        if seen != world:
            raise RuntimeError("RCCL formed a communicator of %d ranks, the job has %d" % (seen, world))
        site_all, mod_all = engine.gather(site_prob, mod_ratio, cuts, dst=0)
        engine.comm_destroy()
    elif mode == "host":
        if rank != 0:
            _publish(os.path.join(xdir, "shard%d.bin" % rank), mod_ratio.tobytes() + site_prob.tobytes())
        else:
            site_all, mod_all = np.empty(int(cuts[-1]), np.float32), np.empty(int(cuts[-1]), np.float64)
            site_all[a:b], mod_all[a:b] = site_prob, mod_ratio
            for r in range(1, world):
                p = os.path.join(xdir, "shard%d.bin" % r)
                _wait_for(p, "rank %d's results" % r, parent)
                ns = int(cuts[r + 1] - cuts[r])
                raw = np.fromfile(p, np.uint8)
                if raw.size != 12 * ns:
                    raise RuntimeError("rank %d delivered %d bytes, expected %d" % (r, raw.size, 12 * ns))
                mod_all[cuts[r]:cuts[r + 1]] = raw[:8 * ns].view(np.float64)
                site_all[cuts[r]:cuts[r + 1]] = raw[8 * ns:12 * ns].view(np.float32)
    engine.close()

    n_write = batch.n_sites
    if getattr(args, "drop_unflushed_tail", False):
        n_write = reference_written_sites(batch.n_sites, args.batch_size, args.save_per_batch)
    write_rows_sharded(batch.native, args.out_dir, xdir, rank, world, a, b, read_prob, site_prob, mod_ratio, n_write, parent)
    if rank == 0 and site_all is not None:
        # the caller asked for the gather: rank 0 checks the gathered site results against what it can see of them (its own
        # shard) -- its exit code is the job's
        if not (np.array_equal(site_all[a:b], site_prob) and np.array_equal(mod_all[a:b], mod_ratio, equal_nan=True)):
            raise RuntimeError("the gathered site results do not contain rank 0's own shard")
