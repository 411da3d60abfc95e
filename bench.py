#!/usr/bin/env python3
"""bench.py -- DRACH sites/s of the m6A inference hot path on MI355X.

Workloads
  uniform (default; BASELINE.json configs[2], the config the metric is quoted on): per GPU
      1,000,000 synthetic DRACH sites x 20 reads, HCT116_RNA002 weights, num_iterations=1000.
      With N GPUs: configs[3]'s shape (N x 1M sites, site-sharded).
  ragged (BASELINE.json configs[4], per-GPU shape): per GPU 125,000 sites x 50..500 reads,
      HEK293T_RNA004 weights, num_iterations=1000.
Both: exact NumPy-stream replay (batch_size 16, save_per_batch 2, seed 0).  One step = one pass of the
hot path (read encoder -> site pooling; read_prob, site_prob, mod_ratio all produced) over the rank's
sites, inputs resident in HBM.  With N GPUs every rank holds its own shard of an N-times-larger job
(weak scaling; shards are flush-group aligned and carry the job offset, so the job's results do not
depend on N) and each step ends with ONE gather of site_prob + mod_ratio to rank 0 over RCCL.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload uniform|ragged] [--sites S] [--iters T]

`--gpus N` without a launcher starts its own N ranks (re-executes under torch.distributed.run on
127.0.0.1); under torch.distributed.run (RANK/WORLD_SIZE set) it runs as one rank.  The 8-GPU run is
    python bench.py --gpus 8        (or: python -m torch.distributed.run --nnodes=1 --nproc-per-node 8
                                      --master-addr 127.0.0.1 --master-port 29500 bench.py --gpus 8)
M6A_BENCH_BACKEND=gloo is a debugging aid: ranks may then share one GPU and the gather is staged
through host memory.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

ENC_FLOP_PER_READ = 14164      # 2*(15*150 + 150*32 + 32)           SURVEY.md section 8(d)
ENC_BYTES_PER_READ = 40        # 9 f32 in + 1 f32 out               SURVEY.md section 8(d)
PEAK_F32_TFLOPS = 157.3        # MI355X_MICROARCH.md: f32 MFMA = f32 vector peak
PEAK_HBM_GBPS = 8000.0
# ds_read_b32 gathers, conflict-free: 64 lanes per 2 LDS cycles per CU (MI355X_MICROARCH.md, LDS table)
PEAK_LDS_GATHERS = 256 * 2.4e9 * 32

# one multiply per draw; packed f32: 2 per lane per 4-cycle issue slot = the 157.3 TFLOP/s vector peak / 2 flops per FMA
PEAK_VALU_MULS = PEAK_F32_TFLOPS * 1e12 / 2


def pool_roofline(variant, draws, avg_ms, launches):
    """The pooling kernel against the resource that bounds it: the register kernel of uniform bags issues one
    v_pk_mul_f32 per two draws and no LDS gather at all (VALU issue); every other variant gathers one 4-byte value
    per draw from LDS.  DESIGN.md sections 4.2r / 4.3 derive both and the practical ceilings under them."""
    rate = draws / (avg_ms * 1e-3) if avg_ms else None
    if variant == "table-reg":
        bound, peak = "valu", PEAK_VALU_MULS
        note = ("peak = packed float32 multiply rate (2 per lane per issue slot, 78.6 T/s); on this part v_pk_mul_f32 issues every "
                "~7 cycles, not 4 (profiles/r02_gpr_variants.txt), which puts the practical ceiling of bags-in-registers at 44.8 T draws/s")
    else:
        bound, peak = "l1-lds", PEAK_LDS_GATHERS
        note = ("a draw moves 2 index bytes through the vector L1 (64 B/clk/CU) and one 4-byte LDS gather (32 banks/clk/CU): both pipes "
                "peak at 19.7 T draws/s; knock-out builds show the index rows are the binding one for the table kernel "
                "(DESIGN.md section 4.3; the preparation kernel runs on a side stream under the encoder and is not in avg_launch_ms)"
                if variant == "ragged-table" else
                "peak = conflict-free ds_read_b32 gather rate (one 4-byte gather per draw)")
    return {"kernel": "site pooling (%s)" % variant, "bound": bound, "achieved": rate / 1e12 if rate else None, "peak": peak / 1e12,
            "unit": "T draws/s", "frac": rate / peak if rate else None, "note": note, "avg_launch_ms": avg_ms, "launches": launches,
            "draws_per_launch": draws}


WORKLOADS = {
    "uniform": dict(model="HCT116_RNA002", sites=1_000_000, bag=20, config="BASELINE.json configs[2]"),
    "ragged": dict(model="HEK293T_RNA004", sites=125_000, bag=(50, 500), config="BASELINE.json configs[4] per-GPU shape"),
}


def measured_traffic(S, bag):
    """HBM bytes per encoder launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE, separate runs; profiles/*_enc_traffic.json), when they were taken on this workload.
    The newest profile wins; the bench line names the file so a stale figure is visible."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "*_enc_traffic.json"))):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        w = d.get("workload", {})
        if w.get("sites") == S and w.get("reads_per_site") == (list(bag) if isinstance(bag, tuple) else bag):
            best = dict(d, file=os.path.relpath(f, REPO))
    return best


def live_traffic(workload, kernel_name, timeout_s=240):
    """HBM bytes per encoder launch measured NOW: two short re-runs of this script under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, --kernel-trace only, as
    MI355X_MICROARCH.md prescribes), read back from the rocpd database.  FETCH_SIZE x2: gfx950 tallies the 128-byte
    requests of wide streaming reads at 64 bytes.  Returns None when rocprofv3 is unavailable or a pass fails."""
    import glob
    import shutil
    import sqlite3
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="m6a_pmc_")
        try:
            cmd = ["rocprofv3", "--pmc", ctr, "--kernel-trace", "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
                   "--workload", workload, "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-live-traffic"]
            subprocess.run(cmd, capture_output=True, timeout=timeout_s, env=dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp")))
            dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
            if not dbs:
                return None
            con = sqlite3.connect(dbs[0])
            cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
            kn = "kernel_name" if "kernel_name" in cols else "name"
            row = con.execute("select avg(value), count(*) from counters_collection where counter_name = ? and %s like ?" % kn,
                              (ctr, kernel_name + "%")).fetchone()
            con.close()
            if not row or not row[1]:
                return None
            vals[ctr] = float(row[0])
        except (subprocess.SubprocessError, OSError, sqlite3.Error):
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {"traffic_bytes_per_launch": (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0,
            "source": "measured by this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over 4 launches of %s "
                      "(FETCH_SIZE %.0f KiB x2 gfx950 correction + WRITE_SIZE %.0f KiB)" % (kernel_name, vals["FETCH_SIZE"], vals["WRITE_SIZE"])}


# ---------------------------------------------------------------------------------------------
# CPU baseline (runs in its own process: it forks worker pools, which must not happen after HIP is up)
# ---------------------------------------------------------------------------------------------
def _pool_site_worker(job):
    """One site of a flush group, as a Pool task (the reference's imap over sites, inference_utils.py:103-104)."""
    from oracle import m6a_oracle as orc
    p, T, thr = job
    off = np.array([0, len(p)], np.int64)
    site, _ = orc.site_pool(p, off, T, thr, batch_size=1, save_per_batch=1)
    return float(site[0])


def cpu_baseline_main(workload, T, budget_s):
    """The oracle (a port of the reference's algorithm, oracle/m6a_oracle.c) on this host's cores, on a
    bounded sample of the bench workload.  Three figures, as BASELINE.md section 3 asks:
      reference_shaped  the loop `m6anet inference` runs: encoder per 16-site batch, then per flush group
                        of <= 32 sites a NEW multiprocessing.Pool(n) and one task per site
                        (inference_utils.py:33-54,102-104), at n = 1, 25 (the reference's default) and all cores;
      best_case         the same arithmetic with one persistent set of threads over all sites (what the
                        reference could do at best) -- `value`, the figure most favourable to the CPU;
      single_thread     one thread in-process."""
    import multiprocessing as mp
    from m6anet_amd import synthetic
    from m6anet_amd.constants import DEFAULT_READ_THRESHOLD
    from m6anet_amd.engine import load_weights
    from oracle import m6a_oracle as orc
    orc.build()
    spec = WORKLOADS[workload]
    cores = os.cpu_count() or 1
    thr = np.float32(DEFAULT_READ_THRESHOLD)
    weights = load_weights(spec["model"])
    n_sample = 600_000 if workload == "uniform" else 60_000
    d = synthetic.make_sites(n_sample, spec["bag"], seed=20250328)
    S = n_sample

    def run(n, threads):
        off = d["off"][:n + 1]
        t0 = time.perf_counter()
        p = orc.encode_reads(weights, d["X"][:off[-1]], d["site_kmers"][:n], off, n_threads=threads)
        orc.site_pool(p, off, T, thr, n_threads=threads)
        return time.perf_counter() - t0

    # (iii) one thread in-process
    n1 = 256 if workload == "uniform" else 64
    t1 = run(n1, 1)
    single = n1 / t1
    # (ii) persistent threads over all sites
    probe = min(S, 64 * cores)
    t = run(probe, cores)
    n = int(min(S, max(probe, probe * (budget_s * 0.5) / max(t, 1e-6))))
    n -= n % 32
    n = max(n, min(S, 32))
    t = run(n, cores)
    best = n / t

    # (i) reference-shaped: per batch of 16 sites the encoder, per flush group a fresh Pool
    def ref_shaped(n_proc, n_groups):
        ctx = mp.get_context("fork")
        bs, gsz = 16, 32
        done = 0
        t0 = time.perf_counter()
        for g in range(n_groups):
            a, b = g * gsz, min(S, (g + 1) * gsz)
            probs = []
            for s0 in range(a, b, bs):
                s1 = min(b, s0 + bs)
                off = d["off"][s0:s1 + 1] - d["off"][s0]
                p = orc.encode_reads(weights, d["X"][d["off"][s0]:d["off"][s1]], d["site_kmers"][s0:s1], off)
                probs.extend(p[off[i]:off[i + 1]] for i in range(s1 - s0))
            with ctx.Pool(n_proc) as pool:
                list(pool.imap(_pool_site_worker, [(p, T, thr) for p in probs]))
            done += b - a
        return done / (time.perf_counter() - t0)

    shaped = {}
    for n_proc, groups in ((1, 12), (25, 8), (cores, 3)):
        if n_proc > cores:
            continue
        shaped["n_processes=%d" % n_proc] = ref_shaped(n_proc, groups)
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": best, "unit": "sites/s", "cores": cores, "kind": "port", "cpu_model": model,
            "per_core_value": best / cores, "single_thread_value": single,
            "reference_shaped_value": shaped,
            "sample": "first %d sites of the same workload (encoder + T=%d sampling) on %d host threads, %.1f s; "
                      "single thread: %d sites; reference-shaped (fresh Pool per 32-site flush): 3-12 flush groups "
                      "per setting" % (n, T, cores, t, n1)}


def run_cpu_baseline_subprocess(workload, T):
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--workload", workload, "--iters", str(T)]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
        for line in reversed(out.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"error": (out.stderr or out.stdout)[-400:]}
    except (subprocess.SubprocessError, OSError, ValueError) as e:
        return {"error": repr(e)}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="uniform")
    ap.add_argument("--sites", type=int, default=None, help="sites per GPU (default: the workload's)")
    ap.add_argument("--reads", type=int, default=None, help="uniform workload: reads per site (default 20)")
    ap.add_argument("--iters", type=int, default=1000, help="num_iterations")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not re-run under rocprofv3 for roofline.traffic; quote the committed profiles/*_enc_traffic.json instead")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--enc-variant", type=int, default=0, help="0 auto, 1 general 16-slot, 2 12-slot encoder kernel")
    ap.add_argument("--scan-driver", type=int, default=0, help="ragged bags: 0 auto, 1 per group, 2 per site, 3 index tables")
    ap.add_argument("--verify", action="store_true",
                    help="after the timed steps rank 0 recomputes the WHOLE job unsharded on its GPU and checks that the gathered "
                         "site_prob / mod_ratio equal it bit for bit (small --sites only: rank 0 holds the whole job)")
    args = ap.parse_args()

    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline_main(args.workload, args.iters, budget_s=20.0)))
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start our own ranks, one per GPU, rendezvous on 127.0.0.1
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import torch
    from m6anet_amd import dist as mdist, synthetic
    from m6anet_amd.constants import DEFAULT_READ_THRESHOLD
    from m6anet_amd.engine import M6ANetEngine, load_weights, shard_plan

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    backend = os.environ.get("M6A_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        mdist.init_from_env(backend, device_id=dev if backend == "nccl" else None)

    spec = WORKLOADS[args.workload]
    S = args.sites or spec["sites"]
    bag = spec["bag"] if args.workload == "ragged" else (args.reads or spec["bag"])
    T = args.iters
    thr = np.float32(DEFAULT_READ_THRESHOLD)
    weights = load_weights(spec["model"])

    # this rank's shard of the world*S-site job: flush-group-aligned cut of the global site range, balanced by reads
    n_reads_job = synthetic.bag_sizes(world * S, bag)
    off_job = np.zeros(world * S + 1, np.int64)
    np.cumsum(n_reads_job, out=off_job[1:])
    cuts = shard_plan(off_job, world)
    a, b = int(cuts[rank]), int(cuts[rank + 1])
    d = synthetic.make_sites(b - a, seed=20250328 + rank, n_reads=n_reads_job[a:b])
    X = torch.from_numpy(d["X"]).to(dev)
    km = torch.from_numpy(d["site_kmers"]).to(dev)
    off = torch.from_numpy(d["off"]).to(dev)
    Sr, R = b - a, int(d["off"][-1])

    eng = M6ANetEngine(weights=weights, device=local_rank)
    if args.enc_variant:
        eng.set_encoder_variant(args.enc_variant)
    if args.scan_driver:
        eng.set_scan_driver(args.scan_driver)
    eng.use_torch_stream()
    eng.set_job_offset(a)
    rp = torch.empty(R, dtype=torch.float32, device=dev)
    site = torch.empty(Sr, dtype=torch.float32, device=dev)
    mod = torch.empty(Sr, dtype=torch.float64, device=dev)
    # the job's one exchange: site_prob + mod_ratio to rank 0, one packed gather per step (RCCL);
    # issued async and double-buffered so the exchange of step i overlaps the compute of step i+1
    # (M6A_BENCH_GATHER=native: the same exchange on the C ABI's own RCCL communicator, m6a_gather)
    native_gather = os.environ.get("M6A_BENCH_GATHER", "torch") == "native" and backend == "nccl"
    if world > 1 and native_gather:
        gather = mdist.NativeGather(eng, cuts, dev, dst=0)
    else:
        gather = mdist.SiteGather(cuts, dev if backend == "nccl" else "cpu", dst=0) if world > 1 else None

    off_host = np.ascontiguousarray(d["off"], dtype=np.int64)
    host_offsets = os.environ.get("M6A_BENCH_HOST_OFFSETS", "1") != "0"

    def step():
        # the loader's host copy of the CSR offsets rides along (every step: its statistics are recomputed from it on
        # the host, the device array is checked against them on the GPU), so the call has nothing to read back and the
        # steps queue back to back; M6A_BENCH_HOST_OFFSETS=0: the library reads the statistics back itself (one stream
        # sync per step)
        if host_offsets:
            eng.set_host_offsets(off_host)
        eng.infer(X, km, off, T, 20, thr, 0, 16, 2, out=(rp, site, mod))
        if world > 1:
            if backend == "nccl":
                gather.start(site, mod)
            else:
                eng.sync()
                gather.start(site.cpu(), mod.cpu())

    def fence():
        if gather is not None:
            gather.drain()          # every gather issued so far has completed
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # the cold call: MT19937 stream / index tables are built for this (seed, T, bag sizes), then cached
    fence()
    t0 = time.perf_counter()
    step()
    fence()
    first_call_ms = (time.perf_counter() - t0) * 1e3
    for _ in range(args.warmup):
        step()
    fence()
    # HIP events around every launch of the dominant kernel (the encoder) inside the timed region -- roofline.achieved comes
    # from them; the pooling kernel is timed the same way over a few extra steps after it (every pair of events costs
    # the stream ~10 us per step, and the headline pays only for the pair it needs)
    eng.profile("encoder")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    enc_ms, enc_n = eng.profile_read(0)
    eng.profile("pooling")
    for _ in range(min(args.steps, 10)):
        step()
    fence()
    pool_ms, pool_n = eng.profile_read(1)
    eng.profile(False)
    eng.sync()
    if world > 1:
        tmax = torch.tensor([dt, first_call_ms], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt, first_call_ms = float(tmax[0].item()), float(tmax[1].item())

    verified = None
    if args.verify:
        got = gather.finish() if gather is not None else (site, mod)
        if rank == 0:
            parts = [synthetic.make_sites(int(cuts[r + 1] - cuts[r]), seed=20250328 + r, n_reads=n_reads_job[int(cuts[r]):int(cuts[r + 1])])
                     for r in range(world)]
            wX = torch.from_numpy(np.concatenate([p["X"] for p in parts])).to(dev)
            wk = torch.from_numpy(np.concatenate([p["site_kmers"] for p in parts])).to(dev)
            whole = M6ANetEngine(weights=weights, device=local_rank)
            _, w_site, w_mod = whole.infer(wX, wk, torch.from_numpy(off_job).to(dev), T, 20, thr, 0, 16, 2, want_read_probs=False)
            whole.sync()
            verified = bool(np.array_equal(got[0].cpu().numpy(), w_site.cpu().numpy()) and
                            np.array_equal(got[1].cpu().numpy(), w_mod.cpu().numpy()))
            whole.close()

    if rank == 0:
        total_sites = int(cuts[-1])
        enc_avg_ms = enc_ms / max(enc_n, 1)
        pool_avg_ms = pool_ms / max(pool_n, 1)
        enc_tflops = ENC_FLOP_PER_READ * R / (enc_avg_ms * 1e-3) / 1e12
        enc_gbps = ENC_BYTES_PER_READ * R / (enc_avg_ms * 1e-3) / 1e9
        draws = Sr * T * 20
        bag_txt = "%d reads" % bag if not isinstance(bag, tuple) else "%d..%d reads" % bag
        enc_kernel = {"csite12": "enc_csite_kernel", "general16": "enc_kernel"}.get(eng.last_encoder_variant, "enc_kernel")
        tr = None
        if world == 1:
            default_shape = S == spec["sites"] and T == 1000 and (args.workload == "ragged" or bag == spec["bag"])
            if default_shape and not args.no_live_traffic:
                tr = live_traffic(args.workload, enc_kernel)
            if tr is None:
                tr = measured_traffic(S, bag)
                if tr is not None:
                    tr = dict(tr, source="%s (%s) -- committed, not measured by this run" % (tr["file"], tr["source"]))
        pool_kernel = {"table-reg": "pool_reg_kernel", "table": "pool_table_kernel", "ragged-table": "pool_rtab_kernel"}.get(
            eng.last_pool_variant, "pool_scan_kernels")
        out = {
            "metric": "DRACH sites/sec at num_iterations=%d" % T,
            "value": total_sites * args.steps / dt,
            "unit": "sites/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "synthetic %d DRACH sites x %s per GPU, %s weights, num_iterations=%d, numpy-stream replay "
                                   "(batch_size 16, save_per_batch 2, seed 0); %s%s"
                                   % (S, bag_txt, spec["model"], T, spec["config"],
                                      " x%d GPUs%s" % (world, " (configs[3])" if args.workload == "uniform" else " (configs[4])") if world > 1 else ""),
                       "sites_per_gpu": S, "reads_per_site": list(bag) if isinstance(bag, tuple) else bag, "reads_rank0": R,
                       "num_iterations": T, "bag_statistics": "host copy of off[] per step, device-checked (m6a_set_host_offsets)" if host_offsets else "read back per step",
                       "pool_kernel": eng.last_pool_variant, "encoder_kernel": eng.last_encoder_variant,
                       "sharding": "contiguous flush-group-aligned site shards balanced by reads, 1 %s gather/step"
                                   % (("RCCL (m6a_gather)" if native_gather else "RCCL (torch.distributed)") if backend == "nccl" else backend)
                                   if world > 1 else "none"},
            "first_call_ms": first_call_ms,
            "verify": verified,
            "roofline": {"kernel": "read encoder (%s)" % eng.last_encoder_variant, "bound": "mfma", "achieved": enc_tflops,
                         "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s", "frac": enc_tflops / PEAK_F32_TFLOPS,
                         "traffic": tr["traffic_bytes_per_launch"] if tr else None,
                         "traffic_source": tr["source"] if tr else None,
                         "algorithmic_bytes_per_launch": ENC_BYTES_PER_READ * R,
                         "avg_launch_ms": enc_avg_ms, "launches": enc_n,
                         "algorithmic_flop_per_read": ENC_FLOP_PER_READ, "reads_per_launch": R,
                         "hbm_view": {"achieved": enc_gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                      "frac": enc_gbps / PEAK_HBM_GBPS, "algorithmic_bytes_per_read": ENC_BYTES_PER_READ}},
            "pool_roofline": pool_roofline(eng.last_pool_variant, draws, pool_avg_ms, pool_n),
            "kernels": {enc_kernel: {"avg_ms": enc_avg_ms, "launches": enc_n},
                        pool_kernel: {"avg_ms": pool_avg_ms, "launches": pool_n, "timed_in": "extra steps after the timed region",
                                      "Gdraws_per_s": draws / (pool_avg_ms * 1e-3) / 1e9 if pool_avg_ms else None}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = run_cpu_baseline_subprocess(args.workload, T)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
