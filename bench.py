#!/usr/bin/env python3
"""bench.py -- DRACH sites/s of the m6A inference hot path on MI355X.

Workload (BASELINE.json configs[2], the config the metric is quoted on): per GPU 1,000,000
synthetic DRACH sites x 20 reads, HCT116_RNA002 weights, num_iterations=1000, exact
NumPy-stream replay (batch_size 16, save_per_batch 2, seed 0).  One step = one pass of the hot
path (read encoder -> site pooling; read_prob, site_prob, mod_ratio all produced) over that
batch, inputs resident in HBM.  With N GPUs every rank holds its own 1M-site shard of an
N x 1M-site job (weak scaling; shards are flush-group aligned so the job's results do not depend
on N) and each step ends with one gather of site_prob + mod_ratio to rank 0 over RCCL.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--sites S] [--iters T]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

ENC_FLOP_PER_READ = 14164      # 2*(15*150 + 150*32 + 32)           SURVEY.md section 8(d)
ENC_BYTES_PER_READ = 40        # 9 f32 in + 1 f32 out               SURVEY.md section 8(d)
PEAK_F32_TFLOPS = 157.3        # MI355X_MICROARCH.md: f32 MFMA = f32 vector peak
PEAK_HBM_GBPS = 8000.0


def measured_traffic(S, n):
    """HBM bytes per enc_kernel launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE, separate runs; profiles/*_enc_traffic.json), when they were taken on this workload."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "*_enc_traffic.json"))):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("workload") == {"sites": S, "reads_per_site": n}:
            best = d
    return best


def cpu_baseline(d, T, thr, weights, budget_s=15.0):
    """The oracle (a port of the reference's algorithm) on this host's cores, bounded sample."""
    from oracle import m6a_oracle as orc
    orc.build()
    cores = os.cpu_count() or 1
    S = len(d["off"]) - 1

    def run(n):
        off = d["off"][:n + 1]
        X = d["X"][:off[-1]]
        t0 = time.perf_counter()
        p = orc.encode_reads(weights, X, d["site_kmers"][:n], off, n_threads=cores)
        orc.site_pool(p, off, T, thr, n_threads=cores)
        return time.perf_counter() - t0

    # single-thread rate on a small sample, for scale
    t1 = time.perf_counter()
    off1 = d["off"][:257]
    p1 = orc.encode_reads(weights, d["X"][:off1[-1]], d["site_kmers"][:256], off1)
    orc.site_pool(p1, off1, T, thr)
    single = 256 / (time.perf_counter() - t1)

    probe = min(S, 64 * cores)
    t = run(probe)
    n = int(min(S, max(probe, probe * budget_s / max(t, 1e-6)), 600_000))
    n -= n % 32
    n = max(n, min(S, 32))
    t = run(n)
    return {"value": n / t, "unit": "sites/s", "cores": cores, "kind": "port", "single_thread_value": single,
            "sample": "first %d sites of the same workload (encoder + T=%d sampling), %d host threads, %.1f s"
                      % (n, T, cores, t)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--sites", type=int, default=1_000_000, help="sites per GPU")
    ap.add_argument("--reads", type=int, default=20, help="reads per site")
    ap.add_argument("--iters", type=int, default=1000, help="num_iterations")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--enc-variant", type=int, default=0, help="0 auto, 1 general 16-slot, 2 12-slot encoder kernel")
    args = ap.parse_args()

    import torch
    from m6anet_amd import dist as mdist, synthetic
    from m6anet_amd.constants import DEFAULT_READ_THRESHOLD
    from m6anet_amd.engine import M6ANetEngine, load_weights, shard_plan

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    # M6A_BENCH_BACKEND=gloo is a debugging aid (ranks may then share a GPU, the gather is staged
    # through host memory); the real multi-GPU run is RCCL: backend "nccl", one GPU per rank
    backend = os.environ.get("M6A_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        mdist.init_from_env(backend, device_id=dev if backend == "nccl" else None)

    S, n, T = args.sites, args.reads, args.iters
    thr = np.float32(DEFAULT_READ_THRESHOLD)
    weights = load_weights("HCT116_RNA002")

    # this rank's shard of the N*S-site job: group-aligned cut of the global site range
    off_global = np.arange(world * S + 1, dtype=np.int64) * n
    cuts = shard_plan(off_global, world)
    a, b = int(cuts[rank]), int(cuts[rank + 1])
    d = synthetic.make_sites(b - a, n, seed=20250328 + rank)
    X = torch.from_numpy(d["X"]).to(dev)
    km = torch.from_numpy(d["site_kmers"]).to(dev)
    off = torch.from_numpy(d["off"]).to(dev)
    Sr, R = b - a, int(d["off"][-1])

    eng = M6ANetEngine(weights=weights, device=local_rank)
    if args.enc_variant:
        eng.set_encoder_variant(args.enc_variant)
    eng.use_torch_stream()
    eng.set_job_offset(a)
    rp = torch.empty(R, dtype=torch.float32, device=dev)
    site = torch.empty(Sr, dtype=torch.float32, device=dev)
    mod = torch.empty(Sr, dtype=torch.float64, device=dev)
    # the job's one exchange: site_prob + mod_ratio to rank 0, one packed gather per step (RCCL);
    # issued async and double-buffered so the exchange of step i overlaps the compute of step i+1
    gather = mdist.SiteGather(cuts, dev if backend == "nccl" else "cpu", dst=0) if world > 1 else None

    def step():
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

    for _ in range(args.warmup):
        step()
    fence()
    eng.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    enc_ms, enc_n = eng.profile_read(0)
    pool_ms, pool_n = eng.profile_read(1)
    eng.profile(False)
    eng.sync()
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        total_sites = int(cuts[-1])
        enc_avg_ms = enc_ms / max(enc_n, 1)
        pool_avg_ms = pool_ms / max(pool_n, 1)
        enc_tflops = ENC_FLOP_PER_READ * R / (enc_avg_ms * 1e-3) / 1e12
        enc_gbps = ENC_BYTES_PER_READ * R / (enc_avg_ms * 1e-3) / 1e9
        tr = measured_traffic(S, n) if world == 1 else None
        out = {
            "metric": "DRACH sites/sec at num_iterations=%d" % T,
            "value": total_sites * args.steps / dt,
            "unit": "sites/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "synthetic %d DRACH sites x %d reads per GPU, HCT116_RNA002 weights, "
                                   "num_iterations=%d, numpy-stream replay (batch_size 16, save_per_batch 2, seed 0); "
                                   "BASELINE.json configs[2]%s" % (S, n, T, " x%d GPUs (configs[3] shape)" % world if world > 1 else ""),
                       "sites_per_gpu": S, "reads_per_site": n, "num_iterations": T,
                       "pool_kernel": eng.last_pool_variant, "encoder_kernel": eng.last_encoder_variant, "sharding": "site shards, 1 RCCL gather/step" if world > 1 else "none"},
            "roofline": {"kernel": "read encoder (%s)" % eng.last_encoder_variant, "bound": "mfma", "achieved": enc_tflops,
                         "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s", "frac": enc_tflops / PEAK_F32_TFLOPS,
                         "traffic": tr["traffic_bytes_per_launch"] if tr else None,
                         "traffic_source": tr["source"] if tr else None,
                         "algorithmic_bytes_per_launch": ENC_BYTES_PER_READ * R,
                         "avg_launch_ms": enc_avg_ms, "launches": enc_n,
                         "algorithmic_flop_per_read": ENC_FLOP_PER_READ, "reads_per_launch": R,
                         "hbm_view": {"achieved": enc_gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                      "frac": enc_gbps / PEAK_HBM_GBPS, "algorithmic_bytes_per_read": ENC_BYTES_PER_READ}},
            "kernels": {{"csite12": "enc_csite_kernel", "general16": "enc_kernel"}.get(eng.last_encoder_variant, "enc_kernel"):
                            {"avg_ms": enc_avg_ms, "launches": enc_n},
                        {"table-reg": "pool_reg_kernel", "table": "pool_table_kernel"}.get(eng.last_pool_variant, "pool_scan_kernels"): {
                            "avg_ms": pool_avg_ms, "launches": pool_n,
                            "Gdraws_per_s": Sr * T * 20 / (pool_avg_ms * 1e-3) / 1e9 if pool_avg_ms else None}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(d, T, thr, weights)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
