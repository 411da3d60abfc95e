#!/usr/bin/env python3
"""End-to-end `m6anet_amd inference` on the bundled data replicated N times (N=1400 ~ the size of the
dataset the reference publishes its 408 s for: 95k sites / 8M reads, README.md:206,245-249)."""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import measure_io  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1400
    with tempfile.TemporaryDirectory() as d:
        size = measure_io.replicate(n, d)
        out = os.path.join(d, "out")
        from m6anet_amd.__main__ import main as cli
        from m6anet_amd import data_utils, engine
        t0 = time.perf_counter()
        cli(["inference", "--input_dir", d, "--out_dir", out, "--num_iterations", "1000", "--n_processes", "0"])
        wall = time.perf_counter() - t0
        # phase split (second run, warm)
        t = time.perf_counter()
        batch = data_utils.load_sites_native([d], 20, "norm_hct116.npz")
        t_load = time.perf_counter() - t
        eng = engine.M6ANetEngine()
        eng.prepare_host_io()              # as the CLI does on its engine thread while the loader parses
        t = time.perf_counter()
        rp, sp, mr = eng.infer(batch.X, batch.site_kmers, batch.off, 1000)
        t_gpu = time.perf_counter() - t     # cold: MT19937 stream + index tables are built inside
        t = time.perf_counter()
        rp, sp, mr = eng.infer(batch.X, batch.site_kmers, batch.off, 1000)
        t_gpu_warm = time.perf_counter() - t
        t = time.perf_counter()
        batch.native.write_csv(out, rp, sp, mr, write_header=True)
        t_csv = time.perf_counter() - t
        # the same dataset packed once into a binary site store: opening it is a mmap
        store = os.path.join(d, "data.m6astore")
        t = time.perf_counter()
        data_utils.pack_sites([d], store, 20, "norm_hct116.npz")
        t_pack = time.perf_counter() - t
        t = time.perf_counter()
        sb = data_utils.open_store(store, "norm_hct116.npz", 20)
        t_open = time.perf_counter() - t
        t = time.perf_counter()
        eng.infer(sb.X, sb.site_kmers, sb.off, 1000)
        t_gpu_store = time.perf_counter() - t
        t0 = time.perf_counter()
        cli(["inference", "--input_dir", store, "--out_dir", out, "--num_iterations", "1000", "--n_processes", "0"])
        wall_store = time.perf_counter() - t0
        # the N-GPU split of the same job, as far as one GPU can show it: two ranks sharing the GPU, results through the
        # exchange directory (M6A_EXCHANGE=host) -- what the launcher, the rank start-up and the gather to rank 0 cost
        import subprocess
        t0 = time.perf_counter()
        subprocess.run([sys.executable, "-m", "m6anet_amd", "inference", "--input_dir", store, "--out_dir", os.path.join(d, "out2"),
                        "--num_iterations", "1000", "--n_processes", "0", "--gpus", "2"], check=True,
                       env=dict(os.environ, M6A_EXCHANGE="host"), cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        wall_gpus2 = time.perf_counter() - t0
        t0 = time.perf_counter()
        subprocess.run([sys.executable, "-m", "m6anet_amd", "inference", "--input_dir", store, "--out_dir", os.path.join(d, "out1"),
                        "--num_iterations", "1000", "--n_processes", "0"], check=True,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        wall_gpus1_process = time.perf_counter() - t0
        same = all(open(os.path.join(d, "out1", f), "rb").read() == open(os.path.join(d, "out2", f), "rb").read()
                   for f in ("data.site_proba.csv", "data.indiv_proba.csv"))
        sites, reads = batch.n_sites, int(batch.off[-1])
        print(json.dumps({"copies": n, "json_MB": size / 1e6, "sites": sites, "reads": reads,
                          "cli_wall_s": wall, "sites_per_s_end_to_end": sites / wall,
                          "load_s": t_load, "gpu_infer_host_pointers_s": t_gpu, "gpu_infer_host_pointers_warm_s": t_gpu_warm, "csv_s": t_csv,
                          "store_pack_s": t_pack, "store_bytes": os.path.getsize(store), "store_open_s": t_open,
                          "gpu_infer_from_mapped_store_s": t_gpu_store, "cli_wall_from_store_s": wall_store,
                          "cli_process_from_store_s": wall_gpus1_process, "cli_process_from_store_gpus2_host_exchange_s": wall_gpus2,
                          "gpus2_csv_bytes_equal_gpus1": same,
                          "pool_kernel": eng.last_pool_variant,
                          "site_csv_bytes": os.path.getsize(os.path.join(out, "data.site_proba.csv")),
                          "indiv_csv_bytes": os.path.getsize(os.path.join(out, "data.indiv_proba.csv")),
                          "reference_published": "408.17 s for 95,030 sites / 8,019,824 reads at num_iterations=1000 (EPYC 7R32, 25 processes)"},
                         indent=1))


if __name__ == "__main__":
    main()
