"""N>1 path on CPU: world_size-2 gloo.  Each rank owns a flush-group-aligned shard of one job,
computes it with the job-offset semantics (the CPU oracle stands in for the HIP engine, which
needs a GPU), and one gather brings everything to rank 0 -- which must equal the unsharded job
bit for bit.  Exercises m6anet_amd.dist (plan, padded ragged gather) and the first_site rule."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, bag, n_sites, out_path):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    from m6anet_amd import dist as mdist, synthetic
    from m6anet_amd.engine import load_weights
    from oracle import m6a_oracle as orc
    r, w = mdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    d = synthetic.make_sites(n_sites, bag, seed=4)          # every rank sees the same job description
    weights = load_weights()
    thr = np.float32(0.033379376)
    cuts = mdist.shard_plan(d["off"], world)
    a, b = mdist.my_shard(cuts, rank)
    off = d["off"][a:b + 1] - d["off"][a]
    X = d["X"][d["off"][a]:d["off"][b]]
    p = orc.encode_reads(weights, X, d["site_kmers"][a:b], off)
    site, mod = orc.site_pool(p, off, 50, thr, first_site=a)
    bufs = {}
    for _ in range(2):                                   # twice: staging buffers are reused
        site_all, mod_all = mdist.gather_sites(torch.from_numpy(site), torch.from_numpy(mod), cuts, dst=0, buffers=bufs)
    if rank == 0:
        p_full = orc.encode_reads(weights, d["X"], d["site_kmers"], d["off"])
        f_site, f_mod = orc.site_pool(p_full, d["off"], 50, thr)
        ok = np.array_equal(site_all.numpy(), f_site) and np.array_equal(mod_all.numpy(), f_mod)
        np.save(out_path, np.array([int(ok), len(site_all), int(cuts[1])]))
    else:
        assert site_all is None and mod_all is None
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


# 1501 sites: the last shard is the larger one and has an odd site count (the packed gather buffer must
# keep its float64 block 8-byte aligned)
@pytest.mark.parametrize("bag,n_sites", [(20, 1500), ((20, 120), 1500), (20, 1501), ((20, 120), 1489)])
def test_two_rank_shards_equal_the_whole_job(tmp_path, bag, n_sites):
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(2, _free_port(), bag, n_sites, out), nprocs=2, join=True)
    ok, n, cut = np.load(out)
    assert ok == 1 and n == n_sites and 0 < cut < n_sites


def test_eight_rank_shards_equal_the_whole_job(tmp_path):
    """The size of the node the driver runs: eight gloo ranks, ragged bags, a site count that leaves the shards unequal."""
    out = str(tmp_path / "res8.npy")
    mp.spawn(_worker, args=(8, _free_port(), (20, 120), 4003, out), nprocs=8, join=True)
    ok, n, cut = np.load(out)
    assert ok == 1 and n == 4003 and 0 < cut < 4003


def test_site_gather_single_rank_odd_count():
    """world = 1 with an odd site count: the float64 view of the packed buffer must stay aligned."""
    import torch
    import torch.distributed as dist
    from m6anet_amd import dist as mdist
    os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    mdist.init_from_env("gloo")
    try:
        site = torch.arange(1009, dtype=torch.float32)
        mod = torch.arange(1009, dtype=torch.float64) * 0.5
        s_all, m_all = mdist.gather_sites(site, mod, np.array([0, 1009]))
        assert torch.equal(s_all, site) and torch.equal(m_all, mod)
    finally:
        dist.destroy_process_group()


def _sharded_writer_rank(rank, world, xdir, out_dir, cuts, seed):
    import numpy as np
    from m6anet_amd import _io, data_utils, multi_gpu
    data = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_tests_data")
    nat = _io.NativeSites([data], 20, data_utils.load_norm_factors("norm_hct116.npz"), 1)
    S, R = nat.tx_pos.size, nat.X.shape[0]
    g = np.random.Generator(np.random.PCG64(seed))
    rp, sp, mr = g.random(R, dtype=np.float32), g.random(S, dtype=np.float32), g.random(S)
    a, b = cuts[rank], cuts[rank + 1]
    off = nat.off
    multi_gpu.write_rows_sharded(nat, out_dir, xdir, rank, world, a, b, rp[off[a]:off[b]].copy(), sp[a:b].copy(), mr[a:b].copy(),
                                 n_write=90, parent=os.getppid())


def test_ranks_write_their_rows_concurrently(tmp_path):
    """The output protocol of `inference --gpus N` on CPU, real processes: four ranks map the same sites, publish the byte
    counts of their rows in the exchange directory, and pwrite() concurrently into the two CSVs -- with the reference's row set
    ending inside the third shard (n_write = 90 of 101 sites: the last rank writes nothing).  The files are m6a_io_write_csv's."""
    import multiprocessing as mp
    import numpy as np
    from m6anet_amd import _io, data_utils
    data = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_tests_data")
    nat = _io.NativeSites([data], 20, data_utils.load_norm_factors("norm_hct116.npz"), 1)
    S, R = nat.tx_pos.size, nat.X.shape[0]
    g = np.random.Generator(np.random.PCG64(11))
    rp, sp, mr = g.random(R, dtype=np.float32), g.random(S, dtype=np.float32), g.random(S)
    one = tmp_path / "one"
    one.mkdir()
    nat.write_csv(str(one), rp, sp, mr, write_header=True, n_sites=90)
    xdir, out = tmp_path / "x", tmp_path / "out"
    xdir.mkdir()
    out.mkdir()
    cuts = [0, 32, 64, 96, 101]
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_sharded_writer_rank, args=(r, 4, str(xdir), str(out), cuts, 11)) for r in (3, 1, 2, 0)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for fn in ("data.site_proba.csv", "data.indiv_proba.csv"):
        assert (out / fn).read_bytes() == (one / fn).read_bytes(), fn
