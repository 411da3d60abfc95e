#!/usr/bin/env python3
"""One rank of tests/test_gpu_comm_stub.py (test infrastructure): W of these share the one GPU of a test box and talk through
tests/stub_rccl's stand-in transport (M6A_RCCL_LIB), driving the PRODUCT's m6a_comm_init / m6a_gather / m6a_gather_reads unchanged.

    worker.py <rank> <world> <xdir> <scenario>

Scenarios (every rank runs the same one; rank `dst` checks and writes <xdir>/ok<rank>):
  planned   the job cut by m6a_shard_plan; every rank runs m6a_infer on ITS shard with its job offset, then gathers site_prob +
            mod_ratio (device pointers, then host pointers) and the read probabilities to rank 0, which has computed the WHOLE
            job unsharded: gathered == unsharded, bit for bit.
  ragged    hand-made cuts with empty, one-site and odd-sized shards; destinations 0, world-1 and the middle rank; device and host
            pointers; values that encode their global index, so a wrong receive offset or count cannot go unnoticed.
  failsend  rank 0 (the destination) has STUB_RCCL_FAIL_SEND=1: its m6a_gather must fail with the RCCL error text, leave NO open
            group on the thread (stub_rccl_group_depth() == 0), and after m6a_comm_destroy + a new communicator the gather works.
"""
import ctypes as C
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)


def publish(path, data):
    tmp = "%s.tmp%d" % (path, os.getpid())
    with open(tmp, "wb") as f:
        f.write(data)
    os.rename(tmp, path)


def wait_for(path, limit=120.0):
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > limit:
            raise TimeoutError(path)
        time.sleep(0.002)
    return open(path, "rb").read()


def form(eng, rank, world, xdir, tag):
    from m6anet_amd.engine import comm_unique_id
    p = os.path.join(xdir, "id_" + tag)
    if rank == 0:
        publish(p, comm_unique_id())
    eng.comm_init(wait_for(p), rank, world)
    info = eng.comm_info()
    assert info["ranks_seen"] == world and info["rank"] == rank and info["rccl_version"] == 99999, info     # the stub, not a real RCCL


def ragged_cuts(world):
    sizes = [7, 0, 1233, 1, 0, 64, 3, 4097, 0, 31, 2, 500][:world]
    if world >= 2:
        sizes[-1] = 129                                  # the last rank never empty: it is a destination below
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)


def main():
    rank, world, xdir, scenario = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    import torch
    from m6anet_amd import synthetic
    from m6anet_amd._lib import M6AError
    from m6anet_amd.engine import M6ANetEngine, load_weights, shard_plan
    dev = torch.device("cuda:0")
    eng = M6ANetEngine(weights=load_weights("HCT116_RNA002"), device=0)
    form(eng, rank, world, xdir, "a")

    if scenario == "planned":
        d = synthetic.make_sites(3001, (16, 60), seed=77)
        off = d["off"]
        cuts = shard_plan(off, world)
        a, b = int(cuts[rank]), int(cuts[rank + 1])
        r0, r1 = int(off[a]), int(off[b])
        eng.set_job_offset(a)
        X, km, o = (torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (d["X"][r0:r1], d["site_kmers"][a:b], off[a:b + 1] - r0))
        rp, site, mod = eng.infer(X, km, o, 200)
        site_all, mod_all = eng.gather(site, mod, cuts, dst=0)                       # device pointers: stream-ordered
        rp_all = eng.gather_reads(rp, off[cuts], dst=0)
        eng.sync()
        h_site, h_mod = eng.gather(site.cpu().numpy(), mod.cpu().numpy(), cuts, dst=0)   # host pointers: staged, synchronous
        h_rp = eng.gather_reads(rp.cpu().numpy(), off[cuts], dst=0)
        if rank == 0:
            whole = M6ANetEngine(weights=load_weights("HCT116_RNA002"), device=0)
            w_rp, w_site, w_mod = whole.infer(d["X"], d["site_kmers"], off, 200)
            whole.close()
            assert np.array_equal(site_all.cpu().numpy(), w_site) and np.array_equal(mod_all.cpu().numpy(), w_mod)
            assert np.array_equal(rp_all.cpu().numpy(), w_rp)
            assert np.array_equal(h_site, w_site) and np.array_equal(h_mod, w_mod) and np.array_equal(h_rp, w_rp)
        else:
            assert site_all is None and mod_all is None and rp_all is None and h_site is None and h_rp is None
    elif scenario == "ragged":
        cuts = ragged_cuts(world)
        a, b = int(cuts[rank]), int(cuts[rank + 1])
        idx = np.arange(a, b)
        site = (idx * 0.25 + 1.0).astype(np.float32)                                 # exact in float32: the global index, encoded
        mod = idx.astype(np.float64) * 3.0 + 0.5
        want_site = (np.arange(cuts[-1]) * 0.25 + 1.0).astype(np.float32)
        want_mod = np.arange(cuts[-1]).astype(np.float64) * 3.0 + 0.5
        rcuts = cuts * 5 + (cuts // 7)                                               # a second, different set of cuts for the reads
        rp = (np.arange(rcuts[rank], rcuts[rank + 1]) * 0.5).astype(np.float32)
        want_rp = (np.arange(rcuts[-1]) * 0.5).astype(np.float32)
        for dst in sorted({0, world - 1, world // 2}):
            for on_dev in (True, False):
                s_in = torch.from_numpy(site).to(dev) if on_dev else site
                m_in = torch.from_numpy(mod).to(dev) if on_dev else mod
                p_in = torch.from_numpy(rp).to(dev) if on_dev else rp
                sa, ma = eng.gather(s_in, m_in, cuts, dst=dst)
                ra = eng.gather_reads(p_in, rcuts, dst=dst)
                eng.sync()
                if rank == dst:
                    got = [x.cpu().numpy() if on_dev else x for x in (sa, ma, ra)]
                    assert np.array_equal(got[0], want_site) and np.array_equal(got[1], want_mod) and np.array_equal(got[2], want_rp), (dst, on_dev)
                else:
                    assert sa is None and ma is None and ra is None
    elif scenario == "failsend":
        cuts = np.arange(world + 1, dtype=np.int64) * 10
        site = torch.full((10,), float(rank), dtype=torch.float32, device=dev)
        mod = torch.full((10,), float(rank) + 0.5, dtype=torch.float64, device=dev)
        stub = C.CDLL(os.environ["M6A_RCCL_LIB"])
        if rank == 0:
            try:
                eng.gather(site, mod, cuts, dst=0)
                raise SystemExit("the injected ncclSend failure was not reported")
            except M6AError as e:
                assert "RCCL send/recv of site_prob" in str(e) and "stub" in str(e), str(e)
            assert stub.stub_rccl_group_depth() == 0, "a failing Send left the thread's RCCL group open"
        else:
            eng.gather(site, mod, cuts, dst=0)                                       # the other ranks' sends go out; nobody waits on rank 0
            eng.sync()
            assert stub.stub_rccl_group_depth() == 0
        eng.comm_destroy()
        form(eng, rank, world, xdir, "b")                                            # a NEW communicator on the same context
        sa, ma = eng.gather(site, mod, cuts, dst=0)
        eng.sync()
        if rank == 0:
            assert np.array_equal(sa.cpu().numpy(), np.repeat(np.arange(world, dtype=np.float32), 10))
            assert np.array_equal(ma.cpu().numpy(), np.repeat(np.arange(world, dtype=np.float64) + 0.5, 10))
    else:
        raise SystemExit("unknown scenario " + scenario)
    eng.comm_destroy()
    eng.close()
    publish(os.path.join(xdir, "ok%d" % rank), b"1")


if __name__ == "__main__":
    main()
