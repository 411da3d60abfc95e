"""Multi-GPU plumbing for the hot path.

Sites are independent, so a job is cut into contiguous, flush-group-aligned site shards (one per
rank / GPU, `shard_plan`), every rank runs encode + pool on its shard with
`engine.set_job_offset(first_site)` so the flush groups and RNG restarts are those of the whole job,
and ONE collective at the end brings site_prob + mod_ratio to rank 0 (`gather_sites`).  There is no
data-path collective.  torch.distributed is only the transport: backend "nccl" is RCCL over xGMI on
MI355X; the same code runs on "gloo" with CPU tensors (tests/test_dist_gloo.py).
"""
import os

import numpy as np

from .engine import shard_plan  # noqa: F401  (re-export)


def init_from_env(backend, device_id=None):
    """Process group from the RANK / WORLD_SIZE / MASTER_* environment torch.distributed.run sets."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if not dist.is_initialized():
        kw = {}
        if device_id is not None:
            kw["device_id"] = device_id
        dist.init_process_group(backend, **kw)
    return dist.get_rank(), dist.get_world_size()


def my_shard(cuts, rank):
    return int(cuts[rank]), int(cuts[rank + 1])


class SiteGather:
    """The job's one exchange: every rank's (site_prob float32 [S_r], mod_ratio float64 [S_r]) to rank
    `dst`, as ONE collective on a packed 12-byte-per-site buffer (mod_ratio block, then site_prob block) padded to the largest shard (so a
    plain gather works for ragged cuts).  Double-buffered: `start()` may be called for the next step
    while the previous gather is still in flight (async_op), which lets the exchange of step i
    overlap the compute of step i+1; `finish()` waits and returns the tensors on rank `dst`."""

    def __init__(self, cuts, device, dst=0, group=None):
        import torch
        import torch.distributed as dist
        self.dist, self.torch = dist, torch
        self.group, self.dst = group, dst
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.sizes = [int(x) for x in np.diff(np.asarray(cuts, np.int64))]
        self.smax = max(self.sizes)
        self.n = self.sizes[self.rank]
        nbytes = 12 * self.smax
        self.send = [torch.zeros(nbytes, dtype=torch.uint8, device=device) for _ in range(2)]
        self.recv = [[torch.empty(nbytes, dtype=torch.uint8, device=device) for _ in range(self.world)]
                     if self.rank == dst else None for _ in range(2)]
        self.work = [None, None]
        self.slot = 0

    def start(self, site, mod):
        k = self.slot
        if self.work[k] is not None:          # buffer k is two steps old: its gather must be done
            self.work[k].wait()
        buf = self.send[k]
        assert site.numel() == self.n and mod.numel() == self.n
        # float64 block first: a .view(float64) needs a storage offset that is a multiple of 8, which
        # 4 * smax is not when the largest shard has an odd site count
        buf[:8 * self.n].view(self.torch.float64).copy_(mod)
        buf[8 * self.smax:8 * self.smax + 4 * self.n].view(self.torch.float32).copy_(site)
        self.work[k] = self.dist.gather(buf, self.recv[k], dst=self.dst, group=self.group, async_op=True)
        self.slot ^= 1
        return k

    def finish(self, k=None):
        """Waits for gather `k` (default: the most recent); returns (site_all, mod_all) on dst."""
        if k is None:
            k = self.slot ^ 1
        if self.work[k] is not None:
            self.work[k].wait()
            self.work[k] = None
        if self.rank != self.dst:
            return None, None
        t = self.torch
        mod = t.cat([self.recv[k][r][:8 * self.sizes[r]].view(t.float64) for r in range(self.world)])
        site = t.cat([self.recv[k][r][8 * self.smax:8 * self.smax + 4 * self.sizes[r]].view(t.float32)
                      for r in range(self.world)])
        return site, mod

    def drain(self):
        for k in (0, 1):
            if self.work[k] is not None:
                self.work[k].wait()
                self.work[k] = None


def gather_sites(site, mod, cuts, dst=0, group=None, buffers=None):
    """Blocking convenience wrapper around SiteGather: returns (site_all, mod_all) tensors of length
    cuts[-1] on rank `dst`, (None, None) elsewhere.  `buffers` = dict reused across calls."""
    buffers = buffers if buffers is not None else {}
    key = (tuple(int(c) for c in cuts), str(site.device), dst)
    if buffers.get("key") != key:
        buffers.clear()
        buffers["key"] = key
        buffers["g"] = SiteGather(cuts, site.device, dst=dst, group=group)
    g = buffers["g"]
    return g.finish(g.start(site, mod))


class NativeGather:
    """The same exchange on the C ABI's own communicator (include/m6a.h: m6a_comm_init / m6a_gather): one grouped
    ncclSend/ncclRecv over xGMI, shards written at their offsets without padding.  The 128-byte RCCL id travels
    over the launcher's process group (any backend).  `start`/`drain` mirror SiteGather so bench.py can use either.
    Construction is collective and ends the same way on every rank: all succeed or all raise."""

    def __init__(self, engine, cuts, device, dst=0):
        import torch
        import torch.distributed as dist
        from .engine import comm_unique_id
        self.engine, self.cuts, self.dst, self.device = engine, np.asarray(cuts, np.int64), dst, device
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        ident = [None]
        if self.rank == 0:
            try:
                ident[0] = comm_unique_id()
            except Exception as e:                          # e.g. librccl not found: tell everybody instead of leaving them waiting
                ident[0] = "error: %s" % e
        dist.broadcast_object_list(ident, src=0)
        if not isinstance(ident[0], (bytes, bytearray)):
            raise RuntimeError("rank 0 could not make an RCCL id (%s)" % ident[0])
        ok, why = 1, ""
        try:
            engine.comm_init(ident[0], self.rank, self.world)
        except Exception as e:
            ok, why = 0, str(e)
        if not self._all_ok(ok):
            if ok:
                engine.comm_destroy()
            raise RuntimeError("m6a_comm_init failed on a rank%s" % (": " + why if why else ""))
        total = int(self.cuts[-1])
        n = int(self.cuts[self.rank + 1] - self.cuts[self.rank])
        self.out = [(torch.empty(total, dtype=torch.float32, device=device), torch.empty(total, dtype=torch.float64, device=device))
                    if self.rank == dst else None for _ in range(2)]
        # The exchange runs on a side stream so that it is not a link in the compute stream's chain (pooling of step i ->
        # exchange -> encoder of step i+1): the shard's results are copied into one of two send buffers on the compute
        # stream, the side stream waits for that copy and carries the RCCL group, and a buffer is reused only after the
        # exchange that read it has finished.
        self.send = [(torch.empty(n, dtype=torch.float32, device=device), torch.empty(n, dtype=torch.float64, device=device)) for _ in range(2)]
        self.side = torch.cuda.Stream(device=device)
        self.copied = [torch.cuda.Event() for _ in range(2)]
        self.sent = [torch.cuda.Event() for _ in range(2)]
        self.in_flight = [False, False]
        self.slot = 0
        self.timing = False          # bench.py: HIP events around every exchange on the side stream
        self.timed_pairs = []

    def exchange_ms(self):
        """(average ms per exchange, exchanges) over the exchanges issued while `timing` was on; drains first."""
        self.drain()
        ms = [a.elapsed_time(b) for a, b in self.timed_pairs]
        self.timed_pairs = []
        return (sum(ms) / len(ms) if ms else None), len(ms)

    def _all_ok(self, ok):
        dev = self.device if self.dist.get_backend() == "nccl" else "cpu"
        flag = self.torch.tensor([int(ok)], dtype=self.torch.int32, device=dev)
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN)
        return bool(flag.item())

    def self_test(self):
        """One exchange of a known pattern, checked on dst.  Collective; returns None when every rank is happy, else
        the reason (the same verdict on every rank)."""
        torch = self.torch
        a, b = int(self.cuts[self.rank]), int(self.cuts[self.rank + 1])
        idx = torch.arange(a, b, device=self.device)
        ok, why = 1, None
        try:
            k = self.start((idx % 1000).to(torch.float32) + self.rank * 1000.0, idx.to(torch.float64) * 0.5)
            site, mod = self.finish(k)
            if self.rank == self.dst:
                n = int(self.cuts[-1])
                want = torch.arange(0, n, device=self.device)
                owner = torch.bucketize(want, torch.as_tensor(self.cuts[1:], device=self.device), right=True)
                if not (torch.equal(site, (want % 1000).to(torch.float32) + owner.to(torch.float32) * 1000.0) and
                        torch.equal(mod, want.to(torch.float64) * 0.5)):
                    ok, why = 0, "m6a_gather delivered wrong data in its self-test"
        except Exception as e:
            ok, why = 0, "m6a_gather failed: %s" % e
        if self._all_ok(ok):
            return None
        return why or "m6a_gather self-test failed on another rank"

    def start(self, site, mod):
        torch = self.torch
        k = self.slot
        cur = torch.cuda.current_stream(self.device)
        if self.in_flight[k]:
            cur.wait_event(self.sent[k])                      # the exchange two steps ago has read this send buffer
        self.send[k][0].copy_(site)
        self.send[k][1].copy_(mod)
        self.copied[k].record(cur)
        self.side.wait_event(self.copied[k])
        self.engine.set_stream(self.side.cuda_stream)         # m6a_gather is enqueued on the context's stream
        if self.timing:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(self.side)
        try:
            self.engine.gather(self.send[k][0], self.send[k][1], self.cuts, self.dst, out=self.out[k])
        finally:
            self.engine.use_torch_stream()
        if self.timing:
            e1.record(self.side)
            self.timed_pairs.append((e0, e1))
        self.sent[k].record(self.side)
        self.in_flight[k] = True
        self.slot ^= 1
        return k

    def finish(self, k=None):
        self.side.synchronize()
        self.engine.sync()
        o = self.out[self.slot ^ 1 if k is None else k]
        return o if o is not None else (None, None)

    def drain(self):
        self.side.synchronize()
        self.engine.sync()
