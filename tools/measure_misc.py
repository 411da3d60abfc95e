#!/usr/bin/env python3
"""Side measurements for DESIGN.md (not the bench line): ragged bags (BASELINE configs[4] shape)
through the scan kernel, configs[1] (100k sites, T=100), and the PCIe-inclusive host-pointer rate."""
import json
import os

import numpy as np
import sys
import time


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from m6anet_amd import synthetic  # noqa: E402
from m6anet_amd.engine import M6ANetEngine, load_weights  # noqa: E402


def timed(eng, X, km, off, T, reps=5):
    eng.infer(X, km, off, T)
    eng.sync()
    eng.profile(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.infer(X, km, off, T)
    eng.sync()
    dt = (time.perf_counter() - t0) / reps
    e, en = eng.profile_read(0)
    p, pn = eng.profile_read(1)
    eng.profile(False)
    return dt, e / max(en, 1), p / max(pn, 1)


def main():
    dev = torch.device("cuda:0")
    out = {}
    for tag, model, S, bag, T in [("config1_100k_T100", "HCT116_RNA002", 100_000, 20, 100),
                                  ("config4_ragged_200k_T1000", "HEK293T_RNA004", 200_000, (50, 500), 1000),
                                  ("uniform33_200k_T1000_scan", "HCT116_RNA002", 200_000, 33, 1000)]:
        eng = M6ANetEngine(weights=load_weights(model))
        d = synthetic.make_sites(S, bag, seed=1)
        X, km, off = (torch.from_numpy(d[k]).to(dev) for k in ("X", "site_kmers", "off"))
        eng.use_torch_stream()
        dt, e_ms, p_ms = timed(eng, X, km, off, T)
        R = int(d["off"][-1])
        out[tag] = {"sites": S, "reads": R, "T": T, "pool": eng.last_pool_variant, "ms_per_call": dt * 1e3,
                    "sites_per_s": S / dt, "enc_ms": e_ms, "pool_ms": p_ms,
                    "enc_TFLOPs": 14164 * R / (e_ms * 1e-3) / 1e12, "pool_Gdraws_per_s": S * T * 20 / (p_ms * 1e-3) / 1e9}
        if tag.startswith("config1"):
            eng.set_stream(None)
            eng.infer(d["X"], d["site_kmers"], d["off"], T)
            t0 = time.perf_counter()
            for _ in range(3):
                eng.infer(d["X"], d["site_kmers"], d["off"], T)
            out[tag]["host_pointer_sites_per_s"] = S * 3 / (time.perf_counter() - t0)
        eng.close()
    # PCIe-inclusive rate of the bench workload (host numpy arrays in, host arrays out); the box's host cores are
    # shared, so: best and median of 7 calls
    eng = M6ANetEngine(weights=load_weights())
    d = synthetic.make_sites(1_000_000, 20, seed=2)
    eng.infer(d["X"], d["site_kmers"], d["off"], 1000)

    def rates(**kw):
        ts, keep = [], []
        for _ in range(7):
            t0 = time.perf_counter()
            keep.append(eng.infer(d["X"], d["site_kmers"], d["off"], 1000, **kw))    # a caller keeps its results:
            ts.append(time.perf_counter() - t0)                                        # freeing them is not timed
        ts.sort()
        return 1e6 / ts[0], 1e6 / ts[3]

    best, med = rates()
    outs = (np.empty(20_000_000, np.float32), np.empty(1_000_000, np.float32), np.empty(1_000_000, np.float64))
    eng.infer(d["X"], d["site_kmers"], d["off"], 1000, out=outs)
    best_r, med_r = rates(out=outs)
    out["bench_workload_host_pointers"] = {
        "sites_per_s": best, "sites_per_s_median": med,
        "sites_per_s_reused_output_arrays": best_r, "sites_per_s_reused_output_arrays_median": med_r,
        "note": "pageable numpy buffers in; fresh numpy arrays out per call (first touch of 92 MB inside the call) or reused "
                "output arrays; chunks through the pinned staging ring, H2D / encoder / D2H overlapped; best and median of 7"}
    # the INTEGRATION.md stub's shape: the reference's own loop, one m6a_encode_reads per 16-site batch and one
    # m6a_site_pool per flush group of <= 32 sites, host arrays in and out (what a maintainer gets by swapping the
    # three call sites and nothing else)
    eng = M6ANetEngine(weights=load_weights())
    d = synthetic.make_sites(20_000, (20, 90), seed=6)
    off = d["off"]
    t0 = time.perf_counter()
    for s0 in range(0, 20_000, 16):
        s1 = min(20_000, s0 + 16)
        eng.get_read_probability(d["X"][off[s0]:off[s1]], d["site_kmers"][s0:s1], off[s0:s1 + 1] - off[s0])
    t_enc = time.perf_counter() - t0
    rp = eng.get_read_probability(d["X"], d["site_kmers"], off)
    t0 = time.perf_counter()
    n_calls = 0
    for g0 in range(0, 20_000, 32):
        g1 = min(20_000, g0 + 32)
        eng.calculate_site_proba(rp[off[g0]:off[g1]], off[g0:g1 + 1] - off[g0], 1000, batch_size=g1 - g0)
        n_calls += 1
    t_pool = time.perf_counter() - t0
    out["reference_loop_through_the_stub"] = {"sites": 20_000, "encode_calls_of_16_sites_s": t_enc, "pool_calls_of_32_sites_s": t_pool,
                                              "sites_per_s": 20_000 / (t_enc + t_pool), "us_per_pool_call": t_pool / n_calls * 1e6,
                                              "note": "per-batch / per-flush-group calls with host arrays, as INTEGRATION.md section 2 wires them"}
    # the same loop STREAMED (m6a_job_begin / feed / end, INTEGRATION.md section 2's recommended wiring): the batches are
    # fed as a DataLoader would produce them, the GPU works behind the loop, one pooling at the end
    import ctypes as C
    X, km = d["X"], d["site_kmers"]
    batches = [(np.ascontiguousarray(X[off[s0]:off[min(20_000, s0 + 16)]]), np.ascontiguousarray(km[s0:s0 + 16]),
                np.ascontiguousarray(off[s0:min(20_000, s0 + 16) + 1] - off[s0])) for s0 in range(0, 20_000, 16)]
    eng.prepare_host_io()
    want = eng.infer(X, km, off, 1000)
    streamed = {}
    for label in ("engine.job_feed (numpy batches)", "raw ctypes calls (pointers prepared)"):
        best = None
        for _ in range(4):
            t0 = time.perf_counter()
            eng.job_begin(1000)
            if label.startswith("engine"):
                for bx, bk, bo in batches:
                    eng.job_feed(bx, bk, bo)
            else:
                L, h = eng._L, eng._h
                for bx, bk, bo in batches:
                    L.m6a_job_feed(h, bx.ctypes.data, bk.ctypes.data, bo.ctypes.data, bo.size - 1)
            t_feed = time.perf_counter() - t0
            got = eng.job_end()
            t_all = time.perf_counter() - t0
            assert all(np.array_equal(a, b) for a, b in zip(got, want))
            if best is None or t_all < best[1]:
                best = (t_feed, t_all)
        streamed[label] = {"feed_loop_s": best[0], "begin_to_end_s": best[1], "sites_per_s": 20_000 / best[1],
                           "us_per_feed_call": best[0] / len(batches) * 1e6}
    # ... and through INTEGRATION.md's stub exactly as a reference-side caller would hold the batches (torch tensors,
    # k-mers per READ as inference_collate builds them)
    import re
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    stub = [b for b in re.findall(r"```python\n(.*?)```", text, flags=re.S) if "class HipModel" in b][0]
    os.environ["M6A_HIP_LIB"] = eng._L._name
    ns = {}
    exec(compile(stub, "hip_backend.py", "exec"), ns)
    from m6anet_amd.engine import load_weights as _lw
    w = _lw()
    shapes = [(66, 2), (150, 15), (150,), (150,), (150,), (150,), (150,), (32, 150), (32,), (1, 32), (1,)]
    sd, at = {}, 0
    for k, shp in zip(ns["_ORDER"], shapes):
        n = int(np.prod(shp))
        sd[k] = torch.from_numpy(w[at:at + n].reshape(shp).copy())
        at += n
    model = ns["HipModel"](sd)
    eng.set_encoder_variant(1)                      # the stub selects the 16-slot encoder (the reference's bits): compare like with like
    want = eng.infer(X, km, off, 1000)
    eng.set_encoder_variant(0)
    tb = [(torch.from_numpy(bx), torch.from_numpy(np.repeat(bk.astype(np.int64), np.diff(bo), axis=0)), torch.from_numpy(np.diff(bo)))
          for bx, bk, bo in batches]
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        model.begin(1000, 0.033379376, 0, 16, 2)
        for f, k, n in tb:
            model.feed(f, k, n)
        t_feed = time.perf_counter() - t0
        got = model.end()
        t_all = time.perf_counter() - t0
        assert all(np.array_equal(a, b) for a, b in zip(got, want))
        if best is None or t_all < best[1]:
            best = (t_feed, t_all)
    streamed["INTEGRATION.md stub (torch batches as the reference's collate builds them)"] = {
        "feed_loop_s": best[0], "begin_to_end_s": best[1], "sites_per_s": 20_000 / best[1], "us_per_feed_call": best[0] / len(tb) * 1e6}
    t0 = time.perf_counter()
    eng.infer(X, km, off, 1000)
    streamed["one m6a_infer over the whole job (host arrays)"] = {"sites_per_s": 20_000 / (time.perf_counter() - t0)}
    out["reference_loop_streamed"] = dict(streamed, sites=20_000, reads=int(off[-1]), batches=len(batches),
                                          note="16-site batches, host arrays, T = 1000; results bit-identical to one m6a_infer")
    # validation-style forward: 5 passes over 200 k ragged sites, device tensors
    eng = M6ANetEngine(weights=load_weights("HEK293T_RNA004"))
    d = synthetic.make_sites(200_000, (50, 500), seed=1)
    X, km, off = (torch.from_numpy(d[k]).to(dev) for k in ("X", "site_kmers", "off"))
    eng.validate_forward(X, km, off, n_iterations=5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.validate_forward(X, km, off, n_iterations=5)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["validate_200k_ragged_5_passes"] = {"sites": 200_000, "reads": int(d["off"][-1]), "passes": 5, "s_per_call": dt,
                                            "site_passes_per_s": 1e6 / dt,
                                            "note": "host sampler (sequential shuffle of every bag, every pass) dominates"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
