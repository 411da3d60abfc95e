#!/usr/bin/env python3
"""Where does the first call's time go?  (a one-shot job is ONE call on a fresh context.)
For both bench workloads: time of m6a_create's background set-up (create -> first entry point returns), then the wall
time and the HIP-event durations of encoder / pooling for calls 1..4 on a fresh context -- once with the GPU idle for
two seconds before call 1, once right after a busy-spin of unrelated GPU work (clocks up)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from m6anet_amd import synthetic  # noqa: E402
from m6anet_amd.engine import M6ANetEngine, load_weights  # noqa: E402

dev = torch.device("cuda:0")
out = {}
for tag, model, S, bag in (("uniform", "HCT116_RNA002", 1_000_000, 20), ("ragged", "HEK293T_RNA004", 125_000, (50, 500))):
    d = synthetic.make_sites(S, bag, seed=20250328)
    X, km, off = (torch.from_numpy(d[k]).to(dev) for k in ("X", "site_kmers", "off"))
    R = int(d["off"][-1])
    bufs = (torch.empty(R, dtype=torch.float32, device=dev), torch.empty(S, dtype=torch.float32, device=dev),
            torch.empty(S, dtype=torch.float64, device=dev))
    spin = torch.randn(4096, 4096, device=dev)
    for mode in ("idle_2s_before", "idle_then_tiny_kernel", "h2d_before", "busy_before"):
        for warm in ("1", "0"):
            os.environ["M6A_WARMUP"] = warm
            t0 = time.perf_counter()
            eng = M6ANetEngine(weights=load_weights(model))
            t_create = time.perf_counter() - t0
            eng.use_torch_stream()
            t_settle = time.perf_counter() - t0
            torch.cuda.synchronize()
            wake_ms = None
            if mode == "idle_2s_before":
                time.sleep(2.0)
            elif mode == "idle_then_tiny_kernel":           # how long does the first kernel after idle take by itself?
                time.sleep(2.0)
                t1 = time.perf_counter()
                spin[:64, :64].add_(1.0)
                torch.cuda.synchronize()
                wake_ms = (time.perf_counter() - t1) * 1e3
            elif mode == "h2d_before":                       # what bench.py and the CLI do: the features cross PCIe, then the call
                time.sleep(2.0)
                X2 = torch.from_numpy(d["X"]).to(dev)
                torch.cuda.synchronize()
                del X2
            else:
                t1 = time.perf_counter()
                while time.perf_counter() - t1 < 0.5:
                    for _ in range(20):
                        spin = torch.tanh(spin @ spin * 1e-3)
                    torch.cuda.synchronize()
            calls = []
            for i in range(4):
                eng.profile(True)
                eng.set_host_offsets(d["off"])
                t0 = time.perf_counter()
                eng.infer(X, km, off, 1000, out=bufs)
                ret = (time.perf_counter() - t0) * 1e3
                torch.cuda.synchronize()
                wall = (time.perf_counter() - t0) * 1e3
                e, en = eng.profile_read(0)
                p, pn = eng.profile_read(1)
                calls.append({"wall_ms": round(wall, 3), "call_returns_ms": round(ret, 3), "enc_ms": round(e, 3), "pool_ms": round(p, 3)})
            out["%s/%s/warmup=%s" % (tag, mode, warm)] = {"create_ms": round(t_create * 1e3, 2), "create_to_first_entry_ms": round(t_settle * 1e3, 2),
                                                          "tiny_kernel_after_idle_ms": wake_ms, "calls": calls}
            eng.close()
print(json.dumps(out, indent=1))
