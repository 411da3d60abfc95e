#!/usr/bin/env python3
"""Time of m6a_random_stream (the NumPy MT19937 stream on the GPU) for a ladder of lengths, segmented generator vs the
single 623-words-per-step chain (M6A_MT_SEGMENTS=0).  usage: stream_probe.py  (re-executes itself for the second mode)"""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure():
    import torch
    from m6anet_amd.engine import M6ANetEngine
    e = M6ANetEngine()
    out = {}
    for n in (1_328_128, 2_097_152, 8_000_000, 33_000_000, 380_000_000):
        buf = torch.empty(n, dtype=torch.int32, device="cuda")
        ts = []
        for _ in range(4):
            torch.cuda.synchronize()
            t = time.perf_counter()
            e._chk(e._L.m6a_random_stream(e._h, 5, n, buf.data_ptr()))
            e.sync()
            ts.append(time.perf_counter() - t)
        out[str(n)] = round(min(ts) * 1e3, 3)
        del buf
    return out


if __name__ == "__main__":
    if "--child" in sys.argv:
        print(json.dumps(measure()))
    else:
        res = {}
        for mode, env in (("segmented", "1"), ("single_chain", "0")):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], capture_output=True, text=True,
                               env=dict(os.environ, M6A_MT_SEGMENTS=env))
            res[mode + "_ms"] = json.loads(r.stdout.strip().splitlines()[-1])
        res["note"] = "ms per stream of n 32-bit words, best of 4, host-timed around one m6a_random_stream call + sync (device buffer)"
        print(json.dumps(res, indent=1))
