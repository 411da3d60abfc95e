#!/usr/bin/env python3
"""Throughput of the native loader / CSV writers (libm6a_io.so) against the Python mirror, on the
reference's bundled data.json replicated N times (unique transcript ids).  Host-only.
Reference figure for comparison: NanopolishDS.__getitem__ = 0.52 ms/site."""
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m6anet_amd import data_utils  # noqa: E402

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_tests_data")


def replicate(n, out):
    info = open(os.path.join(SRC, "data.info")).read().splitlines()[1:]
    blob = open(os.path.join(SRC, "data.json"), "rb").read()
    with open(os.path.join(out, "data.json"), "wb") as fj, open(os.path.join(out, "data.info"), "w") as fi:
        fi.write("transcript_id,transcript_position,start,end,n_reads\n")
        pos = 0
        for k in range(n):
            for row in info:
                tx, p, a, b, nr = row.split(",")
                rec = blob[int(a):int(b)]
                new_tx = "%s_c%d" % (tx, k)
                rec = rec.replace(('"%s"' % tx).encode(), ('"%s"' % new_tx).encode(), 1)
                fj.write(rec)
                fi.write("%s,%s,%d,%d,%s\n" % (new_tx, p, pos, pos + len(rec), nr))
                pos += len(rec)
    return pos


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    with tempfile.TemporaryDirectory() as d:
        size = replicate(n, d)
        out = {"copies": n, "json_MB": size / 1e6, "host_threads": os.cpu_count()}
        for threads in (1, 0):
            t0 = time.perf_counter()
            b = data_utils.load_sites_native([d], 20, "norm_hct116.npz", n_threads=threads)
            dt = time.perf_counter() - t0
            key = "native_load_%s" % ("1thread" if threads == 1 else "all_threads")
            out[key] = {"s": dt, "sites": b.n_sites, "reads": int(b.off[-1]), "ms_per_site": dt * 1e3 / b.n_sites,
                        "MB_per_s": size / 1e6 / dt}
            if threads == 1:
                b.native.close()
        rp = np.random.default_rng(0).random(int(b.off[-1]), dtype=np.float32)
        sp = np.random.default_rng(1).random(b.n_sites, dtype=np.float32)
        mr = np.random.default_rng(2).random(b.n_sites)
        for threads in (1, 0):
            t0 = time.perf_counter()
            b.native.write_csv(d, rp, sp, mr, write_header=True, n_threads=threads)
            dt = time.perf_counter() - t0
            out["native_csv_%s" % ("1thread" if threads == 1 else "all_threads")] = {
                "s": dt, "rows_per_s": (int(b.off[-1]) + b.n_sites) / dt}
        # python mirror on one copy's worth of sites
        with tempfile.TemporaryDirectory() as d1:
            replicate(1, d1)
            t0 = time.perf_counter()
            p = data_utils.load_sites([d1], 20, "norm_hct116.npz")
            dt = time.perf_counter() - t0
            out["python_load"] = {"s": dt, "sites": p.n_sites, "ms_per_site": dt * 1e3 / p.n_sites}
        out["reference_ms_per_site"] = 0.52
        print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
