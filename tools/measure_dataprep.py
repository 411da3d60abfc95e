#!/usr/bin/env python3
"""Throughput of the native dataprep on the reference's bundled eventalign.txt replicated N times
(distinct transcript ids per copy).  Reference figure in the build container: 1.3 s index + 34.4 s
preprocess for ONE copy (2.09 MB)."""
import gzip
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m6anet_amd import _io  # noqa: E402

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_tests_data",
                   "eventalign.txt.gz")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    text = gzip.open(SRC, "rt").read()
    header, body = text.split("\n", 1)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "eventalign.txt")
        with open(path, "w") as f:
            f.write(header + "\n")
            for k in range(n):
                f.write(body.replace("ENST", "C%dENST" % k) if k else body)
        size = os.path.getsize(path)
        res = {"copies": n, "eventalign_MB": size / 1e6, "host_threads": os.cpu_count()}
        for threads in (1, 0):
            t0 = time.perf_counter()
            _io.dataprep(path, os.path.join(d, "out%d" % threads), n_threads=threads, readcount_min=1,
                         readcount_max=1000, min_segment_count=20)
            dt = time.perf_counter() - t0
            sites = len(open(os.path.join(d, "out%d" % threads, "data.info")).read().splitlines()) - 1
            res["threads_%s" % ("1" if threads == 1 else "all")] = {"s": dt, "MB_per_s": size / 1e6 / dt, "sites": sites}
        res["reference_here"] = "35.7 s for one copy (2.09 MB) = 0.06 MB/s"
        print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
