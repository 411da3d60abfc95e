#!/usr/bin/env python3
"""Throughput and memory of the native dataprep on the reference's bundled eventalign.txt replicated to a target size
(distinct transcript ids per copy).  Reference figure in the build container: 1.3 s index + 34.4 s preprocess for ONE
copy (2.09 MB) = 0.06 MB/s.

    python tools/measure_dataprep.py [GB=1.0] [--single]     # --single also times one thread on (at most) the first 0.25 GB

Prints one JSON object: file size, all-thread seconds and GB/s (index and transcript phases from M6A_IO_TRACE), sites
written, bytes of data.json, the process's peak ANONYMOUS memory during the call (RssAnon sampled from /proc/self/status:
the mapped file's pages are page cache, not the process's memory) and the CPUs the process may use."""
import gzip
import json
import os
import subprocess
import sys
import tempfile
import threading
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
SRC = os.path.join(REPO, "tests", "golden", "ref_tests_data", "eventalign.txt.gz")


def rss_anon_kb():
    for line in open("/proc/self/status"):
        if line.startswith("RssAnon:"):
            return int(line.split()[1])
    return 0


def run_one(path, out, threads):
    """In a fresh process (so the peak is this call's): returns dict(s, peak_anon_MB, phases)."""
    code = r"""
import json, os, sys, threading, time
sys.path.insert(0, %r)
from m6anet_amd import _io
def anon():
    for line in open('/proc/self/status'):
        if line.startswith('RssAnon:'):
            return int(line.split()[1])
    return 0
peak, stop = [anon()], [False]
def sample():
    while not stop[0]:
        peak[0] = max(peak[0], anon()); time.sleep(0.02)
t = threading.Thread(target=sample); t.start()
base = anon()
t0 = time.perf_counter()
_io.dataprep(%r, %r, n_threads=%d, readcount_min=1, readcount_max=1000, min_segment_count=20)
dt = time.perf_counter() - t0
stop[0] = True; t.join()
print(json.dumps({'s': dt, 'peak_anon_MB': peak[0] / 1024.0, 'anon_before_MB': base / 1024.0}))
""" % (REPO, path, out, threads)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, M6A_IO_TRACE="1"))
    if r.returncode != 0:
        raise SystemExit(r.stderr[-2000:])
    res = json.loads(r.stdout.strip().splitlines()[-1])
    phases = {}
    for line in r.stderr.splitlines():
        if line.startswith("m6a_io: dataprep"):
            body = line[len("m6a_io: "):]
            if body.rstrip().endswith("ms"):
                name, ms = body.rsplit(None, 2)[0], float(body.rsplit(None, 2)[1])
                phases[name.strip()] = ms / 1e3
            else:
                phases["note"] = body
    res["phases_s"] = phases
    return res


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    gb = float(args[0]) if args else 1.0
    text = gzip.open(SRC, "rt").read()
    header, body = text.split("\n", 1)
    n = max(1, int(gb * 1e9 / len(body)))
    from bench import host_cpu_facts
    with tempfile.TemporaryDirectory(dir=os.environ.get("M6A_MEASURE_TMP")) as d:
        path = os.path.join(d, "eventalign.txt")
        t0 = time.perf_counter()
        with open(path, "w", buffering=16 << 20) as f:
            f.write(header + "\n")
            for k in range(n):
                f.write(body.replace("ENST", "C%dENST" % k) if k else body)
        size = os.path.getsize(path)
        res = {"copies": n, "eventalign_GB": size / 1e9, "file_written_in_s": time.perf_counter() - t0, "host": host_cpu_facts()}
        r = run_one(path, os.path.join(d, "out_all"), 0)
        info = os.path.join(d, "out_all", "data.info")
        r.update({"GB_per_s": size / 1e9 / r["s"], "sites": sum(1 for _ in open(info)) - 1,
                  "data_json_GB": os.path.getsize(os.path.join(d, "out_all", "data.json")) / 1e9,
                  "index_MB": os.path.getsize(os.path.join(d, "out_all", "eventalign.index")) / 1e6})
        res["threads_all"] = r
        if "--single" in sys.argv:
            small = os.path.join(d, "small.txt")
            with open(path, "rb") as f, open(small, "wb") as g:
                blob = f.read(min(size, 250_000_000))
                g.write(blob[:blob.rfind(b"\n") + 1])
            s1 = os.path.getsize(small)
            r1 = run_one(small, os.path.join(d, "out_1"), 1)
            r1.update({"GB": s1 / 1e9, "GB_per_s": s1 / 1e9 / r1["s"]})
            res["threads_1"] = r1
        res["reference_here"] = "35.7 s for one copy (2.09 MB) = 0.00006 GB/s"
        print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
