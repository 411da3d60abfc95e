#!/usr/bin/env python3
"""Rooflines for the rows either side of the hot path: dataprep, the loader, the site
store and the CSV writers, each as bytes/s ACHIEVED against a ceiling MEASURED on the same box in the same run
(tools/host_ceilings.c: a newline scan of the same file over an mmap, memcpy, pwrite -- at the same thread counts), plus the
1 -> N thread scaling of each, so a row reads as "parse-bound" (scales with threads, far under the scan) or "memory/I-O-bound"
(at the ceiling, flat).  Host-only; run it on the GPU box's host so the numbers are those of the box the bench line is from.

    python tools/host_rooflines.py [eventalign_GB=2.0] [json_copies=300]   ->  one JSON object (profiles/r06_host_rooflines.json)

Reference anchors: dataprep m6anet/utils/dataprep_utils.py:269-325,399-488; loader m6anet/utils/data_utils.py:169-190;
writers m6anet/utils/inference_utils.py:59-67."""
import gzip
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
EXE = os.path.join(REPO, "tools", "host_ceilings")


def ceilings(*args):
    out = subprocess.run([EXE] + [str(a) for a in args], capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        return {"error": (out.stderr or out.stdout)[-300:]}
    return json.loads(out.stdout.strip().splitlines()[-1])


def thread_ladder(cores):
    t, out = 1, []
    while t < cores:
        out.append(t)
        t *= 2
    return out + [cores]


def timed(fn, reps=1):
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return best, r


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    gb = float(args[0]) if args else 2.0
    copies = int(args[1]) if len(args) > 1 else 300
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "tools"), "host_ceilings"])
    from bench import host_cpu_facts
    from m6anet_amd import _io, data_utils
    import measure_io
    facts = host_cpu_facts()
    cores = facts["effective_cores"]
    ladder = thread_ladder(cores)
    res = {"host": facts, "threads_ladder": ladder, "rows": {}}
    src = os.path.join(REPO, "tests", "golden", "ref_tests_data", "eventalign.txt.gz")
    with tempfile.TemporaryDirectory(dir=os.environ.get("M6A_MEASURE_TMP")) as d:
        res["scratch_dir"] = d
        res["scratch_fs"] = subprocess.run(["df", "-T", d], capture_output=True, text=True).stdout.splitlines()[-1].split()[1]
        # ---- machine ceilings that need no input file
        res["memcpy_GBps"] = {str(t): ceilings("memcpy", 2048, t).get("GBps_copied") for t in ladder}
        res["pwrite_GBps"] = {str(t): ceilings("pwrite", os.path.join(d, "pw.bin"), 1024, t).get("GBps_written") for t in ladder}

        # ---- dataprep: eventalign.txt -> eventalign.index + data.json + data.info
        text = gzip.open(src, "rt").read()
        header, body = text.split("\n", 1)
        n = max(1, int(gb * 1e9 / len(body)))
        ev = os.path.join(d, "eventalign.txt")
        with open(ev, "w", buffering=16 << 20) as f:
            f.write(header + "\n")
            for k in range(n):
                f.write(body.replace("ENST", "C%dENST" % k) if k else body)
        E = os.path.getsize(ev)
        scan = {str(t): ceilings("scan", ev, t) for t in ladder}
        runs = []
        for t in ladder:
            out = os.path.join(d, "dp_%d" % t)
            dt, _ = timed(lambda: _io.dataprep(ev, out, n_threads=t, readcount_min=1, readcount_max=1000, min_segment_count=20))
            written = sum(os.path.getsize(os.path.join(out, fn)) for fn in ("data.json", "data.info", "eventalign.index"))
            runs.append({"threads": t, "s": dt, "GBps_in": E / dt / 1e9, "GBps_in_plus_out": (E + written) / dt / 1e9, "bytes_out": written})
            if t != ladder[-1]:
                for fn in os.listdir(out):
                    os.remove(os.path.join(out, fn))
        for r in runs:
            r["efficiency_vs_1_thread"] = r["GBps_in"] / (runs[0]["GBps_in"] * r["threads"])
        top, sc = runs[-1], scan[str(cores)]
        res["rows"]["dataprep"] = {
            "bytes_in": E, "bytes_out": top["bytes_out"], "achieved_GBps_in": top["GBps_in"], "threads": cores,
            "ceiling": {"what": "memchr('\\n') over the same mmap'd file, same threads, page cache warm: the least a line-oriented parser does",
                        "GBps": sc.get("memchr_newline_GBps"), "memory_only_sum_GBps": sc.get("sum_words_GBps")},
            "frac_of_newline_scan": top["GBps_in"] / sc["memchr_newline_GBps"] if sc.get("memchr_newline_GBps") else None,
            "frac_of_memory_scan": top["GBps_in"] / sc["sum_words_GBps"] if sc.get("sum_words_GBps") else None,
            "scaling": runs, "newline_scan_by_threads": {t: s.get("memchr_newline_GBps") for t, s in scan.items()},
            "lines": sc.get("lines"), "ns_per_line_per_thread": top["s"] * cores / sc["lines"] * 1e9 if sc.get("lines") else None}
        dp_out = os.path.join(d, "dp_%d" % cores)

        # ---- loader: data.info + data.json -> flat arrays (parse, filter, float64 normalisation, float32 cast)
        jd = os.path.join(d, "json")
        os.makedirs(jd)
        J = measure_io.replicate(copies, jd)
        jscan = {str(t): ceilings("scan", os.path.join(jd, "data.json"), t) for t in ladder}
        runs, b = [], None
        for t in ladder:
            if b is not None:
                b.native.close()
            dt, b = timed(lambda: data_utils.load_sites_native([jd], 20, "norm_hct116.npz", n_threads=t))
            runs.append({"threads": t, "s": dt, "GBps_json": J / dt / 1e9, "sites": b.n_sites, "reads": int(b.off[-1])})
        for r in runs:
            r["efficiency_vs_1_thread"] = r["GBps_json"] / (runs[0]["GBps_json"] * r["threads"])
        top, sc = runs[-1], jscan[str(cores)]
        out_bytes = int(b.off[-1]) * (36 + 8 + 4) + b.n_sites * (3 + 8 + 8)
        res["rows"]["loader"] = {
            "bytes_in": J, "bytes_out_arrays": out_bytes, "achieved_GBps_json": top["GBps_json"], "threads": cores,
            "ceiling": {"what": "memchr('\\n') over data.json, same threads", "GBps": sc.get("memchr_newline_GBps"), "memory_only_sum_GBps": sc.get("sum_words_GBps")},
            "frac_of_newline_scan": top["GBps_json"] / sc["memchr_newline_GBps"] if sc.get("memchr_newline_GBps") else None,
            "numbers_parsed": int(b.off[-1]) * 10, "ns_per_number_per_thread": top["s"] * cores / (int(b.off[-1]) * 10) * 1e9,
            "scaling": runs}

        # ---- site store: pack once, map afterwards
        store = os.path.join(d, "job.m6astore")
        dt_pack, _ = timed(lambda: data_utils.pack_sites([jd], store, 20, "norm_hct116.npz", n_threads=cores))
        Sz = os.path.getsize(store)
        dt_open, sb = timed(lambda: data_utils.open_store(store, "norm_hct116.npz", 20))
        t0 = time.perf_counter()
        touched = float(np.asarray(sb.X).sum(dtype=np.float64))                  # first touch of every feature page (page cache -> mapping)
        dt_touch = time.perf_counter() - t0
        res["rows"]["site_store"] = {
            "store_bytes": Sz, "pack_s": dt_pack, "pack_GBps_json": J / dt_pack / 1e9, "open_s": dt_open,
            "first_touch_of_X_s": dt_touch, "first_touch_GBps": sb.X.nbytes / dt_touch / 1e9, "checksum": touched,
            "ceiling": {"what": "sum of 8-byte words over an mmap of a same-sized file, 1 thread (open maps, it copies nothing)",
                        "GBps": ceilings("scan", store, 1).get("sum_words_GBps")}}

        # ---- CSV writers: %.16f rows of every read and every site
        rp = np.random.default_rng(0).random(int(b.off[-1]), dtype=np.float32)
        sp = np.random.default_rng(1).random(b.n_sites, dtype=np.float32)
        mr = np.random.default_rng(2).random(b.n_sites)
        runs = []
        for t in ladder:
            od = os.path.join(d, "csv_%d" % t)
            os.makedirs(od)
            dt, _ = timed(lambda: b.native.write_csv(od, rp, sp, mr, write_header=True, n_threads=t))
            W = sum(os.path.getsize(os.path.join(od, fn)) for fn in ("data.site_proba.csv", "data.indiv_proba.csv"))
            runs.append({"threads": t, "s": dt, "GBps_written": W / dt / 1e9, "rows_per_s": (int(b.off[-1]) + b.n_sites) / dt, "bytes": W})
            for fn in os.listdir(od):
                os.remove(os.path.join(od, fn))
        for r in runs:
            r["efficiency_vs_1_thread"] = r["GBps_written"] / (runs[0]["GBps_written"] * r["threads"])
        top = runs[-1]
        pw = res["pwrite_GBps"][str(cores)]
        res["rows"]["csv_writers"] = {
            "bytes_out": top["bytes"], "achieved_GBps": top["GBps_written"], "threads": cores, "rows": int(b.off[-1]) + b.n_sites,
            "ceiling": {"what": "pwrite() of as many bytes from a warm buffer into a new file of the same directory, same threads", "GBps": pw},
            "frac_of_pwrite": top["GBps_written"] / pw if pw else None,
            "ns_per_row_per_thread": top["s"] * cores / (int(b.off[-1]) + b.n_sites) * 1e9, "scaling": runs}
        b.native.close()
        _ = dp_out
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
