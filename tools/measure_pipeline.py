#!/usr/bin/env python3
"""eventalign.txt -> dataprep -> inference, end to end, as fresh processes, on the bundled fixture replicated N times (N = 1400:
2.9 GB of eventalign.txt, 141 400 sites / 7.8 M reads -- the size of the dataset the reference publishes 408 s of inference for,
README.md:206,245-249; its dataprep of ONE copy takes 35.7 s in the build container).   python tools/measure_pipeline.py [N]"""
import gzip
import json
import os
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "tests", "golden", "ref_tests_data", "eventalign.txt.gz")


def run(argv):
    t0 = time.perf_counter()
    subprocess.run([sys.executable, "-m", "m6anet_amd"] + argv, check=True, cwd=REPO)
    return time.perf_counter() - t0


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1400
    text = gzip.open(SRC, "rt").read()
    header, body = text.split("\n", 1)
    with tempfile.TemporaryDirectory(dir=os.environ.get("M6A_MEASURE_TMP")) as d:
        ev = os.path.join(d, "eventalign.txt")
        with open(ev, "w", buffering=16 << 20) as f:
            f.write(header + "\n")
            for k in range(n):
                f.write(body.replace("ENST", "C%dENST" % k) if k else body)
        prep, out, store = os.path.join(d, "prep"), os.path.join(d, "out"), os.path.join(d, "data.m6astore")
        res = {"copies": n, "eventalign_GB": os.path.getsize(ev) / 1e9}
        res["dataprep_s"] = run(["dataprep", "--eventalign", ev, "--out_dir", prep, "--n_processes", "0"])
        res["data_json_GB"] = os.path.getsize(os.path.join(prep, "data.json")) / 1e9
        res["inference_from_json_s"] = run(["inference", "--input_dir", prep, "--out_dir", out, "--num_iterations", "1000", "--n_processes", "0"])
        sites = sum(1 for _ in open(os.path.join(out, "data.site_proba.csv"))) - 1
        reads = sum(1 for _ in open(os.path.join(out, "data.indiv_proba.csv"))) - 1
        res.update({"sites": sites, "reads": reads, "site_csv_MB": os.path.getsize(os.path.join(out, "data.site_proba.csv")) / 1e6,
                    "indiv_csv_MB": os.path.getsize(os.path.join(out, "data.indiv_proba.csv")) / 1e6})
        res["pack_s"] = run(["pack", "--input_dir", prep, "--out", store])
        res["inference_from_store_s"] = run(["inference", "--input_dir", store, "--out_dir", os.path.join(d, "out2"), "--num_iterations", "1000", "--n_processes", "0"])
        same = all(open(os.path.join(out, fn), "rb").read() == open(os.path.join(d, "out2", fn), "rb").read()
                   for fn in ("data.site_proba.csv", "data.indiv_proba.csv"))
        res["store_run_writes_the_same_bytes"] = same
        res["eventalign_to_csv_s"] = res["dataprep_s"] + res["inference_from_json_s"]
        res["reference"] = "dataprep 35.7 s per 2.09 MB copy here (single process) => ~14 h for this file; inference 408 s published for a dataset this size"
        print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
