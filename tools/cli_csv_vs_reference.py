#!/usr/bin/env python3
"""GPU box: `m6anet_amd inference` on the reference's bundled dataset (BASELINE.json configs[0]: num_iterations = 5) with each
encoder, against the exact CSVs the reference wrote (tests/golden/config1_*.csv, captured by tests/golden/make_golden.py):
how many LINES of data.site_proba.csv / data.indiv_proba.csv are byte-identical.
    python tools/cli_csv_vs_reference.py"""
import os, sys, gzip, tempfile
sys.path.insert(0, os.getcwd())
from m6anet_amd.__main__ import main
gold = os.path.join(os.getcwd(), "tests", "golden")
data = os.path.join(gold, "ref_tests_data")
for enc in ("reference", "fast"):
    out = tempfile.mkdtemp()
    os.environ.pop("M6A_ENCODER", None)
    main(["inference", "--input_dir", data, "--out_dir", out, "--n_processes", "1", "--num_iterations", "5", "--encoder", enc])
    ours = open(os.path.join(out, "data.indiv_proba.csv")).read().splitlines()
    ref = gzip.open(os.path.join(gold, "config1_indiv_proba.csv.gz"), "rt").read().splitlines()
    same = sum(a == b for a, b in zip(ours, ref))
    s_ours = open(os.path.join(out, "data.site_proba.csv")).read().splitlines()
    s_ref = open(os.path.join(gold, "config1_site_proba.csv")).read().splitlines()
    s_same = sum(a == b for a, b in zip(s_ours, s_ref))
    print(enc, "indiv lines byte-identical: %d of %d;" % (same, len(ref)), "site lines byte-identical: %d of %d" % (s_same, len(s_ref)), flush=True)
    for a, b in list(zip(s_ours, s_ref)):
        if a != b and enc == "reference":
            print("   ours:", a); print("   ref :", b); break
