import gzip, os, sys, time, tempfile, resource, subprocess, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
if len(sys.argv) > 2:
    from m6anet_amd import _io
    t0 = time.perf_counter()
    _io.dataprep(sys.argv[1], sys.argv[2], n_threads=int(sys.argv[3]), readcount_min=1, readcount_max=1000, min_segment_count=20)
    dt = time.perf_counter() - t0
    ru = resource.getrusage(resource.RUSAGE_SELF)
    print(json.dumps({"threads": int(sys.argv[3]), "s": dt, "user": ru.ru_utime, "sys": ru.ru_stime, "ctx_invol": ru.ru_nivcsw, "ctx_vol": ru.ru_nvcsw}))
else:
    SRC = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests/golden/ref_tests_data/eventalign.txt.gz")
    text = gzip.open(SRC, "rt").read(); header, body = text.split("\n", 1)
    d = tempfile.mkdtemp(dir="/dev/shm")
    path = os.path.join(d, "e.txt")
    with open(path, "w") as f:
        f.write(header + "\n")
        for k in range(1500): f.write(body.replace("ENST", "C%dENST" % k) if k else body)
    print("GB", os.path.getsize(path) / 1e9)
    for th in (1, 2, 4, 8, 16, 32):
        r = subprocess.run([sys.executable, __file__, path, os.path.join(d, "o%d" % th), str(th)], capture_output=True, text=True, env=dict(os.environ, M6A_IO_TRACE="1"))
        print(r.stdout.strip(), [l for l in r.stderr.splitlines() if "transcripts" in l or "index" in l])
    import shutil; shutil.rmtree(d)
