cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, time, subprocess, tempfile, json
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import measure_io
from m6anet_amd import data_utils
d = tempfile.mkdtemp()
measure_io.replicate(1400, d)
store = os.path.join(d, "data.m6astore")
data_utils.pack_sites([d], store, 20, "norm_hct116.npz")
res = {}
# one GPU box: the ranks share the GPU (M6A_SHARE_GPU=1); the default performs no device exchange, `_host` adds the debugging gather
for tag, extra, env in (("gpus1", [], {}), ("gpus2", ["--gpus", "2"], {"M6A_SHARE_GPU": "1"}), ("gpus4", ["--gpus", "4"], {"M6A_SHARE_GPU": "1"}),
                        ("gpus8", ["--gpus", "8"], {"M6A_SHARE_GPU": "1"}), ("gpus8_host", ["--gpus", "8"], {"M6A_EXCHANGE": "host"})):
    ts = []
    for rep in range(4):
        t0 = time.perf_counter()
        subprocess.run([sys.executable, "-m", "m6anet_amd", "inference", "--input_dir", store, "--out_dir", os.path.join(d, "o" + tag), "--num_iterations", "1000", "--n_processes", "0"] + extra, check=True, env=dict(os.environ, **env))
        ts.append(time.perf_counter() - t0)
    res[tag] = {"best_s": min(ts), "all_s": ts}
a = open(os.path.join(d, "ogpus1", "data.indiv_proba.csv"), "rb").read()
res["bytes_equal"] = all(open(os.path.join(d, "o" + t, "data.indiv_proba.csv"), "rb").read() == a for t in ("gpus2", "gpus4", "gpus8", "gpus8_host"))
print(json.dumps(res))
PY
