#!/usr/bin/env python3
"""Is the encoder power/clock-limited?  Samples rocm-smi (sclk, power) while the encoder runs back to back."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from m6anet_amd import synthetic
from m6anet_amd.engine import M6ANetEngine, load_weights
dev = torch.device("cuda:0")
e = M6ANetEngine(weights=load_weights("HCT116_RNA002"))
d = synthetic.make_sites(1_000_000, 20, seed=1)
X, km, off = (torch.from_numpy(d[k]).to(dev) for k in ("X", "site_kmers", "off"))
e.use_torch_stream()
samples = []
stop = False
def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=10).stdout
            samples.append(o.strip().splitlines()[-1][:300])
        except Exception as ex:
            samples.append(repr(ex))
        time.sleep(0.3)
def run(label, fn, secs=4.0):
    global stop, samples
    samples, stop = [], False
    th = threading.Thread(target=sampler); th.start()
    t0 = time.time(); n = 0
    e.profile(True)
    while time.time() - t0 < secs:
        for _ in range(20): fn()
        e.sync(); n += 20
    ms, k = e.profile_read(0); mp, kp = e.profile_read(1); e.profile(False)
    stop = True; th.join()
    print(label, "enc avg ms %.3f" % (ms / max(k, 1)), "pool avg ms %.3f" % (mp / max(kp, 1)))
    for s in samples[1::3][:5]: print("   ", s)
print(subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True).stdout.strip().splitlines()[0][:300])
run("idle->encoder only", lambda: e.get_read_probability(X, km, off))
rp = e.get_read_probability(X, km, off)
run("pool only", lambda: e.calculate_site_proba(rp, off, 1000))
