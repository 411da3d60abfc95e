import os, sys, subprocess, json, numpy as np
REPO=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0, REPO)
if len(sys.argv) > 1:
    from m6anet_amd import synthetic
    from m6anet_amd.engine import M6ANetEngine
    from m6anet_amd.constants import asset_path
    out = {}
    for tag, bag, S in (("u", 20, 50000), ("r", (50, 500), 4000)):
        d = synthetic.make_sites(S, bag, seed=7)
        for name in ("hct116", "arabidopsis", "hek293t_glori", "hek293t_m6ace"):
            e = M6ANetEngine(weights=np.fromfile(asset_path("weights_%s.bin" % name), np.float32))
            for mode in (1, 2):
                e.set_encoder_variant(mode)
                out["%s_%s_%d" % (tag, name, mode)] = e.get_read_probability(d["X"], d["site_kmers"], d["off"])
            e.close()
    np.savez(sys.argv[1], **out)
else:
    subprocess.check_call([sys.executable, __file__, "/tmp/a.npz"])
    subprocess.check_call([sys.executable, __file__, "/tmp/b.npz"], env=dict(os.environ, M6A_HIP_LIB=os.path.join(REPO, "tools/ko/libm6a_addabs.so")))
    a, b = np.load("/tmp/a.npz"), np.load("/tmp/b.npz")
    for k in a.files:
        print(k, "bit-identical" if np.array_equal(a[k], b[k]) else "DIFFERENT %d max %g" % ((a[k] != b[k]).sum(), np.abs(a[k] - b[k]).max()))
