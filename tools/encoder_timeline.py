#!/usr/bin/env python3
"""Where a tile's wall time goes inside enc_site16_kernel, from the kernel's own clock: a diagnostic build (-DM6A_AB_STAMPS, never the
product) stamps s_memtime at ten points of 16 consecutive tiles of EVERY wave, with the wave's hardware slot, so the two waves that
share a SIMD can be laid side by side.

    python tools/encoder_timeline.py --build     (here: cross-compiles tools/ko/libm6a_ab_stamps.so, which travels to the GPU box)
    python tools/encoder_timeline.py             (GPU box: one JSON object)

Sections of a tile (stamp i -> i+1) and the MFMAs this wave issues in each:
  0-1  link0 (scalar loads of off[]) + layer 1 of unit tile 0                      8
  1-2  layer 1 of unit tile 1, link1 (x loads), batch norm + layer 2 of unit tile 0 24
  2-3  layer 1 of unit tile 2, link2, batch norm + layer 2 of unit tile 1          24
  3-4  layer 1 of unit tile 3, batch norm + layer 2 of unit tile 2                 24
  4-5  layer 1 of unit tile 4, link3 (ds_bpermute), batch norm + layer 2 of unit tile 3   24
  5-6  batch norm + layer 2 of unit tile 4 (12 hidden units)                       12
  6-7  32 -> 1 layer (relu, products, butterfly, two exchanges between the lane halves)   0
  7-8  sigmoid (Sleef expf, IEEE divide) + the store                                0
  8-9  site base for the next tile (shuffle + readfirstlane), feature registers handed over   0
  9-0' loop back (in the stamped tiles also: lane 0 stores the ten stamps)          0
With two waves per SIMD taking turns on the matrix pipe, a section with n MFMAs lasts >= 128 n cycles while the partner is in its body too.
"""
import ctypes
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
KO = os.path.join(REPO, "tools", "ko")
VARIANTS = {"stamps": [], "stamps_noprio": ["-DM6A_AB_NO_PRIO"], "stamps_body3_epi0": ["-DM6A_AB_PRIO_BODY=3", "-DM6A_AB_PRIO_EPI=0"],
            "stamps_phase": ["-DM6A_AB_PHASE=1"], "stamps_phase_noprio": ["-DM6A_AB_PHASE=1", "-DM6A_AB_NO_PRIO"],
            "stamps_phase_body3_epi0": ["-DM6A_AB_PHASE=1", "-DM6A_AB_PRIO_BODY=3", "-DM6A_AB_PRIO_EPI=0"],
            "stamps_phase_eqprio": ["-DM6A_AB_PHASE=1", "-DM6A_AB_PRIO_BODY=1", "-DM6A_AB_PRIO_EPI=1"]}


def lib(name):
    return os.path.join(KO, "libm6a_ab_%s.so" % name)
TILES, POINTS, FIRST, MAXW = 16, 10, 100, 4096
MFMA = [8, 24, 24, 24, 24, 12, 0, 0, 0, 0]
NAMES = ["link0 + L1(0)", "L1(1) + link1 + BN/L2(0)", "L1(2) + link2 + BN/L2(1)", "L1(3) + BN/L2(2)", "L1(4) + link3 + BN/L2(3)", "BN/L2(4)",
         "32->1 layer", "sigmoid + store", "site base + hand-over", "loop back (+ stamp stores)"]


def build():
    from m6anet_amd import build as B
    os.makedirs(KO, exist_ok=True)
    jump = os.path.join(B.PKG, "assets", "mt19937_jump.bin")
    for name, flags in VARIANTS.items():
        cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               '-DM6A_MT_JUMP_PATH="%s"' % jump, "-DM6A_AB_STAMPS", "-I" + B.INCLUDE, "-I" + B.CSRC] + flags + [os.path.join(B.CSRC, s) for s in B.SOURCES] + ["-o", lib(name)]
        subprocess.check_call(cmd)
        print("built", lib(name))


def med(v):
    v = sorted(v)
    return v[len(v) // 2] if v else None


def main():
    if "--build" in sys.argv:
        build()
        return
    if "--one" not in sys.argv:
        out = {}
        for name in VARIANTS:
            if not os.path.exists(lib(name)):
                continue
            o = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=dict(os.environ, M6A_HIP_LIB=lib(name)), capture_output=True, text=True, timeout=600)
            try:
                out[name] = json.loads(o.stdout.strip().splitlines()[-1])
                out[name]["build_flags"] = ["-DM6A_AB_STAMPS"] + VARIANTS[name]
            except (ValueError, IndexError):
                out[name] = {"error": (o.stderr or o.stdout)[-400:]}
        print(json.dumps(out))
        return
    LIB = os.environ["M6A_HIP_LIB"]
    import numpy as np
    import torch
    from m6anet_amd import synthetic
    from m6anet_amd.engine import M6ANetEngine, load_weights
    eng = M6ANetEngine(weights=load_weights("HCT116_RNA002"))
    d = synthetic.make_sites(1_000_000, 20, seed=20250328)
    X, km, off = (torch.from_numpy(d[k]).cuda() for k in ("X", "site_kmers", "off"))
    rp = torch.empty(int(d["off"][-1]), dtype=torch.float32, device="cuda")
    eng.set_encoder_variant(0)
    for _ in range(4):
        eng.get_read_probability(X, km, off, out=rp)
    eng.sync()
    eng.profile("encoder")
    for _ in range(20):
        eng.get_read_probability(X, km, off, out=rp)
    ms, n = eng.profile_read(0)
    eng.profile(False)
    assert eng.last_encoder_kernel == "enc_site16_kernel"
    dll = ctypes.CDLL(LIB)
    st = np.zeros((MAXW, TILES, POINTS), dtype=np.uint64)
    hw = np.zeros((MAXW, 2), dtype=np.uint32)
    rc = dll.m6a_ab_read_stamps(st.ctypes.data_as(ctypes.c_void_p), hw.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    live = np.nonzero(st[:, 0, 0])[0]
    st = st.astype(np.int64)
    # sections
    sec = []
    for i in range(POINTS):
        if i < POINTS - 1:
            dur = (st[live, :, i + 1] - st[live, :, i]).ravel()
        else:
            dur = (st[live, 1:, 0] - st[live, :-1, POINTS - 1]).ravel()
        sec.append({"section": "%d-%s" % (i, i + 1 if i < POINTS - 1 else "0'"), "what": NAMES[i], "mfma": MFMA[i], "median_cycles": float(np.median(dur)),
                    "mean_cycles": float(dur.mean()), "p10": float(np.percentile(dur, 10)), "p90": float(np.percentile(dur, 90)),
                    "cycles_per_own_mfma": float(np.median(dur)) / MFMA[i] if MFMA[i] else None})
    tile = (st[live, 1:, 0] - st[live, :-1, 0]).ravel()
    # pairs: waves in the same (xcc, se, sh, cu, simd)
    key = {}
    for w in live:
        h, x = int(hw[w, 0]), int(hw[w, 1]) & 0xf
        k = (x, (h >> 13) & 7, (h >> 12) & 1, (h >> 8) & 0xf, (h >> 4) & 3)
        key.setdefault(k, []).append(int(w))
    pairs = [v for v in key.values() if len(v) == 2]
    both_epi, a_epi_only, phase, idle_model = [], [], [], []
    for a, b in pairs:
        t0 = max(st[a, 0, 0], st[b, 0, 0])
        t1 = min(st[a, TILES - 1, 9], st[b, TILES - 1, 9])
        if t1 <= t0:
            continue
        # intervals in which a wave issues no MFMA: stamp 6 (end of the body) .. the next tile's stamp 0
        def gaps(w):
            g = [(st[w, k, 6], st[w, k + 1, 0]) for k in range(TILES - 1)]
            return [(max(s, t0), min(e, t1)) for s, e in g if min(e, t1) > max(s, t0)]
        ga, gb = gaps(a), gaps(b)
        la, lb = sum(e - s for s, e in ga), sum(e - s for s, e in gb)
        ov = sum(max(0, min(e1, e2) - max(s1, s2)) for s1, e1 in ga for s2, e2 in gb)
        span = float(t1 - t0)
        both_epi.append(ov / span)
        a_epi_only.append((la + lb - 2 * ov) / span)
        per = float(np.median(st[a, 1:, 0] - st[a, :-1, 0]))
        phase.append(float(((st[b, 8, 0] - st[a, 8, 0]) % per) / per))
    ex = None
    if pairs:
        a, b = pairs[len(pairs) // 2]
        t0 = int(min(st[a, 4, 0], st[b, 4, 0]))
        ex = {"waves": [a, b], "stamps_minus_t0": {"a": (st[a, 4:8] - t0).tolist(), "b": (st[b, 4:8] - t0).tolist()}}
    out = {"what": "enc_site16_kernel, 20 M reads (1 M sites x 20), s_memtime stamps of tiles %d..%d of every wave (diagnostic build -DM6A_AB_STAMPS)" % (FIRST, FIRST + TILES - 1),
           "kernel_ms": ms / n, "waves": int(len(live)), "tile_cycles_median": float(np.median(tile)), "tile_cycles_mean": float(tile.mean()),
           "mfma_cycles_per_tile_of_both_waves": 2 * 116 * 64,
           "matrix_pipe_busy_from_stamps": 2 * 116 * 64 / float(np.median(tile)),
           "sections": sec,
           "pairs": {"simd_slots_with_two_stamped_waves": len(pairs), "slots_seen": len(key),
                     "share_of_time_both_waves_outside_their_mfma_body": med(both_epi),
                     "share_of_time_exactly_one_wave_outside_its_body": med(a_epi_only),
                     "hw_wave_id_of_the_two": [[int(hw[a, 0]) & 0xf, int(hw[b, 0]) & 0xf] for a, b in pairs[:8]],
                     "phase_of_partner_tile_start_in_own_tile_deciles": [float(np.percentile(phase, q)) for q in range(0, 101, 10)] if phase else None},
           "example_pair": ex}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
