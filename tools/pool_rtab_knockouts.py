#!/usr/bin/env python3
"""Knock-out builds of pool_rtab_kernel (RESULTS WRONG, timing only) for VERDICT r3 item 7: how much of the ragged kernel's
time is its index rows, i.e. what could nibble / 6-bit / 9-bit packed rows buy at best?

    allbytes  every site reads 20-byte rows, also bags over 256 reads (which really need u16 rows of 40 bytes): what a 9-bit
              packing of the big bags (22.5 bytes a row) could approach on configs[4]'s shape
    rows16    byte rows at a stride of 16 bytes instead of 20: what 6-bit indices (15 bytes a row) would fetch for bags <= 64

    python tools/pool_rtab_knockouts.py --build      # build container
    python tools/pool_rtab_knockouts.py              # GPU box: configs[4]'s per-GPU shape and 200 k sites x 33 reads
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(REPO, "m6anet_amd", "csrc")
KO = os.path.join(HERE, "ko")


def rep(s, a, b, count=1):
    assert s.count(a) >= 1, a
    return s.replace(a, b, count)


def allbytes(s):
    return rep(s, "    const bool bytes = n <= M6A_RTAB_U8_MAX_N;", "    const bool bytes = true;")


def rows16(s):
    s = rep(s, "        constexpr int AL = MODE - 2, ND = (AL + 20 + 3) / 4;", "        constexpr int AL = MODE - 2, ND = (AL + 16 + 3) / 4;")
    s = rep(s, "        constexpr int b = J + MODE - 2;", "        constexpr int b = (J & 15) + MODE - 2;")
    s = rep(s, "    const int64_t row_bytes = (int64_t)K * esz;", "    const int64_t row_bytes = bytes ? 16 : (int64_t)K * esz;")
    return s


VARIANTS = {"asis": lambda s: s, "allbytes": allbytes, "rows16": rows16}


def build():
    os.makedirs(KO, exist_ok=True)
    src = open(os.path.join(CSRC, "m6a_pool_rtab.hip")).read()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-w",
             '-DM6A_MT_JUMP_PATH="%s"' % os.path.join(REPO, "m6anet_amd", "assets", "mt19937_jump.bin"),
             "-I" + os.path.join(REPO, "include"), "-I" + CSRC]
    objs = []
    sys.path.insert(0, REPO)
    from m6anet_amd.build import SOURCES
    for f in [x for x in SOURCES if x != "m6a_pool_rtab.hip"]:
        o = os.path.join(KO, "rt_" + f.replace(".hip", ".o"))
        if not os.path.exists(o) or os.path.getmtime(o) < os.path.getmtime(os.path.join(CSRC, f)):
            subprocess.check_call([hipcc] + flags + ["-c", os.path.join(CSRC, f), "-o", o])
        objs.append(o)
    for name, fn in VARIANTS.items():
        p = os.path.join(KO, "pool_rtab_%s.hip" % name)
        open(p, "w").write(fn(src))
        o = p.replace(".hip", ".o")
        subprocess.check_call([hipcc] + flags + ["-c", p, "-o", o])
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + [o, "-o", os.path.join(KO, "libm6a_rt_%s.so" % name)])
        print("built", name)


def time_one():
    import numpy as np
    import torch
    sys.path.insert(0, REPO)
    from m6anet_amd import synthetic
    from m6anet_amd.engine import M6ANetEngine, load_weights
    T = 1000
    eng = M6ANetEngine(weights=load_weights("HEK293T_RNA004"))
    out = {}
    for tag, S, bag in (("configs4_per_gpu", 125_000, (50, 500)), ("200k_x_33", 200_000, 33), ("200k_x_20_90", 200_000, (20, 90))):
        nr = synthetic.bag_sizes(S, bag)
        off_h = np.zeros(S + 1, np.int64)
        np.cumsum(nr, out=off_h[1:])
        g = torch.Generator(device="cuda").manual_seed(1)
        p = torch.rand(int(off_h[-1]), device="cuda", generator=g) ** 4
        off = torch.from_numpy(off_h).cuda()
        eng.set_scan_driver(3)
        for _ in range(3):
            eng.calculate_site_proba(p, off, T)
        eng.sync()
        eng.profile("pooling")
        for _ in range(20):
            eng.calculate_site_proba(p, off, T)
        ms, n = eng.profile_read(1)
        eng.profile(False)
        out[tag] = {"pool_ms": ms / n, "variant": eng.last_pool_variant, "T_draws_per_s": S * T * 20 / (ms / n * 1e-3) / 1e12}
    print(json.dumps(out))


def main():
    if "--build" in sys.argv:
        return build()
    if "--one" in sys.argv:
        return time_one()
    rows = {}
    for name in VARIANTS:
        lib = os.path.join(KO, "libm6a_rt_%s.so" % name)
        if not os.path.exists(lib):
            continue
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=dict(os.environ, M6A_HIP_LIB=lib),
                             capture_output=True, text=True, timeout=900)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        rows[name] = json.loads(line[-1]) if line else {"error": out.stderr[-300:]}
        print(name, rows[name], file=sys.stderr)
    print(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()
