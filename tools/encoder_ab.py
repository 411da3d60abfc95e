#!/usr/bin/env python3
"""A/B of encoder-kernel builds on the bench shape (20 M reads): HIP-event time per launch of the two scalar-chain kernels, per build,
legs interleaved A B C A B C ... so clock drift hits every build alike.

    python tools/encoder_ab.py --build          (here: cross-compiles the variants into tools/ko/, which travels to the GPU box)
    python tools/encoder_ab.py [legs]           (GPU box: one JSON line)

Variants = the product source with ONE macro each (m6a_kernels.hip, `#ifdef M6A_AB_*`):
  base               the product
  csite_scalar_fma   VERDICT r5 item 3(a): link3's 30 v_pk_fma_f32 as 60 plain v_fma_f32, in place between the MFMA groups (same bits)
  csite_pin          the 30 v_pk_fma_f32 kept where they are written (hipcc otherwise sinks them behind the epilogue; same bits)
  addr64             link1's loads addressed the old way (signed lane offset: 64-bit VALU address arithmetic) instead of scalar base + unsigned 32-bit lane offset
  links_abc          enc_site16_kernel: links 1 / 2 / 3 of the input chain at m = a / b / c of the unit-tile loop instead of 1 / 2 / 3 (rounds 2-6a: 0 / 1 / 3); bn_block2 / 8: batch norm woven in blocks of 2 / 8 hidden
                     units; l2_first: batch norm + layer 2 of a unit tile in front of the next unit tile's layer 1 (same bits)
  w3                 enc_site16_kernel at THREE waves per SIMD: layer 2's A operands from LDS (one float4 per block of four hidden units, a block ahead) instead of 80 registers (same bits)
  no_prio            the product WITHOUT its wave-priority split (s_setprio 0 through a tile's MFMA body, 3 through its epilogue; same bits)
  prio_body3_epi0    the round's first split, the other way round (body 3, epilogue 0); p01 / p13 / p23: body / epilogue priorities 0/1, 1/3, 2/3
  bn_pk              batch norm + ReLU of two hidden units per instruction: 38 v_pk_fma_f32 ... clamp instead of 76 v_fma_f32 ... clamp (same bits)
  prio_block_valu_low / _high   inside the body: a block's four batch-norm fmas at priority 0 and its MFMAs at 3 / the fmas at 3 and the MFMAs at 1 (same bits)
  phase / phase_hwid enc_site16_kernel only: the second workgroup of a CU (or the wave in hardware slot 1) starts its tile loop half a tile late
                     (same bits) -- tools/encoder_timeline.py shows the phases
  no_bn / no_links / no_bn_no_epilogue / no_bn_no_epilogue_no_links   knock-outs, WRONG results: layer 2 straight on layer 1's accumulators (no clamped fmas, no pair
                     loads); enc_site16_kernel without its input chain (every tile on the first tile's features); combinations -- what each part of the non-MFMA work costs in place
  no_epilogue        knock-out, WRONG results: the 32 -> 1 layer + sigmoid removed from enc_site16_kernel (what the epilogue costs in
                     place = the most that hiding it under the next tile's MFMAs could buy)
"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
KO = os.path.join(REPO, "tools", "ko")
def _links(a, b, c):
    return ["-DM6A_AB_LINK1_AT=%d" % a, "-DM6A_AB_LINK2_AT=%d" % b, "-DM6A_AB_LINK3_AT=%d" % c]


VARIANTS = {"base": [], "w3": ["-DM6A_AB_W3"], "links_013": _links(0, 1, 3), "links_012": _links(0, 1, 2), "links_024": _links(0, 2, 4), "links_124": _links(1, 2, 4), "links_014": _links(0, 1, 4),
            "links_234": _links(2, 3, 4), "links_134": _links(1, 3, 4), "links_034": _links(0, 3, 4), "links_023": _links(0, 2, 3), "bn_block2": ["-DBN_BLOCK=2"], "bn_block8": ["-DBN_BLOCK=8"], "l2_first": ["-DM6A_AB_L2_FIRST"], "prev": [], "addr64": ["-DM6A_AB_ADDR64"], "no_bn": ["-DM6A_AB_NO_BN"], "no_links": ["-DM6A_AB_NO_LINKS"], "no_bn_no_epilogue": ["-DM6A_AB_NO_BN", "-DM6A_AB_NO_EPILOGUE"],
            "no_bn_no_epilogue_no_links": ["-DM6A_AB_NO_BN", "-DM6A_AB_NO_EPILOGUE", "-DM6A_AB_NO_LINKS"],
            "no_prio": ["-DM6A_AB_NO_PRIO"], "csite_scalar_fma": ["-DM6A_AB_CSITE_SCALAR_FMA"], "csite_pin": ["-DM6A_AB_CSITE_PIN"],
            "no_epilogue": ["-DM6A_AB_NO_EPILOGUE"], "bn_pk": ["-DM6A_AB_BN_PK"],
            "prio_block_valu_low": ["-DM6A_AB_PRIO_BLOCK_VALU=0", "-DM6A_AB_PRIO_BLOCK_MFMA=3"], "prio_block_valu_high": ["-DM6A_AB_PRIO_BLOCK_VALU=3", "-DM6A_AB_PRIO_BLOCK_MFMA=1"],
            "p01": ["-DM6A_AB_PRIO_BODY=0", "-DM6A_AB_PRIO_EPI=1"], "p13": ["-DM6A_AB_PRIO_BODY=1", "-DM6A_AB_PRIO_EPI=3"], "p23": ["-DM6A_AB_PRIO_BODY=2", "-DM6A_AB_PRIO_EPI=3"],
            "p03_bn2": ["-DM6A_AB_PRIO_BLOCK_VALU=2", "-DM6A_AB_PRIO_BLOCK_MFMA=0"],
            "p13_bn0": ["-DM6A_AB_PRIO_BODY=1", "-DM6A_AB_PRIO_EPI=3", "-DM6A_AB_PRIO_BLOCK_VALU=0", "-DM6A_AB_PRIO_BLOCK_MFMA=1"],
            "prio_body3_epi0": ["-DM6A_AB_PRIO_BODY=3", "-DM6A_AB_PRIO_EPI=0"], "phase": ["-DM6A_AB_PHASE=1"],
            "phase_hwid": ["-DM6A_AB_PHASE=1", "-DM6A_AB_PHASE_HWID"],
            "no_epilogue_phase": ["-DM6A_AB_NO_EPILOGUE", "-DM6A_AB_PHASE=1", "-DM6A_AB_PHASE_HWID"],
            "no_epilogue_phase_quarter": ["-DM6A_AB_NO_EPILOGUE", "-DM6A_AB_PHASE=1", "-DM6A_AB_PHASE_HWID", "-DM6A_AB_PHASE_SLEEP=64"],
            "no_epilogue_phase_eighth": ["-DM6A_AB_NO_EPILOGUE", "-DM6A_AB_PHASE=1", "-DM6A_AB_PHASE_HWID", "-DM6A_AB_PHASE_SLEEP=32"]}
EXTRA = [a for a in sys.argv[1:] if a.startswith("+")]       # +name=-DMACRO adds a variant from the command line
ONLY = [a[5:].split(",") for a in sys.argv[1:] if a.startswith("only=")]          # only=base,phase: build / run just these


def lib(name):
    return os.path.join(KO, "libm6a_ab_%s.so" % name)


def build():
    from m6anet_amd import build as B
    os.makedirs(KO, exist_ok=True)
    jump = os.path.join(B.PKG, "assets", "mt19937_jump.bin")
    for name, flags in VARIANTS.items():
        if ONLY and name not in ONLY[0]:
            continue
        cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               '-DM6A_MT_JUMP_PATH="%s"' % jump, "-I" + B.INCLUDE, "-I" + B.CSRC] + flags + [os.path.join(B.CSRC, s) for s in B.SOURCES] + ["-o", lib(name)]
        subprocess.check_call(cmd)
        print("built", lib(name))


def one(legs_unused):
    import torch
    from m6anet_amd import synthetic
    from m6anet_amd.engine import M6ANetEngine, load_weights
    eng = M6ANetEngine(weights=load_weights("HCT116_RNA002"))
    d = synthetic.make_sites(1_000_000, 20, seed=20250328)
    X, km, off = (torch.from_numpy(d[k]).cuda() for k in ("X", "site_kmers", "off"))
    rp = torch.empty(int(d["off"][-1]), dtype=torch.float32, device="cuda")
    out = {}
    for mode, label in ((0, "enc_site16_kernel"), (2, "enc_csite_kernel"), (3, "enc_kernel")):
        eng.set_encoder_variant(mode)
        for _ in range(5):
            eng.get_read_probability(X, km, off, out=rp)
        eng.sync()
        eng.profile("encoder")
        for _ in range(40):
            eng.get_read_probability(X, km, off, out=rp)
        ms, n = eng.profile_read(0)
        clk = eng.profile_clock(0)
        eng.profile(False)
        assert eng.last_encoder_kernel == label
        out[label] = [round(ms / n, 5), round(clk["ghz"], 4) if clk else None]
    print(json.dumps(out))


def main():
    if "--build" in sys.argv:
        for a in EXTRA:
            n, f = a[1:].split("=", 1)
            VARIANTS[n] = f.split(",")
        build()
        return
    if "--one" in sys.argv:
        one(0)
        return
    nums = [a for a in sys.argv[1:] if a.isdigit()]
    legs = int(nums[0]) if nums else 4
    names = [n for n in list(VARIANTS) + [a[1:].split("=")[0] for a in EXTRA] if os.path.exists(lib(n)) and (not ONLY or n in ONLY[0])]
    res = {n: {} for n in names}
    for leg in range(legs):
        for n in names:
            o = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=dict(os.environ, M6A_HIP_LIB=lib(n)), capture_output=True, text=True, timeout=600)
            try:
                r = json.loads(o.stdout.strip().splitlines()[-1])
            except (ValueError, IndexError):
                res[n].setdefault("error", []).append((o.stderr or o.stdout)[-300:])
                continue
            for k, v in r.items():
                res[n].setdefault(k, []).append(v)
    summary = {}
    for n, r in res.items():
        summary[n] = {k: {"median_ms": sorted(x[0] for x in v)[len(v) // 2], "ms_of_each_leg": [x[0] for x in v], "ghz_of_each_leg": [x[1] for x in v]}
                      for k, v in r.items() if k != "error"}
        if "error" in r:
            summary[n]["error"] = r["error"]
    print(json.dumps({"legs": legs, "launches_per_leg": 40, "workload": "20 M reads (1 M sites x 20), HCT116", "builds": summary}))


if __name__ == "__main__":
    main()
