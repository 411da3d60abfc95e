#!/usr/bin/env python3
"""Knock-out builds of pool_reg_kernel (RESULTS WRONG, timing only): what do the per-draw SALU instructions, the scalar
index loads and the index mode itself cost next to the two v_pk_mul_f32 a draw of four sites needs?  (VERDICT r3 item 4:
tools/valu_rate_bench shows the part delivers a v_pk_mul_f32 per 4.6 SIMD cycles at two waves per SIMD, i.e. 9.2 cycles
per draw; the kernel takes 14.0.)

    python tools/pool_reg_knockouts.py --build      # build container: cross-compiles tools/ko/libm6a_<variant>.so
    python tools/pool_reg_knockouts.py              # GPU box: times the pooling of 1 M sites x 20 reads, T = 1000, per variant
"""
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(REPO, "m6anet_amd", "csrc")
KO = os.path.join(HERE, "ko")


def sub(pattern, repl, s, count=1, flags=0):
    out, n = re.subn(pattern, repl, s, count=count, flags=flags)
    assert n >= 1, pattern
    return out


def no_loop_loads(s):
    a = s.index('"1:\\n"')
    b = s.index('"3:\\n"')
    body = s[a:b]
    body = re.sub(r'\s*"s_load_dwordx16 s\[\d+:\d+\], s\[80:81\], \d+\\n"', "", body)
    body = re.sub(r'\s*"s_load_dwordx4 s\[\d+:\d+\], s\[80:81\], \d+\\n"', "", body)
    return s[:a] + body + s[b:]


def vprefetch(look):
    """Three dummy VECTOR loads per round, `look` bytes ahead of the round's indices: they pull the lines of the index row
    into L2 long before the scalar loads ask for them (a scalar load cannot be waited for selectively: lgkmcnt(0) waits for
    all, so the look-ahead itself cannot be a scalar load; vmcnt is never waited for inside the loop)."""
    def f(s):
        s = s.replace('"s_mov_b32 s78, 0\\n"', '"s_mov_b32 s78, 0\\n"\n        "v_mov_b32 v36, 0\\n"', 1)
        s = s.replace('"1:\\n"', '"1:\\n"\n        "global_load_dword v37, v36, s[80:81] offset:%d\\n"\n        "global_load_dword v37, v36, s[80:81] offset:%d\\n"\n'
                                  '        "global_load_dword v37, v36, s[80:81] offset:%d\\n"' % (look, look + 64, look + 128), 1)
        s = s.replace('"v_mov_b32 %[o0], v60\\n"', '"s_waitcnt vmcnt(0)\\n"\n        "v_mov_b32 %[o0], v60\\n"', 1)
        s = s.replace('"v32", "v33", "v34", "v35", "v40"', '"v32", "v33", "v34", "v35", "v36", "v37", "v40"', 1)
        assert "global_load_dword v37" in s and '"v36", "v37"' in s and "v_mov_b32 v36, 0" in s
        return s
    return f


def stamps(s):
    """Per wave, delivered through the four site results of lanes 0..63: q0 = prologue ticks and q1 = loop ticks (s_memtime: the
    shader clock, whose base differs from one part of the chip to the next -- good for durations only), q2 = start and q3 = end on
    s_memrealtime (the 100 MHz constant clock, one base for the whole device; low 24 bits)."""
    s = s.replace('"v_mov_b32 v32, %[b0]\\n"', '"s_memtime s[86:87]\\n"\n        "s_memrealtime s[94:95]\\n"\n        "v_mov_b32 v32, %[b0]\\n"', 1)
    s = s.replace('"82:\\n"', '"82:\\n"\n        "s_waitcnt lgkmcnt(0)\\n"\n        "s_memtime s[88:89]\\n"', 1)
    s = s.replace('"v_mov_b32 %[o0], v60\\n"\n        "v_mov_b32 %[o1], v61\\n"\n        "v_mov_b32 %[o2], v62\\n"\n        "v_mov_b32 %[o3], v63\\n"',
                  '"s_memtime s[90:91]\\n"\n        "s_memrealtime s[96:97]\\n"\n        "s_waitcnt lgkmcnt(0)\\n"\n'
                  '        "s_sub_u32 s92, s88, s86\\n"\n        "s_sub_u32 s93, s90, s88\\n"\n'
                  '        "s_and_b32 s94, s94, 0xffffff\\n"\n        "s_and_b32 s96, s96, 0xffffff\\n"\n'
                  '        "v_cvt_f32_u32 %[o0], s92\\n"\n        "v_cvt_f32_u32 %[o1], s93\\n"\n        "v_cvt_f32_u32 %[o2], s94\\n"\n        "v_cvt_f32_u32 %[o3], s96\\n"', 1)
    s = s.replace('"s84", "s85",', '"s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97",', 1)
    assert "s_memtime s[90:91]" in s and "s_memtime s[88:89]" in s and "s_memrealtime s[96:97]" in s and '"s97",' in s
    return s


VARIANTS = {
    "asis": lambda s: s,
    "stamps": stamps,
    # round 5's loop has ONE scalar instruction per draw (the draw's M0 value straight out of a 16-bit table entry); without it
    # (index mode stays on with the first draw's index): what is that one still worth?
    "nom0": lambda s: sub(r'#define HI\(r\) .*', '#define HI(r) ""', sub(r'#define LO\(r\) .*', '#define LO(r) ""', s)),
    # CORRECT variants of the one scalar instruction: the low half through s_set_gpr_idx_idx (M0[7:0] only; the enables were set by
    # s_set_gpr_idx_on), and a s_nop behind every M0 write (does the indexed multiply wait on M0?)
    "lowidx": lambda s: sub(r'#define LO\(r\) .*', lambda m: r'#define LO(r) "s_set_gpr_idx_idx s[" S(r) "]\n"', s),
    "m0nop": lambda s: sub(r'#define HI\(r\) .*', lambda m: r'#define HI(r) "s_lshr_b32 m0, s[" S(r) "], 16\ns_nop 0\n"',
                            sub(r'#define LO\(r\) .*', lambda m: r'#define LO(r) "s_sext_i32_i16 m0, s[" S(r) "]\ns_nop 0\n"', s)),
    # the scalar index loads of the round loop removed (stale registers)
    "noload": no_loop_loads,
    "pkonly": lambda s: no_loop_loads(VARIANTS["nom0"](s)),
    # the shift moved between the two multiplies of the PREVIOUS draw (same instruction count, SALU never back to back)
    # CORRECT variants (round 6): wave priority -- the wave in hardware slot 1 of its SIMD at priority 3 (its partner at 0), or every wave at 3
    "prio_slot1": lambda s: s.replace("    m6a_clk_stamp(a.clk, 0);", "    m6a_clk_stamp(a.clk, 0);\n    if (__builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11)) & 1) __builtin_amdgcn_s_setprio(3);", 1),
    "prio_all3": lambda s: s.replace("    m6a_clk_stamp(a.clk, 0);", "    m6a_clk_stamp(a.clk, 0);\n    __builtin_amdgcn_s_setprio(3);", 1),
    "vpre640": vprefetch(640),
    "vpre1600": vprefetch(1600),
    "vpre3200": vprefetch(3200),
}


def build():
    os.makedirs(KO, exist_ok=True)
    src = open(os.path.join(CSRC, "m6a_pool_reg.hip")).read()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-w",
             '-DM6A_MT_JUMP_PATH="%s"' % os.path.join(REPO, "m6anet_amd", "assets", "mt19937_jump.bin"),
             "-I" + os.path.join(REPO, "include"), "-I" + CSRC]
    objs = []
    sys.path.insert(0, REPO)
    from m6anet_amd.build import SOURCES
    for f in [x for x in SOURCES if x != "m6a_pool_reg.hip"]:
        o = os.path.join(KO, f.replace(".hip", ".o"))
        if not os.path.exists(o) or os.path.getmtime(o) < os.path.getmtime(os.path.join(CSRC, f)):
            subprocess.check_call([hipcc] + flags + ["-c", os.path.join(CSRC, f), "-o", o])
        objs.append(o)
    for name, fn in VARIANTS.items():
        p = os.path.join(KO, "pool_reg_%s.hip" % name)
        open(p, "w").write(fn(src))
        o = p.replace(".hip", ".o")
        subprocess.check_call([hipcc] + flags + ["-c", p, "-o", o])
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + [o, "-o", os.path.join(KO, "libm6a_%s.so" % name)])
        print("built", name)


def time_one():
    import numpy as np
    import torch
    sys.path.insert(0, REPO)
    from m6anet_amd.engine import M6ANetEngine, load_weights
    T = 1000
    eng = M6ANetEngine(weights=load_weights("HCT116_RNA002"))
    out = {}
    for S in (1_000_000, 524_288, 2_097_152):
        g = torch.Generator(device="cuda").manual_seed(1)
        p = torch.rand(S * 20, device="cuda", generator=g) ** 4
        off = torch.arange(0, S * 20 + 1, 20, device="cuda", dtype=torch.int64)
        scratch = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
        for _ in range(3):
            eng.calculate_site_proba(p, off, T)
        eng.sync()
        eng.profile("pooling")
        for _ in range(20):
            scratch.zero_()                               # what the encoder does to the caches between two poolings
            eng.calculate_site_proba(p, off, T)
        ms, n = eng.profile_read(1)
        eng.profile(False)
        out["pool_ms_%d" % S] = ms / n
        if S == 1_000_000:
            out.update({"pool_ms": ms / n, "launches": n, "variant": eng.last_pool_variant,
                        "cycles_per_draw_of_4_at_2.4GHz": ms / n * 1e-3 * 2.4e9 * 1024 / (S * T * 20 / 4 / 64)})
        del p, off, scratch
    print(json.dumps(out))


def read_stamps():
    """Run the `stamps` build: per wave prologue / loop cycles and start / end times, for one round and for the bench shape."""
    import numpy as np
    import torch
    sys.path.insert(0, REPO)
    from m6anet_amd.engine import M6ANetEngine, load_weights
    T = 1000
    eng = M6ANetEngine(weights=load_weights("HCT116_RNA002"))
    out = {}
    for S in (262_144, 524_288, 1_000_000, 2_097_152):
        g = torch.Generator(device="cuda").manual_seed(1)
        p = torch.rand(S * 20, device="cuda", generator=g) ** 4
        off = torch.arange(0, S * 20 + 1, 20, device="cuda", dtype=torch.int64)
        for _ in range(3):
            site, _ = eng.calculate_site_proba(p, off, T)
        eng.sync()
        v = np.rint(site.cpu().numpy().astype(np.float64) * T)         # the kernel divides by T
        # flush groups: the first has batch_size = 16 sites, the others 32 (m6anet's inverted flush test): site s of
        # group g at position j
        G = 1 + (S - 16 + 31) // 32
        full = np.full((G, 32), np.nan)
        full[0, :16] = v[:16]
        rest = v[16:]
        full[1:1 + rest.size // 32] = rest[:rest.size // 32 * 32].reshape(-1, 32)
        v = full
        # lane = group (g0 + q*64 + lane): q = (group % 256) // 64
        q = (np.arange(G) % 256) // 64
        v = np.where(np.isnan(v), np.nanmedian(v, axis=1, keepdims=True), v)
        pro, loop = v[q == 0].ravel(), v[q == 1].ravel()
        # start / end are 24-bit counters in units of 256 ticks: they wrap, so everything is taken relative to one wave's
        # start on the circle (a kernel lasts < 2^23 units)
        ref = v[q == 2].ravel()[0]

        def rel(x):
            return (((x - ref) + (1 << 23)) % (1 << 24) - (1 << 23)) * 10.0          # ns
        # one value per WAVE: lane 0 of class 2 / 3 of every (256-group block, j)
        gb = np.arange(G) // 256
        first2 = np.array([np.flatnonzero((gb == b) & (q == 2))[0] for b in range(gb.max() + 1) if ((gb == b) & (q == 2)).any()])
        first3 = np.array([np.flatnonzero((gb == b) & (q == 3))[0] for b in range(gb.max() + 1) if ((gb == b) & (q == 3)).any()])
        start, end = rel(v[first2].ravel()), rel(v[first3].ravel())
        t0 = start.min()
        start, end = start - t0, end - t0
        cl = np.zeros(start.size, int)
        np.savez(os.path.join(REPO, "gpurun_out", "pool_reg_stamps_%d.npz" % S), start=start, end=end, cluster=cl,
                 prologue=v[np.array([np.flatnonzero((gb == b) & (q == 0))[0] for b in range(gb.max() + 1) if ((gb == b) & (q == 0)).any()])].ravel(),
                 loop=v[np.array([np.flatnonzero((gb == b) & (q == 1))[0] for b in range(gb.max() + 1) if ((gb == b) & (q == 1)).any()])].ravel())
        order = np.sort(start)
        out[S] = {"waves": 32 * ((G + 255) // 256),
                  "prologue_ticks": {"median": float(np.median(pro)), "p95": float(np.quantile(pro, 0.95)), "max": float(pro.max())},
                  "loop_ticks": {"median": float(np.median(loop)), "p5": float(np.quantile(loop, 0.05)), "p95": float(np.quantile(loop, 0.95))},
                  # when the k-th wave (in start order) began, ticks after the first: how long the dispatcher takes to fill the machine
                  "start_of_wave_k_after_first": {str(k): float(order[min(k, order.size - 1)]) for k in (1, 256, 1024, 2000, 2047, 2048, 2100, 3000, order.size - 1)},
                  "end_quantiles": {str(qq): float(np.quantile(end, qq)) for qq in (0.0, 0.25, 0.5, 0.52, 0.75, 0.9, 0.96, 1.0)},
                  "first_end": float(end.min()), "median_end": float(np.median(end)), "last_end": float(end.max()),
                  "waves_ending_in_last_10pct_of_span": int((end > 0.9 * end.max()).sum())}
        print(S, out[S], file=sys.stderr)
    print(json.dumps(out, indent=1))


def main():
    if "--stamps" in sys.argv:
        read_stamps()
        return
    if "--build" in sys.argv:
        build()
        return
    if "--one" in sys.argv:
        time_one()
        return
    rows = {}
    for name in VARIANTS:
        lib = os.path.join(KO, "libm6a_%s.so" % name)
        if not os.path.exists(lib):
            continue
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=dict(os.environ, M6A_HIP_LIB=lib),
                             capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        rows[name] = json.loads(line[-1]) if line else {"error": out.stderr[-300:]}
        print(name, rows[name], file=sys.stderr)
    print(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()
