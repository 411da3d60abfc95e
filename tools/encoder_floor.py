#!/usr/bin/env python3
"""Reads the PMC passes of tools/encoder_floor.sh and prints, per encoder kernel, where a SIMD's cycles go:
matrix pipe busy, other VALU, waiting -- per launch and per 32-read tile.  usage: encoder_floor.py <dir> <tag>"""
import glob
import os
import sqlite3
import sys

TILES = 625_000            # 20 M reads / 32
SIMDS = 1024               # 256 CUs x 4
MFMA_CYCLES = 64           # v_mfma_f32_32x32x2_f32: 16 passes x 4 cycles on its SIMD


def read(dbdir):
    out, times = {}, {}
    for db in glob.glob(os.path.join(dbdir, "**", "*_results.db"), recursive=True):
        con = sqlite3.connect(db)
        cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
        kn = "name" if "name" in cols else "kernel_name"
        for name, n, avg, mn in con.execute("select %s, count(*), avg(end-start), min(end-start) from kernels group by %s" % (kn, kn)):
            times[name.split("(")[0]] = (n, avg / 1e3, mn / 1e3)
        ccols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
        ck = "kernel_name" if "kernel_name" in ccols else "name"
        for name, ctr, n, avg in con.execute("select %s, counter_name, count(*), avg(value) from counters_collection group by %s, counter_name" % (ck, ck)):
            out.setdefault(name.split("(")[0], {})[ctr] = avg
    return out, times


def main():
    base, tag = sys.argv[1], sys.argv[2]
    print("# %s: where the read encoder's SIMD cycles go (tools/encoder_floor.sh; bench default workload, 20 M reads = %d tiles of 32 reads;" % (tag, TILES))
    print("# rocprofv3 --pmc <group> --kernel-trace, one group per run; profiled runs clock lower than un-profiled ones: read ratios)")
    for enc, kernel in (("auto", "enc_site16_kernel"), ("fast", "enc_csite_kernel")):
        c, t = {}, {}
        for g in range(1, 5):
            cc, tt = read(os.path.join(base, "%s_floor_%s_g%d" % (tag, enc, g)))
            if kernel in cc:
                c.update(cc[kernel])
            if kernel in tt:
                t[g] = tt[kernel]
        print("\n== %s  (M6A_ENCODER=%s)" % (kernel, enc))
        for g, (n, avg, mn) in sorted(t.items()):
            print("   pass %d: %d launches, avg %.1f us, min %.1f us" % (g, n, avg, mn))
        for k in sorted(c):
            print("   %-30s %18.1f" % (k, c[k]))
        need = ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES")
        if not all(k in c for k in need):
            print("   (counters missing: no split)")
            continue
        # SQ_BUSY_CYCLES counts per shader engine (32): kernel cycles = /32; SIMD cycles available = kernel cycles x 1024
        kcyc = c["SQ_BUSY_CYCLES"] / 32.0
        simd_cyc = kcyc * SIMDS
        mfma = c.get("SQ_INSTS_MFMA") or c["SQ_VALU_MFMA_BUSY_CYCLES"] / MFMA_CYCLES
        other = c["SQ_INSTS_VALU"] - mfma
        avg_us = t.get(1, (0, 0, 0))[1]
        print("   -- split (pass 1 clocked at %.3f GHz = SQ_BUSY_CYCLES / 32 / avg duration)" % (kcyc / (avg_us * 1e3) if avg_us else 0))
        print("   MFMA instructions per tile            %8.1f" % (mfma / TILES))
        print("   other VALU instructions per tile      %8.1f" % (other / TILES))
        print("   SIMD cycles per tile                  %8.1f   (floor = MFMAs x 64: %.0f)" % (simd_cyc / TILES, mfma / TILES * MFMA_CYCLES))
        print("   matrix pipe busy                      %8.3f   of SIMD cycles (SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES/32 x 1024))" % (c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cyc))
        if "SQ_INST_CYCLES_VALU" in c:
            print("   SQ_INST_CYCLES_VALU / SIMD cycles     %8.3f   (quad-cycles x 4 if the counter is in quad-cycles: %.3f)" % (
                c["SQ_INST_CYCLES_VALU"] / simd_cyc, 4 * c["SQ_INST_CYCLES_VALU"] / simd_cyc))
        rest = simd_cyc - c["SQ_VALU_MFMA_BUSY_CYCLES"]
        print("   not-MFMA SIMD cycles per tile         %8.1f   = %.2f cycles per other VALU instruction if they were all of it" % (rest / TILES, rest / other))
        wc = c["SQ_WAVE_CYCLES"]
        for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA",
                  "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_FLAT"):
            if k in c:
                print("   %-28s / SQ_WAVE_CYCLES %6.3f" % (k, c[k] / wc))
        if "GRBM_GUI_ACTIVE" in c and 4 in t:
            print("   effective clock, pass 4               %8.3f GHz (GRBM_GUI_ACTIVE / 8 XCDs / avg duration)" % (c["GRBM_GUI_ACTIVE"] / 8.0 / (t[4][1] * 1e3)))


if __name__ == "__main__":
    main()
