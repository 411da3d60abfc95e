#!/usr/bin/env python3
"""Timeline of bench.py's FIRST call from a rocprofv3 trace (csv output of --hip-trace --kernel-trace --memory-copy-trace):
every HIP API call, kernel and copy from 3 ms before the first encoder launch to the end of the first pooling kernel, on
one time axis (ms, 0 = start of the first encoder kernel).   usage: first_call_timeline.py <dir with *_trace.csv>"""
import csv
import glob
import os
import sys


def rows(pattern):
    out = []
    for f in glob.glob(os.path.join(sys.argv[1], "**", pattern), recursive=True):
        with open(f, newline="") as fh:
            out.extend(csv.DictReader(fh))
    return out


def main():
    ker = rows("*kernel_trace.csv")
    api = rows("*hip_api_trace.csv")
    cop = rows("*memory_copy_trace.csv")
    enc = sorted((int(k["Start_Timestamp"]), int(k["End_Timestamp"])) for k in ker if k["Kernel_Name"].startswith("enc_"))
    pool = sorted((int(k["Start_Timestamp"]), int(k["End_Timestamp"])) for k in ker if k["Kernel_Name"].startswith("pool_reg_kernel") or
                  k["Kernel_Name"].startswith("pool_rtab_kernel"))
    t0 = enc[0][0]
    t_end = [p for p in pool if p[0] > t0][0][1] + 300_000
    ev = []
    for k in ker:
        ev.append((int(k["Start_Timestamp"]), int(k["End_Timestamp"]), "KERNEL", k["Kernel_Name"][:60]))
    for a in api:
        ev.append((int(a["Start_Timestamp"]), int(a["End_Timestamp"]), "api t%s" % a.get("Thread_Id", "?")[-4:], a["Function"]))
    for c in cop:
        ev.append((int(c["Start_Timestamp"]), int(c["End_Timestamp"]), "COPY", "%s %s B" % (c.get("Direction", ""), c.get("Bytes", c.get("Size", "?")))))
    ev.sort()
    print("%10s %10s  %-10s %s" % ("start_ms", "dur_ms", "what", "name"))
    for s, e, kind, name in ev:
        if t0 - 3_000_000 <= s <= t_end:
            print("%10.3f %10.3f  %-10s %s" % ((s - t0) / 1e6, (e - s) / 1e6, kind, name))
    print("# second encoder launch starts at %.3f ms" % ((enc[1][0] - t0) / 1e6))


if __name__ == "__main__":
    main()
