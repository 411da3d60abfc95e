#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (bench_results.db): per-kernel launch count / average
duration, and per-kernel averages of any PMC counters collected.  Output is the text kept under
profiles/.   usage: rocpd_summary.py <results.db> [more.db ...]"""
import sqlite3
import sys


def main():
    for path in sys.argv[1:]:
        con = sqlite3.connect(path)
        print("== %s" % path)
        cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
        name_col = "name" if "name" in cols else "kernel_name"
        rows = con.execute(
            "select %s, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) "
            "from kernels group by %s order by 6 desc" % (name_col, name_col)).fetchall()
        tot = sum(r[5] for r in rows) or 1
        print("%-60s %8s %12s %12s %12s %7s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct"))
        for r in rows:
            print("%-60s %8d %12.2f %12.2f %12.2f %6.1f%%" % (r[0][:60], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3,
                                                             100.0 * r[5] / tot))
        try:
            ccols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
            if ccols:
                kn = "kernel_name" if "kernel_name" in ccols else "name"
                rows = con.execute(
                    "select %s, counter_name, count(*), avg(value), sum(value) from counters_collection "
                    "group by %s, counter_name order by 1, 2" % (kn, kn)).fetchall()
                if rows:
                    print("%-60s %-28s %8s %18s" % ("kernel", "counter", "samples", "avg_per_dispatch"))
                    for r in rows:
                        print("%-60s %-28s %8d %18.1f" % (r[0][:60], r[1], r[2], r[3]))
        except sqlite3.Error as e:
            print("(no counters: %s)" % e)


if __name__ == "__main__":
    main()
