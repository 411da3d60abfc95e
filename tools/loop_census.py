#!/usr/bin/env python3
"""Instruction census of a kernel's hot loop from hipcc's assembly (hipcc -S --cuda-device-only):
    python tools/loop_census.py file.s kernel_substring
The hot loop = the span between a label and the backward branch to it that holds the most v_mfma instructions."""
import collections
import re
import sys


def kernel_body(lines, name):
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % name, l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start:end]


def main():
    lines = open(sys.argv[1]).read().splitlines()
    body = kernel_body(lines, sys.argv[2])
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    best = None
    for i, l in enumerate(body):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            span = body[labels[m.group(1)]:i + 1]
            n = sum("v_mfma" in x for x in span)
            if best is None or n > best[0]:
                best = (n, span)
    n, span = best
    ops = collections.Counter()
    for l in span:
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        ops[t.split()[0]] += 1
    valu = sum(v for k, v in ops.items() if k.startswith("v_") and not k.startswith("v_mfma"))
    print("loop: %d instructions, %d v_mfma, %d other VALU, %d SALU, %d s_waitcnt, %d s_nop, %d ds_, %d global_/buffer_, %d s_load" % (
        sum(ops.values()), n, valu, sum(v for k, v in ops.items() if k.startswith("s_") and k not in ("s_waitcnt", "s_nop") and not k.startswith("s_load")),
        ops["s_waitcnt"], ops["s_nop"], sum(v for k, v in ops.items() if k.startswith("ds_")),
        sum(v for k, v in ops.items() if k.startswith(("global_", "buffer_"))), sum(v for k, v in ops.items() if k.startswith("s_load"))))
    for k, v in sorted(ops.items(), key=lambda kv: -kv[1]):
        print("  %4d  %s" % (v, k))


if __name__ == "__main__":
    main()
