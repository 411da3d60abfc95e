#!/usr/bin/env python3
"""Fuzz of the read encoder against the CPU oracle (test infrastructure, like tests/): random jobs -- bag structures
from one read per site to thousands, empty sites, read counts that are not multiples of the 32-read tile, site
boundaries at every position of a tile, all four checkpoints and randomly perturbed weights, host and device
pointers, the 16-slot and the 12-slot kernel (forced where its precondition holds) -- against the oracle's read
probabilities at the reference's own bar (rtol 1e-5, atol 1e-8; m6anet/tests/test_inference.py:32).

What the bar can and cannot say at this sample size.  Two float32 evaluations of this network in different summation
orders differ by rounding noise that is a sizeable share of rtol 1e-5 (small probabilities: one ulp of a logit near -18
is 1e-6 relative): on the reference's own captures the worst read uses 0.3-0.6 of it, over 10 M random reads 0.30 / 0.81 /
0.94 / 0.60 (12-slot kernel) and 0.31 / 0.90 / 0.97 / 0.56 (16-slot kernel) for the four checkpoints, and the tail keeps
growing with the sample: about one read in 10^7 crosses 1.0 on the HEK293T weights (1.03 seen; since the kernels follow the reference's order
(below) a few in 10^10, 1.08 seen in round 6: profiles/r06_encoder_fuzz_final.json).  The exact float64 result
itself sits at 0.9-1.0 of the bar against the float32 oracle (HISTORY.md section 8).  So this fuzz fails on a GROSS error --
any read beyond 1.5x the bar (5x for randomly perturbed weights, where the same noise is larger still) -- and reports how
many reads went beyond 1.0 and the worst one; the committed tests hold the fixtures and seeded samples to the bar itself.

Since the second half of round 4 the kernels and the oracle both add in the reference's order: what
separates them is the 32 -> 1 sum, expf, and in the 12-slot kernel the pre-summed site constants -- 11 G reads, none beyond
the bar, worst 0.94 (profiles/r04_encoder_fuzz_final.json).  The thresholds above stay as they are: a gross-error net.

    python tests/fuzz_encoder.py [seconds] [seed]      -> one JSON line with the case counts
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m6anet_amd.constants import DEFAULT_PRETRAINED_MODELS  # noqa: E402
from m6anet_amd.engine import M6ANetEngine, load_weights  # noqa: E402
from oracle import m6a_oracle as orc  # noqa: E402


def main():
    import torch
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    g = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 1))
    orc.build()
    models = list(DEFAULT_PRETRAINED_MODELS)
    dev = torch.device("cuda:0")
    t_end = time.time() + budget
    n_cases = n_runs = 0
    worst = worst_perturbed = 0.0
    n_reads = n_over = 0
    by_kernel = {}
    g16_by_kernel = {}
    g16_reads = 0
    engines = {}
    while time.time() < t_end:
        name = models[int(g.integers(0, len(models)))]
        w = load_weights(name).copy()
        perturbed = g.random() < 0.3
        if perturbed:                                              # other weights than the four the fixtures pin
            w *= (1.0 + 0.05 * g.standard_normal(w.size)).astype(np.float32)
            w[2982:3132] = np.abs(w[2982:3132]) + 1e-3             # BatchNorm running variance stays positive
        key = name if not perturbed else None
        eng = engines.get(key) if key else None
        if eng is None:
            eng = M6ANetEngine(weights=w)
            if key:
                engines[key] = eng
        kind = int(g.integers(0, 6))
        S = int(g.integers(1, 3000))
        if kind == 0:
            bags = np.full(S, 20)
        elif kind == 1:
            bags = g.integers(16, 120, size=S)
        elif kind == 2:
            bags = g.integers(1, 40, size=S)                       # bags under 16 reads: the 16-slot kernel
        elif kind == 3:
            bags = g.choice([1, 15, 16, 17, 31, 32, 33, 63, 64, 65, 500, 1000], size=S)
        elif kind == 4:
            bags = np.where(g.random(S) < 0.02, g.integers(1000, 5000, size=S), g.integers(16, 64, size=S))
        else:
            bags = g.integers(20, 700, size=S)
        if g.random() < 0.15:
            bags = np.where(g.random(S) < 0.2, 0, bags)            # empty sites
        if bags.sum() == 0 or bags.sum() > 3_000_000:
            continue
        off = np.concatenate([[0], np.cumsum(bags)]).astype(np.int64)
        R = int(off[-1])
        # z-scores (data_utils.py:216-218): N(0,1) clipped to +-6 as SURVEY.md section 8(d) specifies, sometimes narrower or a
        # little wider (at 2.5 sigma the logits grow and plain float32 rounding noise alone crosses the bar: 1.03 seen)
        X = np.clip(g.standard_normal((R, 9)) * float(g.choice([0.3, 1.0, 1.0, 1.3])), -6, 6).astype(np.float32)
        km = g.integers(0, 66, size=(S, 3), dtype=np.uint8)
        want = orc.encode_reads(w, X, km, off, n_threads=8)
        n_cases += 1
        # 0: the automatic choice (16 slots: what the product runs); 2: the opt-in 12-slot kernel; 3: the 16-slot arithmetic behind
        # the per-lane walk; 4: "fast" (12 slots where every bag has >= 16 reads, else the 16-slot kernels -- then held to bits)
        variants = [0, 4] + ([2, 3] if bags.min() >= 16 else [])
        for v in variants:
            eng.set_encoder_variant(v)
            for on_dev in (False, True):
                if on_dev:
                    got = eng.get_read_probability(torch.from_numpy(X).to(dev), torch.from_numpy(km).to(dev), torch.from_numpy(off).to(dev))
                    eng.sync()
                    got = got.cpu().numpy()
                else:
                    got = eng.get_read_probability(X, km, off)
                n_runs += 1
                by_kernel[eng.last_encoder_kernel] = by_kernel.get(eng.last_encoder_kernel, 0) + 1
                g16_by_kernel[eng.last_encoder_kernel] = g16_by_kernel.get(eng.last_encoder_kernel, 0) + (R if eng.last_encoder_variant == "general16" else 0)
                if eng.last_encoder_variant == "general16":
                    # the 16-slot kernel and the oracle restate the reference operation for operation: the same bits, always --
                    # perturbed weights included
                    n_bits = int(np.count_nonzero(got.view(np.uint32) != want.view(np.uint32)))
                    g16_reads += R
                    if n_bits:
                        print(json.dumps({"FAIL": name, "kernel": "general16", "reads_not_bit_identical_to_the_oracle": n_bits, "S": S, "R": R,
                                          "perturbed": perturbed, "device_pointers": on_dev, "kind": kind}))
                        sys.exit(1)
                used = float(np.max(np.abs(got - want) / (1e-8 + 1e-5 * np.abs(want))))
                if perturbed:
                    worst_perturbed = max(worst_perturbed, used)
                else:
                    worst = max(worst, used)
                    n_reads += R
                    n_over += int(np.count_nonzero(np.abs(got - want) > 1e-8 + 1e-5 * np.abs(want)))
                if not used <= (5.0 if perturbed else 1.5):
                    i = int(np.argmax(np.abs(got - want) / (1e-8 + 1e-5 * np.abs(want))))
                    print(json.dumps({"FAIL": name, "perturbed": perturbed, "variant": v, "kernel": eng.last_encoder_variant, "device_pointers": on_dev,
                                      "kind": kind, "S": S, "R": R, "read": i, "got": float(got[i]), "want": float(want[i]), "tolerance_used": used}))
                    sys.exit(1)
        eng.set_encoder_variant(0)
    print(json.dumps({"cases": n_cases, "kernel_runs": n_runs, "by_kernel": by_kernel, "seconds": budget, "worst_tolerance_used": worst,
                      "reads_checked_real_checkpoints": n_reads, "reads_beyond_the_bar": n_over,
                      "worst_tolerance_used_perturbed_weights": worst_perturbed,
                      "general16_reads_bit_identical_to_the_oracle": g16_reads, "of_which_by_kernel": {k: v for k, v in g16_by_kernel.items() if v},
                      "result": "no gross error: every read within 1.5x (real checkpoints) / 5x (perturbed weights) of rtol 1e-5, atol 1e-8"}))


if __name__ == "__main__":
    main()
