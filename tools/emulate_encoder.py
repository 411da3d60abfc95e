#!/usr/bin/env python3
"""The encoder's float32 arithmetic, operation by operation, in NumPy -- to ask questions the GPU cannot answer cheaply:
which ORDER of the same multiply-adds lands closest to the reference's values?

torch's CPU path was pinned first (build container, hooks on the reference's modules, 320 reads x hek293t_glori; every
candidate order compared bit for bit with torch's own intermediate tensors):
    Linear 15->150   : acc = 0; for k = 0..14: acc = fma(x[k], W1[j][k], acc);  then acc + b1[j]          100.00 % identical
    BatchNorm (eval) : alpha = g * (1/sqrt(var + eps)); beta = fma(-mean, alpha, bias); fma(y, alpha, beta) 100.00 %
    Linear 150->32   : acc = 0; for k = 0..149: acc = fma(h[k], W2[j][k], acc); then acc + b2[j]          100.00 %
    Linear 32->1     : an MKL gemv (rows in groups of 4): s = x0*w0; 16 lanes of products k = 1..16 with lane 0 = fma(x1, w1, s),
                       butterfly l+8, l+4, l+2, l+1; again for k = 17..31 with lane 0 carrying the sum; + b3                  100.00 %
                       -- on this CPU's AVX-512 path; MKL_ENABLE_INSTRUCTIONS=AVX2 gives other logits (26-57 % identical) and
                       93-98 % of the hidden layer's bits: layers 1-2 are what the paths share, so that is what is followed
    Sigmoid          : 1 / (1 + exp(-z)) with Sleef's 1-ulp exp (92 % against a correctly rounded exp)

    python tools/emulate_encoder.py [model]      # needs tests/golden/reference_at_scale.npz; gpurun_out/read_probs_<model>.npz if present
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from m6anet_amd import synthetic                      # noqa: E402
from m6anet_amd.constants import asset_path           # noqa: E402

f64, f32 = np.float64, np.float32


def fma(a, b, c):
    # the product of two floats is exact in double; the sum is rounded to double, then to float (double rounding can differ
    # from a true fma in ~1e-9 of cases: irrelevant here)
    return (a.astype(f64) * b.astype(f64) + c.astype(f64)).astype(f32)


def unpack(w):
    return dict(E=w[0:132].reshape(66, 2), W1=w[132:2382].reshape(150, 15), b1=w[2382:2532], g=w[2532:2682], be=w[2682:2832],
                mu=w[2832:2982], var=w[2982:3132], W2=w[3132:7932].reshape(32, 150), b2=w[7932:7964], W3=w[7964:7996], b3=w[7996])


def chain(x, W, ks, acc=None):
    """acc[r][j] = fma(x[r][k], W[j][k], acc[r][j]) for k in ks, in that order."""
    if acc is None:
        acc = np.zeros((x.shape[0], W.shape[0]), f32)
    for k in ks:
        acc = fma(x[:, k:k + 1], W[None, :, k], acc)
    return acc


def sigmoid(z):
    return (f32(1) / (f32(1) + np.exp(-z.astype(f64)).astype(f32))).astype(f32)


def sleef_expf(d):
    """Sleef's expf (u10) as torch's vectorised sigmoid calls it: 99.9997 % identical to torch.sigmoid's over 2 M random floats."""
    d = d.astype(f32)
    sh = d.shape
    c = lambda x: np.broadcast_to(f32(x), sh)                                      # noqa: E731
    qf = np.rint((d * f32(1.442695040888963407359924681001892137426645954152985934135449406931)).astype(f32)).astype(f32)
    q = qf.astype(np.int32)
    s_ = fma(qf, c(-0.693145751953125), d)
    s_ = fma(qf, c(-1.428606765330187045e-06), s_)
    u = c(0.000198527617612853646278381).copy()
    for k in (0.00139304355252534151077271, 0.00833336077630519866943359, 0.0416664853692054748535156, 0.166666671633720397949219, 0.5):
        u = fma(u, s_, c(k))
    u = (f32(1.0) + fma((s_ * s_).astype(f32), u, s_)).astype(f32)
    q1 = q >> 1
    u = ((u * np.ldexp(f32(1), q1).astype(f32)).astype(f32) * np.ldexp(f32(1), q - q1).astype(f32)).astype(f32)
    return np.where(d < -104, f32(0), u).astype(f32)


def sigmoid_sleef(z):
    return (f32(1) / (f32(1) + sleef_expf(-z))).astype(f32)


def torch_like(P, inp, upto="p"):
    y = chain(inp, P["W1"], range(15)) + P["b1"]
    invstd = (f32(1) / np.sqrt(P["var"] + f32(1e-5))).astype(f32)
    alpha = (P["g"] * invstd).astype(f32)
    beta = fma(-P["mu"], alpha, P["be"])
    h = np.maximum(fma(y, np.broadcast_to(alpha, y.shape), np.broadcast_to(beta, y.shape)), 0)
    h2 = np.maximum(chain(h, P["W2"], range(150)) + P["b2"], 0)
    if upto == "h2":
        return h2
    z = chain(h2, P["W3"][None, :], range(32)) + P["b3"]
    return sigmoid(z[:, 0])


def mkl_avx512_gemv_32(h2, W3, b3):
    """Linear(32, 1) as MKL's AVX-512 sgemv computes it for rows in groups of 4 (found with cancellation triples, see the module
    docstring): s = x0*w0; lanes l = 0..15 hold the products of k = 1 + l, lane 0 = fma(x1, w1, s); butterfly l+8, l+4, l+2, l+1;
    the same again for k = 17 + l (lane 15 empty) with lane 0 carrying the sum so far; + b3."""
    def butterfly(v):
        for step in (8, 4, 2, 1):
            v = [v[l] + v[l + step] for l in range(step)]
        return v[0]
    p = [(h2[:, k] * W3[k]).astype(f32) for k in range(32)]
    v = [p[1 + l] for l in range(16)]
    v[0] = fma(h2[:, 1], np.broadcast_to(W3[1], h2[:, 1].shape), p[0])
    s = butterfly(v)
    v = [p[17 + l] if 17 + l < 32 else np.zeros_like(s) for l in range(16)]
    v[0] = fma(h2[:, 17], np.broadcast_to(W3[17], h2[:, 17].shape), s)
    return (butterfly(v) + f32(b3)).astype(f32)


def bn_pairs(P):
    invstd = (f32(1) / np.sqrt(P["var"] + f32(1e-5))).astype(f32)
    alpha = (P["g"] * invstd).astype(f32)
    return alpha, fma(-P["mu"], alpha, P["be"])


def folded(P):
    """Rounds 1-3: batch norm folded into layer 1's weights in float32."""
    alpha, _ = bn_pairs(P)
    shift = (P["be"] - (P["mu"] * alpha).astype(f32)).astype(f32)
    W1f = np.zeros((150, 16), f32)
    W1f[:, :15] = (alpha[:, None] * P["W1"]).astype(f32)
    W1f[:, 15] = ((alpha * P["b1"]).astype(f32) + shift).astype(f32)
    return W1f


def old_unit_order(m):
    """Rounds 1-3, layer 2: the k pairs of unit tile m in issue order (register q held units (q&3) + 8(q>>2) + 4 half)."""
    out = []
    for q in range(12 if m == 4 else 16):
        u = 32 * m + (q & 3) + 8 * (q >> 2)
        out += [u, u + 4]
    return out


def out_unit(q, half):
    """Layer 2's output unit in accumulator register q of lane half `half` (m6a_api.hip build_fragments)."""
    if half == 0:
        return 2 * q + 1 if q < 8 else 2 * (q - 8) + 17
    return 2 * q + 2 if q < 8 else (2 * (q - 8) + 18 if q < 15 else 0)


def layer3_halves(P, h2):
    """The 12-slot kernel: each lane half sums the 16 output units in its registers (out_unit) as one fma chain in register
    order, the halves are added, then b3.  (Rounds 1-3 and the first half of round 4: rows (q&3) + 8(q>>2) + 4 half.)"""
    rows = [[out_unit(q, hf) for q in range(16)] for hf in (0, 1)]
    z0 = chain(h2[:, rows[0]], P["W3"][None, rows[0]], range(16))
    z1 = chain(h2[:, rows[1]], P["W3"][None, rows[1]], range(16))
    return ((z0 + z1) + P["b3"])[:, 0]


def layer2(P, h, order):
    W2a = np.zeros((32, 160), f32)
    W2a[:, :150] = P["W2"]
    W2a[:, 150] = P["b2"]
    ha = np.zeros((h.shape[0], 160), f32)
    ha[:, :150] = h
    ha[:, 150] = 1
    return np.maximum(chain(ha, W2a, order), 0)


def kernel_r3(P, inp):
    """general16 as rounds 1-3 built it: folded layer 1 in slot pairs (st, st+8), layer 2 in accumulator order."""
    x16 = np.concatenate([inp, np.ones((inp.shape[0], 1), f32)], 1)
    ks = sum(([st, st + 8] for st in range(8)), [])
    h = np.maximum(chain(x16, folded(P), ks), 0)
    h2 = layer2(P, h, sum((old_unit_order(m) for m in range(5)), []))
    rows = [[(q & 3) + 8 * (q >> 2) + 4 * hf for q in range(16)] for hf in (0, 1)]
    z0 = chain(h2[:, rows[0]], P["W3"][None, rows[0]], range(16))
    z1 = chain(h2[:, rows[1]], P["W3"][None, rows[1]], range(16))
    return sigmoid(((z0 + z1) + P["b3"])[:, 0])


def kernel_general16(P, inp):
    """enc_kernel now: the reference's operations all the way (k = 0..14, + b1; fma batch norm; k = 0..149, + b2; the AVX-512
    gemv's order; Sleef's exp)."""
    alpha, beta = bn_pairs(P)
    y = chain(inp, P["W1"], range(15)) + P["b1"]
    h = np.maximum(fma(y, np.broadcast_to(alpha, y.shape), np.broadcast_to(beta, y.shape)), 0)
    return sigmoid_sleef(mkl_avx512_gemv_32(layer2(P, h, range(152)), P["W3"], P["b3"]))


def kernel_csite12(P, inp):
    """enc_csite_kernel now: x0..x8 as a chain, then ONE addition of the site's c = b1 + sum_e W1[:, 9+e] * e (a chain that
    starts at b1), then as enc_kernel."""
    alpha, beta = bn_pairs(P)
    c = chain(inp[:, 9:15], P["W1"][:, 9:15], range(6), acc=np.broadcast_to(P["b1"], (inp.shape[0], 150)).astype(f32).copy())
    y = chain(inp, P["W1"], range(9)) + c
    h = np.maximum(fma(y, np.broadcast_to(alpha, y.shape), np.broadcast_to(beta, y.shape)), 0)
    return sigmoid(layer3_halves(P, layer2(P, h, range(152))))


def use(got, want):
    want = want.astype(f64)
    return np.abs(got.astype(f64) - want) / (1e-8 + 1e-5 * np.abs(want))


def line(label, got, ref):
    u = use(got, ref)
    print("  %-58s identical %.4f   rms use %.4f   p99.99 %.4f   worst %.4f   beyond %d"
          % (label, float((got.view(np.uint32) == ref.view(np.uint32)).mean()), float(np.sqrt((u * u).mean())), float(np.quantile(u, 0.9999)),
             float(u.max()), int((u > 1).sum())), flush=True)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "hek293t_glori"
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000
    P = unpack(np.fromfile(asset_path("weights_%s.bin" % name), np.float32))
    d = synthetic.make_sites(1_000_000, 20, seed=20250328, prefix_sites=S)
    R = int(d["off"][S])
    emb = P["E"][np.repeat(d["site_kmers"][:S].astype(np.int64), 20, axis=0)].reshape(-1, 6)
    inp = np.concatenate([d["X"][:R].reshape(-1, 9), emb], 1).astype(f32)
    ref = np.load(os.path.join(REPO, "tests", "golden", "reference_at_scale.npz"))["uniform_%s_readprob" % name][:R]
    print(name, R, "reads; against the REFERENCE's read probabilities:")
    line("torch's order, layer 3 as one chain of 32", torch_like(P, inp), ref)
    line("torch's order, layer 3 as MKL's AVX-512 gemv, Sleef's exp", sigmoid_sleef(mkl_avx512_gemv_32(torch_like(P, inp, upto="h2"), P["W3"], P["b3"])), ref)
    line("rounds 1-3 general16 (emulated)", kernel_r3(P, inp), ref)
    eg, ec = kernel_general16(P, inp), kernel_csite12(P, inp)
    line("general16 (emulated)", eg, ref)
    line("csite12 (emulated)", ec, ref)
    hip = os.path.join(REPO, "gpurun_out", "read_probs_%s.npz" % name)
    if os.path.exists(hip):
        h = np.load(hip)
        line("general16 on the GPU", h["general16"][:R], ref)
        line("csite12 on the GPU", h["csite12"][:R], ref)
        for lab, e, g in (("general16", eg, h["general16"][:R]), ("csite12", ec, h["csite12"][:R])):
            ulp = np.abs(e.view(np.int32).astype(np.int64) - g.view(np.int32).astype(np.int64))
            print("  %s emulated vs on the GPU: identical %.4f, within 1 ulp %.4f, worst %d ulp (the GPU's expf against a correctly rounded exp)"
                  % (lab, float((ulp == 0).mean()), float((ulp <= 1).mean()), int(ulp.max())))


if __name__ == "__main__":
    main()
