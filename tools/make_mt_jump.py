#!/usr/bin/env python3
"""Writes m6anet_amd/assets/mt19937_jump.bin: the GF(2) jump polynomials of MT19937 that let the GPU generate the NumPy
legacy stream (np.random.seed -> RandomState: the stream m6anet's site sampling draws from, m6anet/scripts/inference.py:86,
m6anet/utils/inference_utils.py:85) in PARALLEL SEGMENTS instead of one 623-words-per-step chain.

Published facts used (Matsumoto & Nishimura 1998; Haramoto et al. 2008, "Efficient jump ahead for F2-linear RNGs"):
the word sequence x[k] of MT19937 (x[0..623] the seeded state, outputs = temper(x[624+j])) is linear over GF(2) with a
characteristic polynomial phi(t) of degree 19937, the same for every bit position, so for any D
        x[k + D] = XOR over the set bits t of (t^D mod phi) of x[k + t]          (k >= 1),
and because tempering is a linear bijection on words the outputs obey the same relation.  phi is recovered here with
Berlekamp-Massey from one bit sequence (135 terms, degree 19937), t^D mod phi by square-and-multiply on Python integers.

File layout (little-endian): magic "M6AMTJP1", u32 n_regimes, u32 n_per_regime, u32 words_per_poly (312 u64), u32 back (512);
then per regime: u64 G (segment length in words), followed by n_per_regime polynomials r_i = t^(i*G - back) mod phi, i = 1..n,
each 312 u64 (bit t of word t/64 = coefficient of t^t).  Segment i of a stream starts at output word i*G; its 1078-word
history x[i*G - 454 .. i*G + 623] is the XOR above taken over x[58 + m + t] (m = 0..1077): mt_jump_kernel, m6a_pool_rtab.hip.

    python tools/make_mt_jump.py            (about a minute; verifies every regime against numpy before writing)
"""
import os
import struct
import sys

import numpy as np

DEG = 19937
BACK = 512
REGIMES = [1 << 16, 1 << 20, 1 << 24]
N_PER = 31


def mt_words(seed, n):
    x = [0] * n
    x[0] = seed
    for i in range(1, 624):
        x[i] = (1812433253 * (x[i - 1] ^ (x[i - 1] >> 30)) + i) & 0xffffffff
    for k in range(n - 624):
        y = (x[k] & 0x80000000) | (x[k + 1] & 0x7fffffff)
        x[k + 624] = x[k + 397] ^ (y >> 1) ^ (0x9908b0df if y & 1 else 0)
    return x


def berlekamp_massey(bits):
    C, B, L, m, S = 1, 1, 0, 1, 0
    for i, b in enumerate(bits):
        S = (S << 1) | b
        if bin(C & S).count("1") & 1:
            T = C
            C ^= B << m
            if 2 * L <= i:
                L, B, m = i + 1 - L, T, 1
            else:
                m += 1
        else:
            m += 1
    return C, L


def characteristic_polynomial():
    x = mt_words(5489, 2 * DEG + 1300)
    C, L = berlekamp_massey([v & 1 for v in x[1:1 + 2 * DEG + 600]])
    assert L == DEG
    # connection polynomial C(t) = 1 + c1 t + ... (s[i] = sum c_j s[i-j])  ->  phi(t) = t^L C(1/t)
    phi = 0
    for j in range(L + 1):
        if (C >> j) & 1:
            phi |= 1 << (L - j)
    assert bin(phi).count("1") == 135 and phi >> DEG == 1
    return phi


def reduce_mod(a, phi):
    while True:
        d = a.bit_length() - 1
        if d < DEG:
            return a
        a ^= phi << (d - DEG)


def mulmod(a, b, phi):
    r = 0
    while b:
        low = b & -b
        r ^= a << (low.bit_length() - 1)
        b ^= low
    return reduce_mod(r, phi)


def powmod_t(e, phi):
    """t^e mod phi"""
    r, base = 1, 2                 # the polynomials "1" and "t"
    while e:
        if e & 1:
            r = mulmod(r, base, phi)
        base = mulmod(base, base, phi)
        e >>= 1
    return r


def main():
    out_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "m6anet_amd", "assets", "mt19937_jump.bin")
    phi = characteristic_polynomial()
    print("phi: degree %d, %d terms" % (phi.bit_length() - 1, bin(phi).count("1")), file=sys.stderr)
    blob = [b"M6AMTJP1", struct.pack("<IIII", len(REGIMES), N_PER, 312, BACK)]
    # the check stream: numpy's own (outputs obey the same relation as the untempered words)
    n_check = REGIMES[1] * 3 + DEG + 4096
    out = np.frombuffer(np.random.RandomState(12345).bytes(4 * n_check), dtype=np.uint32)
    for G in REGIMES:
        step = powmod_t(G, phi)
        r = powmod_t(G - BACK, phi)
        polys = []
        for i in range(1, N_PER + 1):
            polys.append(r)
            D = i * G - BACK
            if D + 2000 + DEG < n_check:                     # verify against numpy where the check stream reaches
                taps = [t for t in range(DEG) if (r >> t) & 1]
                ks = np.arange(0, 1500)
                acc = np.zeros(len(ks), np.uint32)
                for t in taps:
                    acc ^= out[ks + t]
                assert np.array_equal(acc, out[ks + D]), (G, i)
            r = mulmod(r, step, phi)
        # every regime's polynomials are powers of the same t: cross-check the first of this regime against plain powering
        assert polys[1] == powmod_t(2 * G - BACK, phi)
        blob.append(struct.pack("<Q", G))
        for p in polys:
            blob.append(p.to_bytes(312 * 8, "little"))
        print("G = 2^%d: %d polynomials, %d..%d terms" % (G.bit_length() - 1, len(polys), min(bin(p).count("1") for p in polys),
                                                           max(bin(p).count("1") for p in polys)), file=sys.stderr)
    data = b"".join(blob)
    with open(out_path, "wb") as f:
        f.write(data)
    print("%s: %d bytes" % (out_path, len(data)), file=sys.stderr)


if __name__ == "__main__":
    main()
