#!/usr/bin/env python3
"""Numerical experiment behind HISTORY.md section 8 (test infrastructure: it uses the oracle): would layer 2 of the read
encoder (150 -> 32, 68 % of the FLOPs) on the bf16 matrix pipe -- float32 operands split exactly into three bf16 parts,
6 or 9 partial products, float32 accumulation -- stay inside the reference's read-probability bar (np.allclose,
rtol 1e-5, atol 1e-8)?  Prints, per checkpoint, the largest |a-b| / (1e-8 + 1e-5 |b|) ("tolerance used") against the
reference's captured probabilities on the bundled reads, and against the float32 oracle on the bundled and on synthetic
reads, for: layer 2 in float64 (the exact answer), the 6- and 9-product splits, and the plain 3-product bf16 split.
NumPy emulation (each 16-deep block of products is summed exactly and rounded once, which flatters the hardware).

    python tests/experiment_bf16_split.py
"""
import numpy as np, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from m6anet_amd import synthetic
from oracle import m6a_oracle as orc
f32=np.float32
def bf16_trunc(x):
    return (x.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)
def split3(x):
    h=bf16_trunc(x); r=(x-h).astype(f32); m=bf16_trunc(r); l=(r-m).astype(f32)
    # l must be exactly representable in bf16
    assert np.array_equal(bf16_trunc(l), l)
    return h,m,l
O_E,O_W1,O_B1,O_G,O_BE,O_MU,O_VAR,O_W2,O_B2,O_W3,O_B3=0,132,2382,2532,2682,2832,2982,3132,7932,7964,7996
def encode(w, X, km, off, mode):
    E=w[O_E:O_W1].reshape(66,2); W1=w[O_W1:O_B1].reshape(150,15); b1=w[O_B1:O_G]
    gam,bet,mu,var=w[O_G:O_BE],w[O_BE:O_MU],w[O_MU:O_VAR],w[O_VAR:O_W2]
    W2=w[O_W2:O_B2].reshape(32,150); b2=w[O_B2:O_W3]; W3=w[O_W3:O_B3]; b3=w[O_B3]
    alpha=(gam*(f32(1)/np.sqrt(var+f32(1e-5)))).astype(f32); shift=(bet-mu*alpha).astype(f32)
    W1f=(alpha[:,None]*W1).astype(f32); b1f=(alpha*b1+shift).astype(f32)
    n=np.diff(off); site=np.repeat(np.arange(len(n)),n)
    emb=E[km[site]].reshape(len(site),6)
    F=np.concatenate([X,emb],1).astype(f32)
    # layer 1 as the kernel does it: BatchNorm folded, one float32 FMA chain over the 15 inputs, bias last
    def fma(a, b, c):                       # fmaf: the product is exact in float64, one rounding to float32
        return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)
    H1=np.zeros((len(F),150),f32)
    for k in range(15): H1=fma(F[:,k:k+1], W1f[:,k][None,:], H1)
    H1=np.maximum((H1+b1f).astype(f32),0)
    if mode=='f32chain':                    # the kernel today: exact float32 FMA chain (v_mfma_f32_32x32x2_f32)
        acc=np.zeros((len(H1),32),f32)
        for k in range(150): acc=fma(H1[:,k:k+1], W2[:,k][None,:], acc)
        acc=(acc+b2).astype(f32)
    elif mode=='f64':
        acc=(H1.astype(np.float64)@W2.T.astype(np.float64)+b2).astype(f32)
    else:
        hh,hm,hl=split3(H1); wh,wm,wl=split3(W2)
        acc=np.zeros((len(H1),32),f32)
        prods=[(hh,wh),(hh,wm),(hm,wh),(hm,wm),(hh,wl),(hl,wh)]
        if mode=='split9': prods+= [(hm,wl),(hl,wm),(hl,wl)]
        if mode=='split3': prods=prods[:3]
        # K blocks of 16: each MFMA: acc = round_f32(acc + sum16 exact products) (optimistic single rounding)
        order=list(range(len(prods)))[::-1]  # small terms first
        for kb in range(0,160,16):
            ks=slice(kb,min(kb+16,150))
            for pi in order:
                a,b=prods[pi]
                acc=(acc.astype(np.float64)+a[:,ks].astype(np.float64)@b[:,ks].T.astype(np.float64)).astype(f32)
        acc=(acc+b2).astype(f32)
    H2=np.maximum(acc,0)
    z=(H2.astype(np.float64)@W3.astype(np.float64)+b3).astype(f32)
    return (f32(1)/(f32(1)+np.exp(-z).astype(f32))).astype(f32)
def used(a,b): return float(np.max(np.abs(a.astype(np.float64)-b)/(1e-8+1e-5*np.abs(b))))
g=np.load(__import__('os').path.dirname(__import__('os').path.abspath(__file__))+'/golden/bundled_inputs.npz'); ref=np.load(__import__('os').path.dirname(__import__('os').path.abspath(__file__))+'/golden/bundled_readprob.npz')
d=synthetic.make_sites(3000,(20,90),seed=3)
for name in ['hct116','arabidopsis','hek293t_glori','hek293t_m6ace']:
    w=np.fromfile(__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))),'m6anet_amd','assets','weights_%s.bin'%name),f32)
    o_b=orc.encode_reads(w,g['X'],g['site_kmers'],g['off']); o_s=orc.encode_reads(w,d['X'],d['site_kmers'],d['off'])
    row=[name]
    for mode in ['f32chain','f64','split6','split9','split3']:
        pb=encode(w,g['X'],g['site_kmers'],g['off'],mode); ps=encode(w,d['X'],d['site_kmers'],d['off'],mode)
        row.append('%s: ref %.2f orc_b %.2f orc_s %.2f'%(mode,used(pb,ref[name]),used(pb,o_b),used(ps,o_s)))
    row.append('oracle vs ref %.2f'%used(o_b,ref[name]))
    print(' | '.join(row))
