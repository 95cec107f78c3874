#!/usr/bin/env python3
"""Histogram of y = -|a - b| over the five `plus` operations of a Forward cell (log_sum_exp, src/include/coati/utils.hpp:134-156),
per LANE and per WAVEFRONT-instruction of forward_l1's lane mapping (lane t does row k - t of its W columns at step k): how
often could a wave-uniform early-out skip log1pf (all 64 lanes y <= -16), skip expf too (all y < -104), or skip the k = 1
route of log1pf (every lane either y <= -16 or e < 0.41422)?  CPU only: the oracle's Forward matrices, numpy.
usage: forward_y_hist.py [columns per lane W = 8]    ->  profiles/r04/forward_y_histogram.txt"""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from coati_amd import host
from oracle import pyoracle as orc
table, consts = host.set_subst("mar-mg"), host.gap_consts()
ng, gs, go, ge = [np.float32(x) for x in consts]
print("consts", consts)
W = int(sys.argv[1]) if len(sys.argv)>1 else 8
tot = {}
def acc(name, y, wave_all):
    d = tot.setdefault(name, np.zeros(8))
    d[0] += y.size
    d[1] += (y <= -16).sum()
    d[2] += (y < -103.97).sum()
    d[3] += (y > -0.8814).sum()   # not small: e >= 0.41422
    # wave-level (steps, 64 lanes)
    d[4] += wave_all.shape[0]
    d[5] += (wave_all <= -16).all(axis=1).sum()
    d[6] += (wave_all < -103.97).all(axis=1).sum()
    d[7] += ((wave_all <= -16) | (wave_all < -0.8814)).all(axis=1).sum()  # all lanes either skip or small route
for p in range(3):
    a_cat, a_off, b_cat, b_off = host.synth_encoded(p, 1)
    a, b = a_cat, b_cat
    M, D, I = orc.fill("log", table, consts, 1, a, b)[:3]
    la, lb = len(a), len(b)
    # matrices are (la+1) x (lb+1)
    f = np.float32
    dgM, dgD, dgI = M[:-1,:-1], D[:-1,:-1], I[:-1,:-1]
    upM, upD, upI = M[:-1,1:], D[:-1,1:], I[:-1,1:]
    lfM, lfI = M[1:,:-1], I[1:,:-1]
    s = table[a][:, b].astype(f)
    with np.errstate(invalid='ignore', over='ignore'):
        m2m = ((dgM+ng)+ng)+s; d2m = (dgD+gs)+s; i2m = ((dgI+gs)+ng)+s
        m2d = (upM+ng)+go; i2d = (upI+gs)+go; d2d = upD+ge
        m2i = lfM+go; i2i = lfI+ge
        def lp(a_, b_):
            hi = np.maximum(a_, b_); y = -np.abs(a_-b_)
            y = np.where(np.isnan(y), -np.inf, y)
            e = np.exp(y.astype(np.float64)).astype(f)
            return (hi + np.where(y <= -16, e, np.log1p(e.astype(np.float64)).astype(f))).astype(f), y
        p1, y1 = lp(m2m, d2m); _, y2 = lp(p1, i2m)
        p3, y3 = lp(m2d, d2d); _, y4 = lp(p3, i2d)
        _, y5 = lp(m2i, i2i)
    # wave layout: lane t handles columns t*W..t*W+W-1 of strip; at step k row k-t.  For each (strip, step k, c) the 64 lanes' y
    for name, y in (("M1", y1), ("M2", y2), ("D1", y3), ("D2", y4), ("I", y5)):
        waves = []
        nstrip = (lb + 64*W - 1)//(64*W)
        for st in range(nstrip):
            col0 = st*64*W
            for c in range(W):
                cols = col0 + np.arange(64)*W + c
                ok = cols < lb
                # steps k = 0..la+62 ; row = k - t
                k = np.arange(0, la+63)[:,None]; t = np.arange(64)[None,:]
                r = k - t
                valid = (r>=0)&(r<la)&ok[None,:]
                yy = np.where(valid, y[np.clip(r,0,la-1), np.clip(cols,0,lb-1)[None,:]], -np.inf)  # idle lanes: treat as -inf (can be forced)
                waves.append(yy)
        acc(name, y, np.concatenate(waves))
for name, d in tot.items():
    print(f"{name}: lanes y<=-16 {d[1]/d[0]:.3f}  y<-104 {d[2]/d[0]:.3f}  e>=0.414 {d[3]/d[0]:.3f} | waves all<=-16 {d[5]/d[4]:.3f} all<-104 {d[6]/d[4]:.3f} all(skip|small) {d[7]/d[4]:.3f}")
