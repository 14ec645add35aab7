#!/usr/bin/env python3
"""Brute-force the LDS slot swizzle of the slab kernels: pixel Pl's 16-byte slot s is stored at slot s ^ f(Pl).
Counts extra LDS cycles (bank conflicts) of the ds_read_b128 fragment reads for the DENSE lane->pixel maps of the
four ResNet-18 feature-map widths, over all tile phases and tap shifts, for a family of candidate f."""
import itertools
import sys

GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]


def lane_pixels(W, H, i0):
    """PF position (relative) of dense pixel i0 + l, l = 0..31"""
    P, S = W + 1, (H + 1) * (W + 1)
    out = []
    for l in range(32):
        i = i0 + l
        n, rem = divmod(i, H * W)
        y, x = divmod(rem, W)
        out.append(n * S + y * P + x)
    return out


def conflicts(f, W, H, slot=0):
    extra = total = 0
    period = H * W * 32
    for i0 in range(0, min(period, 4096 * 4), 32):
        px = lane_pixels(W, H, i0)
        for shift in range(0, 2 * (W + 1) + 3):
            for g in GROUPS:
                cols = {}
                for l in g:
                    Pl = px[l] + shift
                    col = (Pl & 1) * 8 + (slot ^ f(Pl))
                    cols[col] = cols.get(col, 0) + 1
                extra += max(cols.values()) - 1
                total += 1
    return extra / total


cands = {'(Pl>>1)&7 [current]': lambda p: (p >> 1) & 7}
for a, b in itertools.combinations(range(1, 8), 2):
    cands['((Pl>>%d)^(Pl>>%d))&7' % (a, b)] = (lambda a, b: lambda p: ((p >> a) ^ (p >> b)) & 7)(a, b)
for a, b, c in itertools.combinations(range(1, 8), 3):
    cands['((Pl>>%d)^(Pl>>%d)^(Pl>>%d))&7' % (a, b, c)] = (lambda a, b, c: lambda p: ((p >> a) ^ (p >> b) ^ (p >> c)) & 7)(a, b, c)
for k in (3, 5, 7, 9, 11, 13):
    cands['((Pl*%d)>>1)&7' % k] = (lambda k: lambda p: ((p * k) >> 1) & 7)(k)
    cands['((Pl*%d)>>2)&7' % k] = (lambda k: lambda p: ((p * k) >> 2) & 7)(k)
    cands['((Pl*%d)>>3)&7' % k] = (lambda k: lambda p: ((p * k) >> 3) & 7)(k)
res = []
for name, f in cands.items():
    r = [conflicts(f, W, W) for W in (8, 16, 32, 64)]
    res.append((sum(r), r, name))
res.sort()
for tot, r, name in res[:12]:
    print('%-34s extra cycles per group-read: W=8 %.3f  W=16 %.3f  W=32 %.3f  W=64 %.3f' % (name, *r))
cur = [x for x in res if 'current' in x[2]][0]
print('%-34s extra cycles per group-read: W=8 %.3f  W=16 %.3f  W=32 %.3f  W=64 %.3f' % (cur[2], *cur[1]))
