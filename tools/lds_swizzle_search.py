#!/usr/bin/env python3
"""Brute-force search of conflict-free LDS layouts for the wave-local exchanges of the 8-points-per-thread row stage
(csrc/fft_rowqe8.hpp).  Bank model: /opt/skills/guides/MI355X_MICROARCH.md section LDS (lane groups per instruction)."""
import itertools, sys

RD128 = [list(range(0,4))+list(range(12,16))+list(range(20,28)), list(range(4,12))+list(range(16,20))+list(range(28,32))]
RD128 = RD128 + [[x+32 for x in g] for g in RD128]
WR128 = [list(range(8*i, 8*i+8)) for i in range(8)]
RD64 = [list(range(0,32)), list(range(32,64))]
WR64 = [list(range(16*i, 16*i+16)) for i in range(4)]

def conflicts(groups, pos, esz, nbanks):
    """max number of distinct addresses on one bank within a lane group (1 = conflict-free); pos: lane -> element index"""
    worst = 1
    for g in groups:
        per = {}
        for l in g:
            a = pos[l] * esz
            for d in range(esz // 4):
                b = ((a // 4) + d) % nbanks
                per.setdefault(b, set()).add(a)
        worst = max(worst, max(len(s) for s in per.values()))
    return worst

def check(posfn, wr_lane, rd_lane, esz):
    """posfn(c1, a, b); wr: fixed reg r -> lane l holds element wr_lane(l, r); rd likewise.  returns (worst write, worst read)"""
    rdg, wrg = (RD128, WR128) if esz == 16 else (RD64, WR64)
    rdb, wrb = (64, 32)
    ww = rw = 1
    for r in range(8):
        ww = max(ww, conflicts(wrg, [posfn(*wr_lane(l, r)) for l in range(64)], esz, wrb))
        rw = max(rw, conflicts(rdg, [posfn(*rd_lane(l, r)) for l in range(64)], esz, rdb))
    return ww, rw

def bijective(posfn):
    s = {posfn(c1, a, b) for c1 in range(8) for a in range(8) for b in range(8)}
    return len(s) == 512 and min(s) == 0 and max(s) == 511

# GF(2)-linear 3x3 maps as tables
def lin_maps():
    out = []
    for cols in itertools.product(range(8), repeat=3):
        out.append(tuple((cols[0] if v & 1 else 0) ^ (cols[1] if v & 2 else 0) ^ (cols[2] if v & 4 else 0) for v in range(8)))
    return out
LM = lin_maps()
SIMPLE = [m for m in LM if sum(1 for v in m if v) <= 8][:]

def search(name, wr_lane, rd_lane, esz, limit=3):
    # element (c1, a, b) -> 64*c1 + 8*hi + lo ; hi = a ^ P(b) ^ Q(c1), lo = b ^ R(a) ^ S(c1)
    found = []
    cand = [LM[0]] + [m for m in LM if m != LM[0]]
    small = [m for m in cand if m in (LM[0],) or True]
    # restrict to a modest family: zero, identity, bit-reversal, shifts
    fam = {}
    ident = tuple(range(8)); zero = (0,)*8
    fam['0'] = zero; fam['I'] = ident
    fam['rev'] = tuple(((v&1)<<2)|(v&2)|((v>>2)&1) for v in range(8))
    fam['shl1'] = tuple((v<<1)&7 for v in range(8)); fam['shr1'] = tuple(v>>1 for v in range(8))
    fam['shl2'] = tuple((v<<2)&7 for v in range(8)); fam['shr2'] = tuple(v>>2 for v in range(8))
    fam['b0'] = tuple(v&1 for v in range(8)); fam['b1'] = tuple((v>>1)&1 for v in range(8)); fam['b2'] = tuple((v>>2)&1 for v in range(8))
    fam['b0s1'] = tuple((v&1)<<1 for v in range(8)); fam['b0s2'] = tuple((v&1)<<2 for v in range(8))
    fam['b1s1'] = tuple(((v>>1)&1)<<1 for v in range(8)); fam['b1s2'] = tuple(((v>>1)&1)<<2 for v in range(8))
    fam['b2s1'] = tuple(((v>>2)&1)<<1 for v in range(8)); fam['b2s2'] = tuple(((v>>2)&1)<<2 for v in range(8))
    keys = list(fam)
    for p, q, r, s in itertools.product(keys, repeat=4):
        P, Q, Rm, S = fam[p], fam[q], fam[r], fam[s]
        fn = lambda c1, a, b: 64 * c1 + 8 * (a ^ P[b] ^ Q[c1]) + (b ^ Rm[a] ^ S[c1])
        if not bijective(fn): continue
        ww, rw = check(fn, wr_lane, rd_lane, esz)
        if ww == 1 and rw == 1:
            found.append((p, q, r, s))
            if len(found) >= limit: break
    print(name, 'esz', esz, '->', found if found else 'NONE in family')
    return found

if __name__ == '__main__':
    # E1: element (c1, hi = t2-index l>>3, lo = l0 = l&7).  write: reg c1, lane l -> (c1, l>>3, l&7); read: reg t2, lane (l0 + 8 c1) -> (c1, t2, l0)
    e1w = lambda l, r: (r, l >> 3, l & 7)
    e1r = lambda l, r: (l >> 3, r, l & 7)
    # E2: element (c1, c2, l0).  write: reg c2, lane (l0 + 8 c1) -> (c1, c2, l0); read: reg l0, lane (c2 + 8 c1) -> (c1, c2, l0)
    e2w = lambda l, r: (l >> 3, r, l & 7)
    e2r = lambda l, r: (l >> 3, l & 7, r)
    for esz in (16, 8):
        # the transposed exchanges of the forward transform swap the roles of read and write
        f1 = search('E1 ', e1w, e1r, esz)
        f1t = search('E1T', e1r, e1w, esz)
        f2 = search('E2 ', e2w, e2r, esz)
        f2t = search('E2T', e2r, e2w, esz)
