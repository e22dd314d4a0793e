"""LDS bank-conflict model of the fused DCT kernels (bifurcationkit.jl_amd/csrc/dct_fast.hip, index math of dct_core.h),
CPU only.  The banking rules are the ones of /opt/skills/guides/MI355X_MICROARCH.md, section LDS:

  ds_read_b128   4 lane groups {0-3,12-15,20-27} {4-11,16-19,28-31} {32-35,44-47,52-59} {36-43,48-51,60-63}, one LDS cycle
                 each when conflict-free; a 16-byte access occupies one of the 16 slots of the 256-byte bank row
  ds_write_b128  8 contiguous groups of 8 lanes, banks modulo 32 dwords (8 slots of a 128-byte row)
  ds_read_b64    2 groups of 32 lanes, 32 8-byte slots of the 256-byte row
  identical addresses broadcast; every further distinct address on a busy slot within a group costs one more cycle.

For every LDS instruction of a phase the script lists lanes -> byte addresses exactly as the kernel forms them and reports
ideal cycles, extra (conflict) cycles and their share -- the quantity rocprofv3 reports as SQ_LDS_BANK_CONFLICT /
SQ_LDS_IDX_ACTIVE (measured 35-44 % for these kernels, profiles/r2_sq_stall_breakdown.txt).  It then tries alternative
table layouts.  Usage: python sim.py [N] [LT]"""
import sys
from collections import defaultdict

R128 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
R128 = R128 + [[l + 32 for l in g] for g in R128]
W128 = [list(range(8 * j, 8 * j + 8)) for j in range(8)]
R64 = [list(range(0, 32)), list(range(32, 64))]


def cycles(kind, addrs):
    """addrs: 64 byte addresses (None = inactive lane).  Returns (ideal, extra) LDS-array cycles."""
    groups, slot_bytes, nslots = {"r128": (R128, 16, 16), "w128": (W128, 16, 8), "r64": (R64, 8, 32)}[kind]
    ideal = extra = 0
    for g in groups:
        per = defaultdict(set)
        for l in g:
            a = addrs[l]
            if a is not None:
                per[(a // slot_bytes) % nslots].add(a)
        if per:
            ideal += 1
            extra += max(len(v) for v in per.values()) - 1
    return ideal, extra


def swz(i):
    return i ^ (((i >> 4) ^ (i >> 8)) & 15)


def bitrev(i, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (i & 1)
        i >>= 1
    return r


class Layout:
    """LDS layout of one tile: pair lines of (N + 1) complex values, then the tables."""
    def __init__(self, N, npairs, pstride=None, tw_map=lambda j: j, ew_map=lambda k: k, swz=swz):
        self.N, self.npairs = N, npairs
        self.pstride = N + 1 if pstride is None else pstride
        self.z0 = 0
        self.tw0 = 16 * npairs * self.pstride
        self.tw_map, self.ew_map, self.swz = tw_map, ew_map, swz
        self.tw_len = max(tw_map(j) for j in range(N // 2)) + 1
        self.ew0 = self.tw0 + 16 * self.tw_len
        self.ew_len = max(ew_map(k) for k in range(N // 2 + 1)) + 2
        self.lam0 = self.ew0 + 16 * self.ew_len

    def z(self, pair, idx):          # idx: logical position BEFORE the swizzle
        return 16 * (pair * self.pstride + self.swz(idx))

    def zs(self, pair, sidx):        # already swizzled position
        return 16 * (pair * self.pstride + sidx)

    def tw(self, j):
        return self.tw0 + 16 * self.tw_map(j)

    def ew(self, k):
        return self.ew0 + 16 * self.ew_map(k)

    def lam(self, k):
        return self.lam0 + 8 * k


def run_waves(nthreads, lane_fn):
    """lane_fn(tid) -> list of (kind, addr or None) of equal length for every lane; returns dict kind -> [ideal, extra]"""
    tot = defaultdict(lambda: [0, 0])
    per_lane = [lane_fn(t) for t in range(nthreads)]
    ninstr = len(per_lane[0])
    for w0 in range(0, nthreads, 64):
        for i in range(ninstr):
            kind = per_lane[w0][i][0]
            addrs = [per_lane[w0 + l][i][1] for l in range(64)]
            a, b = cycles(kind, addrs)
            tot[kind][0] += a
            tot[kind][1] += b
    return tot


def zpass_phases(L, N, LT, NT, split=False, ax0=False):
    """Phases of dct_fused_kernel<NT, 2, false> (axis >= 1 lane maps) on one tile; split: the 512-lane experiment;
    ax0: the lane maps of the contiguous-axis kernels (a wave stays on one pair line; an item owns two first-stage groups)."""
    bits = N.bit_length() - 1
    G = N >> 3
    npairs = LT // 2
    pb = npairs.bit_length() - 1
    out = {}

    hbits = bits - 4

    def first(tid, store=True):
        kind = "w128" if store else "r128"
        if ax0:
            pair, gp = tid >> hbits, tid & ((1 << hbits) - 1)
            ins = []
            for g in (gp, G - 1 - gp):
                sb = L.swz(bitrev(g, bits - 3) << 3)
                ins += [(kind, L.zs(pair, sb ^ q)) for q in range(8)]
            return ins
        pair, gp = tid & (npairs - 1), tid >> pb
        sb = L.swz(bitrev(gp, bits - 3) << 3)
        return [(kind, L.zs(pair, sb ^ q)) for q in range(8)]
    nfirst = npairs * (G >> 1) if ax0 else npairs * G
    for rep in range(max(1, nfirst // NT)):
        t = run_waves(min(NT, nfirst), lambda tid, rep=rep: first(tid + rep * NT))
        for k, v in t.items():
            o = out.setdefault("first stage (stores)", defaultdict(lambda: [0, 0]))
            o[k][0] += v[0]; o[k][1] += v[1]
        t = run_waves(min(NT, nfirst), lambda tid, rep=rep: first(tid + rep * NT, store=False))
        for k, v in t.items():
            o = out.setdefault("last stage (loads)", defaultdict(lambda: [0, 0]))
            o[k][0] += v[0]; o[k][1] += v[1]

    def middle(w, lh, R):
        gbits = bits - R
        pair, g = w >> gbits, w & ((1 << gbits) - 1)
        lo = g & ((1 << lh) - 1)
        base = ((g >> lh) << (lh + R)) + lo
        ins = [("r128", L.z(pair, base + (q << lh))) for q in range(1 << R)]
        sh = bits - lh - R
        if R == 3:
            ins += [("r128", L.tw(lo << (sh + 2))), ("r128", L.tw(lo << (sh + 1))), ("r128", L.tw(lo << sh))]
        else:
            for s in range(R):
                for q in range(1 << R):
                    if q & (1 << s):
                        continue
                    pos = ((q & ((1 << s) - 1)) << lh) + lo
                    ins.append(("r128", L.tw(pos << (bits - (lh + s) - 1))))
        ins += [("w128", L.z(pair, base + (q << lh))) for q in range(1 << R)]
        return ins
    lh = 3
    while lh < bits - 3:
        R = 3 if bits - 3 - lh >= 3 else bits - 3 - lh
        ngr = npairs << (bits - R)
        name = "LDS stage lh=%d R=%d (x2: forward and inverse)" % (lh, R)
        for rep in range(max(1, ngr // NT)):
            t = run_waves(min(NT, ngr), lambda tid, rep=rep: middle(tid + rep * NT, lh, R))
            o = out.setdefault(name, defaultdict(lambda: [0, 0]))
            for k, v in t.items():
                o[k][0] += 2 * v[0]; o[k][1] += 2 * v[1]
        lh += R

    def merged(tid):
        if split:
            h, pair, t = tid & 1, (tid >> 1) & (npairs - 1), tid >> (pb + 1)
            groups = [t if h == 0 else ((G >> 1) if t == 0 else G - t)]
        elif ax0:
            pair, t = tid >> hbits, tid & ((1 << hbits) - 1)
            groups = [t, (G >> 1) if t == 0 else G - t]
        else:
            pair, t = tid & (npairs - 1), tid >> pb
            groups = [t, (G >> 1) if t == 0 else G - t]
        ins = []
        for g in groups:
            ins += [("r128", L.z(pair, g + q * G)) for q in range(8)]
            ins += [("r128", L.tw(g << 2)), ("r128", L.tw(g << 1)), ("r128", L.tw(g))]
        for g in groups:
            for q in range(8):
                k = g + q * G
                ins.append(("r128", L.ew(k if k <= N // 2 else N - k)))
                ins.append(("r64", L.lam(k)))
        for g in groups:
            if split:
                ins += [("r128", L.tw(g << 2)), ("r128", L.tw(g << 1)), ("r128", L.tw(g))]
            ins += [("w128", L.z(pair, g + q * G)) for q in range(8)]
        return ins
    nmid = npairs * (G >> 1) * (2 if split else 1)
    t = run_waves(nmid, merged)
    out["merged middle"] = t
    return out


def report(title, phases):
    print(title)
    ti = te = 0
    for name, t in phases.items():
        i = sum(v[0] for v in t.values()); e = sum(v[1] for v in t.values())
        ti += i; te += e
        detail = "  ".join("%s %d+%d" % (k, v[0], v[1]) for k, v in sorted(t.items()))
        print("  %-48s ideal %6d  extra %6d  (%4.1f %% of cycles)   %s" % (name, i, e, 100.0 * e / (i + e), detail))
    print("  %-48s ideal %6d  extra %6d  (%4.1f %% of cycles)" % ("TOTAL per tile", ti, te, 100.0 * te / (ti + te)))
    return ti, te


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    LT = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    base = Layout(N, LT // 2)
    report("product z round trip: N=%d, %d lines, 256 lanes, current layout" % (N, LT), zpass_phases(base, N, LT, 256))
    # twiddle tables with one padding slot per 8 / 16 entries: power-of-two strides then spread over the slots
    for name, f in (("tw / ew padded j + (j >> 3)", lambda j: j + (j >> 3)), ("tw / ew padded j + (j >> 4)", lambda j: j + (j >> 4)),
                    ("tw / ew XOR-folded j ^ ((j >> 4) & 15)", lambda j: j ^ ((j >> 4) & 15)),
                    ("tw / ew XOR-folded j ^ (((j >> 4) ^ (j >> 8)) & 15)", lambda j: j ^ (((j >> 4) ^ (j >> 8)) & 15))):
        L2 = Layout(N, LT // 2, tw_map=f, ew_map=f)
        report("same kernel, " + name, zpass_phases(L2, N, LT, 256))
    report("experiment: 512 lanes, lane-pair split, current layout", zpass_phases(base, N, LT, 512, split=True))


def total(L, N, LT, NT=256, **kw):
    ph = zpass_phases(L, N, LT, NT, **kw)
    i = sum(v[0] for t in ph.values() for v in t.values()); e = sum(v[1] for t in ph.values() for v in t.values())
    return i, e
