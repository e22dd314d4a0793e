"""Search over XOR swizzles of the LDS line index (low 4 bits ^ a hash of the bits above bit 2), twiddle-table padding and the
pair-line stride for the layout with the fewest modelled bank-conflict cycles, both lane maps (axis >= 1 and contiguous axis)."""
import itertools
import sys
import sim

N = int(sys.argv[1]) if len(sys.argv) > 1 and __name__ == "__main__" else 512
LT = int(sys.argv[2]) if len(sys.argv) > 2 and __name__ == "__main__" else 16


def make_swz(shifts, mask=15):
    def f(i):
        h = 0
        for s in shifts:
            h ^= i >> s
        return i ^ (h & mask)
    return f


def bijective(f, n):
    return len({f(i) for i in range(n)}) == n and all((f(8 * g + q) == f(8 * g) ^ q) for g in range(n // 8) for q in range(8))


if __name__ == "__main__":
    pad = lambda j: j + (j >> 4)
    results = []
    for r in range(1, 5):
        for shifts in itertools.combinations(range(3, 10), r):
            f = make_swz(shifts)
            if not bijective(f, N):
                continue
            L = sim.Layout(N, LT // 2, tw_map=pad, ew_map=pad, swz=f)
            a = sim.total(L, N, LT)
            b = sim.total(L, N, LT, ax0=True)
            results.append((a[1] + b[1], a[1], b[1], shifts))
    results.sort()
    cur = sim.Layout(N, LT // 2)
    print("current layout: extra cycles axis>=1 %d, contiguous axis %d (ideal %d each)" % (sim.total(cur, N, LT)[1], sim.total(cur, N, LT, ax0=True)[1], sim.total(cur, N, LT)[0]))
    for tot, a, b, sh in results[:12]:
        print("shifts %-14s padded tables: extra axis>=1 %5d  contiguous %5d  sum %5d" % (sh, a, b, tot))
