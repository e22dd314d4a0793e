"""Summarise a rocprofv3 SQ-counter pass over the 512^3 bench into a per-kernel stall breakdown.
Usage: python scripts/sq_summary.py gpurun_out/sq_<tag> [profiles/<tag>_sq_stall_breakdown.txt]
Counters (one pass, 8 SQ slots; MI355X_MICROARCH.md "rocprofv3 PMC slots"): SQ_WAVE_CYCLES = ACTIVE_INST_ANY + WAIT_ANY
(wave parked on s_waitcnt / barrier) + WAIT_INST_ANY (issue stall), in quad-cycles summed over waves."""
import collections
import glob
import sqlite3
import statistics
import sys

d = sys.argv[1]
f = glob.glob(d + "/**/*.db", recursive=True)[0]
cur = sqlite3.connect(f).cursor()
rows = cur.execute("select kernel_name, counter_name, value, (end-start) from counters_collection where (end-start) > 300000").fetchall()


def clean(k):
    k = k.replace("void ", "").replace("bk::(anonymous namespace)::", "").replace("bk::", "")
    return k.split("(")[0]


agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for k, c, v, dt in rows:
    agg[clean(k)][c].append(v)
    dur[clean(k)].append(dt / 1e3)
names = sorted({c for k in agg for c in agg[k]})
lines = ["# rocprofv3 --pmc " + " ".join(names) + " --kernel-trace -- python bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-steady",
         "# SH3d 512^3, dispatches > 300 us; medians per launch; percentages are of SQ_WAVE_CYCLES (quad-cycles summed over waves):",
         "# active = an instruction of the wave issued; wait_any = parked on s_waitcnt / barrier; wait_inst = issue stall;",
         "# valu / lds / vmem = share of wave cycles in which that unit's instruction was issuing; lds_conf = bank-conflict share of LDS cycles", ""]
hdr = f"{'kernel':44s} {'n':>4s} {'us(prof)':>9s} {'active%':>8s} {'wait_any%':>10s} {'wait_inst%':>11s} {'valu%':>7s} {'lds%':>6s} {'lds_conf%':>10s}"
lines.append(hdr)
for k in sorted(agg, key=lambda k_: -sum(dur[k_])):
    m = {c: statistics.median(v) for c, v in agg[k].items()}
    wc = m.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    pct = lambda c: 100.0 * m.get(c, 0.0) / wc
    ldsa = m.get("SQ_LDS_IDX_ACTIVE", 0.0)
    conf = 100.0 * m.get("SQ_LDS_BANK_CONFLICT", 0.0) / ldsa if ldsa else 0.0
    lines.append(f"{k[:44]:44s} {len(dur[k]) // max(1, len(names)):4d} {statistics.median(dur[k]):9.1f} {pct('SQ_ACTIVE_INST_ANY'):8.1f} "
                 f"{pct('SQ_WAIT_ANY'):10.1f} {pct('SQ_WAIT_INST_ANY'):11.1f} {pct('SQ_ACTIVE_INST_VALU'):7.1f} "
                 f"{pct('SQ_ACTIVE_INST_LDS'):6.1f} {conf:10.1f}")
txt = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt)
print(txt)
