"""Summarise the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) into
profiles/<tag>_pmc_hbm_traffic.{txt,json}: measured HBM bytes per launch of every kernel of the 512^3 bench.
Usage: python scripts/pmc_summary.py gpurun_out/pmc_fetch_<tag> gpurun_out/pmc_write_<tag> profiles/<tag>"""
import collections
import glob
import hashlib
import json
import os
import sqlite3
import statistics
import sys

fetch_dir, write_dir, out_prefix = sys.argv[1:4]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sources_sha():
    """Fingerprint of the kernel sources the counters belong to (bench.py refuses a traffic file whose fingerprint differs
    from the sources it runs: a PMC pass must be regenerated whenever a kernel changes)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "bifurcationkit.jl_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def clean(k):
    k = k.replace("void ", "").replace("bk::(anonymous namespace)::", "").replace("bk::", "")
    return k.split("(")[0]


med = collections.defaultdict(dict)
for d, cname in ((fetch_dir, "FETCH_SIZE"), (write_dir, "WRITE_SIZE")):
    f = glob.glob(d + "/**/*.db", recursive=True)[0]
    cur = sqlite3.connect(f).cursor()
    rows = cur.execute("select kernel_name, value from counters_collection where (end-start) > 300000 and counter_name = ?",
                       (cname,)).fetchall()
    agg = collections.defaultdict(list)
    for k, v in rows:
        agg[clean(k)].append(v)
    allv = [x for k, v in agg.items() if k.startswith("dct_f") for x in v]
    if allv:
        agg["dct_pass (all dct_f* kernels)"] = allv
    for k, v in agg.items():
        # per kernel instantiation: the median launch; the dct_pass FAMILY mixes 16 B/point passes with the 24 B/point x passes that
        # carry the pointwise factor / the shift axpy of the stencil-free operator (round 5): launch-weighted MEAN, which is what
        # bench.py's `alg_bytes_per_launch` of the family is
        mid = statistics.fmean(v) if k.startswith("dct_pass") else statistics.median(v)
        med[k][cname] = (len(v), mid, min(v), max(v))

lines = ["# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over",
         "#   python bench.py --steps 1 --warmup 0 --cpu-sample 0      (SH3d 512^3, MI355X, ROCm 7.2)",
         "# Dispatches longer than 300 us only (the 512^3 launches; the one-cell launches are filtered out).",
         "# Units: the counters are in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports 1/2 of",
         "# the bytes of wide coalesced reads -> read_bytes = 2 * FETCH_SIZE * 1024.  WRITE_SIZE * 1024 matches the known",
         "# 1 GiB output of every kernel exactly (axpbyz / copyBuffer write 1 GiB -> 1048576.0 KiB), so it is used as is.",
         "# Read-side calibration: every DCT pass kernel reads its 1 GiB input exactly once -> FETCH_SIZE = 0.500 GiB.",
         "", f"{'kernel':58s} {'n':>4s} {'FETCH_KiB(med)':>15s} {'read_GiB(x2)':>13s} {'WRITE_KiB(med)':>15s} {'write_GiB':>10s}"]
out = {}
for k in sorted(med):
    f_ = med[k].get("FETCH_SIZE", (0, 0.0, 0, 0))
    w_ = med[k].get("WRITE_SIZE", (0, 0.0, 0, 0))
    lines.append(f"{k[:58]:58s} {f_[0]:4d} {f_[1]:15.1f} {2 * f_[1] / 1048576:13.3f} {w_[1]:15.1f} {w_[1] / 1048576:10.3f}")
    out[k] = dict(n=f_[0], read_bytes=2 * f_[1] * 1024, write_bytes=w_[1] * 1024, read_bytes_min=2 * f_[2] * 1024,
                  read_bytes_max=2 * f_[3] * 1024)
lines += ["", "# multidot<KB> / multiaxpy<KB>: the median launch has k+1 (k+2) = read_GiB vectors; compare with the algorithmic bytes.",
          "# dct_fused_kernel<NT, MODE, AX0>: algorithmic 1 GiB read + 1 GiB written per pass (MODE 2 = forward + symbol +",
          "#   inverse of the last axis in one pass); the <..., AX0 = true, ..., FZ = true> instantiations are the x passes of the stencil-free",
          "#   operator: 2 GiB read (the vector + u, or the spectrum + the shifted vector) + 1 GiB written.  The family row is the launch-weighted",
          "#   MEAN.  sh_stream_kernel<true, VL>: algorithmic 2 GiB read (v, u) + 1 GiB written."]
out["_meta"] = dict(sources_sha=sources_sha(), command="rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py "
                    "--steps 1 --warmup 0 --cpu-sample 0 --no-steady")
lines.append(f"# kernel sources fingerprint (bifurcationkit.jl_amd/csrc/*.hip, *.h): {out['_meta']['sources_sha']}")
open(out_prefix + "_pmc_hbm_traffic.txt", "w").write("\n".join(lines) + "\n")
json.dump(out, open(out_prefix + "_pmc_hbm_traffic.json", "w"), indent=1)
print("\n".join(lines))
