"""One line per bench.py JSON line in the given files (the fields a reader compares between runs)."""
import json
import sys

for path in sys.argv[1:]:
    for line in open(path):
        line = line.strip()
        if not line.startswith("{"):
            continue
        try:
            d = json.loads(line)
            c = d.get("config", {})
            r = d.get("roofline") or {}
            comm = d.get("comm") or {}
            k = d.get("kernels") or {}
            fam = {n: (round(v["gbs"] / 8000.0, 3), round(v["ms_total"] / max(d.get("steps", 1), 1), 2)) for n, v in k.items() if isinstance(v, dict) and "gbs" in v}
            print(path.split("/")[-1], c.get("grid"), "ranks", d.get("n_gpus"), "ms %.2f" % d["ms_per_step"], "steps/s %.3f" % d["value"],
                  "itlin", c.get("itlinear_per_step"), "p", (c.get("full_corrector") or {}).get("p"),
                  "roof", (r.get("kernel"), round(r.get("frac", 0.0), 3), r.get("traffic")), "families (frac of 8 TB/s, ms per step)", fam,
                  "comm", comm.get("backend"), comm.get("ranks_in_communicator"), "blocks", c.get("gmres_blocks"))
        except Exception as e:  # noqa: BLE001
            print("unparsed", path, e, line[:200])
