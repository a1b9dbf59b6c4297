"""Average kernel durations over the bench's TIMED steps from a rocprofv3 kernel trace (the per-kernel statistics file
averages every dispatch of the process, including warm-up steps and the candidate launches of tune_placement):
for each library kernel the last `steps` x (dispatches per step) dispatches.
usage: trace_timed_region.py <kernel_trace.csv> <steps> <out.json>"""
import csv, json, sys, re
from collections import defaultdict

path, steps, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
rows = list(csv.DictReader(open(path)))
by = defaultdict(list)
for r in rows:
    name = r.get("Kernel_Name") or r.get("Name") or ""
    m = re.search(r"\b(k_[a-z_0-9]+(<[^>]*>)?)", name)
    if not m or m.group(1).startswith("k_probe"):
        continue
    by[m.group(1)].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
res = {}
for k, v in by.items():
    v.sort()
    # dispatches per step: the d2h leg (3 steps) and the CPU-free tail follow the timed steps; find the per-step count
    # from the statistics of the whole run is fragile, so use the known schedule: k_reduce runs twice (three times with
    # the refinement pass of N > 16), everything else once per step
    per = 1
    if k == "k_reduce":
        per = 3 if any(n.startswith("k_gram<") and n.endswith("true>") for n in by if not n.startswith("k_gram<8")
                       and not n.startswith("k_gram<4") and not n.startswith("k_gram<12") and not n.startswith("k_gram<16")) else 2
    if k.startswith("k_eig") and any(n.startswith("k_gram<2") or n.startswith("k_gram<3") for n in by):
        per = 2
    tail = 3 * per                      # the three steps of the small-D2H leg after the timed region
    sel = v[-(steps * per + tail):-tail] if len(v) >= steps * per + tail else v[-steps * per:]
    d = [x[1] for x in sel]
    res[k] = {"dispatches_averaged": len(d), "avg_ms": round(sum(d) / len(d) / 1e6, 4),
              "min_ms": round(min(d) / 1e6, 4), "max_ms": round(max(d) / 1e6, 4),
              "all_dispatches": len(v), "avg_all_ms": round(sum(x[1] for x in v) / len(v) / 1e6, 4)}
json.dump({"note": "per-kernel durations over the bench's timed steps only (last dispatches before the 3-step small-D2H leg); "
                   "avg_all_ms is what the kernel_stats.csv of the same trace averages (warm-up and tune_placement included)",
           "steps": steps, "kernels": res}, open(out, "w"), indent=1)
for k, r in sorted(res.items(), key=lambda kv: -kv[1]["avg_ms"]):
    print(f"{k:40s} timed-region avg {r['avg_ms']:.4f} ms over {r['dispatches_averaged']} (all {r['all_dispatches']}: {r['avg_all_ms']:.4f})")
