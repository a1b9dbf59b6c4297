"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counter unit KiB).
FETCH_SIZE is doubled (gfx950 tallies the 128-B requests of wide coalesced reads as 64 B: MI355X_MICROARCH.md, HBM
section); WRITE_SIZE is exact for 16-B-per-lane streaming stores.   usage: pmc_traffic.py fetch.csv write.csv out.json [bench arguments, for the record]"""
import collections, csv, json, sys, time


def means(path, counter):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        name = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        if name.startswith("k_"):
            acc[name].append(float(row["Counter_Value"]))
    # the first dispatches are warm-up; keep the mean over all (same bytes every launch)
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


fetch, write = means(sys.argv[1], "FETCH_SIZE"), means(sys.argv[2], "WRITE_SIZE")
out = {"note": __doc__.strip().split("usage")[0].strip(), "measured": time.strftime("%Y-%m-%d %H:%M UTC", time.gmtime()),
       "command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE> -- python3 bench.py --steps 3 --warmup 1 --no-cpu "
                  "--placement-candidates 1 " + " ".join(sys.argv[4:]) + " (1 x MI355X; each pass run once)", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(k, (0.0, 0))
    w, nw = write.get(k, (0.0, 0))
    rd, wr = int(f * 1024 * 2), int(w * 1024)
    out["kernels"][k] = {"FETCH_SIZE_KiB_avg": f, "WRITE_SIZE_KiB_avg": w, "dispatches": max(nf, nw),
                         "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out["kernels"].items():
    print(f"{k:40s} read {v['hbm_read_bytes'] / 1e9:8.3f} GB  write {v['hbm_write_bytes'] / 1e9:8.3f} GB  ({v['dispatches']} dispatches)")
