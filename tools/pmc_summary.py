"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel, per counter, the mean over dispatches."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    name = row["Kernel_Name"].split("(")[0]
    if not ("k_" in name):
        continue
    acc[name[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n=%d" % len(next(iter(cs.values()))))
