#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per dispatch of each kernel."""
import csv, sys, collections, glob
for path in sys.argv[1:]:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:36], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            if len(v) >= 20:
                v = v[len(v) // 5:]   # drop warm-up dispatches
                print("%s | %-36s %-12s n=%5d mean=%14.1f" % (path.split("/")[-1], k, c, len(v), sum(v) / len(v)))
