#!/usr/bin/env python3
"""aggregate a rocprofv3 kernel_trace.csv: mean/min kernel duration and mean start-to-start period per (kernel, grid)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
g = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0][-40:]
    key = (name, int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]))
    g[key].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for key, v in sorted(g.items(), key=lambda kv: kv[0][1]):
    if len(v) < 20: continue
    v.sort()
    d = [e - s for s, e in v]
    per = [v[i + 1][0] - v[i][0] for i in range(len(v) - 1)]
    per.sort()
    gap = sorted(v[i + 1][0] - v[i][1] for i in range(len(v) - 1))
    print("%-42s grid=%9d n=%5d  dur mean %7.2f us min %7.2f | period median %7.2f us | gap median %6.2f us" % (
        key[0], key[1], len(v), sum(d) / len(d) / 1e3, min(d) / 1e3, per[len(per) // 2] / 1e3, gap[len(gap) // 2] / 1e3))
