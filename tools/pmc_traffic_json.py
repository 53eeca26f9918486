#!/usr/bin/env python3
"""profiles/<round>/pmc_traffic.json from the four rocprofv3 --pmc passes of tools/pmc.sh: HBM-side bytes per step of the
step kernel (FETCH_SIZE x2: the gfx950 half-count of MI355X_MICROARCH.md, re-calibrated here at 2 M envs; + WRITE_SIZE)."""
import collections, csv, glob, json, re, sys
base, out = sys.argv[1], sys.argv[2]


def mean(path, counter):
    vals = collections.defaultdict(list)
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "k_env" in r["Kernel_Name"]:
                vals[re.search(r"k_env[a-z_]*<[^>]*>", r["Kernel_Name"]).group(0)].append(float(r["Counter_Value"]))
    name, v = max(vals.items(), key=lambda kv: len(kv[1]))
    v = v[len(v) // 5:]
    return name, sum(v) / len(v), len(v)


k1, f_small, n1 = mean(base + "/FETCH_SIZE_65536", "FETCH_SIZE")
_, w_small, _ = mean(base + "/WRITE_SIZE_65536", "WRITE_SIZE")
try:
    _, f_prv, n_prv = mean(base + "/FETCH_SIZE_65536_private", "FETCH_SIZE")
    _, w_prv, _ = mean(base + "/WRITE_SIZE_65536_private", "WRITE_SIZE")
except Exception:
    f_prv = w_prv = None
k2, f_big, _ = mean(base + "/FETCH_SIZE_2097152", "FETCH_SIZE")
_, w_big, _ = mean(base + "/WRITE_SIZE_2097152", "WRITE_SIZE")
n_big = 2097152
fetch_ratio = f_big * 1024 / (176 * n_big)
corr = 1.0 / fetch_ratio
write_ratio = w_big * 1024 / (216 * n_big)
traffic = corr * f_small * 1024 + w_small * 1024
extra = {}
if f_prv is not None:
    t_prv = corr * f_prv * 1024 + w_prv * 1024
    extra = {"private_queue": {"fetch_size_kib": f_prv, "write_size_kib": w_prv, "traffic_bytes_per_step": int(t_prv),
                               "ratio": t_prv / (392 * 65536), "dispatches_averaged": n_prv,
                               "note": "step launches without the end-of-kernel release: the tiles' state stays dirty in the XCDs' L2s, so "
                                       "fewer bytes reach the memory side than the step moves algorithmically (the average includes the ~190 "
                                       "fenced launches of bench.py's verification run)"}}
json.dump({**extra, "kernel": k1.strip(), "envs": 65536, "dispatches_averaged": n1, "fetch_size_kib": f_small, "fetch_correction": corr,
           "write_size_kib": w_small, "traffic_bytes_per_step": int(traffic), "algorithmic_bytes_per_step": 392 * 65536,
           "ratio": traffic / (392 * 65536),
           "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes (tools/pmc.sh); FETCH_SIZE corrected by the "
                     "ratio measured at 2 M envs (far beyond L2 + Infinity Cache)",
           "calibration_2M_envs": {"kernel": k2.strip(), "fetch_kib": f_big, "fetch_ratio_vs_176B": fetch_ratio, "write_kib": w_big,
                                   "write_ratio_vs_216B": write_ratio}}, open(out, "w"), indent=1)
print(open(out).read())
