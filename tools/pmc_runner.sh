#!/bin/bash
# SQ counters of the Runner kernels (tools/runner_once.py: 4 launches each of f32 / bf16x3, T = 32, 65 536 envs), one rocprofv3
# pass per counter group; prints per-kernel averages.   bash tools/pmc_runner.sh [tag]
TAG=${1:-r02}
R=$(pwd); OUT=$R/gpurun_out/pmc_runner_$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/g$i -- python3 $R/tools/runner_once.py > $OUT/g$i.log 2>&1 || echo "pass $i ($C) failed: $(tail -2 $OUT/g$i.log)"
done
cd $R
python - "$OUT" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "runner" not in k: continue
        name = ("split " if "runner_split" in k else "serial ") + ("bf16x3" if k.rstrip(")").split(",")[-1].strip().startswith(("true", "1")) or "Lb1EEE" in k else "f32")
        acc[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
