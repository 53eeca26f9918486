# One GPU-box round of the evidence DESIGN.md section 5 quotes (round 3).  Run via gpurun from the repo root, AFTER
# tools/build_stamp.sh in the build container (the stamped twins travel with the snapshot).
#   1. in-kernel s_memrealtime timelines of the step chain: HIP stream, 1 / 2 / 3 private queues (light stamps: period / span / gap of
#      the unperturbed chain; full stamps: the phase budget)
#   2. rocprofv3 --kernel-trace --stats of the same command in both launch paths, each with a sidecar json that records the launch
#      shape the profile was TAKEN with (bench.py computes rocprof_*.kernel_frac from it)
#   3. PMC traffic passes (tools/pmc.sh)
set -o pipefail
R=$(pwd); OUT=$R/gpurun_out/r03; mkdir -p $OUT
export TMPDIR=/tmp
L2=$R/quadsim_amd/csrc/libquadsim_hip_stamp2.so; L1=$R/quadsim_amd/csrc/libquadsim_hip_stamp.so
for Q in 1 2 3; do
  QUADSIM_HIP_LIB=$L2 timeout -k 10 120 python3 tools/stamp_timeline.py --queue-mode private --queues $Q --ordering host --pool 512 \
      --json $OUT/step_kernel_timeline_private_q$Q.json > $OUT/step_kernel_timeline_private_q$Q.txt 2>&1 || echo "stamp private q$Q failed"
done
QUADSIM_HIP_LIB=$L2 timeout -k 10 120 python3 tools/stamp_timeline.py --queue-mode hip --pool 512 --json $OUT/step_kernel_timeline_hip.json > $OUT/step_kernel_timeline_hip.txt 2>&1 || echo "stamp hip failed"
QUADSIM_HIP_LIB=$L1 timeout -k 10 120 python3 tools/stamp_timeline.py --queue-mode private --queues 1 --ordering host --pool 64 > $OUT/step_kernel_phases_private_q1.txt 2>&1 || echo "phases private failed"
QUADSIM_HIP_LIB=$L1 timeout -k 10 120 python3 tools/stamp_timeline.py --queue-mode hip --pool 64 > $OUT/step_kernel_phases_hip.txt 2>&1 || echo "phases hip failed"
QUADSIM_HIP_LIB=$L2 timeout -k 10 120 python3 tools/stamp_timeline.py --envs 131072 --queue-mode private --queues 2 --ordering host --pool 128 \
      --json $OUT/step_kernel_timeline_131072_private_q2.json > $OUT/step_kernel_timeline_131072_private_q2.txt 2>&1 || echo "stamp 131072 failed"
tail -n 6 $OUT/step_kernel_timeline_private_q1.txt $OUT/step_kernel_timeline_private_q2.txt $OUT/step_kernel_timeline_hip.txt
cd /tmp
X="--steps 500 --warmup 50 --min-timed-steps 500 --no-cpu-baseline --no-extras --no-parity"
for M in private hip; do
  rm -rf $OUT/prof_$M
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$M -- python3 $R/bench.py $X --queue-mode $M --queues 1 > $OUT/prof_$M.log 2>&1
  echo "rocprof $M rc=$?"
  find $OUT/prof_$M -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $OUT/step_api_kernel_stats_$M.csv
  head -3 $OUT/step_api_kernel_stats_$M.csv
  python3 - <<PY
import json
m = "$M"
json.dump({"queue_mode": m, "queues": 1 if m == "private" else 0, "envs": 65536, "envs_per_launch": 65536,
           "command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py $X --queue-mode $M --queues 1",
           "note": ("bytes of ONE 65 536-env launch / its average duration in the PROFILED process; rocprofv3 wraps every HSA queue and adds a "
                    "completion signal per dispatch, which a private-queue launch otherwise does not carry: the profiled process steps slower "
                    "than the unprofiled chain (see the stamp timeline for the latter)") if m == "private" else
                   "bytes of ONE 65 536-env launch / its average duration; HIP-stream launches (agent-scope release after every kernel)"},
          open("$OUT/step_api_kernel_stats_%s.json" % m, "w"), indent=1)
PY
done
cd $R
bash tools/pmc.sh r03 > $OUT/pmc_r03.log 2>&1; tail -5 $OUT/pmc_r03.log
python tools/pmc_traffic_json.py $R/gpurun_out/pmc_r03 $OUT/pmc_traffic.json
