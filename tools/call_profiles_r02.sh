# One GPU-box round of the measurements DESIGN.md section 5 quotes (round 2).  Run via gpurun from the repo root.
set -o pipefail
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf $OUT/prof_r02
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r02 -- python3 $R/bench.py --steps 500 --warmup 50 --min-timed-steps 500 --no-cpu-baseline --no-extras --no-parity --queue-mode private --queues 1 > $OUT/prof_r02.log 2>&1
echo "rocprof stats rc=$?"; tail -1 $OUT/prof_r02.log | head -c 400; echo
cd $R
find $OUT/prof_r02 -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $OUT/step_api_kernel_stats.csv
head -4 $OUT/step_api_kernel_stats.csv
bash tools/pmc.sh r02 > $OUT/pmc_r02.log 2>&1; tail -5 $OUT/pmc_r02.log
python tools/pmc_traffic_json.py $OUT/pmc_r02 $OUT/pmc_traffic.json
