# One GPU-box round of the Runner-kernel measurements DESIGN.md section 8 / profiles/r02 quote.  Run via gpurun from the repo root.
set -o pipefail
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
# 1. A/B of the two kernel flavours (us per step, T = 32, 65 536 envs)
{ for i in 1 2 3; do
    echo "role-split  $(timeout -k 10 200 python tools/ab_policy.py 2>&1 | tail -1)"
    echo "one-wave    $(QUADSIM_RUNNER_SERIAL=1 timeout -k 10 200 python tools/ab_policy.py 2>&1 | tail -1)"
  done; } > $OUT/runner_ab.txt 2>&1
cat $OUT/runner_ab.txt
# 2. phase budget of the role-split kernel (stamped build)
{ for P in f32 bf16x3; do QUADSIM_HIP_LIB=$R/quadsim_amd/csrc/libquadsim_hip_stamp.so timeout -k 10 200 python tools/runner_phases.py 65536 64 $P 2>&1 | grep -v amdgpu.ids || exit 1; done; } > $OUT/runner_phases.txt 2>&1
cat $OUT/runner_phases.txt
# 3. SQ counters
bash tools/pmc_runner.sh r02 > $OUT/pmc_runner.txt 2>&1; tail -40 $OUT/pmc_runner.txt
# 4. Runner.run() at T = 600 under rocprofv3 --stats
cd /tmp; rm -rf $OUT/prof_runner_r02
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_runner_r02 -- python3 $R/tools/runner_trace.py > $OUT/prof_runner_r02.log 2>&1
echo "rocprof runner rc=$?"
cd $R
find $OUT/prof_runner_r02 -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $OUT/ppo2_collect_kernel_stats.csv
head -6 $OUT/ppo2_collect_kernel_stats.csv | cut -c1-200
