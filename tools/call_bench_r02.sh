# bench lines of round 2: the default run, the driver's 20-step form, the single-GPU shares of BASELINE configs 2-5
set -o pipefail
R=$(pwd); OUT=$R/gpurun_out/bench_r02; mkdir -p $OUT
timeout -k 10 900 python bench.py > $OUT/bench_final.json 2> $OUT/bench_final.err; echo "default rc=$?"
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err; echo "steps20 rc=$?"
X="--no-extras --no-cpu-baseline --no-parity"
timeout -k 10 300 python bench.py $X --envs-per-gpu 4096 --randomise 0 > $OUT/bench_c2.json 2>/dev/null; echo "c2 rc=$?"
timeout -k 10 300 python bench.py $X --envs-per-gpu 4096 --randomise 0 --integrator rk4 > $OUT/bench_c2_rk4.json 2>/dev/null; echo "c2 rk4 rc=$?"
timeout -k 10 300 python bench.py $X > $OUT/bench_c3.json 2>/dev/null; echo "c3 rc=$?"
timeout -k 10 300 python bench.py $X --env docking-v2 > $OUT/bench_c4_per_gpu.json 2>/dev/null; echo "c4 rc=$?"
timeout -k 10 300 python bench.py $X --env docking-v2 --envs-per-gpu 131072 --randomise 2 > $OUT/bench_c5_per_gpu.json 2>/dev/null; echo "c5 rc=$?"
timeout -k 10 300 python bench.py $X --queue-mode hip > $OUT/bench_c3_hip_stream.json 2>/dev/null; echo "c3 hip rc=$?"
timeout -k 10 300 python bench.py $X --envs-per-gpu 131072 > $OUT/bench_131072_v0.json 2>/dev/null; echo "131072 v0 rc=$?"
timeout -k 10 300 python bench.py --gpus 2 --backend gloo $X > $OUT/bench_gloo2_selflaunch.json 2>/dev/null; echo "gloo2 rc=$?"
QS_BENCH_FORCE_DIST=1 MASTER_PORT=29544 timeout -k 10 600 python bench.py --no-cpu-baseline --no-parity > $OUT/bench_rccl_world1.json 2>/dev/null; echo "rccl world1 rc=$?"
for f in $OUT/*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']
print('%-32s %.3f G/s  period %.2f us  frac %.3f read %.3f timeline %.3f  [%s x%s]' % ('$(basename $f)', d['value']/1e9, r['step_period_us'], r['frac'], r['read_frac'], r['gpu_timeline_frac'], d['config']['queue_mode'], d['config']['private_queues']))"; done
