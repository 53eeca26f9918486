# Multi-rank rehearsals of bench.py on the ONE-GPU box (round 3).  No scaling number can come
# out of these -- the ranks share one GPU -- they show that `python bench.py --gpus N` produces a line that counts: the BASELINE
# config for N, non-null roofline and cpu_baseline, RCCL initialised and used where the backend is nccl.
set -o pipefail
R=$(pwd); OUT=$R/gpurun_out/r03; mkdir -p $OUT
timeout -k 10 500 python3 bench.py --gpus 2 --backend gloo > $OUT/bench_gloo2.json 2> $OUT/bench_gloo2.err; echo "gloo2 rc=$?"
timeout -k 10 400 python3 bench.py --gpus 4 --backend gloo --no-extras --cpu-seconds 6 > $OUT/bench_gloo4.json 2> $OUT/bench_gloo4.err; echo "gloo4 rc=$?"
QS_BENCH_FORCE_DIST=1 timeout -k 10 500 python3 bench.py > $OUT/bench_rccl_world1.json 2> $OUT/bench_rccl_world1.err; echo "rccl world 1 rc=$?"
for f in bench_gloo2 bench_gloo4 bench_rccl_world1; do python3 - <<PY
import json
try:
    d = json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1])
    print("$f", d["n_gpus"], "%.4g" % d["value"], d["config"]["baseline_config"], d["config"]["env"], d["config"]["envs_per_gpu"], d["config"]["randomise"],
          "frac %.3f" % d["roofline"]["frac"], "cpu_baseline", None if d.get("cpu_baseline") is None else "%.3g" % d["cpu_baseline"]["value"], "allgather", d.get("allgather"))
except Exception as ex:
    print("$f", "ERR", ex)
PY
done
