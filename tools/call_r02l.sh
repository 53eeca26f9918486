set -o pipefail
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu_r02l.log 2>&1
echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu_r02l.log
bash tools/ab_libs.sh quadsim_amd/csrc/libquadsim_hip_prev.so quadsim_amd/csrc/libquadsim_hip.so 3
bash tools/ab_libs.sh quadsim_amd/csrc/libquadsim_hip_prev.so quadsim_amd/csrc/libquadsim_hip.so 1 --envs-per-gpu 4096
bash tools/ab_libs.sh quadsim_amd/csrc/libquadsim_hip_prev.so quadsim_amd/csrc/libquadsim_hip.so 1 --envs-per-gpu 131072
bash tools/ab_libs.sh quadsim_amd/csrc/libquadsim_hip_prev.so quadsim_amd/csrc/libquadsim_hip.so 1 --randomise 2 --env docking-v2 --envs-per-gpu 131072
