set -o pipefail
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
bash tools/ab_libs.sh quadsim_amd/csrc/libquadsim_hip.so quadsim_amd/csrc/libquadsim_hip_sc1.so 3
bash tools/ab_libs.sh quadsim_amd/csrc/libquadsim_hip.so quadsim_amd/csrc/libquadsim_hip_sc1.so 1 --envs-per-gpu 131072
bash tools/ab_libs.sh quadsim_amd/csrc/libquadsim_hip.so quadsim_amd/csrc/libquadsim_hip_sc1.so 1 --envs-per-gpu 4096
