bash tools/ab_libs.sh quadsim_amd/csrc/libquadsim_hip.so quadsim_amd/csrc/libquadsim_hip_noreset.so 2
bash tools/ab_libs.sh quadsim_amd/csrc/libquadsim_hip.so quadsim_amd/csrc/libquadsim_hip_noreset.so 1 --envs-per-gpu 4096
