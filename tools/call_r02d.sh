set -o pipefail
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
export QUADSIM_HIP_LIB=$R/quadsim_amd/csrc/libquadsim_hip_stamp.so
rm -f $OUT/stamp_timeline.txt
for cfg in "65536 1" "4096 1" "131072 1"; do
  timeout -k 10 120 python tools/stamp_timeline.py $cfg 2>&1 | grep -v amdgpu.ids | tee -a $OUT/stamp_timeline.txt
  echo | tee -a $OUT/stamp_timeline.txt
done
