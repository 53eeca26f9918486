R=$(pwd); OUT=$R/gpurun_out
for L in libquadsim_hip_stamp.so libquadsim_hip_stamp_sc1.so; do
  echo "== $L"
  QUADSIM_HIP_LIB=$R/quadsim_amd/csrc/$L timeout -k 10 120 python tools/stamp_timeline.py 65536 1 2>&1 | grep -v amdgpu.ids | head -19
done
