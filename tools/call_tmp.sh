R=$(pwd)
QUADSIM_HIP_LIB=$R/quadsim_amd/csrc/libquadsim_hip_dbg.so timeout -k 10 900 python -m pytest tests -m gpu -q 2>&1 | tail -3
QUADSIM_HIP_LIB=$R/quadsim_amd/csrc/libquadsim_hip_dbg.so timeout -k 10 600 python tools/soak_shape_debug.py 2>&1 | tail -4
