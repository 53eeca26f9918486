timeout -k 10 300 python -m pytest tests/test_gpu_groups_and_rollout.py -m gpu -q -x -k "guard or private" 2>&1 | tail -8
