timeout -k 10 300 python -m pytest tests/test_gpu_groups_and_rollout.py -m gpu -q -x -k "private" 2>&1 | tail -3
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --steps 200 --no-extras --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value']/1e9, d['config']['queue_mode'], d['config']['private_queues'], d['config']['queue_note'])"
