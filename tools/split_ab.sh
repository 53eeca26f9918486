#!/bin/bash
# A/B of the serial and the role-split step kernel (QS_SPLIT switch), interleaved rounds
for r in 1 2 3; do for m in 0 1; do echo "split=$m: $(QS_SPLIT=$m python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f G/s  %.2f us' % (d['value']/1e9, d['roofline']['launch_period_us']))")"; done; done
for m in 0 1; do echo "split=$m"; QS_SPLIT=$m python tools/ab_rollout.py | tail -4; done
