set -o pipefail
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_groups_and_rollout.py -m gpu -q -x -k "gae or episode or runner" > $OUT/pytest_gpu_r02j.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu_r02j.log
export TMPDIR=/tmp; cd /tmp
rm -rf $OUT/runner_trace3
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/runner_trace3 -- python3 $R/tools/runner_trace.py > $OUT/runner_trace3.log 2>&1
echo "runner trace rc=$?"
cd $R
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/runner_trace3/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
i0=[i for i,r in enumerate(rows) if 'k_runner_rollout' in r['Kernel_Name']][0]
for r in rows[i0:i0+6]:
    print('%.1f'%((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3), r['Kernel_Name'].replace('(anonymous namespace)::','')[:70])
PY
