"""serial vs role-split step kernel over env counts (QS_SPLIT forced per process): step API and fused roll-out"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.sweep import run
for n in (4096, 65536, 131072, 262144, 524288, 1048576):
    us, eps = run(n, K=300 if n > 100000 else 1000)
    us2, eps2 = run(n, rollout_T=64, K=128)
    print("QS_SPLIT=%s N=%8d step %7.2f us %7.3f G/s | rollout %6.2f us/step %7.3f G/s" % (os.environ.get("QS_SPLIT", "auto"), n, us, eps / 1e9, us2, eps2 / 1e9))
