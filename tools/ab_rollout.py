#!/usr/bin/env python3
"""A/B of library builds on the fused roll-out and the step kernel (set QUADSIM_HIP_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.sweep import run
print("lib:", os.environ.get("QUADSIM_HIP_LIB", "default"))
for n in (65536, 1048576):
    us, eps = run(n, K=300 if n > 100000 else 1000)
    print("  step    N=%8d %8.2f us %7.3f G/s" % (n, us, eps / 1e9))
    us, eps = run(n, rollout_T=64, K=256)
    print("  rollout N=%8d %8.2f us/step %7.3f G/s" % (n, us, eps / 1e9))
