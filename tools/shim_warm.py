"""per-window step time of a fresh single-env shim: how long the cold phase lasts"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import quadsim_amd as qa
e = qa.DockingEnv(); e.reset()
a = np.full(4, -0.5)
for w in range(12):
    t0 = time.perf_counter()
    for _ in range(250):
        o, r, d, info = e.step(a)
        if d: e.reset()
    print("window %2d: %.1f us/step" % (w, (time.perf_counter() - t0) / 250 * 1e6))
