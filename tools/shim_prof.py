"""where the time of the single-env gym shim (QS_IO_HOST) goes: C-ABI calls vs Python"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import quadsim_amd as qa
from quadsim_amd import _lib
e = qa.DockingEnv(); e.reset()
lib = e._lib
a = np.zeros((1,4), np.float32); obs = np.zeros((1,12), np.float32); rew = np.zeros(1, np.float32)
done = np.zeros(1, np.uint8); flags = np.zeros(1, np.uint8)
p = lambda x: x.ctypes.data_as(C.c_void_p)
args = (e._h, p(a), p(obs), p(rew), p(done), p(flags), None)
def tm(f, K=500):
    for _ in range(20): f()
    t0 = time.perf_counter()
    for _ in range(K): f()
    return (time.perf_counter() - t0) / K * 1e6
def stp():
    o, r, d, info = e.step(np.zeros(4))
    if d: e.reset()
print("env.step (python, first) %.1f us" % tm(stp, 300))
print("qs_step (host, N=1)      %.1f us" % tm(lambda: lib.qs_step(*args)))
sc = np.zeros((1,13), np.float32); st = np.zeros((1,13), np.float32); ls = np.zeros(1, np.float32); t = np.zeros(1, np.float32)
print("qs_get_state             %.1f us" % tm(lambda: lib.qs_get_state(e._h, p(sc), p(st), None, None, p(ls), p(t))))
print("qs_sync only             %.1f us" % tm(lambda: lib.qs_sync(e._h)))
print("_pull_state (python)     %.1f us" % tm(e._pull_state))
e.reset()
print("qs_get_state again       %.1f us" % tm(lambda: lib.qs_get_state(e._h, p(sc), p(st), None, None, p(ls), p(t))))
print("env.step (python)        %.1f us" % tm(stp, 300))
