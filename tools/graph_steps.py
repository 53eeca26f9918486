"""step API issued three ways: Python loop, native loop (qs_rollout_stepwise), one captured graph of T step launches"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quadsim_amd as qa
N, T = 65536, 64
env = qa.VecDockingEnv("docking-v0", num_envs=N, randomise=1, seed=0, init_range=qa.C3_INIT_RANGE)
env.reset()
acts = env.random_actions(T)
def timeit(f, reps=20):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps / T * 1e6
def py_loop():
    for t in range(T): env.step(acts[t])
print("python loop      %.2f us/step" % timeit(py_loop))
print("native stepwise  %.2f us/step" % timeit(lambda: env.rollout(acts, stepwise=True)))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    py_loop(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        py_loop()
    print("graph replay     %.2f us/step" % timeit(g.replay))
env.close()
