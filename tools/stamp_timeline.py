"""In-kernel timeline of the role-split step kernel in the REAL launch chain (QS_STAMP builds, tools/build_stamp.sh):

    QUADSIM_HIP_LIB=quadsim_amd/csrc/libquadsim_hip_stamp.so  python tools/stamp_timeline.py --envs 65536                   (phases)
    QUADSIM_HIP_LIB=quadsim_amd/csrc/libquadsim_hip_stamp2.so python tools/stamp_timeline.py --queue-mode private --queues 2  (light)

Stamps (100 MHz real-time counter s_memrealtime, kept in registers until the wave ends) per workgroup and step: chaser wave 0
start, 1 state loads landed, 2 drone step done (before barrier #1), 3 after #1, 4 obs + reward done (before #2), 5 after #2,
6 after the reset branch, 7 stores drained; target wave 8 + {0 start, 1 loads landed, 2 before #1, 3 after #1, 4 before #2,
5 after #2, 6 stores issued}.  The light build (-DQS_STAMP=2) records only the first and last stamp of each wave: period, span
and gap of the chain without the ~0.7-0.9 us per step that the full build costs (eight scalar-memory round trips, load-landed
waits, and a flush of eight stores behind the drained wave end), and only in ONE WORKGROUP IN 64 (two stamps in every wave
still cost ~0.4 us per step); its last stamp is "all stores issued", not "drained", and its first / last workgroup are those
of the stamped sample, so its span / gap split is approximate (span low by the drain and the unsampled stragglers) while its
PERIOD -- consecutive starts of the same workgroups -- is the chain's own.

The stamp buffer travels in StepArgs, so launches from the handle's private AQL queues (a second copy of the code object,
loaded through HSA) record too -- unlike rocprofv3, which wraps every HSA queue and thereby changes these launches, the stamps
leave the queue alone.  --json writes the summary for bench.py (roofline.stamp_*)."""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import quadsim_amd as qa

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--env", default="docking-v0")
ap.add_argument("--randomise", type=int, default=1)
ap.add_argument("--groups", type=int, default=1)
ap.add_argument("--queue-mode", default="hip", choices=["hip", "private"])
ap.add_argument("--queues", type=int, default=1)
ap.add_argument("--ordering", default="host", choices=["host", "stream"],
                help="private queues: host = back-to-back packets (pre-staged actions), stream = GPU-side hand-shake per step")
ap.add_argument("--pool", type=int, default=64, help="distinct action batches cycled")
ap.add_argument("--json", default=None)
args = ap.parse_args()
n, G = args.envs, args.groups
lib = qa._lib.load()
if not hasattr(lib, "qs_debug_set_stamps"):
    raise SystemExit("this library has no stamps: build tools/build_stamp.sh and set QUADSIM_HIP_LIB")
env = qa.VecDockingEnv(args.env, num_envs=n, randomise=args.randomise, seed=1234, init_range=qa.C3_INIT_RANGE,
                       mass_scale=(0.8, 1.2), inertia_scale=(0.8, 1.2), copy=False)
tiles = (n + 63) // 64
buf = torch.zeros((64, tiles, 16), dtype=torch.int64, device="cuda")
lib.qs_debug_set_stamps.argtypes = [C.c_void_p, C.c_uint64]
assert lib.qs_debug_set_stamps(C.c_void_p(buf.data_ptr()), buf.numel()) == 0
env.reset()
if G > 1:
    env.set_groups(G, threads=True)
if args.queue_mode == "private":
    env.set_queue_mode(True, args.queues, ordering=args.ordering)
pool = env.random_actions(args.pool, step0=0)
h = env._h
p = lambda t: C.c_void_p(t.data_ptr())            # noqa: E731
io = (p(env._obs), p(env._rew), p(env._done), p(env._flags), p(env._term), None)
torch.cuda.synchronize()
aptr = [p(pool[i]) for i in range(args.pool)]
seq = [aptr[k % args.pool] for k in range(6 * 64 + 640)]   # laid out beforehand: a tensor index per step would make the host the limit
step = lib.qs_step_groups
for a in seq:                              # the last 64 steps overwrite the earlier ones (slot = step counter mod 64)
    step(h, a, *io)
env.sync()
torch.cuda.synchronize()
light = "stamp2" in os.environ.get("QUADSIM_HIP_LIB", "")
s = buf.cpu().numpy().astype(np.float64) * 0.01          # us
s = s[8:56]                                              # steady state
if light:
    s = s[:, ::64]                                       # the light build stamps one workgroup in 64 (tile % 64 == 0)
t0 = s[:, :, 0].min(axis=1)                              # first workgroup start of each launch (all groups / queues)
t_end = np.maximum(s[:, :, 7], s[:, :, 14]).max(axis=1)  # last wave drained (chaser) / stores issued (target)
per = np.diff(t0)
mode = args.queue_mode if args.queue_mode == "hip" else "private x%d (%s-ordered)" % (args.queues, args.ordering)
out = {"envs": n, "env": args.env, "randomise": args.randomise, "groups": G, "queue_mode": args.queue_mode,
       "queues": args.queues if args.queue_mode == "private" else 0,
       "ordering": args.ordering if args.queue_mode == "private" else "stream", "build": "light" if light else "full",
       "action_pool": args.pool,
       "stamp_period_us": float(np.median(per)), "stamp_period_min_us": float(per.min()), "stamp_period_max_us": float(per.max()),
       "stamp_start_spread_us": float(np.median(s[:, :, 0].max(axis=1) - t0)),
       "stamp_kernel_span_us": float(np.median(t_end - t0)),
       "stamp_gap_us": float(np.median(t0[1:] - t_end[:-1])),
       "workgroup_lifetime_us": float(np.median(s[:, :, 7] - s[:, :, 0])),
       "span_ends_at": "stores issued" if light else "stores drained"}
print("envs %d %s randomise %d | launch path: %s | groups %d tiles %d | %s stamps" % (n, args.env, args.randomise, mode, G, tiles, out["build"]))
print("step period (first start -> next step's first start): median %.2f us  (min %.2f max %.2f)" % (out["stamp_period_us"], per.min(), per.max()))
print("start spread (first -> last wg start): median %.2f us" % out["stamp_start_spread_us"])
end_name = "last wave's stores issued" if light else "last drained"
print("kernel span (first start -> %s): median %.2f us" % (end_name, out["stamp_kernel_span_us"]))
print("gap (%s -> next step's first start): median %.2f us" % (end_name, out["stamp_gap_us"]))
if not light:
    names = ["start->loads landed", "loads->drone step done", "wait at #1", "#1->obs/reward done", "wait at #2", "reset branch", "stores + drain"]
    out["chaser_wave_us"] = {}
    for j, nm in enumerate(names):
        d = s[:, :, j + 1] - s[:, :, j]
        out["chaser_wave_us"][nm] = float(np.median(d))
        print("  chaser wave %-24s median %.2f  p90 %.2f us" % (nm, np.median(d), np.percentile(d, 90)))
    tn = ["start->loads landed", "loads->step+draw done", "wait at #1", "#1->PID done", "wait at #2", "reset + stores issued"]
    out["target_wave_us"] = {}
    for j, nm in enumerate(tn):
        d = s[:, :, 8 + j + 1] - s[:, :, 8 + j]
        out["target_wave_us"][nm] = float(np.median(d))
        print("  target wave %-24s median %.2f  p90 %.2f us" % (nm, np.median(d), np.percentile(d, 90)))
print("  workgroup lifetime (chaser wave)      median %.2f  p90 %.2f us" % (out["workgroup_lifetime_us"], np.percentile(s[:, :, 7] - s[:, :, 0], 90)))
if args.queue_mode == "private" and args.queues > 1:
    # per queue: each steps a contiguous tile range (whole multiples of 8 tiles)
    cuts = [0] + [min(tiles, ((tiles * q // args.queues) + 7) // 8 * 8) for q in range(1, args.queues)] + [tiles]
    if light:
        cuts = [(c + 63) // 64 for c in cuts]            # indices into the stamped subset
    out["per_queue"] = []
    for q in range(args.queues):
        sl = s[:, cuts[q]:cuts[q + 1]]
        if sl.shape[1] == 0:
            continue
        a, b = sl[:, :, 0].min(axis=1), np.maximum(sl[:, :, 7], sl[:, :, 14]).max(axis=1)
        out["per_queue"].append({"tiles": [cuts[q], cuts[q + 1]], "span_us": float(np.median(b - a)), "period_us": float(np.median(np.diff(a))),
                                 "gap_us": float(np.median(a[1:] - b[:-1]))})
        print("  queue %d (tiles %d..%d): span median %.2f us, period %.2f us, gap %.2f us" % (q, cuts[q], cuts[q + 1], np.median(b - a), np.median(np.diff(a)), np.median(a[1:] - b[:-1])))
if G > 1:
    for g in range(G):
        lo, hi = env.group_range(g)
        sl = s[:, lo // 64:(hi + 63) // 64]
        a, b = sl[:, :, 0].min(axis=1), sl[:, :, 7].max(axis=1)
        print("  group %d: span median %.2f us, period %.2f us, gap %.2f us" % (g, np.median(b - a), np.median(np.diff(a)), np.median(a[1:] - b[:-1])))
if args.json:
    with open(args.json, "w") as f:
        json.dump(out, f, indent=1)
env.close()
