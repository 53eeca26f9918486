"""ctypes binding of libquadsim_hip.so (the C ABI declared in include/quadsim.h).

There is deliberately no fallback of any kind: if the HIP library is missing or
no MI355X is visible, loading / qs_create raise.  Nothing here imports oracle/.
"""
import ctypes as C
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.environ.get("QUADSIM_HIP_LIB") or os.path.join(CSRC, "libquadsim_hip.so")  # override: A/B builds
SOURCES = [os.path.join(CSRC, "quadsim_hip.hip")]
HEADERS = [os.path.join(CSRC, h) for h in ("quadsim_device.hpp", "step_kernels.hpp", "rollout_ops.hpp", "policy_rollout.hpp", "env_groups.hpp",
                                          "private_queue.hpp")] + [os.path.join(HERE, "..", "include", "quadsim.h")]

QS_OK = 0
KIND_V0, KIND_V2, KIND_V1, KIND_HOVER = 0, 1, 2, 3
INTEG_FROZEN, INTEG_RK4 = 0, 1
IO_DEVICE, IO_HOST = 0, 1
RANDOMISE_NONE, RANDOMISE_INIT, RANDOMISE_PARAMS = 0, 1, 2
ORDER_HOST, ORDER_STREAM = 0, 1
FLAG_DOCKED, FLAG_OVERLIMIT, FLAG_OVERTIME, FLAG_CHASER_LIMITED, FLAG_TARGET_LIMITED = 1, 2, 4, 8, 16

EXPORTS = [
    "qs_config_default", "qs_version", "qs_last_error", "qs_create", "qs_destroy", "qs_reset", "qs_step",
    "qs_rollout", "qs_rollout_slab", "qs_rollout_stepwise", "qs_fill_random_actions", "qs_get_state", "qs_set_state", "qs_set_params", "qs_get_params",
    "qs_set_init_state", "qs_get_init_state", "qs_obs_dim", "qs_get_step_counter", "qs_set_step_counter", "qs_set_stream", "qs_sync", "qs_timer_start", "qs_timer_stop",
    "qs_drone_step", "qs_ctrl", "qs_rel_obs", "qs_transform", "qs_gae", "qs_swap_and_flatten", "qs_expert_action", "qs_policy_rollout", "qs_policy_rollout_fast", "qs_policy_rollout_fast_blob_bytes",
    "qs_policy_forward", "qs_policy_forward_fast",
    "qs_runner_rollout", "qs_runner_rollout_fast", "qs_runner_rollout_fast_blob_bytes",
    "qs_step_ex", "qs_set_groups", "qs_group_count", "qs_group_range", "qs_group_stream", "qs_group_set_stream",
    "qs_step_group", "qs_step_groups", "qs_groups_fork", "qs_groups_join",
    "qs_swap_and_flatten_u8", "qs_gae_flatten", "qs_episode_stats", "qs_set_rollout_layout",
    "qs_set_queue_mode", "qs_get_queue_mode", "qs_set_queue_ordering", "qs_get_queue_ordering",
]


class QsConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("kind", C.c_int32), ("num_envs", C.c_int64), ("device", C.c_int32),
        ("integrator", C.c_int32), ("dt", C.c_float), ("auto_reset", C.c_int32), ("randomise", C.c_int32),
        ("io_space", C.c_int32), ("seed", C.c_uint64), ("env_id_offset", C.c_uint64),
        ("init_range", C.c_float * 4), ("mass_scale", C.c_float * 2), ("inertia_scale", C.c_float * 2),
        ("mass", C.c_float), ("inertia", C.c_float * 3), ("stream", C.c_void_p),
        ("external_stream", C.c_int32), ("reserved", C.c_int32),
    ]


class QsActorCritic(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("squash", C.c_int32),
        ("wt1", C.c_void_p), ("b1", C.c_void_p), ("wt2", C.c_void_p), ("b2", C.c_void_p),
        ("wt3", C.c_void_p), ("b3", C.c_void_p), ("wtv2", C.c_void_p), ("bv2", C.c_void_p),
        ("wtv3", C.c_void_p), ("bv3", C.c_void_p), ("logstd", C.c_float * 4),
    ]


class QuadsimError(RuntimeError):
    pass


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise QuadsimError("hipcc not found: cannot build libquadsim_hip.so")


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -shared: cross-compiles without a GPU."""
    deps = SOURCES + HEADERS
    if (not force and os.path.exists(LIB_PATH)
            and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps)):
        return LIB_PATH
    # -ffp-contract=on: a*b+c fuses only where one expression says so, so every kernel that inlines the same device
    # function computes the same bits (the serial and the role-split step kernel, the policy kernels' env step);
    # hipcc's default (fast) fuses across statements depending on the surrounding code.  Same instruction count.
    # -fno-slp-vectorize: SLP packs the scalar f32 chains into v_pk_* pairs at the price of ~300 extra
    # v_mov and +56 VGPRs; measured 5 % slower on the step kernel (profiles/r01/ab_slp.txt)
    cmd = [hipcc_path(), "-std=c++20", "-O3", "-fno-slp-vectorize", "-ffp-contract=on", "--offload-arch=gfx950", "-fPIC", "-shared",
           "-Wno-unused-result", *SOURCES, "-lhsa-runtime64", "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def load():
    """dlopen the library; raises QuadsimError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise QuadsimError(
            "%s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). quadsim_amd has no CPU fallback." % LIB_PATH)
    # torch (the owner of device memory and streams in this package) bundles its own HIP runtime;
    # it must be the one already resident when libquadsim_hip.so resolves libamdhip64, otherwise two
    # runtimes end up in one process and device pointers / streams cannot be shared.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    vp, i64, u64, i32, f32 = C.c_void_p, C.c_int64, C.c_uint64, C.c_int32, C.c_float
    sig = {
        "qs_config_default": [C.POINTER(QsConfig)],
        "qs_version": [],
        "qs_create": [C.POINTER(QsConfig), C.POINTER(vp)],
        "qs_destroy": [vp],
        "qs_reset": [vp, vp, vp],
        "qs_step": [vp, vp, vp, vp, vp, vp, vp],
        "qs_rollout": [vp, i64, vp, vp, vp, vp, vp],
        "qs_rollout_stepwise": [vp, i64, vp, vp, vp, vp, vp],
        "qs_rollout_slab": [vp, i64, vp, vp, vp],
        "qs_fill_random_actions": [vp, i64, u64, vp],
        "qs_get_state": [vp] + [vp] * 6,
        "qs_set_state": [vp] + [vp] * 6,
        "qs_set_params": [vp, vp, vp],
        "qs_get_params": [vp, vp, vp],
        "qs_set_init_state": [vp, vp, vp],
        "qs_get_init_state": [vp, vp, vp],
        "qs_obs_dim": [vp, C.POINTER(i32)],
        "qs_get_step_counter": [vp, C.POINTER(u64)],
        "qs_set_step_counter": [vp, u64],
        "qs_set_stream": [vp, vp, i32],
        "qs_sync": [vp],
        "qs_timer_start": [vp],
        "qs_timer_stop": [vp, C.POINTER(f32)],
        "qs_drone_step": [vp, i64, vp, vp, vp, vp, vp],
        "qs_ctrl": [vp, i64, i32, vp, vp, vp, f32, vp],
        "qs_rel_obs": [vp, i64, vp, vp, vp],
        "qs_transform": [vp, i32, i64, vp, vp],
        "qs_gae": [vp, i64, i64, vp, vp, vp, vp, vp, f32, f32, vp, vp],
        "qs_swap_and_flatten": [vp, i64, i64, i64, vp, vp],
        "qs_expert_action": [vp, vp, f32, f32, vp],
        "qs_policy_rollout": [vp, i64] + [vp] * 11,
        "qs_policy_rollout_fast": [vp, i64] + [vp] * 6,
        "qs_policy_rollout_fast_blob_bytes": [],
        "qs_policy_forward": [vp, i64] + [vp] * 8,
        "qs_policy_forward_fast": [vp, i64, vp, vp, vp],
        "qs_runner_rollout": [vp, i64, C.POINTER(QsActorCritic)] + [vp] * 12,
        "qs_runner_rollout_fast": [vp, i64, vp, C.POINTER(C.c_float), i32] + [vp] * 12,
        "qs_runner_rollout_fast_blob_bytes": [],
        "qs_step_ex": [vp] * 8,
        "qs_set_groups": [vp, i32, i32],
        "qs_group_count": [vp, C.POINTER(i32)],
        "qs_group_range": [vp, i32, C.POINTER(i64), C.POINTER(i64)],
        "qs_group_stream": [vp, i32, C.POINTER(vp)],
        "qs_group_set_stream": [vp, i32, vp],
        "qs_step_group": [vp, i32] + [vp] * 7,
        "qs_step_groups": [vp] * 8,
        "qs_groups_fork": [vp],
        "qs_groups_join": [vp],
        "qs_swap_and_flatten_u8": [vp, i64, i64, vp, vp],
        "qs_gae_flatten": [vp, i64, i64, vp, vp, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp, vp, vp, vp],
        "qs_episode_stats": [vp, i64, i64, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp],
        "qs_set_rollout_layout": [vp, i32],
        "qs_set_queue_mode": [vp, i32],
        "qs_get_queue_mode": [vp, C.POINTER(i32)],
        "qs_set_queue_ordering": [vp, i32],
        "qs_get_queue_ordering": [vp, C.POINTER(i32)],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.qs_last_error.argtypes = []
    lib.qs_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != QS_OK:
        msg = load().qs_last_error().decode("utf-8", "replace")
        raise QuadsimError("%s failed (%d): %s" % (what or "quadsim call", rc, msg))


def default_config():
    cfg = QsConfig()
    check(load().qs_config_default(C.byref(cfg)), "qs_config_default")
    return cfg
