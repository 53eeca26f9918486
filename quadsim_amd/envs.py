"""Single-env shims with the old-gym ``Env`` protocol of the reference:
``DockingEnv`` (gym-docking/gym_docking/envs/docking_env.py:12-254) and
``MovingDockingEnv`` (moving_docking_env.py).  One env (N = 1) on the GPU with
host I/O: ``reset() -> ndarray(12,)``, ``step(a) -> (obs, reward, done, info)``;
``step`` never resets by itself (docking_env.py:104-231).  The attribute surface
the reference scripts rely on is kept.
"""
import ctypes as C

import numpy as np

from . import _lib
from .drone import Drone
from .spaces import docking_spaces


class _ChaserView(Drone):
    """env.chaser / env.target: constants + get_* only (run_expert_policy.py:41, run_trained_docking_ppo2.py:45)"""


class _SingleDockingEnv:
    metadata = {"render.modes": ["human"]}
    _kind = _lib.KIND_V0
    spec = None
    reward_range = (-float("inf"), float("inf"))

    def __init__(self, device=0, integrator="frozen"):
        self._lib = _lib.load()
        cfg = _lib.default_config()
        cfg.kind = self._kind
        cfg.num_envs = 1
        cfg.device = device
        cfg.integrator = _lib.INTEG_RK4 if integrator == "rk4" else _lib.INTEG_FROZEN
        cfg.auto_reset = 0
        cfg.io_space = _lib.IO_HOST
        self._config_hook(cfg)
        self._h = C.c_void_p()
        _lib.check(self._lib.qs_create(C.byref(cfg), C.byref(self._h)), "qs_create")
        self._nominal_chaser = np.array([8, -50, 5, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0], dtype=np.float64)
        self.chaser = _ChaserView(device)
        self.target = _ChaserView(device)
        self.observation_space, self.action_space = docking_spaces()
        self.obs_low, self.obs_high = self.observation_space.low, self.observation_space.high
        self.chaser_ini_state = np.array([8, -50, 5, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0], dtype=np.float64)
        self.target_ini_state = np.array([10, -50, 5, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0], dtype=np.float64)
        self.chaser_dock_port = np.array([0.1, 0.0, 0.0])
        self.target_dock_port = np.array([-0.1, 0.0, 0.0])
        self.action_mean = np.ones(4) * self.chaser.mass * self.chaser.gravity / 2.0
        self.action_std = np.ones(4) * self.chaser.mass * self.chaser.gravity / 2.0
        self.t = 0
        self.done = False
        self.reward = 0.0
        self.last_shaping = 0.0
        self.np_random = None
        self._pull_state()
        self.rel_state = self._obs_now()
        self.seed()

    def _config_hook(self, cfg):
        pass

    # -- helpers -------------------------------------------------------------
    def _pull_state(self):
        sc = np.zeros((1, 13), np.float32); st = np.zeros((1, 13), np.float32)
        ls = np.zeros(1, np.float32); t = np.zeros(1, np.float32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        _lib.check(self._lib.qs_get_state(self._h, p(sc), p(st), None, None, p(ls), p(t)), "qs_get_state")
        self.state_chaser = sc[0].astype(np.float64)
        self.state_target = st[0].astype(np.float64)
        self.last_shaping = float(ls[0])

    def _obs_now(self):
        from .drone import rel_obs_batch
        return rel_obs_batch(self.state_chaser[None], self.state_target[None])[0].astype(np.float64)

    # -- gym.Env protocol ------------------------------------------------------
    def reset(self):
        obs = np.zeros((1, 12), np.float32)
        # honour a script-mutated chaser_ini_state (run_expert_policy.py:44,63-64): reset() restores THAT state
        if not np.array_equal(self.chaser_ini_state, self._nominal_chaser):
            sc = np.ascontiguousarray(self.chaser_ini_state[None], dtype=np.float32)
            st = np.ascontiguousarray(self.target_ini_state[None], dtype=np.float32)
            _lib.check(self._lib.qs_set_init_state(self._h, sc.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p)),
                       "qs_set_init_state")
            self._nominal_chaser = self.chaser_ini_state.copy()
        _lib.check(self._lib.qs_reset(self._h, None, obs.ctypes.data_as(C.c_void_p)), "qs_reset")
        self._pull_state()
        self.done = False
        self.t = 0.0
        self.reward = 0.0
        self.rel_state = obs[0].astype(np.float64)
        return self.rel_state

    def step(self, action):
        a = np.ascontiguousarray(np.asarray(action, dtype=np.float32).reshape(1, 4))
        obs = np.zeros((1, 12), np.float32); rew = np.zeros(1, np.float32)
        done = np.zeros(1, np.uint8); flags = np.zeros(1, np.uint8)
        p = lambda x: x.ctypes.data_as(C.c_void_p)  # noqa: E731
        _lib.check(self._lib.qs_step(self._h, p(a), p(obs), p(rew), p(done), p(flags), None), "qs_step")
        self.t += 1
        self._pull_state()
        self.rel_state = obs[0].astype(np.float64)
        self.reward = float(rew[0])
        self.done = bool(done[0])
        info = {"chaser": self.state_chaser, "target": self.state_target,
                "flag_docking": bool(flags[0] & _lib.FLAG_DOCKED),
                "done_overlimit": bool(flags[0] & _lib.FLAG_OVERLIMIT)}
        return self.rel_state, self.reward, self.done, info

    def render(self, mode="human"):
        return None

    def close(self):
        if self._h:
            self._lib.qs_destroy(self._h)
            self._h = C.c_void_p()
        return None

    def seed(self, seed=None):
        self.np_random = np.random.RandomState(seed)
        return [seed]

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DockingEnv(_SingleDockingEnv):
    """docking-v0"""
    _kind = _lib.KIND_V0


class MovingDockingEnv(_SingleDockingEnv):
    """docking-v2"""
    _kind = _lib.KIND_V2


class ImitatingDockingEnv(_SingleDockingEnv):
    """docking-v1 (imitating_docking_env.py): v0 whose chaser start is jittered once at construction (:34).
    The jitter comes from the rocRAND INIT stream keyed by `seed`; set ``chaser_ini_state`` + reset() to inject one."""
    _kind = _lib.KIND_V1

    def __init__(self, device=0, integrator="frozen", seed=0):
        self._seed0 = seed
        super().__init__(device, integrator)
        sc = np.zeros((1, 13), np.float32); st = np.zeros((1, 13), np.float32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        _lib.check(self._lib.qs_get_init_state(self._h, p(sc), p(st)), "qs_get_init_state")
        self.chaser_ini_state = sc[0].astype(np.float64)
        self._nominal_chaser = self.chaser_ini_state.copy()

    def _config_hook(self, cfg):
        cfg.seed = int(self._seed0)


class HoveringEnv:
    """hovering-v0 (hovering_env.py:10-92): one drone, obs = raw state [13], action in [0,1]^4, step never resets."""
    metadata = {"render.modes": ["human"]}

    def __init__(self, device=0, integrator="frozen", seed=0):
        from .spaces import hovering_spaces
        self._lib = _lib.load()
        cfg = _lib.default_config()
        cfg.kind, cfg.num_envs, cfg.device, cfg.auto_reset, cfg.io_space = _lib.KIND_HOVER, 1, device, 0, _lib.IO_HOST
        cfg.integrator = _lib.INTEG_RK4 if integrator == "rk4" else _lib.INTEG_FROZEN
        cfg.seed = int(seed)
        self._h = C.c_void_p()
        _lib.check(self._lib.qs_create(C.byref(cfg), C.byref(self._h)), "qs_create")
        self.drone = _ChaserView(device)
        self.observation_space, self.action_space = hovering_spaces()
        self.action_max = np.ones(4) * self.drone.mass * self.drone.gravity
        ini = np.zeros((1, 13), np.float32)
        _lib.check(self._lib.qs_get_init_state(self._h, ini.ctypes.data_as(C.c_void_p), None), "qs_get_init_state")
        self.ini_state = ini[0].astype(np.float64)
        self.state_des = np.array([0, 0, 5.0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0])
        self.state = np.zeros(13)
        self.np_random = np.random.RandomState(seed)

    def reset(self):
        ini = np.ascontiguousarray(self.ini_state[None], dtype=np.float32)      # honours a caller-set ini_state
        _lib.check(self._lib.qs_set_init_state(self._h, ini.ctypes.data_as(C.c_void_p), None), "qs_set_init_state")
        obs = np.zeros((1, 13), np.float32)
        _lib.check(self._lib.qs_reset(self._h, None, obs.ctypes.data_as(C.c_void_p)), "qs_reset")
        self.state = obs[0].astype(np.float64)
        return self.state

    def step(self, action):
        a = np.ascontiguousarray(np.asarray(action, dtype=np.float32).reshape(1, 4))
        obs = np.zeros((1, 13), np.float32); rew = np.zeros(1, np.float32); done = np.zeros(1, np.uint8)
        p = lambda x: x.ctypes.data_as(C.c_void_p)  # noqa: E731
        _lib.check(self._lib.qs_step(self._h, p(a), p(obs), p(rew), p(done), None, None), "qs_step")
        self.state = obs[0].astype(np.float64)
        return self.state, float(rew[0]), bool(done[0]), {}

    def render(self, mode="human"):
        return None

    def seed(self, seed=None):
        self.np_random = np.random.RandomState(seed)
        return [seed]

    def close(self):
        if self._h:
            self._lib.qs_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def register_gym_ids():
    """register 'docking-v0' / 'docking-v2' with gym when gym is importable
    (gym-docking/gym_docking/__init__.py:3-17); returns True if registered."""
    try:
        from gym.envs.registration import register
    except Exception:  # noqa: BLE001
        return False
    for gid, cls in (("docking-v0", "DockingEnv"), ("docking-v2", "MovingDockingEnv"),
                     ("docking-v1", "ImitatingDockingEnv"), ("hovering-v0", "HoveringEnv")):
        try:
            register(id=gid, entry_point="quadsim_amd.envs:%s" % cls)
        except Exception:  # already registered
            pass
    return True


def make(env_id, **kw):
    """gym.make stand-in for the two ids of the hot path"""
    name = env_id.split(":")[-1]
    if name == "docking-v0":
        return DockingEnv(**kw)
    if name == "docking-v2":
        return MovingDockingEnv(**kw)
    if name == "docking-v1":
        return ImitatingDockingEnv(**kw)
    if name == "hovering-v0":
        return HoveringEnv(**kw)
    raise ValueError("unknown env id %r" % env_id)
