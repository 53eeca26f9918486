"""Minimal ``Box`` so the envs expose observation_space / action_space exactly as
docking_env.py:85-95 does even when ``gym`` is not installed (it is not, on the
build image).  When gym is importable its own Box is used."""
import numpy as np

try:  # pragma: no cover - gym is absent on the build image
    from gym.spaces import Box as _GymBox
except Exception:  # noqa: BLE001
    _GymBox = None


class _Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        self.low = np.asarray(low, dtype=self.dtype)
        self.high = np.asarray(high, dtype=self.dtype)
        self.shape = tuple(shape) if shape is not None else self.low.shape

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return np.random.uniform(lo, hi).astype(self.dtype)

    def __repr__(self):
        return "Box(%s, %s)" % (self.shape, self.dtype)


Box = _GymBox or _Box


def docking_spaces():
    """(observation_space, action_space) of docking-v0/v2: docking_env.py:85-95"""
    obs_low = np.array([-np.inf, -np.inf, -np.inf, -100, -100, -100, -np.pi, -np.pi / 2, -np.pi,
                        -10 * np.pi, -10 * np.pi, -10 * np.pi])
    obs_high = -obs_low
    action_space = Box(low=np.array([-1.0, -1.0, -1.0, -1.0]), high=np.array([1.0, 1.0, 1.0, 1.0]), dtype=np.float32)
    observation_space = Box(low=obs_low, high=obs_high, dtype=np.float32)
    return observation_space, action_space


def hovering_spaces():
    """(observation_space, action_space) of hovering-v0: hovering_env.py:37-41 with Drone.state_lim_* (quadrotor.py:35-39)"""
    tp = 10 * 2 * np.pi
    low = np.array([-100, -100, 0, -100, -100, -100, -100, -100, -100, -100, -tp, -tp, -tp])
    high = np.array([100, 100, 100, 100, 100, 100, 100, 100, 100, 100, tp, tp, tp])
    return Box(low=low, high=high, dtype=np.float32), Box(low=np.zeros(4), high=np.ones(4), dtype=np.float32)
