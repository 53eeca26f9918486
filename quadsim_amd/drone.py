"""Layer-1 mirrors of the reference's ``dynamics.quadrotor.Drone`` and
``controller.PIDController.controller``, computed on the GPU through the C ABI's
layer-1 entry points (qs_drone_step / qs_ctrl, host I/O).  Same names, argument
meaning and in-place mutation behaviour as the reference so that
run_sim_PID.py:23-54-style loops read unchanged.
"""
import atexit
import ctypes as C

import numpy as np

from . import _lib

_ctx = {}


def _close_contexts():
    """qs_destroy every cached layer-1 handle (registered with atexit; also callable to release the device early)"""
    lib = _lib.load() if _ctx else None
    while _ctx:
        _, h = _ctx.popitem()
        lib.qs_destroy(h)


atexit.register(_close_contexts)


def _context(device=0, integrator=_lib.INTEG_FROZEN, dt=0.02, mass=0.18, inertia=(0.00025, 0.000232, 0.0003738)):
    """small host-I/O handle used as the device/stream context of the layer-1 calls"""
    key = (device, integrator, float(dt), float(mass), tuple(float(x) for x in inertia))
    h = _ctx.get(key)
    if h is None:
        lib = _lib.load()
        cfg = _lib.default_config()
        cfg.num_envs = 1
        cfg.device = device
        cfg.integrator = integrator
        cfg.dt = dt
        cfg.io_space = _lib.IO_HOST
        cfg.mass = mass
        cfg.inertia = (C.c_float * 3)(*inertia)
        h = C.c_void_p()
        _lib.check(lib.qs_create(C.byref(cfg), C.byref(h)), "qs_create")
        _ctx[key] = h
    return h


def _f32(x, shape):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float32).reshape(shape))


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def drone_step_batch(state, u_prev, u, par=None, dt=0.02, integrator="frozen", device=0):
    """n independent Drone.step calls (dynamics/quadrotor.py:126-144).
    -> (state' [n,13], u_prev' [n,4], limited [n] bool)"""
    state = _f32(state, (-1, 13)).copy()
    n = state.shape[0]
    u_prev = _f32(u_prev, (n, 4)).copy()
    u = _f32(u, (n, 4))
    par = None if par is None else _f32(par, (n, 4))
    lim = np.zeros(n, np.uint8)
    h = _context(device, _lib.INTEG_RK4 if integrator == "rk4" else _lib.INTEG_FROZEN, dt)
    _lib.check(_lib.load().qs_drone_step(h, n, _p(state), _p(u_prev), _p(u), _p(par), _p(lim)), "qs_drone_step")
    return state, u_prev, lim.astype(bool)


def ctrl_batch(mode, state_des, state, state_last=None, mass=0.18, device=0):
    """n controller.PID (mode 0) / vel_controller (mode 1) calls.
    -> (u [n,4], mutated state_des [n,13])"""
    state_des = _f32(state_des, (-1, 13)).copy()
    n = state_des.shape[0]
    state = _f32(state, (n, 13))
    sl = None if state_last is None else _f32(state_last, (n, 13))
    u = np.zeros((n, 4), np.float32)
    h = _context(device)
    _lib.check(_lib.load().qs_ctrl(h, n, int(mode), _p(state_des), _p(state), _p(sl), C.c_float(mass), _p(u)), "qs_ctrl")
    return u, state_des


def rel_obs_batch(chaser, target, device=0):
    """state2rel over dock ports for n (chaser, target) pairs -> obs [n,12]"""
    chaser = _f32(chaser, (-1, 13))
    n = chaser.shape[0]
    target = _f32(target, (n, 13))
    obs = np.zeros((n, 12), np.float32)
    _lib.check(_lib.load().qs_rel_obs(_context(device), n, _p(chaser), _p(target), _p(obs)), "qs_rel_obs")
    return obs


_TRANSFORM = {"quat2euler": (0, 4, 3), "euler2quat": (1, 3, 4), "quat2rot": (2, 4, 9), "rot2euler": (3, 9, 3)}


def transform_batch(name, x, device=0):
    """utils/transform.py on the GPU for n inputs: 'quat2euler' [n,4]->[n,3], 'euler2quat' [n,3]->[n,4],
    'quat2rot' [n,4]->[n,3,3], 'rot2euler' [n,3,3]->[n,3]"""
    op, wi, wo = _TRANSFORM[name]
    x = _f32(x, (-1, wi))
    out = np.zeros((x.shape[0], wo), np.float32)
    _lib.check(_lib.load().qs_transform(_context(device), op, x.shape[0], _p(x), _p(out)), "qs_transform")
    return out.reshape(-1, 3, 3) if name == "quat2rot" else out


class Drone:
    """dynamics/quadrotor.py:5-63 attribute surface + reset/step on the GPU."""

    def __init__(self, device=0, integrator="frozen"):
        self.dt = 0.02
        self.t0 = 0
        self.t = self.t0
        self.gravity = 9.81
        self.mass = 0.18
        self.Inertia = np.diag([0.00025, 0.000232, 0.0003738])
        self.arm_length = 0.086
        self.F_max = 4 * self.mass * self.gravity
        self.F_min = 0
        self.dim_state = 13
        self.dim_u = 4
        self.state = np.zeros(self.dim_state)
        self.initial_state = np.array([0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0], dtype=np.float64)
        self.u = np.zeros(self.dim_u)
        self.kf = 6.11e-8
        self.km = 1.5e-9
        self.motor_lambda = self.km / self.kf
        L, lam = self.arm_length, self.motor_lambda
        self.rotor2control = np.array([[1, 1, 1, 1], [0, L, 0, -L], [-L, 0, L, 0], [lam, -lam, lam, -lam]])
        self.dock_port_inB_pos = np.array([0.05, 0, 0])
        self._device = device
        self._integrator = integrator

    def reset(self, reset_state=None, dock_port=None):
        if reset_state is not None:
            self.initial_state = np.asarray(reset_state, dtype=np.float64)
        if dock_port is not None:
            self.dock_port_inB_pos = np.asarray(dock_port, dtype=np.float64)
        self.state = self.initial_state.copy()
        self.u = np.zeros(self.dim_u)
        return self.state

    def step(self, u):
        par = np.array([[self.mass, self.Inertia[0, 0], self.Inertia[1, 1], self.Inertia[2, 2]]], np.float32)
        s, up, _ = drone_step_batch(self.state[None], self.u[None], np.asarray(u)[None], par=par, dt=self.dt,
                                    integrator=self._integrator, device=self._device)
        self.state = s[0].astype(np.float64)
        self.u = up[0].astype(np.float64)
        self.t = self.t + self.dt
        return self.state

    def get_state(self):
        return self.state

    def get_time(self):
        return self.t

    def get_arm_length(self):
        return self.arm_length

    def get_mass(self):
        return self.mass

    def get_dock_port_state(self):
        """dynamics/quadrotor.py:213-224: {'pos', 'vel', 'quat', 'angular_rate'} of the dock port.  quat2rot runs on the
        GPU (qs_transform); the two 3-vector products are layer-1 glue, not the env path (the fused step evaluates the
        ports in-kernel and never needs 'quat', which state2rel ignores, docking_env.py:263-264).  With the reference's
        unit-diagonal "rotation" the trace branch of rot2quat (:259-289) always fires: quat = (1, q0 n1, q0 n2, q0 n3)."""
        R = transform_batch("quat2rot", self.state[6:10][None], device=self._device)[0].astype(np.float64)
        b = R.T @ self.dock_port_inB_pos
        w = self.state[10:13]
        quat = np.array([1.0, (R[2, 1] - R[1, 2]) / 4.0, (R[0, 2] - R[2, 0]) / 4.0, (R[1, 0] - R[0, 1]) / 4.0], np.float32)
        return {"pos": self.state[0:3] + b, "vel": self.state[3:6] + np.cross(w, b), "quat": quat,
                "angular_rate": self.state[10:].copy()}


class controller:
    """controller/PIDController.py:7-50 surface; PID / vel_controller mutate state_des[6:12] in place."""

    def __init__(self, L, mass, device=0):
        self.mass = mass
        self.g = 9.81
        self._device = device

    def PID(self, state_des, state_now):
        u, sd = ctrl_batch(0, np.asarray(state_des)[None], np.asarray(state_now)[None], mass=self.mass, device=self._device)
        state_des[6:12] = sd[0, 6:12]
        return u[0].astype(np.float64)

    def vel_controller(self, state_des, state_now, state_last):
        u, sd = ctrl_batch(1, np.asarray(state_des)[None], np.asarray(state_now)[None], np.asarray(state_last)[None],
                           mass=self.mass, device=self._device)
        state_des[6:12] = sd[0, 6:12]
        return u[0].astype(np.float64)
