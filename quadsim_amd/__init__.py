"""quadsim_amd -- MI355X-native drop-in for QuadSim's env.step() hot path.

Python host code over a C ABI (include/quadsim.h) into hand-written HIP kernels
for gfx950.  No CPU fallback: importing is cheap, but every env needs the built
library and a GPU.
"""
from . import _lib
from ._lib import QuadsimError, build_library
from .drone import Drone, controller, ctrl_batch, drone_step_batch, rel_obs_batch, transform_batch
from .envs import DockingEnv, HoveringEnv, ImitatingDockingEnv, MovingDockingEnv, make, register_gym_ids
from .vec_env import C3_INIT_RANGE, VecDockingEnv, shard_range
from . import distributed
from .policy import MlpPolicy, fused_policy_rollout, rollout_with_policy
from .rollout_buffer import EpisodeTracker, compute_gae, gae_and_flatten, swap_and_flatten
from .expert import PIDExpert, record_expert_dataset
from .runner import ActorCriticPolicy, Runner, fused_runner_rollout

__all__ = ["VecDockingEnv", "DockingEnv", "MovingDockingEnv", "ImitatingDockingEnv", "HoveringEnv", "Drone", "controller", "make", "register_gym_ids",
           "shard_range", "build_library", "QuadsimError", "C3_INIT_RANGE", "drone_step_batch", "ctrl_batch",
           "rel_obs_batch", "transform_batch", "_lib", "distributed", "MlpPolicy", "rollout_with_policy", "fused_policy_rollout", "compute_gae", "swap_and_flatten", "gae_and_flatten", "EpisodeTracker", "PIDExpert", "record_expert_dataset", "ActorCriticPolicy", "Runner", "fused_runner_rollout"]

register_gym_ids()
