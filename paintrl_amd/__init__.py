"""paintrl_amd: MI355X-native batched paint-coverage simulator (the PaintGymEnv.step() hot path of
translearn/PaintRL) behind the reference's Gym-facing API.

Host side (numpy): part_tables (mesh -> static tables), device_tables (device layout), config.
Device side: libpaintrl_hip.so (csrc/paintrl_hip.hip + prl_*.hpp, csrc/policy_mlp.hip) through the C ABI
of include/paintrl.h.  Rollout driver: rollout (RolloutWorker, PPO step), policy (FusedPolicy).
"""
from .config import EXTRA_CONFIG, PaintToolProfile, Part_Dict, make_config  # noqa: F401
from .param_test_env import ParamTestEnv  # noqa: F401

__all__ = ['PaintGymEnv', 'Robot', 'BatchedPaintEnv', 'FusedPolicy', 'ParamTestEnv', 'Part_Dict', 'EXTRA_CONFIG',
           'PaintToolProfile', 'make_config']


def __getattr__(name):            # torch-dependent classes are imported lazily
    if name == 'PaintGymEnv':
        from .robot_gym_env import PaintGymEnv
        return PaintGymEnv
    if name == 'Robot':
        from .robot_gym_env import Robot
        return Robot
    if name == 'BatchedPaintEnv':
        from .batched_env import BatchedPaintEnv
        return BatchedPaintEnv
    if name == 'FusedPolicy':
        from .policy import FusedPolicy
        return FusedPolicy
    raise AttributeError(name)
