"""Stub of ``gym`` built from paintrl_amd.spaces (robot_gym_env.py:8-10, param_test_env.py:1-3)."""
from paintrl_amd.spaces import Env  # noqa: F401
from . import spaces, utils, error, logger  # noqa: F401
