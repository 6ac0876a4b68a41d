"""RLlib ``VectorEnv`` adaptor around BatchedPaintEnv (paint_ppo.py:135-137 registers ONE ``PaintGymEnv`` per rollout worker and
lets RLlib vectorise by process; here one worker drives N envs of one GPU through a single kernel launch per step).

    from ray.tune.registry import register_env
    register_env('paint_vector', lambda env_config: PaintVectorEnv.from_env_config(env_config, num_envs=4096))
    # config: {"env": "paint_vector", "remote_worker_envs": False, ...}

``ray`` is optional: when it is importable the class derives from ``ray.rllib.env.vector_env.VectorEnv`` (so RLlib's sampler
accepts it), otherwise from ``object`` with the same method names and return conventions (lists of per-env numpy observations,
float rewards, bool dones, info dicts with the reference's ``reward`` / ``penalty`` keys, rge:368).  Observations cross to the
host here -- that is RLlib's interface; a rollout that stays on the device uses paintrl_amd.rollout.RolloutWorker instead.
"""
import numpy as np

from . import spaces as _spaces

try:                                                     # pragma: no cover - ray is not installed in the build image
    from ray.rllib.env.vector_env import VectorEnv as _Base
except Exception:  # noqa: BLE001
    _Base = object


class PaintVectorEnv(_Base):
    def __init__(self, batched_env, auto_reset_on_done=False):
        """``batched_env``: a BatchedPaintEnv created WITHOUT in-kernel auto-reset (RLlib resets finished sub-envs itself
        through ``reset_at``); discrete or continuous actions as configured there."""
        self.env = batched_env
        self.num_envs = int(batched_env.n_envs)
        if batched_env.cfg.auto_reset and not auto_reset_on_done:
            raise ValueError('PaintVectorEnv: create the BatchedPaintEnv with auto_reset=False (RLlib calls reset_at)')
        od = batched_env.obs_dim
        self.observation_space = _spaces.Box(low=0.0, high=1.0, shape=(od,), dtype=np.float64)      # rge:166-173
        if batched_env.discrete:
            self.action_space = _spaces.Discrete(int(batched_env.cfg.n_discrete))
        else:
            self.action_space = _spaces.Box(low=-1.0, high=1.0, shape=(int(batched_env.action_dim),), dtype=np.float64)
        if _Base is not object:                          # pragma: no cover
            _Base.__init__(self, self.observation_space, self.action_space, self.num_envs)

    @classmethod
    def from_env_config(cls, env_config, num_envs, device=None):
        """``env_config`` as paint_ppo.py:84-123 builds it for PaintGymEnv (``urdf_root``, ``extra_config`` ...); the part is
        loaded once and shared by the ``num_envs`` sub-environments."""
        from .robot_gym_env import PaintGymEnv
        return cls(PaintGymEnv.make_batched(num_envs, device=device, auto_reset=False, **dict(env_config)))

    # ---- VectorEnv interface
    def vector_reset(self):
        return [row for row in self.env.reset().cpu().numpy()]

    def reset_at(self, index=None):
        return self.env.reset_at(0 if index is None else int(index)).cpu().numpy()

    def vector_step(self, actions):
        a = np.asarray(actions)
        obs, rew, done, info = self.env.step(a)
        obs, rew, done, info = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy(), info.cpu().numpy()
        return ([row for row in obs], [float(r) for r in rew], [bool(d) for d in done],
                [{'reward': float(i[0]), 'penalty': float(i[1])} for i in info])

    def get_sub_environments(self):                      # (no per-env Python objects exist: the batch IS the environment)
        return []

    get_unwrapped = get_sub_environments                 # (older RLlib name)

    def close(self):
        self.env.close()
