"""Minimal stand-ins for ``gym.spaces.Box`` / ``gym.spaces.Discrete`` and ``gym.Env``.

The reference's envs subclass ``gym.Env`` and publish ``action_space`` /
``observation_space`` (``PaintRLEnv/robot_gym_env.py:120-173``,
``PaintRLEnv/param_test_env.py:96-110``).  ``gym`` is not installed on the build
or the GPU box; the real package is used when importable, otherwise these.
"""
import numpy as np

try:  # pragma: no cover - gym is absent in this image
    import gym as _gym
    from gym import spaces as _spaces
    Env = _gym.Env
    Box = _spaces.Box
    Discrete = _spaces.Discrete
    HAVE_GYM = True
except Exception:  # noqa: BLE001
    HAVE_GYM = False

    class Env(object):
        metadata = {'render.modes': []}
        reward_range = (-float('inf'), float('inf'))
        action_space = None
        observation_space = None

        def step(self, action):
            raise NotImplementedError

        def reset(self):
            raise NotImplementedError

        def render(self, mode='human'):
            raise NotImplementedError

        def close(self):
            pass

        def seed(self, seed=None):
            return []

        def __enter__(self):
            return self

        def __exit__(self, *args):
            self.close()
            return False

    class Space(object):
        def __init__(self, shape=None, dtype=None):
            self.shape = None if shape is None else tuple(shape)
            self.dtype = None if dtype is None else np.dtype(dtype)
            self._rng = np.random.default_rng()

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)
            return [seed]

    class Box(Space):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            if shape is None:
                low = np.asarray(low)
                high = np.asarray(high)
                shape = low.shape
            else:
                low = np.full(shape, low)
                high = np.full(shape, high)
            Space.__init__(self, shape, dtype)
            self.low = low.astype(self.dtype)
            self.high = high.astype(self.dtype)

        def sample(self):
            return self._rng.uniform(self.low, self.high).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low)) and bool(np.all(x <= self.high))

        def __repr__(self):
            return 'Box%s' % (self.shape,)

    class Discrete(Space):
        def __init__(self, n):
            Space.__init__(self, (), np.int64)
            self.n = int(n)

        def sample(self):
            return int(self._rng.integers(self.n))

        def contains(self, x):
            try:
                xi = int(x)
            except (TypeError, ValueError):
                return False
            return xi == x and 0 <= xi < self.n

        def __repr__(self):
            return 'Discrete(%d)' % self.n


def np_random(seed=None):
    """``gym.utils.seeding.np_random``: returns (generator, seed)."""
    if seed is None:
        seed = int(np.random.SeedSequence().entropy % (2 ** 31))
    return np.random.RandomState(seed % (2 ** 32)), seed
