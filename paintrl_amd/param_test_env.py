"""ParamTestEnv: the N x N visit-counting toy env used for hyper-parameter tests.

Host-only plumbing (BASELINE.json configs[0]); mirrors ``PaintRLEnv/param_test_env.py``
(ParamTestEnv pte:96-246, SectionObservation pte:66-93) with the same constructor,
spaces, reward / termination / observation arithmetic.  ``zigzag`` and ``spiral``
are this package's scripted drivers in the spirit of pte:279-342.
"""
import numpy as np

from . import spaces


class ParamTestEnv(spaces.Env):
    """``OBS_MODE`` (class attribute, read when an env is constructed, pte:99, 132-139): 'section' (four quadrant ratios,
    pte:66-93), 'grid' (Grid10Observation pte:49-63: 10 x 10 cells of 2 x 2 interior squares -- as in the reference only
    sizes up to 22 fit), 'direct' (DirectObservation pte:23-30: the whole world), anything else (the reference's 'simple'):
    no status part.  The agent's position / size is appended in every mode (pte:199-203)."""
    reward_range = (-1e3, 1e3)
    action_space = spaces.Discrete(4)
    OBS_MODE = 'section'
    observation_space = spaces.Box(low=0.0, high=1.0, shape=(6,), dtype=np.float64)

    @classmethod
    def change_obs_mode(cls, mode, size=22):
        """Set OBS_MODE and the matching observation_space (the reference derives the space from the class attribute
        when the class body runs, pte:101-108; 'direct' there is written for size 42)."""
        cls.OBS_MODE = mode
        shape = {'section': (6,), 'simple': (2,), 'direct': (size * size + 2,), 'grid': (102,)}.get(mode, (2,))
        cls.observation_space = spaces.Box(low=0.0, high=1.0, shape=shape, dtype=np.float64)

    def __init__(self, size, max_len=900, train_mode=True, termination_by_repeat=False):
        self.size = size
        self.EPISODE_MAX_LENGTH = max(max_len, (self.size - 2) ** 2)           # pte:114
        self._mode = train_mode
        self.repeat_termination = termination_by_repeat
        edge = (0, size - 1)
        self._init_world = np.ones((size, size), dtype=np.int64)
        self._init_world[list(edge), :] = 0
        self._init_world[:, list(edge)] = 0
        self.init_reward_counter = int(self._init_world.sum())
        self.ACTION_DEF = {0: '>', 1: '^', 2: '<', 3: 'v'}
        self._obs_mode = self.OBS_MODE                                      # pte:132-139: fixed at construction
        self._grid_max_counter = int(self.init_reward_counter / 100)           # pte:54
        self.reset()

    def get_current_pos(self):
        return self._i, self._j

    def reset(self):                                                            # pte:150-160
        self._i = self._j = 1
        self.world = self._init_world.copy()
        self.visit_table = np.zeros_like(self.world)
        self.visit_table[1, 1] += 1
        self._violated_wall = False
        self._repeat_visit = False
        self._reward_counter = self.init_reward_counter
        self._step_counter = 0
        return self._observation()

    def _collect(self):                                                         # pte:206-211
        if self.world[self._i, self._j] > 0:
            self.world[self._i, self._j] -= 1
            self._reward_counter -= 1
            return 1
        return 0

    def _move(self, action):                                                    # pte:162-183
        immediate = self._collect()
        self._step_counter += 1
        if action == 0:
            self._i += 1
        elif action == 1:
            self._j += 1
        elif action == 2:
            self._i -= 1
        elif action == 3:
            self._j -= 1
        else:
            raise IndexError('No such action!')
        if not (0 <= self._i < self.size and 0 <= self._j < self.size):
            self._i = min(max(self._i, 0), self.size - 1)
            self._j = min(max(self._j, 0), self.size - 1)
            self._violated_wall = True
            return immediate
        self.visit_table[self._i, self._j] += 1
        if self.visit_table[self._i, self._j] > 1:
            self._repeat_visit = True
        return immediate

    def _termination(self):                                                     # pte:192-197
        if self._violated_wall or self._reward_counter <= 0 or self._step_counter >= self.EPISODE_MAX_LENGTH - 1:
            return True
        return bool(self._repeat_visit and self.repeat_termination)

    def _status(self):
        n = self.size
        if self._obs_mode == 'section':                                         # pte:66-93
            x, y = self._i, self._j
            inner = self.world[1:n - 1, 1:n - 1]
            # quadrants of the interior relative to the agent: (i<=x, j<=y), (i<=x, j>y), (i>x, j<=y), (i>x, j>y)
            xi = max(min(x, n - 2), 0)
            yj = max(min(y, n - 2), 0)
            quads = (inner[:xi, :yj], inner[:xi, yj:], inner[xi:, :yj], inner[xi:, yj:])
            return np.asarray([0 if q.size == 0 else q.sum() / q.size for q in quads], dtype=np.float64)
        if self._obs_mode == 'grid':                                            # pte:49-63 Grid10Observation
            obs = np.zeros((10, 10), dtype=np.float64)
            for i in range(1, n - 1):                                           # (row-major like the reference's dict walk:
                for j in range(1, n - 1):                                       #  the float sums are order dependent)
                    obs[int(i / 2 + 0.5) - 1][int(j / 2 + 0.5) - 1] += self.world[i, j] / self._grid_max_counter
            return obs.reshape((100,))
        if self._obs_mode == 'direct':                                          # pte:23-30 DirectObservation
            return self.world.astype(np.float64).reshape(-1)
        return np.array([])                                                     # pte:17-20 NoObservation

    def _observation(self):                                                     # pte:199-203
        return np.append(self._status(), [self._i / self.size, self._j / self.size])

    def step(self, action):                                                     # pte:218-236
        immediate = self._move(action)
        reward = 0 if self._violated_wall else self._collect()
        reward += immediate
        penalty = 0.2
        done = self._termination()
        return self._observation(), reward - penalty, done, {'reward': reward, 'penalty': penalty}

    def render(self, mode='human'):
        pass

    def close(self):
        pass

    def seed(self, seed=None):
        return spaces.np_random(seed)[1]


def zigzag(grid_size=22, env=None):
    """Serpentine sweep; returns (steps, total_return, actions)."""
    env = env or ParamTestEnv(grid_size)
    state = env.reset()
    up, horizontal, done, total, actions = True, 0, False, 0.0, []
    while not done:
        cur = round(grid_size * state[-1])
        edge = (grid_size - 2) if up else 1
        if cur % grid_size != edge:
            a = 1 if up else 3
        elif horizontal < 1:
            a, horizontal = 0, horizontal + 1
        else:
            horizontal, up = 0, not up
            continue
        state, r, done, _ = env.step(a)
        actions.append(a)
        total += r
    return len(actions), total, actions


def spiral(grid_size=22, env=None):
    """Inward spiral; returns (steps, total_return, actions)."""
    env = env or ParamTestEnv(grid_size)
    env.reset()
    done, total, actions = False, 0.0, []
    direction, strait, use_len = 0, grid_size - 3, 3
    cur = strait
    while not done:
        cur -= 1
        _, r, done, _ = env.step(direction % 4)
        actions.append(direction % 4)
        if cur == 0:
            direction += 1
            use_len -= 1
            if use_len <= 0:
                use_len = 2
                strait -= 1
            cur = strait
        total += r
    return len(actions), total, actions
