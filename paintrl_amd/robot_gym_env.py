"""PaintGymEnv: the Gym-facing drop-in for ``PaintRLEnv/robot_gym_env.py`` (rge:120-422).

Same constructor, class attributes, spaces, classmethods, ``step`` 4-tuple and
``reset`` as the reference, so ``env_creator = lambda cfg: PaintGymEnv(**cfg)``
(paint_ppo.py:135-137) keeps working.  Internally it is a one-env view of
``BatchedPaintEnv``: every ``step`` is one launch of the HIP step kernel.  For
throughput use ``BatchedPaintEnv`` directly (thousands of envs per launch).

Differences from the reference, by design:
  * ``with_robot=True`` (KUKA inverse kinematics through Bullet, SURVEY §2 row 2: out of scope)
    raises NotImplementedError -- pass with_robot=False as paint_ppo.py:87 does;
  * ``Robot.PAINT_METHOD`` ('fast' | 'normal', rob:171-172) is this module's ``Robot`` class
    attribute, read when an env is constructed;
  * ``renders`` only controls printing of the replay buffer; there is no GUI;
  * rays hit an explicit collision triangle set (``collision_mode``), see DESIGN.md.
"""
import os
import random

import numpy as np

from . import config as _config
from . import part_tables as _pt
from . import spaces
from .config import EXTRA_CONFIG as _DEFAULT_EXTRA, PaintToolProfile, Part_Dict

_urdf_cache = {}          # (path, mtime, collision_mode) -> PartTables, like bpw._urdf_cache


def load_part_tables(path, collision_mode='hull', obs_grad=4, paint_radius=None):
    radius = PaintToolProfile.PAINT_RADIUS if paint_radius is None else paint_radius
    key = (os.path.abspath(path), os.path.getmtime(path), collision_mode, radius)
    if key not in _urdf_cache:
        _urdf_cache[key] = _pt.build_part_tables(path, obs_grad=obs_grad, collision_mode=collision_mode,
                                                 paint_radius=radius)
    return _urdf_cache[key]


class Robot(object):
    """Class-level switches of rob:163-172 plus the few ``env.robot`` members scripts touch
    (rge:436, spiral.py:38).  The kinematics themselves run inside the step kernel."""

    PAINT_PER_ACTION = 5                 # compile-time constant of the kernel
    NOT_ON_PART_TERMINATE_STEPS = 1000
    PAINT_METHOD = 'fast'                # 'fast' (ball query, bpw:568-570) | 'normal' (cone beams, rob:280-285)
    BETA = 2                             # rob:169: the beta of the thickness profile (COLOR_MODE 'HSI' beam table)

    def __init__(self, env):
        self._env = env
        self.angle_diff = 0.0            # rob:203

    def reset(self, pose):
        """Robot.reset([pose, orn_normal]) (rob:366-372), as spiral.py:38 uses it."""
        self._env._batch.set_pose(0, pose[0], pose[1])
        self._env._refresh_state()

    def get_angle_diff(self):            # rob:374-375: |new turning angle - previous| of the last action
        return self.angle_diff

    def termination_request(self):       # rob:377-378
        return bool(self._env._state['terminate'])

    def get_observation(self):           # rob:380-381 (the tool pose; orientation as the quaternion it is kept as)
        return self._env._state['pose'].copy(), self._env._state['quat'].copy()


_RobotView = Robot                       # name of round 1


class PaintGymEnv(spaces.Env):
    metadata = {'render.modes': ['human', 'rgb_array'], 'video.frames_per_second': 30}
    reward_range = (-1e3, 1e3)

    # Adjust env by hand when using Ray (rge:126-132)
    ACTION_SHAPE = 1
    ACTION_MODE = 'discrete'
    DISCRETE_GRANULARITY = 4
    OBS_MODE = 'section'
    OBS_GRAD = 4
    EXTRA_CONFIG = dict(_DEFAULT_EXTRA)

    action_space = spaces.Discrete(DISCRETE_GRANULARITY)
    observation_space = spaces.Box(low=0.0, high=1.0, shape=(OBS_GRAD + 2,), dtype=np.float64)

    @classmethod
    def change_obs_mode(cls, mode='section', grad=5):
        """rge:176-193 (the observation_space shape follows the actual observation here)."""
        cls.OBS_MODE = mode
        cls.OBS_GRAD = grad
        cls.observation_space = spaces.Box(low=0.0, high=1.0, shape=(_config.obs_dim(mode, grad),), dtype=np.float64)

    @classmethod
    def change_action_mode(cls, shape=2, mode='continuous', discrete_granularity=20):
        """rge:195-205."""
        cls.ACTION_SHAPE = shape
        cls.ACTION_MODE = mode
        if mode == 'continuous':
            cls.action_space = spaces.Box(low=-1.0, high=1.0, shape=(shape,), dtype=np.float64)
        else:
            cls.DISCRETE_GRANULARITY = discrete_granularity
            cls.action_space = spaces.Discrete(discrete_granularity)

    def __init__(self, urdf_root, with_robot=True, renders=False, render_video=False, rollout=False,
                 extra_config=None, collision_mode='hull', device=None):
        from .batched_env import BatchedPaintEnv
        from .device_tables import DeviceTables
        if extra_config is None:
            extra_config = self.EXTRA_CONFIG
        cfg = dict(_DEFAULT_EXTRA)
        cfg.update(extra_config)
        self._setup_extra_config(cfg)
        if with_robot:
            raise NotImplementedError(
                'with_robot=True needs Bullet inverse kinematics and the KUKA URDF of pybullet_data (rob:193-195, '
                '220-233): out of scope for the batched simulator.  Construct with with_robot=False, as '
                'paint_ppo.py:87 does.')
        if Robot.PAINT_METHOD not in _config.PAINT_METHODS:
            raise ValueError("Robot.PAINT_METHOD must be 'fast' or 'normal', not %r" % (Robot.PAINT_METHOD,))
        self._with_robot = False
        self._renders = renders
        self._rollout = rollout
        self._urdf_root = urdf_root
        self._step_counter = 0
        self.replay_buffer = []
        path = os.path.join(urdf_root, 'urdf', 'painting', self._part_name)
        self._tables = load_part_tables(path, collision_mode, self.OBS_GRAD)
        self._start_points = _pt.start_points(self._tables, self.START_POINT_MODE)
        n_disc = self.action_space.n if self.ACTION_MODE != 'continuous' else 4
        # rob:244-249 set_up_paint_params: COLOR_MODE 'RGB' casts the uniform lattice of beams, 'HSI' the beta-profile rings
        # whose radii are drawn with random.uniform (rob:38-69) -- from Python's `random`, like the reference
        self._paint_plain = _pt.beta_plain(self._tables.density, Robot.BETA) if self.COLOR_MODE == 'HSI' else None
        self._batch = BatchedPaintEnv(
            DeviceTables(self._tables, obs_grad=self.OBS_GRAD, start_points=self._start_points, beams=self._paint_plain),
            1, device=device,
            obs_mode=self.OBS_MODE, obs_grad=self.OBS_GRAD, action_mode=self.ACTION_MODE,
            action_dim=self.ACTION_SHAPE if self.ACTION_MODE == 'continuous' else 1, n_discrete=n_disc,
            termination_mode=self.TERMINATION_MODE, turning_penalty=self.TURNING_PENALTY,
            overlap_penalty=self.OVERLAP_PENALTY, paint_method=Robot.PAINT_METHOD, max_episode_len=self.EPISODE_MAX_LENGTH,
            expected_episode_len=self.Expected_Episode_Length, switch_threshold=self.SWITCH_THRESHOLD,
            max_possible_point=self._max_possible_point, paint_radius=PaintToolProfile.PAINT_RADIUS,
            step_size=PaintToolProfile.STEP_SIZE, color_mode=self.COLOR_MODE)
        self.robot = Robot(self)
        # one device buffer for everything a step returns, so that step() costs ONE device-to-host copy:
        # obs[od] | reward | info[2] | state record[16] | done (u8 in the last 8 bytes)
        import torch
        od = self._batch.obs_dim
        self._pack = torch.zeros(od + 3 + 16 + 1, dtype=torch.float64, device=self._batch.device)
        self._pack_obs = self._pack[:od].view(1, od)
        self._pack_reward, self._pack_info = self._pack[od:od + 1], self._pack[od + 1:od + 3].view(1, 2)
        self._pack_state = self._pack[od + 3:od + 19].view(1, 16)
        self._pack_done = self._pack[od + 19:od + 20].view(torch.uint8)[:1]
        self._scratch_final = torch.zeros((1, od), dtype=torch.float64, device=self._batch.device)
        self._act_i32 = torch.zeros(1, dtype=torch.int32, device=self._batch.device)
        self._state = None
        self.reset()

    def _setup_extra_config(self, config):                                    # rge:240-252
        self._part_name = Part_Dict[config['Part_NO']][0]
        self._max_possible_point = Part_Dict[config['Part_NO']][1]
        self.RENDER_WIDTH, self.RENDER_HEIGHT = config['RENDER_WIDTH'], config['RENDER_HEIGHT']
        self.Expected_Episode_Length = config['Expected_Episode_Length']
        self.EPISODE_MAX_LENGTH = config['EPISODE_MAX_LENGTH']
        self.TERMINATION_MODE = config['TERMINATION_MODE']
        self.SWITCH_THRESHOLD = config['SWITCH_THRESHOLD']
        self.START_POINT_MODE = config['START_POINT_MODE']
        self.TURNING_PENALTY = config['TURNING_PENALTY']
        self.OVERLAP_PENALTY = config['OVERLAP_PENALTY']
        self.COLOR_MODE = config['COLOR_MODE']
        if self.COLOR_MODE not in _config.COLOR_MODES:
            raise ValueError("COLOR_MODE must be 'RGB' or 'HSI', not %r" % (self.COLOR_MODE,))

    def _obs_out(self, row):
        return np.array(row.cpu().numpy(), dtype=np.float64)

    def _decode_state(self, rec):
        ints = rec.view(np.int32)
        return {'pose': rec[0:3].copy(), 'quat': rec[3:7].copy(), 'last_turning_angle': float(rec[7]),
                'total_reward': float(rec[8]), 'total_return': float(rec[9]), 'terminate': int(ints[20]),
                'terminate_counter': int(ints[21]), 'last_on_part': int(ints[22]), 'step_counter': int(ints[23])}

    def _refresh_state(self):
        self._batch.state_into(self._pack_state)
        self._state = self._decode_state(self._pack[self._batch.obs_dim + 3:self._batch.obs_dim + 19].cpu().numpy())

    def step(self, action):                                                    # rge:349-368
        b, od = self._batch, self._batch.obs_dim
        if self.ACTION_MODE == 'continuous':
            import torch
            act = torch.as_tensor(np.asarray(action, dtype=np.float64).reshape(1, -1), device=b.device)
        else:
            self._act_i32.fill_(int(action))
            act = self._act_i32
        b.step_into(act, self._pack_obs, self._pack_reward, self._pack_done, self._pack_info, self._scratch_final)
        b.state_into(self._pack_state)
        host = self._pack.cpu().numpy()                  # the step's only device-to-host copy (and sync)
        prev_angle = self._state['last_turning_angle']
        self._state = self._decode_state(host[od + 3:od + 19])
        self.robot.angle_diff = abs(self._state['last_turning_angle'] - prev_angle)      # rob:357
        self._step_counter += 1
        done = bool(host[od + 19:od + 20].view(np.uint8)[0])
        if self._renders and self._rollout:
            self.replay_buffer.append(action)
            if done:
                print(self.replay_buffer)
        return np.array(host[:od], dtype=np.float64), float(host[od]), done, {'reward': float(host[od + 1]),
                                                                               'penalty': float(host[od + 2])}

    def reset(self):                                                           # rge:370-387
        if self._rollout:
            index = 0
            self.replay_buffer = []
        else:
            random.randint(0, 7)              # the reference draws an (unused) pre-paint mode first, rge:378
            index = random.randint(0, len(self._start_points) - 1)
        self._step_counter = 0
        obs = self._obs_out(self._batch.reset(start_idx=[index])[0])
        self._refresh_state()
        self.robot.angle_diff = 0.0
        return obs

    def get_texture_image(self):
        """Part.get_texture_image() (bpw:737-738): the reference's texel list as a uint8 (W, H, 3) array -- texels outside
        the profiles black, the back side's (0, 255, 0), unpainted front texels (191, 191, 191), painted ones (255, 0, 0),
        with the quirks of the labelling kept (part_tables.label_texture); COLOR_MODE 'HSI': front texels carry their
        thickness byte in all three channels (bpw:404-406)."""
        from . import part_tables
        if self.COLOR_MODE == 'HSI':
            return part_tables.texture_image(self._tables, thickness=self._batch.thickness(0), color_mode='HSI')
        return part_tables.texture_image(self._tables, painted=self._batch.painted_bits(0))

    def get_job_status(self):
        """Number of painted samples (bpw:727-732)."""
        return int(self._batch.painted_bits(0).sum())

    def get_job_limit(self):
        return int(self._tables.sample_pos.shape[0])

    def render(self, mode='human'):                                            # rge:389-415
        if mode == 'human':
            raise Exception('please set render parameter to true to see the result')
        return self.get_texture_image()

    def close(self):
        if getattr(self, '_batch', None) is not None:
            self._batch.close()
            self._batch = None

    @classmethod
    def make_batched(cls, n_envs, device=None, auto_reset=False, seed=0, **env_config):
        """The BatchedPaintEnv of ``n_envs`` environments that PaintGymEnv(**env_config) would be ONE of: same part, class
        attributes (OBS_MODE, ACTION_MODE ...), EXTRA_CONFIG and Robot.PAINT_METHOD; the part's tables are built once.
        ``env_config`` is what paint_ppo.py:84-123 hands to ``env_creator`` (use with_robot=False)."""
        from .batched_env import BatchedPaintEnv
        one = cls(device=device, **env_config)
        try:
            kw = dict(one._batch.cfg_kwargs)
            kw.update(auto_reset=bool(auto_reset), seed=int(seed))
            return BatchedPaintEnv(one._batch.parts, int(n_envs), device=one._batch.device, **kw)
        finally:
            one.close()

    def seed(self, seed=None):
        return spaces.np_random(seed)[1]

    def __enter__(self):
        self.reset()
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.close()
