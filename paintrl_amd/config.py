"""PaintGymEnv configuration (PaintRLEnv/robot_gym_env.py:127-157) -> PrlConfig POD."""
import math

import numpy as np

from . import _lib

OBS_MODES = {'section': 0, 'grid': 1, 'simple': 2, 'discrete': 3}
ACTION_MODES = {'discrete': 0, 'continuous': 1}
TERMINATION_MODES = {'late': 0, 'early': 1, 'hybrid': 2}
PAINT_METHODS = {'fast': 0, 'normal': 1}
COLOR_MODES = {'RGB': 0, 'HSI': 1}


class PaintToolProfile(object):
    """The profile of the paint gun (bpw:40-43); read when an env is constructed, like the reference's class."""
    PAINT_RADIUS = 0.051
    STEP_SIZE = PAINT_RADIUS


# rge:106-117 Part_Dict: part number -> [urdf file, max possible points]
Part_Dict = {
    0: ['door_test.urdf', 9148], 1: ['square.urdf', 14350], 2: ['door_lf.urdf', 0], 3: ['door_lr.urdf', 0],
    4: ['door_rf.urdf', 0], 5: ['door_rr.urdf', 17000], 6: ['roof.urdf', 0], 7: ['bonnet.urdf', 0],
    8: ['door_rr_big.urdf', 0], 9: ['test.urdf', 9148],
}

# rge:134-157 EXTRA_CONFIG defaults
EXTRA_CONFIG = {
    'RENDER_HEIGHT': 720, 'RENDER_WIDTH': 960, 'Part_NO': 0, 'Expected_Episode_Length': 245,
    'EPISODE_MAX_LENGTH': 245, 'TERMINATION_MODE': 'late', 'SWITCH_THRESHOLD': 0.9,
    'START_POINT_MODE': 'anchor', 'TURNING_PENALTY': False, 'OVERLAP_PENALTY': False, 'COLOR_MODE': 'RGB',
}


def discrete_action_table(n, step_size=0.051):
    """Discrete action a -> (delta_axis1, delta_axis2, turning angle).

    rge:342-347 (_preprocess_action), rob:151-153 (direction_normalize, 1-D), rob:396-397,
    rob:352-356 (_set_turning_angle) evaluated with the same numpy / math calls the
    reference makes, so e.g. cos(pi/2) stays 6.1e-17 instead of 0.
    """
    d1, d2, ang = [], [], []
    for a in range(n):
        v = 2 * (a - n / 2) / n
        phi = (v + 1) * np.pi
        x, y = 1 * np.cos(phi), 1 * np.sin(phi)
        delta1, delta2 = x * step_size, y * step_size
        d1.append(float(delta1))
        d2.append(float(delta2))
        ang.append(math.atan(abs(delta2 / delta1)) if delta1 != 0 else math.pi / 2)
    return d1, d2, ang


def make_config(obs_mode='section', obs_grad=4, action_mode='discrete', action_dim=1, n_discrete=4,
                termination_mode='late', turning_penalty=False, overlap_penalty=False, paint_method='fast',
                max_episode_len=245, expected_episode_len=245, switch_threshold=0.9, max_possible_point=9148,
                auto_reset=False, seed=0, paint_radius=0.051, step_size=0.051, color_mode='RGB'):
    """Build the PrlConfig POD.  ``max_possible_point`` is a number or one number per part id."""
    c = _lib.PrlConfig()
    c.obs_mode, c.obs_grad = OBS_MODES[obs_mode], int(obs_grad)
    c.action_mode, c.action_dim, c.n_discrete = ACTION_MODES[action_mode], int(action_dim), int(n_discrete)
    c.termination_mode = TERMINATION_MODES[termination_mode]
    c.turning_penalty, c.overlap_penalty = int(bool(turning_penalty)), int(bool(overlap_penalty))
    c.paint_method = PAINT_METHODS[paint_method]
    c.color_mode = COLOR_MODES[color_mode]
    c.max_episode_len, c.expected_episode_len = int(max_episode_len), int(expected_episode_len)
    c.auto_reset = int(bool(auto_reset))
    c.switch_threshold = float(switch_threshold)
    c.paint_radius, c.step_size = float(paint_radius), float(step_size)
    pts = list(np.atleast_1d(max_possible_point).astype(float))
    for k in range(8):
        c.max_possible_point[k] = pts[min(k, len(pts) - 1)]
    c.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    if c.action_mode == 0:
        if not 1 <= n_discrete <= _lib.MAX_DISCRETE:
            raise ValueError('DISCRETE_GRANULARITY must be 1..%d' % _lib.MAX_DISCRETE)
        d1, d2, ang = discrete_action_table(n_discrete, step_size)
        for k in range(n_discrete):
            c.act_delta1[k], c.act_delta2[k], c.act_angle[k] = d1[k], d2[k], ang[k]
    return c


def obs_dim(obs_mode, obs_grad):
    """rge:166-173."""
    if obs_mode == 'section':
        return obs_grad + 2
    if obs_mode == 'grid':
        return obs_grad ** 2
    if obs_mode == 'simple':
        return 2
    return obs_grad + 1
