"""GPU: the Gym-facing PaintGymEnv (N=1 view) behaves like the reference class."""
import os
import random

import numpy as np
import pytest

from conftest import load_episodes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def urdf_root(tmp_path_factory):
    from paintrl_amd import synth_parts
    root = str(tmp_path_factory.mktemp('synth_root'))
    synth_parts.write_synthetic_parts(root)
    return root


def test_gym_env_rollout_replays_golden_sheet_zigzag(urdf_root):
    """zigzag.py-style run: Part_NO=1, 'fixed' start, rollout=True, OBS_MODE='simple'."""
    from paintrl_amd import PaintGymEnv
    ep = load_episodes('sheet')['g2_zigzag']
    PaintGymEnv.change_action_mode(1, 'discrete', 4)
    PaintGymEnv.change_obs_mode('simple', 4)
    cfg = dict(PaintGymEnv.EXTRA_CONFIG, Part_NO=1, START_POINT_MODE='fixed')
    with PaintGymEnv(urdf_root, with_robot=False, renders=False, rollout=True, extra_config=cfg) as env:
        assert env.action_space.n == 4 and env.observation_space.shape == (2,)
        obs = env.reset()
        assert np.array_equal(obs, ep['obs0'])
        total = 0.0
        for k, a in enumerate(ep['actions']):
            obs, r, done, info = env.step(int(a))
            assert np.array_equal(obs, ep['obs'][k]) and r == ep['reward'][k] and done == bool(ep['done'][k])
            assert info == {'reward': ep['info'][k, 0], 'penalty': ep['info'][k, 1]}
            total += r
        assert done and env.get_job_status() == int(np.unpackbits(ep['snaps'][-1], bitorder='little').sum())
        img = env.render(mode='rgb_array')
        assert img.shape == (240, 240, 3) and (img[..., 0] == 255).sum() == env.get_job_status()
    PaintGymEnv.change_obs_mode('section', 4)


def test_gym_env_training_reset_follows_python_random(urdf_root):
    """Non-rollout reset draws randint(0,7) then the start index from the `random` module (rge:377-381)."""
    from paintrl_amd import PaintGymEnv
    ep = load_episodes('door')['g3_random_1']
    PaintGymEnv.change_action_mode(1, 'discrete', 4)
    PaintGymEnv.change_obs_mode('section', 4)
    env = PaintGymEnv(urdf_root, with_robot=False, extra_config=dict(PaintGymEnv.EXTRA_CONFIG, Part_NO=0))
    random.seed(101)                       # the seed make_golden.py used for this episode
    obs = env.reset()
    assert np.array_equal(obs, ep['obs0'])
    for k, a in enumerate(ep['actions']):
        obs, r, done, info = env.step(int(a))
        assert np.array_equal(obs, ep['obs'][k]) and r == ep['reward'][k] and done == bool(ep['done'][k])
    assert len(env._start_points) == 4 and env.get_job_limit() == 9664
    env.close()


def test_missing_library_fails_loudly(monkeypatch):
    from paintrl_amd import _lib, build
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(build, 'LIBRARY', os.path.join(os.path.dirname(build.LIBRARY), 'does_not_exist.so'))
    with pytest.raises(_lib.PaintRLError):
        _lib.load()
