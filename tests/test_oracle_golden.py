"""Pins the CPU oracle to the golden vectors recorded from the reference itself
(tests/golden/make_golden.py): every episode must replay exactly."""
import numpy as np
import pytest

import oracle
from conftest import env_kwargs_from_cfg, load_episodes, start_points_for, synthetic_tables

import os

from conftest import GOLDEN

TAGS = [t for t in ('door', 'sheet', 'sheet_tool', 'door_big', 'door_hsi', 'sparse', 'door_term', 'sheet_term', 'door_hsi_cone', 'seam',
                      'door_every_step', 'sheet_every_step')
        if os.path.isfile(os.path.join(GOLDEN, 'episodes_%s.npz' % t))]
CASES = [(tag, n) for tag in TAGS for n in sorted(load_episodes(tag))]
PART = {'door': 'door_test', 'sheet': 'square', 'sheet_tool': 'square', 'door_big': 'door_rr_big', 'door_hsi': 'door_test',
        'sparse': 'test', 'door_term': 'door_test', 'sheet_term': 'square', 'door_hsi_cone': 'door_test', 'seam': 'door_lf',
        'door_every_step': 'door_test', 'sheet_every_step': 'square'}


def replay(backend_step, backend_reset, ep, exact=True, atol=0.0):
    obs0 = backend_reset(int(ep['start_idx']))
    if exact:
        assert np.array_equal(obs0, ep['obs0'])
    else:
        np.testing.assert_allclose(obs0, ep['obs0'], rtol=0, atol=atol)
    snaps = dict(zip(ep['snap_steps'].tolist(), ep['snaps']))
    for k, a in enumerate(ep['actions']):
        obs, rew, done, info, bits = backend_step(a, (k + 1) in snaps)
        if exact:
            assert np.array_equal(obs, ep['obs'][k]), 'obs differs at step %d' % k
            assert rew == ep['reward'][k] and info[0] == ep['info'][k, 0] and info[1] == ep['info'][k, 1], k
        else:
            np.testing.assert_allclose(obs, ep['obs'][k], rtol=0, atol=atol)
            np.testing.assert_allclose([rew, info[0], info[1]],
                                       [ep['reward'][k], ep['info'][k, 0], ep['info'][k, 1]], rtol=0, atol=atol)
        assert bool(done) == bool(ep['done'][k]), 'done differs at step %d' % k
        if (k + 1) in snaps:
            want = np.unpackbits(snaps[k + 1], bitorder='little')[:bits.size].astype(bool)
            assert np.array_equal(bits, want), 'painted texel set differs at step %d' % (k + 1)


@pytest.mark.parametrize('tag,name', CASES)
def test_oracle_replays_reference_episode(tag, name):
    ep = load_episodes(tag)[name]
    cfg = ep['cfg']
    tables = synthetic_tables(PART[tag], cfg.get('paint_radius', 0.051))
    hsi = cfg.get('color_mode', 'RGB') == 'HSI'
    orc = oracle.Oracle(tables, 1, start_points=start_points_for(tables, cfg['start_mode']),
                        color_mode=cfg.get('color_mode', 'RGB'), beams=ep.get('beams'), **env_kwargs_from_cfg(cfg))
    continuous = cfg['action_mode'] == 'continuous'

    def reset(idx):
        return orc.reset([idx])[0]

    def step(a, want_bits):
        obs, rew, done, info = orc.step([a])
        return obs[0], rew[0], done[0], info[0], orc.painted_bits(0)

    # continuous actions go through libm sin/cos/atan2 on both sides: tolerance, not bit-exact; the float deposit
    # sums of COLOR_MODE='HSI' are added in cKDTree traversal order by the reference: 1e-12
    replay(step, reset, ep, exact=not (continuous or hsi), atol=1e-12 if hsi else 1e-9)
    st = orc.state(0)
    if hsi:
        assert np.array_equal(orc.thick[0], ep['final_thick'])           # the bytes themselves are exact
        assert np.array_equal(st['pose'], ep['final_pose']) and np.array_equal(st['quat'], ep['final_quat'])
    elif not continuous:
        assert np.array_equal(st['pose'], ep['final_pose']) and np.array_equal(st['quat'], ep['final_quat'])
        assert st['total_return'] == float(ep['total_return'])


OFFPART = sorted(load_episodes('sheet_offpart')) if os.path.isfile(os.path.join(GOLDEN, 'episodes_sheet_offpart.npz')) else []


@pytest.mark.parametrize('name', OFFPART)
def test_oracle_replays_the_off_part_counter_limit(name):
    """rob:292-300: the episode ends because more than NOT_ON_PART_TERMINATE_STEPS = 1000 misses were counted -- reached by
    hovering 4 / 6 cm beside the sheet (Robot.reset([pose, orn]) after the env's reset, as spiral.py does), where every
    shot misses the part and still paints: recorded from the reference (make_golden.py --off-part), replayed exactly."""
    ep = load_episodes('sheet_offpart')[name]
    cfg = ep['cfg']
    tables = synthetic_tables('square')
    orc = oracle.Oracle(tables, 1, start_points=start_points_for(tables, cfg['start_mode']), **env_kwargs_from_cfg(cfg))

    def reset(idx):
        obs = orc.reset([idx])[0]
        orc.set_pose(0, ep['set_pose'], ep['set_orn'])
        return obs

    def step(a, want_bits):
        obs, rew, done, info = orc.step([a])
        return obs[0], rew[0], done[0], info[0], orc.painted_bits(0)

    replay(step, reset, ep, exact=True)
    st = orc.state(0)
    assert st['terminate_counter'] == int(ep['terminate_counter']) > 1000 and st['terminate'] == 1
    assert np.array_equal(st['pose'], ep['final_pose']) and np.array_equal(st['quat'], ep['final_quat'])
    assert st['total_return'] == float(ep['total_return'])
