"""COLOR_MODE = 'HSI' (bpw:384-434): thickness bytes per sample, as the reference behaves.

Integer state (bytes, status bits, done flags, poses) is exact; rewards are float sums of quantity / 255 whose
order the reference takes from cKDTree internals, so they are compared to 1e-12 (observations are ratios of
integer counts and poses: exact)."""
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, env_kwargs_from_cfg, load_episodes, start_points_for, synthetic_tables

pytestmark = pytest.mark.gpu
TOL = 1e-12


def _env(tables, n, sp=None, **kw):
    from paintrl_amd.batched_env import BatchedPaintEnv
    from paintrl_amd.device_tables import DeviceTables
    return BatchedPaintEnv(DeviceTables(tables, obs_grad=kw.get('obs_grad', 4), start_points=sp), n, **kw)


@pytest.mark.parametrize('part,kw', [
    ('door_test', dict(obs_mode='section')),
    ('door_test', dict(obs_mode='grid', overlap_penalty=True, turning_penalty=True)),
    ('square', dict(obs_mode='section', obs_grad=6, max_possible_point=14350)),
])
def test_hsi_matches_oracle_on_random_batch(part, kw):
    tables = synthetic_tables(part)
    sp = start_points_for(tables, 'all')
    n, steps = 128, 40
    env = _env(tables, n, sp, color_mode='HSI', **kw)
    orc = oracle.Oracle(tables, n, start_points=sp, threads=8, color_mode='HSI', **kw)
    rng = np.random.RandomState(31)
    start = rng.randint(0, len(sp), size=n)
    assert np.array_equal(env.reset(start_idx=start).cpu().numpy(), orc.reset(start))
    assert (env.thickness() == 255).all() and env.painted_bits(0).all()          # every texel reads "painted" at first
    deposits = 0
    for k in range(steps):
        a = rng.randint(0, 4, size=n)
        o, r, d, i = env.step(a)
        oo, rr, dd, ii = orc.step(a)
        assert np.array_equal(o.cpu().numpy(), oo), 'obs, step %d' % k
        assert np.allclose(r.cpu().numpy(), rr, rtol=0, atol=TOL) and np.allclose(i.cpu().numpy(), ii, rtol=0, atol=TOL)
        assert np.array_equal(d.cpu().numpy(), dd), 'done, step %d' % k
        assert np.array_equal(env.thickness(), orc.thick), 'thickness bytes, step %d' % k
        deposits += int((ii[:, 0] > 0).sum())
        if dd.any():
            new = rng.randint(0, len(sp), size=n)
            o2 = env.reset(mask=dd, start_idx=new).cpu().numpy()
            assert np.array_equal(o2[dd], orc.reset(new, mask=dd)[dd])
    assert deposits > n * 5
    words = env.painted_words().cpu().numpy().view(np.uint64)
    bits = env.parts[0].mask_to_canonical(words)
    assert np.array_equal(bits, np.stack([orc.painted_bits(e) for e in range(n)]))
    assert np.array_equal(bits, env.thickness() == 255)
    env.close()


def test_hsi_bytes_wrap_and_auto_reset():
    """Repainting one spot drives bytes through zero (numpy uint8 wrap, bpw:393) and a byte that lands on 0 stays;
    in-kernel auto-reset restores 255 everywhere."""
    tables = synthetic_tables('square')
    sp = start_points_for(tables, 'anchor')
    n = 8
    env = _env(tables, n, sp, color_mode='HSI', auto_reset=True, max_episode_len=60, max_possible_point=14350)
    orc = oracle.Oracle(tables, n, start_points=sp, color_mode='HSI', max_episode_len=60, max_possible_point=14350)
    start = np.zeros(n, dtype=np.int64)
    env.reset(start_idx=start)
    orc.reset(start)
    acts = np.array(([0] * 1 + [2] * 1) * 29)                       # back and forth over the same samples
    for k, a in enumerate(acts):
        o, r, d, i = env.step(np.full(n, a), start_idx=start)
        oo, rr, dd, ii = orc.step(np.full(n, a))
        assert np.array_equal(env.thickness(), orc.thick), 'step %d' % k
        assert not dd.any()
    th = orc.thick[0]
    assert (th < 60).any() and ((th > 200) & (th < 255)).any()        # low bytes and (after wrapping) high ones
    for a in (0, 2):                                                  # steps 59, 60: the episode ends at max length
        o, r, d, i = env.step(np.full(n, a), start_idx=start)
    assert bool(d.all())
    assert (env.thickness() == 255).all() and env.painted_bits(0).all()
    env.close()


def test_hsi_limits_are_reported():
    from paintrl_amd import _lib
    tables = synthetic_tables('door_test')
    env = _env(tables, 4)
    with pytest.raises(_lib.PaintRLError, match='HSI'):
        env.thickness()
    env.close()


@pytest.mark.skipif(not os.path.isfile(os.path.join(GOLDEN, 'episodes_door_hsi.npz')), reason='fixture not generated')
@pytest.mark.parametrize('name', ['g13_hsi_serpentine', 'g13_hsi_random', 'g13_hsi_grid_overlap'])
def test_hsi_replays_reference_episode(name):
    """Episodes recorded from the reference itself with COLOR_MODE='HSI' on the synthetic door."""
    from test_oracle_golden import replay
    ep = load_episodes('door_hsi')[name]
    cfg = ep['cfg']
    tables = synthetic_tables('door_test')
    env = _env(tables, 1, start_points_for(tables, cfg['start_mode']), color_mode='HSI', **env_kwargs_from_cfg(cfg))

    def reset(idx):
        return env.reset(start_idx=[idx]).cpu().numpy()[0]

    def step(a, want_bits):
        obs, rew, done, info = env.step([a])
        bits = env.painted_bits(0) if want_bits else None
        return obs.cpu().numpy()[0], float(rew[0]), bool(done[0]), info.cpu().numpy()[0], bits

    replay(step, reset, ep, exact=False, atol=TOL)
    assert np.array_equal(env.thickness(0), ep['final_thick'])
    env.close()


def test_hsi_under_cone_beams_matches_oracle():
    """COLOR_MODE='HSI' with PAINT_METHOD='normal' (rob:38-69 beta-profile beam table + bpw:419-434 on the
    list of nearest samples): a sample under k beams receives k deposits per shot.  Bytes, status bits, observations,
    done flags and poses exact; rewards to 1e-12 (the reference sums in beam order, the device in lane order)."""
    import random
    from paintrl_amd import part_tables
    tables = synthetic_tables('door_test')
    sp = start_points_for(tables, 'all')
    random.seed(7)
    beams = part_tables.beta_plain(tables.density)
    assert beams.shape == (450, 3)
    n, steps = 48, 8
    from paintrl_amd.batched_env import BatchedPaintEnv
    from paintrl_amd.device_tables import DeviceTables
    kw = dict(color_mode='HSI', paint_method='normal', obs_mode='grid', overlap_penalty=True)
    env = BatchedPaintEnv(DeviceTables(tables, start_points=sp, beams=beams), n, **kw)
    orc = oracle.Oracle(tables, n, start_points=sp, threads=8, beams=beams, **kw)
    rng = np.random.RandomState(5)
    start = rng.randint(0, len(sp), size=n)
    assert np.array_equal(env.reset(start_idx=start).cpu().numpy(), orc.reset(start))
    multi = 0
    for k in range(steps):
        a = rng.randint(0, 4, size=n)
        o, r, d, i = env.step(a)
        oo, rr, dd, ii = orc.step(a)
        assert np.array_equal(o.cpu().numpy(), oo), 'obs, step %d' % k
        assert np.allclose(r.cpu().numpy(), rr, rtol=0, atol=TOL) and np.allclose(i.cpu().numpy(), ii, rtol=0, atol=TOL)
        assert np.array_equal(d.cpu().numpy(), dd), 'done, step %d' % k
        assert np.array_equal(env.thickness(), orc.thick), 'thickness bytes, step %d' % k
        multi += int((orc.thick < 255 - 26).sum())                    # more than one deposit's worth gone from a byte
        if dd.any():
            new = rng.randint(0, len(sp), size=n)
            assert np.array_equal(env.reset(mask=dd, start_idx=new).cpu().numpy()[dd], orc.reset(new, mask=dd)[dd])
    assert multi > 100
    env.close()

