"""The BASELINE configurations at their per-GPU size (4 096 envs) over MANY steps with the in-kernel auto-reset ON --
what `bench.py` runs -- against the oracle on every row: observation (the terminal one through `final_obs`, the one
after the reset through `obs`), reward, done, info, and at the end every painted bit and the whole motion state.

The short full-size tests (tests/test_gpu_edge_cases.py, tests/test_gpu_configs.py: 4 steps from a fresh reset, no
auto-reset) never see an episode boundary, a mask more than a few per cent full or the off-part termination at this
size; these do: with random discrete-4 actions an episode lasts ~17 steps, so 60 steps are three to four
episode ends per env.  The start point of every new episode is passed explicitly (`start_idx`), so that the oracle can
be reset to the same one (rge:370-387).
"""
import numpy as np
import pytest

import oracle
from conftest import start_points_for, synthetic_tables

pytestmark = pytest.mark.gpu


def _dt(tables, sp, obs_grad=4):
    from paintrl_amd.device_tables import DeviceTables
    return DeviceTables(tables, obs_grad=obs_grad, start_points=sp)


class _Slices(object):
    """The oracles of a (possibly mixed) batch: oracle k holds the envs of `index[k]` (arrays of env numbers)."""

    def __init__(self, oracles, index, n):
        self.oracles, self.index, self.n = oracles, index, n
        self.obs_dim = oracles[0].obs_dim

    def reset(self, start, mask=None):
        out = np.zeros((self.n, self.obs_dim))
        for o, ix in zip(self.oracles, self.index):
            out[ix] = o.reset(start[ix], mask=None if mask is None else mask[ix])
        return out

    def step(self, a):
        obs, rew = np.zeros((self.n, self.obs_dim)), np.zeros(self.n)
        done, info = np.zeros(self.n, dtype=bool), np.zeros((self.n, 2))
        for o, ix in zip(self.oracles, self.index):
            obs[ix], rew[ix], done[ix], info[ix] = o.step(a[ix])
        return obs, rew, done, info

    def painted_bits(self, e):
        for o, ix in zip(self.oracles, self.index):
            k = np.nonzero(ix == e)[0]
            if k.size:
                return o.painted_bits(int(k[0]))
        raise IndexError(e)

    def state(self, e):
        for o, ix in zip(self.oracles, self.index):
            k = np.nonzero(ix == e)[0]
            if k.size:
                return o.state(int(k[0]))
        raise IndexError(e)


def _run(env, orc, n_start_of_env, steps, seed, min_episode_ends, reward_atol=0.0):
    """`steps` batched steps of `env` (auto_reset=True) and of the oracle side by side; returns the number of episodes
    that ended.  n_start_of_env: int array (N,), the size of each env's start-point table."""
    n = env.n_envs
    rng = np.random.RandomState(seed)
    start = (rng.randint(0, 1 << 30, size=n) % n_start_of_env).astype(np.int32)
    assert np.array_equal(env.reset(start_idx=start).cpu().numpy(), orc.reset(start)), 'reset observation'
    ends = 0
    fullest = 0.0
    for k in range(steps):
        a = rng.randint(0, 4, size=n)
        nxt = (rng.randint(0, 1 << 30, size=n) % n_start_of_env).astype(np.int32)
        o, r, d, i = env.step(a, start_idx=nxt)
        o, r, d, i = o.cpu().numpy().copy(), r.cpu().numpy().copy(), d.cpu().numpy().copy(), i.cpu().numpy().copy()
        f = env.final_obs.cpu().numpy()
        oo, rr, dd, ii = orc.step(a)
        assert np.array_equal(d, dd), 'done, step %d' % k
        if reward_atol:                                  # COLOR_MODE 'HSI': a float sum whose order the reference does not pin
            assert np.allclose(r, rr, rtol=0, atol=reward_atol) and np.allclose(i, ii, rtol=0, atol=reward_atol), 'reward / info, step %d' % k
        else:
            assert np.array_equal(r, rr) and np.array_equal(i, ii), 'reward / info, step %d' % k
        assert np.array_equal(o[~dd], oo[~dd]), 'observation, step %d' % k
        assert np.array_equal(f[dd], oo[dd]), 'terminal observation, step %d' % k
        if dd.any():
            o2 = orc.reset(nxt, mask=dd)
            assert np.array_equal(o[dd], o2[dd]), 'observation after the auto-reset, step %d' % k
            ends += int(dd.sum())
        if reward_atol:
            assert np.array_equal(env.thickness(), orc.thick), 'thickness bytes, step %d' % k
    assert ends >= min_episode_ends, ends
    # the state the next step would start from: every painted bit, the tool, the counters
    words = env.painted_words().cpu().numpy().view(np.uint64)
    st = env.state()
    for p, part in enumerate(env.parts):
        ix = np.nonzero(env.env_part_id == p)[0]
        bits = part.mask_to_canonical(words[ix])
        want = np.stack([orc.painted_bits(int(e)) for e in ix])
        assert np.array_equal(bits, want), 'painted bits of part %d' % p
        fullest = max(fullest, float(want.mean(axis=1).max()))
    for e in range(n):
        s = orc.state(e)
        assert np.array_equal(st['pose'][e], s['pose']) and np.array_equal(st['quat'][e], s['quat']), e
        if reward_atol:
            assert abs(st['total_return'][e] - s['total_return']) <= 100 * reward_atol and abs(st['total_reward'][e] - s['total_reward']) <= 100 * reward_atol, e
        else:
            assert st['total_return'][e] == s['total_return'] and st['total_reward'][e] == s['total_reward'], e
        assert st['step_counter'][e] == s['step_counter'] and st['terminate_counter'][e] == s['terminate_counter'], e
        assert st['last_on_part'][e] == s['last_on_part'], e
    return ends, fullest


def test_headline_config_60_steps_with_auto_reset_equals_oracle():
    """BASELINE config 2 as the bench runs it: door, section observation, 4 096 envs, anchor starts, random discrete-4
    actions, auto-reset on -- 60 steps, more than three episode ends per env."""
    from paintrl_amd.batched_env import BatchedPaintEnv
    tables = synthetic_tables('door_test')
    sp = start_points_for(tables, 'anchor')
    n = 4096
    env = BatchedPaintEnv(_dt(tables, sp), n, auto_reset=True)
    orc = oracle.Oracle(tables, n, start_points=sp, threads=16)
    ends, _ = _run(env, orc, np.full(n, len(sp)), 60, 2024, 3 * n)
    assert env.state()['episode'].mean() >= 3.5                # (the counter starts at 1 with the first reset)
    env.close()


def test_headline_config_all_starts_long_episodes_equals_oracle():
    """The same batch from the 'all' start table (episodes from the middle of the door run longer: masks fill up, section
    counts see painted words on both sides of the lines), 60 steps."""
    from paintrl_amd.batched_env import BatchedPaintEnv
    tables = synthetic_tables('door_test')
    sp = start_points_for(tables, 'all')
    n = 4096
    env = BatchedPaintEnv(_dt(tables, sp), n, auto_reset=True)
    orc = oracle.Oracle(tables, n, start_points=sp, threads=16)
    ends, fullest = _run(env, orc, np.full(n, len(sp)), 60, 4048, n // 2)
    assert fullest > 0.1                                 # some env carried a mask more than a tenth full at the end
    env.close()


def test_grid_overlap_turning_30_steps_with_auto_reset_equals_oracle():
    """BASELINE config 3: door, grid observation (16 cells) + OVERLAP_PENALTY (+ TURNING_PENALTY), 4 096 envs, 30 steps."""
    from paintrl_amd.batched_env import BatchedPaintEnv
    tables = synthetic_tables('door_test')
    sp = start_points_for(tables, 'anchor')
    n = 4096
    kw = dict(obs_mode='grid', obs_grad=4, overlap_penalty=True, turning_penalty=True)
    env = BatchedPaintEnv(_dt(tables, sp), n, auto_reset=True, **kw)
    orc = oracle.Oracle(tables, n, start_points=sp, threads=16, **kw)
    _run(env, orc, np.full(n, len(sp)), 30, 303, n)
    env.close()


def test_mixed_door_sheet_30_steps_with_auto_reset_equals_oracle():
    """BASELINE config 5's per-GPU share: env i on the door if i is even, on the sheet if odd, 'all' start tables,
    4 096 envs, 30 steps with auto-reset."""
    from paintrl_amd.batched_env import BatchedPaintEnv
    door, sheet = synthetic_tables('door_test'), synthetic_tables('square')
    sp_d, sp_s = start_points_for(door, 'all'), start_points_for(sheet, 'all')
    n = 4096
    ids = (np.arange(n) % 2).astype(np.int32)
    env = BatchedPaintEnv([_dt(door, sp_d), _dt(sheet, sp_s)], n, env_part_id=ids, max_possible_point=[9148, 14350],
                          auto_reset=True)
    od = oracle.Oracle(door, n // 2, start_points=sp_d, max_possible_point=9148, threads=16)
    os_ = oracle.Oracle(sheet, n // 2, start_points=sp_s, max_possible_point=14350, threads=16)
    orc = _Slices([od, os_], [np.arange(0, n, 2), np.arange(1, n, 2)], n)
    n_start = np.where(ids == 0, len(sp_d), len(sp_s))
    _run(env, orc, n_start, 30, 505, n // 8)
    env.close()


@pytest.mark.parametrize('tex,kw,steps', [
    (480, dict(obs_mode='section', overlap_penalty=True), 24),          # 38 224 samples (door_rf class), the last-shot row in use
    (652, dict(obs_mode='section'), 10),                                # 70 411 samples (door_rr_big, rge:116)
])
def test_large_part_full_size_with_auto_reset_equals_oracle(tex, kw, steps):
    """Seven of the reference's ten parts exceed 16 384 samples (Part_Dict rge:106-117) and take step_kernel_big: the mask rows
    stay in HBM, the painter works on the words of its cell block in place, the last-shot row is kept by its set of non-zero
    words.  4 096 envs with auto-reset from the 'all' start table, every row against the oracle, every painted bit at the end."""
    from paintrl_amd.batched_env import BatchedPaintEnv
    tables = synthetic_tables('door_rr_big', tex_size=(tex, tex))
    sp = start_points_for(tables, 'all')
    n = 4096
    mpp = int(0.95 * tables.sample_pos.shape[0])
    env = BatchedPaintEnv(_dt(tables, sp), n, auto_reset=True, max_possible_point=mpp, **kw)
    assert env.mask_stride > 256
    orc = oracle.Oracle(tables, n, start_points=sp, threads=16, max_possible_point=mpp, **kw)
    _run(env, orc, np.full(n, len(sp)), steps, 909, n // 16)
    env.close()


@pytest.mark.parametrize('tex,n,steps', [(0, 4096, 24), (480, 1024, 10)])
def test_thickness_mode_full_size_equals_oracle(tex, n, steps):
    """COLOR_MODE 'HSI' (bpw:384-434): thickness bytes, float rewards.  The painter finds the five shots' largest distances first
    and then visits each word of their cell block once (paint_shots_hsi_words) -- on the door's register masks at the bench's
    launch shape, and on a 38 224-sample part's rows in HBM; every byte after every step, rewards to 1e-12, observations,
    terminal rows and the painted (status) bits exactly, with auto-reset from the 'all' start table."""
    from paintrl_amd.batched_env import BatchedPaintEnv
    tables = synthetic_tables('door_rr_big', tex_size=(tex, tex)) if tex else synthetic_tables('door_test')
    sp = start_points_for(tables, 'all')
    mpp = int(0.95 * tables.sample_pos.shape[0])
    env = BatchedPaintEnv(_dt(tables, sp), n, auto_reset=True, max_possible_point=mpp, color_mode='HSI')
    orc = oracle.Oracle(tables, n, start_points=sp, threads=16, max_possible_point=mpp, color_mode='HSI')
    _run(env, orc, np.full(n, len(sp)), steps, 1717, 0, reward_atol=1e-12)
    env.close()


@pytest.mark.parametrize('part,tex,grad,n,steps', [('door_test', 0, 6, 2048, 16), ('door_rr_big', 480, 6, 512, 8)])
def test_atan2_sectors_with_auto_reset_equal_oracle(part, tex, grad, n, steps):
    """OBS_GRAD != 4 (bpw:1045-1061) at a launch-sized batch: words inside one wedge counted by popcount, float atan2f
    where it is safe, the reference's float64 expression at the sectors' ends (prl_observe.hpp section_general_wave) -- every
    row against the oracle with the in-kernel auto-reset on, the door's register masks and a 38 224-sample part's rows in HBM."""
    from paintrl_amd.batched_env import BatchedPaintEnv
    tables = synthetic_tables(part, tex_size=(tex, tex)) if tex else synthetic_tables(part)
    sp = start_points_for(tables, 'all')
    mpp = int(0.95 * tables.sample_pos.shape[0])
    kw = dict(obs_mode='section', obs_grad=grad, max_possible_point=mpp)
    from paintrl_amd.device_tables import DeviceTables
    env = BatchedPaintEnv(DeviceTables(tables, obs_grad=grad, start_points=sp), n, auto_reset=True, **kw)
    orc = oracle.Oracle(tables, n, start_points=sp, threads=16, **kw)
    _run(env, orc, np.full(n, len(sp)), steps, 2626, 0)
    env.close()


def test_cone_beams_full_size_8_steps_equals_oracle():
    """PAINT_METHOD 'normal' (rob:251-285, bpw:562-566) at the launch shape of the benchmark: 4 096 envs, 8 steps with
    auto-reset, anchor starts (off-part shots, the window and the rim of the door all occur)."""
    from paintrl_amd.batched_env import BatchedPaintEnv
    tables = synthetic_tables('door_test')
    sp = start_points_for(tables, 'anchor')
    n = 4096
    env = BatchedPaintEnv(_dt(tables, sp), n, auto_reset=True, paint_method='normal')
    orc = oracle.Oracle(tables, n, start_points=sp, threads=16, paint_method='normal')
    _run(env, orc, np.full(n, len(sp)), 8, 808, 1)
    env.close()


SOAK_SLICE_STEPS = 300          # the slice of the soak that every `-m gpu` run takes (headline configuration only)


@pytest.mark.parametrize('part,starts,seed', [('door_test', 'anchor', 11), ('door_test', 'all', 12), ('square', 'all', 13),
                                               ('test', 'all', 14)])         # ('test': the coarse sheet with the reference's stale kd-tree)
def test_soak_many_steps_equals_oracle(part, starts, seed):
    """The headline batch over many steps -- dozens of episode ends per env, every row against the oracle.  The headline
    configuration (door, anchor starts: what bench.py runs) takes SOAK_SLICE_STEPS = 300 steps x 4 096 envs in every `-m gpu`
    run (~20 s, most of it the oracle on the host cores); PAINTRL_SOAK_STEPS sets the length and adds the other two
    configurations (`PAINTRL_SOAK_STEPS=2000 pytest tests/test_gpu_full_size.py -k soak`; the last long run is recorded in
    profiles/)."""
    import os
    from paintrl_amd.batched_env import BatchedPaintEnv
    asked = os.environ.get('PAINTRL_SOAK_STEPS')
    if not asked and (part, starts) != ('door_test', 'anchor'):
        pytest.skip('soak run of the other configurations: set PAINTRL_SOAK_STEPS (e.g. 400)')
    steps = int(asked) if asked else SOAK_SLICE_STEPS
    tables = synthetic_tables(part)
    sp = start_points_for(tables, starts)
    n = 4096
    kw = dict(max_possible_point=14000) if part == 'test' else {}
    env = BatchedPaintEnv(_dt(tables, sp), n, auto_reset=True, **kw)
    orc = oracle.Oracle(tables, n, start_points=sp, threads=16, **kw)
    ends, fullest = _run(env, orc, np.full(n, len(sp)), steps, seed, n)
    print('soak %s / %s: %d steps, %d episode ends (%.1f per env), fullest mask %.2f' % (part, starts, steps, ends, ends / n, fullest))
    if not asked:
        assert ends >= 12 * n                            # (an episode of the random walk lasts ~17 steps)
    env.close()


@pytest.mark.skipif(not __import__('os').environ.get('PAINTRL_SOAK_STEPS'), reason='soak run: set PAINTRL_SOAK_STEPS (e.g. 400)')
@pytest.mark.parametrize('case', ['grid_overlap_turning', 'mixed', 'cone_beams', 'large_part', 'thickness'])
def test_soak_other_configurations_equal_oracle(case):
    """The soak run for BASELINE configs 3 (grid + penalties), 5 (mixed door / sheet) and the cone-beam painter (a twentieth
    of the steps: its oracle is the slow one)."""
    import os
    from paintrl_amd.batched_env import BatchedPaintEnv
    steps = int(os.environ['PAINTRL_SOAK_STEPS'])
    n = 4096
    door = synthetic_tables('door_test')
    if case == 'large_part':                       # 38 224 samples (mask rows in HBM), OVERLAP_PENALTY on, a fifth of the steps
        big = synthetic_tables('door_rr_big', tex_size=(480, 480))
        sp = start_points_for(big, 'all')
        mpp = int(0.95 * big.sample_pos.shape[0])
        steps = max(20, steps // 5)
        env = BatchedPaintEnv(_dt(big, sp), n, auto_reset=True, overlap_penalty=True, max_possible_point=mpp)
        orc = oracle.Oracle(big, n, start_points=sp, threads=16, overlap_penalty=True, max_possible_point=mpp)
        n_start = np.full(n, len(sp))
    elif case == 'mixed':
        sheet = synthetic_tables('square')
        sp_d, sp_s = start_points_for(door, 'all'), start_points_for(sheet, 'all')
        ids = (np.arange(n) % 2).astype(np.int32)
        env = BatchedPaintEnv([_dt(door, sp_d), _dt(sheet, sp_s)], n, env_part_id=ids, max_possible_point=[9148, 14350],
                              auto_reset=True)
        orc = _Slices([oracle.Oracle(door, n // 2, start_points=sp_d, max_possible_point=9148, threads=16),
                       oracle.Oracle(sheet, n // 2, start_points=sp_s, max_possible_point=14350, threads=16)],
                      [np.arange(0, n, 2), np.arange(1, n, 2)], n)
        n_start = np.where(ids == 0, len(sp_d), len(sp_s))
    else:
        sp = start_points_for(door, 'anchor')
        kw = dict(obs_mode='grid', obs_grad=4, overlap_penalty=True, turning_penalty=True) if case == 'grid_overlap_turning' \
            else (dict(color_mode='HSI') if case == 'thickness' else dict(paint_method='normal'))
        if case == 'cone_beams':
            steps = max(8, steps // 20)
        if case == 'thickness':                   # (COLOR_MODE 'HSI': every byte after every step; a quarter of the steps)
            steps = max(20, steps // 4)
        env = BatchedPaintEnv(_dt(door, sp), n, auto_reset=True, **kw)
        orc = oracle.Oracle(door, n, start_points=sp, threads=16, **kw)
        n_start = np.full(n, len(sp))
    ends, fullest = _run(env, orc, n_start, steps, 77, 1, reward_atol=1e-12 if case == 'thickness' else 0.0)
    print('soak %s: %d steps, %d episode ends (%.1f per env), fullest mask %.2f' % (case, steps, ends, ends / n, fullest))
    env.close()
