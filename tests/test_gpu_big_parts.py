"""Parts with more than 16 384 coverage samples (the reference's door_lf ... door_rr_big, Part_Dict rge:106-117):
their masks live in LDS (step_kernel_big / reset_kernel_big / observe_kernel_big).  Same parity bar as the small
parts: every observation, reward, done flag, painted bit and pose equals the CPU oracle, and the episodes recorded
from the reference on the synthetic 480 x 480 door replay bit for bit."""
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, env_kwargs_from_cfg, load_episodes, start_points_for, synthetic_tables

pytestmark = pytest.mark.gpu


def _env(tables, n, sp=None, **kw):
    from paintrl_amd.batched_env import BatchedPaintEnv
    from paintrl_amd.device_tables import DeviceTables
    return BatchedPaintEnv(DeviceTables(tables, obs_grad=kw.get('obs_grad', 4), start_points=sp), n, **kw)


@pytest.mark.parametrize('tex,kw', [
    (320, dict(obs_mode='section')),                                     # 17 098 samples: 276 words, 5 per lane
    (320, dict(obs_mode='grid', overlap_penalty=True, turning_penalty=True)),
    (320, dict(obs_mode='discrete', termination_mode='hybrid', n_discrete=8)),
    (320, dict(obs_mode='section', obs_grad=6)),                         # atan2 sectors on LDS masks
    (480, dict(obs_mode='section')),                                     # 38 224 samples: 605 words
    (480, dict(obs_mode='grid', obs_grad=5, overlap_penalty=True)),
])
def test_big_part_matches_oracle_on_random_batch(tex, kw):
    tables = synthetic_tables('door_rr_big', tex_size=(tex, tex))
    assert tables.sample_pos.shape[0] > 16384
    sp = start_points_for(tables, 'all')
    n, steps = 96, 30
    mpp = int(0.95 * tables.sample_pos.shape[0])
    env = _env(tables, n, sp, max_possible_point=mpp, **kw)
    assert env.mask_stride > 256
    orc = oracle.Oracle(tables, n, start_points=sp, threads=8, max_possible_point=mpp, **kw)
    rng = np.random.RandomState(tex)
    start = rng.randint(0, len(sp), size=n)
    assert np.array_equal(env.reset(start_idx=start).cpu().numpy(), orc.reset(start))
    nd = kw.get('n_discrete', 4)
    for k in range(steps):
        a = rng.randint(0, nd, size=n)
        o, r, d, i = env.step(a)
        oo, rr, dd, ii = orc.step(a)
        assert np.array_equal(o.cpu().numpy(), oo), 'obs, step %d' % k
        assert np.array_equal(r.cpu().numpy(), rr) and np.array_equal(i.cpu().numpy(), ii), 'reward, step %d' % k
        assert np.array_equal(d.cpu().numpy(), dd), 'done, step %d' % k
        if dd.any():
            new = rng.randint(0, len(sp), size=n)
            o2 = env.reset(mask=dd, start_idx=new).cpu().numpy()
            assert np.array_equal(o2[dd], orc.reset(new, mask=dd)[dd])
    words = env.painted_words().cpu().numpy().view(np.uint64)
    bits = env.parts[0].mask_to_canonical(words)
    assert np.array_equal(bits, np.stack([orc.painted_bits(e) for e in range(n)]))
    assert np.array_equal(env.observe().cpu().numpy(), orc.observe())
    st = env.state()
    for e in range(n):
        so = orc.state(e)
        assert np.array_equal(st['pose'][e], so['pose']) and np.array_equal(st['quat'][e], so['quat'])
        assert st['total_return'][e] == so['total_return'] and st['step_counter'][e] == so['step_counter']
    env.close()


def test_big_part_auto_reset_and_episode_statistics():
    """In-kernel auto-reset on LDS-resident masks == done -> manual reset; the episode statistics count the bits."""
    tables = synthetic_tables('door_rr_big', tex_size=(320, 320))
    sp = start_points_for(tables, 'all')
    n, steps = 64, 50
    env_a, env_m = _env(tables, n, sp, auto_reset=True), _env(tables, n, sp)
    rng = np.random.RandomState(5)
    start = rng.randint(0, len(sp), size=n)
    env_a.reset(start_idx=start)
    env_m.reset(start_idx=start)
    finished = 0
    for k in range(steps):
        a = rng.randint(0, 4, size=n)
        nxt = rng.randint(0, len(sp), size=n)
        oa, ra, da, _ = env_a.step(a, start_idx=nxt)
        oa, da, fa = oa.cpu().numpy().copy(), da.cpu().numpy().copy(), env_a.final_obs.cpu().numpy().copy()
        om, rm, dm, _ = env_m.step(a)
        om, dm = om.cpu().numpy().copy(), dm.cpu().numpy().copy()
        assert np.array_equal(da, dm) and np.array_equal(ra.cpu().numpy(), rm.cpu().numpy())
        assert np.array_equal(oa[~dm], om[~dm]) and np.array_equal(fa[dm], om[dm])
        if dm.any():
            words = env_m.painted_words().cpu().numpy().view(np.uint64)
            cov = np.unpackbits(words.view(np.uint8), axis=1).sum(1)
            assert np.array_equal(env_a.state()['last_episode_painted'][dm], cov[dm])
            o2 = env_m.reset(mask=dm, start_idx=nxt).cpu().numpy()
            assert np.array_equal(oa[dm], o2[dm])
            finished += int(dm.sum())
    assert finished > 10
    assert np.array_equal(env_a.painted_words().cpu().numpy(), env_m.painted_words().cpu().numpy())
    env_a.close()
    env_m.close()


def test_mixed_batch_of_small_and_big_parts():
    """A batch is launched for its widest part: the 240 x 240 door and the 320 x 320 one interleaved."""
    from paintrl_amd.batched_env import BatchedPaintEnv
    from paintrl_amd.device_tables import DeviceTables
    small, big = synthetic_tables('door_test'), synthetic_tables('door_rr_big', tex_size=(320, 320))
    sp_s, sp_b = start_points_for(small, 'anchor'), start_points_for(big, 'anchor')
    n = 48
    ids = (np.arange(n) % 2).astype(np.int32)
    env = BatchedPaintEnv([DeviceTables(small, start_points=sp_s), DeviceTables(big, start_points=sp_b)], n,
                          env_part_id=ids, max_possible_point=[9148, 16000])
    o_s = oracle.Oracle(small, n // 2, start_points=sp_s, max_possible_point=9148)
    o_b = oracle.Oracle(big, n // 2, start_points=sp_b, max_possible_point=16000)
    start = np.arange(n) % 4
    obs = env.reset(start_idx=start).cpu().numpy()
    assert np.array_equal(obs[0::2], o_s.reset(start[0::2])) and np.array_equal(obs[1::2], o_b.reset(start[1::2]))
    rng = np.random.RandomState(9)
    for k in range(20):
        a = rng.randint(0, 4, size=n)
        o, r, d, i = env.step(a)
        o, r, d = o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy()
        o1, r1, d1, _ = o_s.step(a[0::2])
        o2, r2, d2, _ = o_b.step(a[1::2])
        assert np.array_equal(o[0::2], o1) and np.array_equal(o[1::2], o2), 'step %d' % k
        assert np.array_equal(r[0::2], r1) and np.array_equal(r[1::2], r2)
        assert np.array_equal(d[0::2], d1) and np.array_equal(d[1::2], d2)
    env.close()


@pytest.mark.parametrize('kw,n,steps', [
    (dict(paint_method='normal'), 24, 6),
    (dict(paint_method='normal', obs_mode='grid', overlap_penalty=True), 16, 5),
    (dict(color_mode='HSI'), 64, 20),
    (dict(color_mode='HSI', obs_mode='section', obs_grad=6, turning_penalty=True), 48, 12),
    (dict(color_mode='HSI', paint_method='normal', _beta=True), 12, 4),
])
def test_big_part_composes_with_cone_beams_and_thickness(kw, n, steps):
    """Seven of the reference's ten parts are 'big': PAINT_METHOD 'normal' and COLOR_MODE 'HSI' run on them too (masks in
    LDS throughout).  HSI rewards to 1e-12, everything else exact."""
    import random
    from paintrl_amd import part_tables
    from paintrl_amd.batched_env import BatchedPaintEnv
    from paintrl_amd.device_tables import DeviceTables
    kw = dict(kw)
    tables = synthetic_tables('door_rr_big', tex_size=(320, 320))
    sp = start_points_for(tables, 'all')
    beams = None
    if kw.pop('_beta', False):
        random.seed(3)
        beams = part_tables.beta_plain(tables.density)
    hsi = kw.get('color_mode') == 'HSI'
    mpp = int(0.95 * tables.sample_pos.shape[0])
    env = BatchedPaintEnv(DeviceTables(tables, obs_grad=kw.get('obs_grad', 4), start_points=sp, beams=beams), n, max_possible_point=mpp, **kw)
    orc = oracle.Oracle(tables, n, start_points=sp, threads=8, max_possible_point=mpp, beams=beams, **kw)
    rng = np.random.RandomState(17)
    start = rng.randint(0, len(sp), size=n)
    assert np.array_equal(env.reset(start_idx=start).cpu().numpy(), orc.reset(start))
    for k in range(steps):
        a = rng.randint(0, 4, size=n)
        o, r, d, i = env.step(a)
        oo, rr, dd, ii = orc.step(a)
        assert np.array_equal(o.cpu().numpy(), oo), 'obs, step %d' % k
        if hsi:
            assert np.allclose(r.cpu().numpy(), rr, rtol=0, atol=1e-12) and np.array_equal(env.thickness(), orc.thick), 'step %d' % k
        else:
            assert np.array_equal(r.cpu().numpy(), rr) and np.array_equal(i.cpu().numpy(), ii), 'reward, step %d' % k
        assert np.array_equal(d.cpu().numpy(), dd), 'done, step %d' % k
        if dd.any():
            new = rng.randint(0, len(sp), size=n)
            assert np.array_equal(env.reset(mask=dd, start_idx=new).cpu().numpy()[dd], orc.reset(new, mask=dd)[dd])
    words = env.painted_words().cpu().numpy().view(np.uint64)
    assert np.array_equal(env.parts[0].mask_to_canonical(words), np.stack([orc.painted_bits(e) for e in range(n)]))
    env.close()


@pytest.mark.parametrize('kw', [dict(), dict(color_mode='HSI'), dict(paint_method='normal', _small=True)])
def test_rollout_entry_points_on_every_configuration(kw):
    """prl_rollout_fragment (given actions, and with the policy) and prl_batch_act_step on a large part (fused kernels on the
    rows in HBM), with COLOR_MODE 'HSI' (their thickness builds, round 5) and under cone beams (which the library takes launch
    by launch): rows bit for bit those of prl_policy_act + prl_batch_step."""
    import torch
    from paintrl_amd.rollout import MLPPolicy, RolloutWorker
    kw = dict(kw)
    small = kw.pop('_small', False)
    tables = synthetic_tables('door_test') if small else synthetic_tables('door_rr_big', tex_size=(320, 320))
    sp = start_points_for(tables, 'all')
    n, T = 40, 6 if small else 12
    mpp = int(0.95 * tables.sample_pos.shape[0])
    mk = lambda: _env(tables, n, sp, auto_reset=True, seed=21, max_possible_point=mpp, **kw)     # noqa: E731
    env_a, env_b = mk(), mk()
    start = np.random.RandomState(1).randint(0, len(sp), size=n)
    o0 = env_a.reset(start_idx=start).clone()
    env_b.reset(start_idx=start)
    dev, od = env_a.device, env_a.obs_dim
    f64 = dict(dtype=torch.float64, device=dev)
    obs, fin = torch.zeros((T + 1, n, od), **f64), torch.zeros((T, n, od), **f64)
    rew, info = torch.zeros((T, n), **f64), torch.zeros((T, n, 2), **f64)
    done = torch.zeros((T, n), dtype=torch.uint8, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(9)
    act = torch.randint(0, 4, (T, n), generator=gen, device=dev, dtype=torch.int32)
    obs[0].copy_(o0)
    env_a.rollout_fragment(T, obs, fin, rew, done, info, act)
    torch.cuda.synchronize()
    for t in range(T):
        o, r, d, i = env_b.step(act[t])
        assert torch.equal(obs[t + 1], o) and torch.equal(rew[t], r) and torch.equal(done[t].bool(), d), 'row %d' % t
    assert torch.equal(env_a.painted_words(), env_b.painted_words())
    env_a.close()
    env_b.close()
    envs = [mk() for _ in range(3)]
    torch.manual_seed(3)
    policy = MLPPolicy(envs[0].obs_dim, 4).to(envs[0].device)
    workers = [RolloutWorker(envs[0], policy, fragment=T, seed=5), RolloutWorker(envs[1], policy, fragment=T, seed=5, persistent=True),
               RolloutWorker(envs[2], policy, fragment=T, seed=5, act_step=True)]
    out = [w.collect() for w in workers]
    torch.cuda.synchronize()
    for other in (1, 2):
        for k in out[0][0]:
            assert torch.equal(out[0][0][k], out[other][0][k]), (other, k)
        assert torch.equal(out[0][1], out[other][1])
    for e in envs:
        e.close()


@pytest.mark.skipif(not os.path.isfile(os.path.join(GOLDEN, 'episodes_door_big.npz')), reason='fixture not generated')
@pytest.mark.parametrize('name', ['g11_big_all_0', 'g11_big_all_1', 'g11_big_all_2', 'g11_big_grid_overlap', 'g11_big_section6'])
def test_big_part_replays_reference_episode(name):
    """Episodes recorded from the reference itself on the synthetic 480 x 480 door (Part_NO 8 layout)."""
    from test_oracle_golden import replay
    ep = load_episodes('door_big')[name]
    cfg = ep['cfg']
    tables = synthetic_tables('door_rr_big')
    env = _env(tables, 1, start_points_for(tables, cfg['start_mode']), **env_kwargs_from_cfg(cfg))

    def reset(idx):
        return env.reset(start_idx=[idx]).cpu().numpy()[0]

    def step(a, want_bits):
        obs, rew, done, info = env.step([a])
        bits = env.painted_bits(0) if want_bits else None
        return obs.cpu().numpy()[0], float(rew[0]), bool(done[0]), info.cpu().numpy()[0], bits

    replay(step, reset, ep, exact=True, atol=0)
    st = env.state()
    assert np.array_equal(st['pose'][0], ep['final_pose']) and np.array_equal(st['quat'][0], ep['final_quat'])
    assert st['total_return'][0] == float(ep['total_return'])
    env.close()


@pytest.mark.parametrize('kw,n,steps', [
    (dict(color_mode='HSI'), 24, 10),                               # four LDS copies of 1 234 words + the kd-walk rows
    (dict(), 24, 10),                                               # three copies
    (dict(color_mode='HSI', paint_method='normal'), 6, 3),          # five copies + the hit list
    (dict(obs_mode='section', obs_grad=6), 16, 6),                  # atan2 sectors: the counters' static LDS on top
])
def test_reference_sized_part_with_the_stale_tree_fits_the_lds(kw, n, steps):
    """A part of the reference's largest size (78 218 samples, mask stride 1 234 words; door_rr_big has 71 k) that also
    carries the stale kd-tree: the masks' LDS copies are sized NEXT TO the kernels' own static LDS (candidate list, shot
    centres, kd-walk rows: ~24 KB), so the launch takes fewer waves per workgroup instead of failing (4 waves x 4 copies x
    1 234 x 8 B + 24 KB > 160 KB)."""
    from paintrl_amd.batched_env import BatchedPaintEnv
    from paintrl_amd.device_tables import DeviceTables
    tables = synthetic_tables('test', tex_size=(560, 560))
    assert tables.sample_pos.shape[0] > 70000 and len(tables.kd_split_dim) > 0
    sp = start_points_for(tables, 'all')
    hsi = kw.get('color_mode') == 'HSI'
    mpp = int(0.95 * tables.sample_pos.shape[0])
    env = BatchedPaintEnv(DeviceTables(tables, obs_grad=kw.get('obs_grad', 4), start_points=sp), n, max_possible_point=mpp, **kw)
    assert env.mask_stride >= 1100
    orc = oracle.Oracle(tables, n, start_points=sp, threads=8, max_possible_point=mpp, **kw)
    rng = np.random.RandomState(71)
    start = rng.randint(0, len(sp), size=n)
    assert np.array_equal(env.reset(start_idx=start).cpu().numpy(), orc.reset(start))
    for k in range(steps):
        a = rng.randint(0, 4, size=n)
        o, r, d, i = env.step(a)
        oo, rr, dd, ii = orc.step(a)
        assert np.array_equal(o.cpu().numpy(), oo), 'obs, step %d' % k
        if hsi:
            assert np.allclose(r.cpu().numpy(), rr, rtol=0, atol=1e-12) and np.array_equal(env.thickness(), orc.thick), 'step %d' % k
        else:
            assert np.array_equal(r.cpu().numpy(), rr) and np.array_equal(i.cpu().numpy(), ii), 'reward, step %d' % k
        assert np.array_equal(d.cpu().numpy(), dd), 'done, step %d' % k
        if dd.any():
            new = rng.randint(0, len(sp), size=n)
            assert np.array_equal(env.reset(mask=dd, start_idx=new).cpu().numpy()[dd], orc.reset(new, mask=dd)[dd])
    words = env.painted_words().cpu().numpy().view(np.uint64)
    assert np.array_equal(env.parts[0].mask_to_canonical(words), np.stack([orc.painted_bits(e) for e in range(n)]))
    assert np.array_equal(env.observe().cpu().numpy(), orc.observe())
    env.close()


def test_fused_rollout_on_a_large_part_with_the_stale_tree():
    """The fused rollout kernels of large parts (k_rollout.hip compiled with -DPRL_KW=0: mask rows in HBM) on a part that also
    carries the stale vertex kd-tree (34 000 samples): the given-action fragment as ONE persistent launch equals the steps launch
    by launch, row by row, painted rows, last-shot rows and states at the end; OVERLAP_PENALTY on."""
    import torch
    tables = synthetic_tables('test', tex_size=(370, 370))
    assert tables.sample_pos.shape[0] > 16384 and len(tables.kd_split_dim) > 0
    sp = start_points_for(tables, 'all')
    n, T = 40, 14
    mpp = int(0.95 * tables.sample_pos.shape[0])
    mk = lambda: _env(tables, n, sp, auto_reset=True, seed=3, overlap_penalty=True, max_possible_point=mpp)     # noqa: E731
    env_a, env_b = mk(), mk()
    start = np.random.RandomState(2).randint(0, len(sp), size=n)
    o0 = env_a.reset(start_idx=start).clone()
    env_b.reset(start_idx=start)
    dev, od = env_a.device, env_a.obs_dim
    f64 = dict(dtype=torch.float64, device=dev)
    obs, fin = torch.zeros((T + 1, n, od), **f64), torch.zeros((T, n, od), **f64)
    rew, info = torch.zeros((T, n), **f64), torch.zeros((T, n, 2), **f64)
    done = torch.zeros((T, n), dtype=torch.uint8, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(12)
    act = torch.randint(0, 4, (T, n), generator=gen, device=dev, dtype=torch.int32)
    obs[0].copy_(o0)
    env_a.rollout_fragment(T, obs, fin, rew, done, info, act)
    torch.cuda.synchronize()
    for t in range(T):
        o, r, d, i = env_b.step(act[t])
        assert torch.equal(obs[t + 1], o) and torch.equal(rew[t], r) and torch.equal(info[t], i) and torch.equal(done[t].bool(), d), 'row %d' % t
    assert torch.equal(env_a.painted_words(), env_b.painted_words())
    assert torch.equal(env_a.last_shot_words()[0], env_b.last_shot_words()[0])
    sa, sb = env_a.state(), env_b.state()
    for k in sa:
        if k != 'facet_hint':
            assert np.array_equal(sa[k], sb[k]), k
    env_a.close()
    env_b.close()
