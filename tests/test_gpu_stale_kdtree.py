"""Parts whose vertex rows the reference moves under its kd-tree (bpw:943-946): the device walks the stale tree the
way scipy does (nearest_vertex_kd), so trajectories stay those of the reference.  Synthetic coarse sheet ('test')."""
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


@pytest.mark.parametrize('kw', [dict(obs_mode='section'), dict(obs_mode='grid', overlap_penalty=True),
                                dict(obs_mode='section', obs_grad=6)])
def test_stale_tree_part_matches_oracle(kw):
    tables = synthetic_tables('test')
    assert len(tables.vertices_mutated) > 0 and len(tables.kd_split_dim) > 0
    sp = start_points_for(tables, 'all')
    n, steps = 192, 40
    env = _env(tables, n, sp, max_possible_point=14000, **kw)
    orc = oracle.Oracle(tables, n, start_points=sp, threads=8, max_possible_point=14000, **kw)
    rng = np.random.RandomState(3)
    start = rng.randint(0, len(sp), size=n)
    assert np.array_equal(env.reset(start_idx=start).cpu().numpy(), orc.reset(start))
    for k in range(steps):
        a = rng.randint(0, 4, size=n)
        o, r, d, i = env.step(a)
        oo, rr, dd, ii = orc.step(a)
        assert np.array_equal(o.cpu().numpy(), oo), 'obs, step %d' % k
        assert np.array_equal(r.cpu().numpy(), rr) and np.array_equal(d.cpu().numpy(), dd), 'step %d' % k
        if dd.any():
            new = rng.randint(0, len(sp), size=n)
            assert np.array_equal(env.reset(mask=dd, start_idx=new).cpu().numpy()[dd], orc.reset(new, mask=dd)[dd])
    st = env.state()
    for e in range(n):
        so = orc.state(e)
        assert np.array_equal(st['pose'][e], so['pose']) and np.array_equal(st['quat'][e], so['quat'])
    words = env.painted_words().cpu().numpy().view(np.uint64)
    assert np.array_equal(env.parts[0].mask_to_canonical(words), np.stack([orc.painted_bits(e) for e in range(n)]))
    env.close()


def test_queries_on_the_split_planes_and_vertex_coordinates():
    """Tools parked so that the sub-shots' hit points lie EXACTLY on the stale tree's split planes (x[dim] < split is false
    there: the greater child is the near one), on vertex coordinates and half-way between neighbouring vertices (equal
    distances: the first point in tree order wins, equal bounds pop in list order), and well off the sheet's outline (long
    walks, many queued cells): one step in each of the four directions, everything equal to the oracle.  The tree's
    lane-parallel query (nearest_vertex_kd_lanes) computes bounds, near / far choices and every leaf's answer at once; this
    is where its comparisons meet scipy's at equality."""
    tables = synthetic_tables('test')
    sd, sp_val = np.asarray(tables.kd_split_dim), np.asarray(tables.kd_split)
    side = np.asarray(tables.vertex_is_side).astype(bool)
    verts = np.asarray(tables.vertices, dtype=np.float64)[side]
    a0 = [k for k in range(3) if k not in (tables.a1, tables.a2)][0]
    lo, hi = verts.min(0), verts.max(0)
    rng = np.random.RandomState(17)
    poses = []
    for node in np.nonzero(sd >= 0)[0]:                             # on each split plane, at random and at vertex coordinates
        for k in range(6):
            p = rng.uniform(lo, hi)
            if k >= 3:
                p = verts[rng.randint(len(verts))].copy()
            p[sd[node]] = sp_val[node]
            poses.append(p)
    for k in range(40):                                             # on vertices, half-way between two, off the outline
        v = verts[rng.randint(len(verts))]
        w = verts[np.argsort(((verts - v) ** 2).sum(1))[1 + k % 3]]
        poses.append(v.copy())
        poses.append(0.5 * (v + w))
        q = rng.uniform(lo, hi)
        q[tables.a1 if k % 2 else tables.a2] += (hi - lo)[tables.a1 if k % 2 else tables.a2] * (0.3 if k % 4 < 2 else -0.3)
        poses.append(q)
    poses = np.repeat(np.asarray(poses), 4, axis=0)                 # every pose steps in all four directions
    for p in poses:
        p[a0] = verts[:, a0].max()
    n = len(poses)
    acts = np.arange(n) % 4
    orn = [0.0, 0.0, 0.0]
    orn[a0] = -1.0
    sp = start_points_for(tables, 'all')
    for kw in (dict(obs_mode='section'), dict(obs_mode='section', paint_method='normal')):
        env = _env(tables, n, sp, max_possible_point=14000, **kw)
        orc = oracle.Oracle(tables, n, start_points=sp, threads=8, max_possible_point=14000, **kw)
        start = np.arange(n) % len(sp)
        assert np.array_equal(env.reset(start_idx=start).cpu().numpy(), orc.reset(start))
        for i in range(n):
            env.set_pose(i, poses[i], orn)
            orc.set_pose(i, poses[i], orn)
        for k in range(3):
            a = (acts + k) % 4
            o, r, d, _ = env.step(a)
            oo, rr, dd, _ = orc.step(a)
            assert np.array_equal(o.cpu().numpy(), oo), 'obs, step %d' % k
            assert np.array_equal(r.cpu().numpy(), rr) and np.array_equal(d.cpu().numpy(), dd), 'step %d' % k
            st = env.state()
            for e in range(n):
                so = orc.state(e)
                assert np.array_equal(st['pose'][e], so['pose']) and np.array_equal(st['quat'][e], so['quat']), (k, e)
        env.close()


def test_stale_tree_on_a_large_part_and_in_a_mixed_batch():
    """The same tree under the LDS-mask kernels (the reference's door_lf / door_rf carry both properties), next to a
    part without a tree in one batch."""
    from paintrl_amd.batched_env import BatchedPaintEnv
    from paintrl_amd.device_tables import DeviceTables
    sparse_big = synthetic_tables('test', tex_size=(360, 360))
    assert sparse_big.sample_pos.shape[0] > 16384 and len(sparse_big.kd_split_dim) > 0
    door = synthetic_tables('door_test')
    sp_b, sp_d = start_points_for(sparse_big, 'all'), start_points_for(door, 'all')
    n = 64
    ids = (np.arange(n) % 2).astype(np.int32)
    env = BatchedPaintEnv([DeviceTables(sparse_big, start_points=sp_b), DeviceTables(door, start_points=sp_d)], n,
                          env_part_id=ids, max_possible_point=[30000, 9148])
    o_b = oracle.Oracle(sparse_big, n // 2, start_points=sp_b, max_possible_point=30000)
    o_d = oracle.Oracle(door, n // 2, start_points=sp_d, max_possible_point=9148)
    rng = np.random.RandomState(4)
    start = np.where(ids == 0, rng.randint(0, len(sp_b), size=n), rng.randint(0, len(sp_d), size=n))
    obs = env.reset(start_idx=start).cpu().numpy()
    assert np.array_equal(obs[0::2], o_b.reset(start[0::2])) and np.array_equal(obs[1::2], o_d.reset(start[1::2]))
    for k in range(25):
        a = rng.randint(0, 4, size=n)
        o, r, d, i = env.step(a)
        o, r, d = o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy()
        o1, r1, d1, _ = o_b.step(a[0::2])
        o2, r2, d2, _ = o_d.step(a[1::2])
        assert np.array_equal(o[0::2], o1) and np.array_equal(o[1::2], o2), 'step %d' % k
        assert np.array_equal(r[0::2], r1) and np.array_equal(r[1::2], r2)
        assert np.array_equal(d[0::2], d1) and np.array_equal(d[1::2], d2)
    env.close()


@pytest.mark.parametrize('kw,n,steps', [(dict(paint_method='normal'), 40, 8),
                                        (dict(paint_method='normal', obs_mode='grid', overlap_penalty=True), 24, 6),
                                        (dict(color_mode='HSI'), 96, 30)])
def test_stale_tree_composes_with_cone_beams_and_thickness(kw, n, steps):
    """The reference's own sheet (square.urdf, Part_NO 1) carries moved vertex rows: PAINT_METHOD 'normal' and
    COLOR_MODE 'HSI' have to work on such parts too.  HSI rewards to 1e-12 (summation order), everything else exact."""
    tables = synthetic_tables('test')
    sp = start_points_for(tables, 'all')
    hsi = kw.get('color_mode') == 'HSI'
    env = _env(tables, n, sp, max_possible_point=14000, **kw)
    orc = oracle.Oracle(tables, n, start_points=sp, threads=8, max_possible_point=14000, **kw)
    rng = np.random.RandomState(13)
    start = rng.randint(0, len(sp), size=n)
    assert np.array_equal(env.reset(start_idx=start).cpu().numpy(), orc.reset(start))
    for k in range(steps):
        a = rng.randint(0, 4, size=n)
        o, r, d, i = env.step(a)
        oo, rr, dd, ii = orc.step(a)
        assert np.array_equal(o.cpu().numpy(), oo), 'obs, step %d' % k
        if hsi:
            assert np.allclose(r.cpu().numpy(), rr, rtol=0, atol=1e-12) and np.array_equal(env.thickness(), orc.thick)
        else:
            assert np.array_equal(r.cpu().numpy(), rr) and np.array_equal(i.cpu().numpy(), ii), 'reward, step %d' % k
        assert np.array_equal(d.cpu().numpy(), dd), 'done, step %d' % k
        if dd.any():
            new = rng.randint(0, len(sp), size=n)
            assert np.array_equal(env.reset(mask=dd, start_idx=new).cpu().numpy()[dd], orc.reset(new, mask=dd)[dd])
    words = env.painted_words().cpu().numpy().view(np.uint64)
    assert np.array_equal(env.parts[0].mask_to_canonical(words), np.stack([orc.painted_bits(e) for e in range(n)]))
    env.close()


@pytest.mark.parametrize('kw,n,steps', [(dict(paint_method='normal'), 16, 5), (dict(color_mode='HSI'), 48, 15)])
def test_stale_tree_on_a_large_part_composes_too(kw, n, steps):
    """Both properties at once -- what the reference's door_lf / door_rf / door_rr are: a stale tree AND more than 16 384
    samples -- under cone beams and under COLOR_MODE 'HSI'."""
    tables = synthetic_tables('test', tex_size=(360, 360))
    assert tables.sample_pos.shape[0] > 16384 and len(tables.kd_split_dim) > 0
    sp = start_points_for(tables, 'all')
    hsi = kw.get('color_mode') == 'HSI'
    env = _env(tables, n, sp, max_possible_point=30000, **kw)
    orc = oracle.Oracle(tables, n, start_points=sp, threads=8, max_possible_point=30000, **kw)
    rng = np.random.RandomState(23)
    start = rng.randint(0, len(sp), size=n)
    assert np.array_equal(env.reset(start_idx=start).cpu().numpy(), orc.reset(start))
    for k in range(steps):
        a = rng.randint(0, 4, size=n)
        o, r, d, i = env.step(a)
        oo, rr, dd, ii = orc.step(a)
        assert np.array_equal(o.cpu().numpy(), oo), 'obs, step %d' % k
        if hsi:
            assert np.allclose(r.cpu().numpy(), rr, rtol=0, atol=1e-12) and np.array_equal(env.thickness(), orc.thick)
        else:
            assert np.array_equal(r.cpu().numpy(), rr), 'reward, step %d' % k
        assert np.array_equal(d.cpu().numpy(), dd), 'done, step %d' % k
        if dd.any():
            new = rng.randint(0, len(sp), size=n)
            assert np.array_equal(env.reset(mask=dd, start_idx=new).cpu().numpy()[dd], orc.reset(new, mask=dd)[dd])
    env.close()


def test_stale_tree_in_the_rollout_kernels():
    """prl_rollout_fragment (given actions and with the policy) and prl_batch_act_step on a part with the stale tree:
    rows bit for bit those of one prl_batch_step (+ prl_policy_act) launch per step."""
    import torch
    from paintrl_amd.rollout import MLPPolicy, RolloutWorker
    tables = synthetic_tables('test')
    sp = start_points_for(tables, 'all')
    n, T = 150, 30
    kw = dict(auto_reset=True, seed=21, max_possible_point=14000)
    env_a, env_b = _env(tables, n, sp, **kw), _env(tables, n, sp, **kw)
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
    envs = [_env(tables, n, sp, **kw) for _ in range(3)]
    torch.manual_seed(3)
    policy = MLPPolicy(envs[0].obs_dim, 4).to(envs[0].device)
    workers = [RolloutWorker(envs[0], policy, fragment=T, seed=5), RolloutWorker(envs[1], policy, fragment=T, seed=5, persistent=True),
               RolloutWorker(envs[2], policy, fragment=T, seed=5, act_step=True)]
    for frag in range(2):
        out = [w.collect() for w in workers]
        torch.cuda.synchronize()
        for other in (1, 2):
            for k in out[0][0]:
                assert torch.equal(out[0][0][k], out[other][0][k]), (frag, other, k)
            assert torch.equal(out[0][1], out[other][1])
    assert torch.equal(envs[0].painted_words(), envs[1].painted_words()) and torch.equal(envs[0].painted_words(), envs[2].painted_words())
    for e in envs:
        e.close()


@pytest.mark.skipif(not os.path.isfile(os.path.join(GOLDEN, 'episodes_sparse.npz')), reason='fixture not generated')
def test_stale_tree_part_replays_reference_episodes():
    from test_oracle_golden import replay
    for name, ep in sorted(load_episodes('sparse').items()):
        cfg = ep['cfg']
        tables = synthetic_tables('test')
        env = _env(tables, 1, start_points_for(tables, cfg['start_mode']), **env_kwargs_from_cfg(cfg))

        def reset(idx):
            return env.reset(start_idx=[idx]).cpu().numpy()[0]

        def step(a, want_bits):
            obs, rew, done, info = env.step([a])
            bits = env.painted_bits(0) if want_bits else None
            return obs.cpu().numpy()[0], float(rew[0]), bool(done[0]), info.cpu().numpy()[0], bits

        replay(step, reset, ep, exact=True)
        st = env.state()
        assert np.array_equal(st['pose'][0], ep['final_pose']) and st['total_return'][0] == float(ep['total_return']), name
        env.close()
