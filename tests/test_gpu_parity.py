"""GPU parity: the HIP path (through the C ABI) against the golden vectors and the CPU oracle."""
import numpy as np
import pytest

import oracle
from conftest import env_kwargs_from_cfg, load_episodes, start_points_for, synthetic_tables
from test_oracle_golden import CASES, PART, replay

pytestmark = pytest.mark.gpu


def _gpu_env(tables, n, start_points, beams=None, **kw):
    from paintrl_amd.batched_env import BatchedPaintEnv
    from paintrl_amd.device_tables import DeviceTables
    obs_grad = kw.get('obs_grad', 4)
    return BatchedPaintEnv(DeviceTables(tables, obs_grad=obs_grad, start_points=start_points, beams=beams), n, **kw)


DEVICE_CASES = CASES


@pytest.mark.parametrize('tag,name', DEVICE_CASES)
def test_gpu_replays_reference_episode(tag, name):
    ep = load_episodes(tag)[name]
    cfg = ep['cfg']
    tables = synthetic_tables(PART[tag], cfg.get('paint_radius', 0.051))
    hsi = cfg.get('color_mode', 'RGB') == 'HSI'
    env = _gpu_env(tables, 1, start_points_for(tables, cfg['start_mode']), beams=ep.get('beams'),
                   color_mode=cfg.get('color_mode', 'RGB'), **env_kwargs_from_cfg(cfg))
    continuous = cfg['action_mode'] == 'continuous'

    def reset(idx):
        return env.reset(start_idx=[idx]).cpu().numpy()[0]

    def step(a, want_bits):
        obs, rew, done, info = env.step([a])
        bits = env.painted_bits(0) if want_bits else None
        return obs.cpu().numpy()[0], float(rew[0]), bool(done[0]), info.cpu().numpy()[0], bits

    # float deposit sums of COLOR_MODE='HSI': 1e-12 (summation order); continuous actions: device libm, 1e-9
    replay(step, reset, ep, exact=not (continuous or hsi), atol=1e-12 if hsi else 1e-9)
    st = env.state()
    if hsi:
        assert np.array_equal(env.thickness(0), ep['final_thick'])
        assert np.array_equal(st['pose'][0], ep['final_pose']) and np.array_equal(st['quat'][0], ep['final_quat'])
    elif not continuous:
        assert np.array_equal(st['pose'][0], ep['final_pose']) and np.array_equal(st['quat'][0], ep['final_quat'])
        assert st['total_return'][0] == float(ep['total_return'])
    env.close()


@pytest.mark.parametrize('part,kw', [
    ('door_test', dict(obs_mode='section')),
    ('door_test', dict(obs_mode='grid', overlap_penalty=True, turning_penalty=True)),
    ('door_test', dict(obs_mode='discrete', termination_mode='hybrid')),
    ('square', dict(obs_mode='section', max_possible_point=14350)),
    ('square', dict(obs_mode='simple', n_discrete=8, max_possible_point=14350)),
    ('door_test', dict(obs_mode='section', paint_method='normal', _n=48, _steps=8)),
    # the cone beams' work lists at a realistic fill (256 sub-lists each of the far list and the ray list: k_cone_beams.hip)
    ('door_test', dict(obs_mode='section', paint_method='normal', _n=1536, _steps=12)),
    ('square', dict(obs_mode='grid', paint_method='normal', overlap_penalty=True, max_possible_point=14350, _n=32,
                    _steps=6)),
    ('square', dict(obs_mode='section', paint_method='normal', max_possible_point=14350, _n=768, _steps=8)),
    # the sheet with a seam of doubled vertices: equally near vertices resolve in the reference tree's order (vertex_rank)
    ('door_lf', dict(obs_mode='section', max_possible_point=14350)),
    ('door_lf', dict(obs_mode='grid', overlap_penalty=True, paint_method='normal', max_possible_point=14350, _n=64, _steps=8)),
])
def test_gpu_matches_oracle_on_random_batch(part, kw):
    tables = synthetic_tables(part)
    sp = start_points_for(tables, 'all')
    kw = dict(kw)
    n, steps = kw.pop('_n', 192), kw.pop('_steps', 40)
    env = _gpu_env(tables, n, sp, **kw)
    orc = oracle.Oracle(tables, n, start_points=sp, threads=8, **kw)
    rng = np.random.RandomState(11)
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
        # finished envs restart from a fresh start point on both sides
        if dd.any():
            new = rng.randint(0, len(sp), size=n)
            o2 = env.reset(mask=dd, start_idx=new).cpu().numpy()
            o3 = orc.reset(new, mask=dd)
            assert np.array_equal(o2[dd], o3[dd])
    words = env.painted_words().cpu().numpy().view(np.uint64)
    bits = env.parts[0].mask_to_canonical(words)
    for e in range(n):
        assert np.array_equal(bits[e], orc.painted_bits(e)), 'painted set, env %d' % e
    st = env.state()
    for e in range(n):
        so = orc.state(e)
        assert np.array_equal(st['pose'][e], so['pose']) and np.array_equal(st['quat'][e], so['quat'])
        assert st['total_return'][e] == so['total_return'] and st['step_counter'][e] == so['step_counter']
    env.close()


def test_gpu_cone_beams_on_a_collision_set_that_is_not_convex():
    """PAINT_METHOD 'normal' with collision_mode='trimesh': no hull to walk, every beam of every trip is left over and the trips
    go through the general code whole (the trip list of k_cone_beams.hip)."""
    from paintrl_amd import part_tables, synth_parts
    from paintrl_amd.batched_env import BatchedPaintEnv
    from paintrl_amd.device_tables import DeviceTables
    tables = part_tables.build_part_tables(mesh=synth_parts.synthetic_mesh('square'), tex_size=(240, 240),
                                           name='square', collision_mode='trimesh')
    sp = start_points_for(tables, 'all')
    dt = DeviceTables(tables, start_points=sp)
    n, steps = 40, 6
    env = BatchedPaintEnv(dt, n, max_possible_point=14350, paint_method='normal')
    orc = oracle.Oracle(tables, n, start_points=sp, max_possible_point=14350, threads=8, paint_method='normal')
    rng = np.random.RandomState(33)
    start = rng.randint(0, len(sp), size=n)
    assert np.array_equal(env.reset(start_idx=start).cpu().numpy(), orc.reset(start))
    for k in range(steps):
        a = rng.randint(0, 4, size=n)
        o, r, d, i = env.step(a)
        oo, rr, dd, ii = orc.step(a)
        assert np.array_equal(o.cpu().numpy(), oo) and np.array_equal(r.cpu().numpy(), rr), 'step %d' % k
        assert np.array_equal(d.cpu().numpy(), dd)
    words = env.painted_words().cpu().numpy().view(np.uint64)
    bits = dt.mask_to_canonical(words)
    assert all(np.array_equal(bits[e], orc.painted_bits(e)) for e in range(n))
    env.close()


def test_gpu_auto_reset_matches_manual_reset():
    """auto_reset inside the step kernel == done -> reset with the same start index."""
    tables = synthetic_tables('door_test')
    sp = start_points_for(tables, 'all')
    n, steps = 128, 60
    env_a = _gpu_env(tables, n, sp, auto_reset=True)
    env_m = _gpu_env(tables, n, sp)
    rng = np.random.RandomState(5)
    start = rng.randint(0, len(sp), size=n)
    env_a.reset(start_idx=start)
    env_m.reset(start_idx=start)
    for k in range(steps):
        a = rng.randint(0, 4, size=n)
        nxt = rng.randint(0, len(sp), size=n)
        oa, ra, da, ia = env_a.step(a, start_idx=nxt)
        fa = env_a.final_obs.cpu().numpy()
        oa, ra, da = oa.cpu().numpy().copy(), ra.cpu().numpy().copy(), da.cpu().numpy().copy()
        om, rm, dm, im = env_m.step(a)
        om, dm = om.cpu().numpy().copy(), dm.cpu().numpy().copy()
        assert np.array_equal(da, dm) and np.array_equal(ra, rm.cpu().numpy())
        assert np.array_equal(oa[~dm], om[~dm]) and np.array_equal(fa[dm], om[dm])
        if dm.any():
            o2 = env_m.reset(mask=dm, start_idx=nxt).cpu().numpy()
            assert np.array_equal(oa[dm], o2[dm])
    assert np.array_equal(env_a.painted_words().cpu().numpy(), env_m.painted_words().cpu().numpy())
    assert (env_a.state()['episode'] >= 1).all()
    env_a.close()
    env_m.close()


def test_gpu_ray_batch_matches_numpy_and_oracle():
    from paintrl_amd import geometry as geo
    tables = synthetic_tables('door_test')
    env = _gpu_env(tables, 1, None)
    rng = np.random.RandomState(2)
    n = 3000
    o = np.stack([rng.uniform(-0.2, 0.3, n), rng.uniform(-0.8, 0.6, n), rng.uniform(0.1, 1.3, n)], axis=1)
    d = o + np.stack([-rng.uniform(0.2, 1.0, n), rng.normal(0, 0.2, n), rng.normal(0, 0.2, n)], axis=1)
    idx, t, pos = geo.ray_closest_hit(tables.col_v0, tables.col_e1, tables.col_e2, o, d)
    gi, gt, gp = env.ray_test_batch(o, d)
    gi, gt, gp = gi.cpu().numpy(), gt.cpu().numpy(), gp.cpu().numpy()
    assert (idx >= 0).sum() > 500 and (idx < 0).sum() > 100
    assert np.array_equal(gi, idx)
    hit = idx >= 0
    assert np.array_equal(gt[hit], t[hit]) and np.array_equal(gp[hit], pos[hit])
    oi, ot, op = oracle.Oracle(tables, 1).ray_batch(o, d)
    assert np.array_equal(oi, idx) and np.array_equal(ot[hit], t[hit]) and np.array_equal(op[hit], pos[hit])
    env.close()


@pytest.mark.parametrize('part', ['door_test', 'square'])
def test_gpu_rays_grazing_the_outline(part):
    """The general ray search certifies a miss by the collision set's outline (prl_ray.hpp beam_outside_outline_wave,
    margin 1e-6 m): rays along and near the silhouette -- nanometres to millimetres inside and outside every outline
    edge, parallel to the third axis and tilted -- must agree with the brute-force numpy ray and the oracle."""
    from scipy.spatial import ConvexHull
    from paintrl_amd import geometry as geo
    tables = synthetic_tables(part)
    env = _gpu_env(tables, 1, None)
    a0, a1, a2 = tables.a0, tables.a1, tables.a2
    v0, e1, e2 = (np.asarray(x) for x in (tables.col_v0, tables.col_e1, tables.col_e2))
    corners = np.concatenate([v0, v0 + e1, v0 + e2])
    hull = ConvexHull(corners[:, [a1, a2]])
    poly = corners[hull.vertices][:, [a1, a2]]                       # counter-clockwise
    zlo, zhi = corners[:, a0].min(), corners[:, a0].max()
    rng = np.random.RandomState(5)
    o_list, d_list = [], []
    for i in range(len(poly)):
        p0, p1 = poly[i], poly[(i + 1) % len(poly)]
        e = p1 - p0
        nrm = np.array([e[1], -e[0]]) / np.linalg.norm(e)            # outward of a CCW polygon
        for s in (0.0, 0.03, 0.5, 0.97, 1.0):
            for off in (-1e-3, -1e-5, -2e-6, -1e-7, -1e-9, 0.0, 1e-9, 1e-7, 9e-7, 1.1e-6, 2e-6, 1e-5, 1e-3):
                q = p0 + s * e + off * nrm
                for tilt in (0.0, 1e-3, -0.05):
                    o = np.zeros(3)
                    o[a1], o[a2], o[a0] = q[0], q[1], zhi + 0.1
                    d = o.copy()
                    d[a0] = zlo - 0.1
                    d[a1] += tilt * rng.uniform(-1, 1)
                    d[a2] += tilt * rng.uniform(-1, 1)
                    o_list.append(o)
                    d_list.append(d)
    o, d = np.array(o_list), np.array(d_list)
    idx, t, pos = geo.ray_closest_hit(v0, e1, e2, o, d)
    gi, gt, gp = env.ray_test_batch(o, d)
    gi, gt, gp = gi.cpu().numpy(), gt.cpu().numpy(), gp.cpu().numpy()
    assert (idx >= 0).sum() > 100 and (idx < 0).sum() > 100
    assert np.array_equal(gi, idx)
    hit = idx >= 0
    assert np.array_equal(gt[hit], t[hit]) and np.array_equal(gp[hit], pos[hit])
    oi, ot, op = oracle.Oracle(tables, 1).ray_batch(o, d)
    assert np.array_equal(oi, idx) and np.array_equal(ot[hit], t[hit])
    env.close()


def test_gpu_mixed_part_batch():
    """Config 5: door and sheet envs interleaved in one batch, each equal to its oracle."""
    from paintrl_amd.batched_env import BatchedPaintEnv
    from paintrl_amd.device_tables import DeviceTables
    door, sheet = synthetic_tables('door_test'), synthetic_tables('square')
    sp_d, sp_s = start_points_for(door, 'anchor'), start_points_for(sheet, 'anchor')
    n = 64
    ids = (np.arange(n) % 2).astype(np.int32)
    env = BatchedPaintEnv([DeviceTables(door, start_points=sp_d), DeviceTables(sheet, start_points=sp_s)], n,
                          env_part_id=ids, max_possible_point=[9148, 14350])
    od = oracle.Oracle(door, n // 2, start_points=sp_d, max_possible_point=9148)
    os_ = oracle.Oracle(sheet, n // 2, start_points=sp_s, max_possible_point=14350)
    start = np.arange(n) % 4
    obs = env.reset(start_idx=start).cpu().numpy()
    assert np.array_equal(obs[0::2], od.reset(start[0::2])) and np.array_equal(obs[1::2], os_.reset(start[1::2]))
    rng = np.random.RandomState(9)
    for k in range(25):
        a = rng.randint(0, 4, size=n)
        o, r, d, i = env.step(a)
        o, r, d = o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy()
        o1, r1, d1, _ = od.step(a[0::2])
        o2, r2, d2, _ = os_.step(a[1::2])
        assert np.array_equal(o[0::2], o1) and np.array_equal(o[1::2], o2)
        assert np.array_equal(r[0::2], r1) and np.array_equal(r[1::2], r2)
        assert np.array_equal(d[0::2], d1) and np.array_equal(d[1::2], d2)
    env.close()


def test_gpu_trimesh_collision_mode_matches_oracle():
    """collision_mode='trimesh': rays against every mesh triangle (5 344 on the sheet -> 100+ chunks,
    i.e. more than one 64-chunk pass of the two-level cull)."""
    from paintrl_amd import part_tables, synth_parts
    from paintrl_amd.batched_env import BatchedPaintEnv
    from paintrl_amd.device_tables import DeviceTables
    tables = part_tables.build_part_tables(mesh=synth_parts.synthetic_mesh('square'), tex_size=(240, 240),
                                           name='square', collision_mode='trimesh')
    assert tables.col_v0.shape[0] == tables.tri_side.shape[0]
    sp = start_points_for(tables, 'all')
    dt = DeviceTables(tables, start_points=sp)
    assert dt.n_col_chunks > 64
    n, steps = 96, 25
    env = BatchedPaintEnv(dt, n, max_possible_point=14350)
    orc = oracle.Oracle(tables, n, start_points=sp, max_possible_point=14350, threads=8)
    rng = np.random.RandomState(21)
    start = rng.randint(0, len(sp), size=n)
    assert np.array_equal(env.reset(start_idx=start).cpu().numpy(), orc.reset(start))
    for k in range(steps):
        a = rng.randint(0, 4, size=n)
        o, r, d, i = env.step(a)
        oo, rr, dd, ii = orc.step(a)
        assert np.array_equal(o.cpu().numpy(), oo) and np.array_equal(r.cpu().numpy(), rr), 'step %d' % k
        assert np.array_equal(d.cpu().numpy(), dd)
    words = env.painted_words().cpu().numpy().view(np.uint64)
    bits = dt.mask_to_canonical(words)
    assert all(np.array_equal(bits[e], orc.painted_bits(e)) for e in range(n))
    # the rayTestBatch drop-in agrees with numpy on this triangle soup too
    from paintrl_amd import geometry as geo
    o3 = np.stack([rng.uniform(-0.3, 0.2, 500), rng.uniform(-0.7, 0.5, 500), rng.uniform(0.2, 1.3, 500)], axis=1)
    e3 = o3 + np.stack([-rng.uniform(0.1, 1.0, 500), rng.normal(0, 0.1, 500), rng.normal(0, 0.1, 500)], axis=1)
    idx, t, pos = geo.ray_closest_hit(tables.col_v0, tables.col_e1, tables.col_e2, o3, e3)
    gi, gt, gp = env.ray_test_batch(o3, e3)
    hit = idx >= 0
    assert hit.sum() > 50 and np.array_equal(gi.cpu().numpy(), idx) and np.array_equal(gt.cpu().numpy()[hit], t[hit])
    env.close()


@pytest.mark.parametrize('name', ['t4_offpart_counter_04', 't4_offpart_mixed_06'])
def test_gpu_replays_the_off_part_counter_limit(name):
    """rob:292-300 on the device: more than 1000 counted misses end the episode (S.terminate_counter > NOT_ON_PART_TERMINATE,
    prl_step.hpp sub_shot) -- the reference's own run beside the sheet, with the tool moved mid-episode through
    prl_batch_set_pose (Robot.reset([pose, orn]), rob:366-372); every row and the final state equal the recording."""
    from test_oracle_golden import replay
    ep = load_episodes('sheet_offpart')[name]
    cfg = ep['cfg']
    tables = synthetic_tables('square')
    sp = start_points_for(tables, cfg['start_mode'])
    env = _gpu_env(tables, 3, sp, **env_kwargs_from_cfg(cfg))         # env 1 is the one replayed; 0 and 2 stay put

    def reset(idx):
        obs = env.reset(start_idx=np.full(3, idx, dtype=np.int32)).cpu().numpy()[1].copy()
        env.set_pose(1, ep['set_pose'], ep['set_orn'])
        return obs

    def step(a, want_bits):
        obs, rew, done, info = env.step(np.array([1, a, 1], dtype=np.int32))
        return (obs.cpu().numpy()[1].copy(), float(rew[1]), bool(done[1]), info.cpu().numpy()[1].copy(),
                env.painted_bits(1) if want_bits else None)

    replay(step, reset, ep, exact=True)
    st = env.state()
    assert st['terminate_counter'][1] == int(ep['terminate_counter']) > 1000 and st['terminate'][1] == 1
    assert np.array_equal(st['pose'][1], ep['final_pose']) and np.array_equal(st['quat'][1], ep['final_quat'])
    assert st['total_return'][1] == float(ep['total_return'])
    env.close()
