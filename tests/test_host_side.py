"""CPU tests of the host side: table builder vs the reference's tables (G0), ParamTestEnv (G1),
device layout invariants, the C ABI surface, numpy ray vs oracle ray."""
import hashlib
import os
import re

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, REPO, synthetic_tables
from paintrl_amd import _lib, config, geometry, part_tables
from paintrl_amd.device_tables import CELL, DeviceTables
from paintrl_amd.param_test_env import ParamTestEnv, spiral, zigzag


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize('tag,name,radius', [('door', 'door_test', 0.051), ('sheet', 'square', 0.051),
                                             ('sheet_r040', 'square', 0.04), ('seam', 'door_lf', 0.051)])
def test_tables_match_reference_digests(tag, name, radius):
    """G0: every static table equals what the reference built for the same synthetic mesh (the last case
    with PaintToolProfile.PAINT_RADIUS = 0.04)."""
    g = np.load(os.path.join(GOLDEN, 'g0_tables_%s.npz' % tag))
    t = synthetic_tables(name, radius)
    front = t.tri_side == 1
    assert int(g['P']) == t.sample_pos.shape[0] and int(g['T']) == t.tri_side.shape[0]
    assert int(g['V']) == t.vertices.shape[0]
    assert list(g['side_counts']) == [int((t.tri_side == k).sum()) for k in (1, 2, 3)]
    assert list(g['axes']) == [t.a1, t.a2, t.a0]
    assert np.array_equal(g['ranges'], np.array(t.ranges)) and float(g['lwr']) == t.lwr
    assert np.array_equal(g['grid_lo'], t.grid_lo) and np.array_equal(g['grid_hi'], t.grid_hi)
    assert float(g['density']) == t.density and np.array_equal(g['beams'], t.beams)
    assert str(g['sha_pix']) == sha(t.sample_pix.astype(np.int32))
    assert str(g['sha_pos']) == sha(t.sample_pos)
    assert str(g['sha_sides']) == sha(t.tri_side.astype(np.int8))
    assert str(g['sha_front_normals']) == sha(t.tri_normal[front])
    assert str(g['sha_side_vertices']) == sha(t._side_data)
    assert str(g['sha_cells4']) == sha(part_tables.grid_observation_cells(t, 4).astype(np.int32))
    sp = np.array(part_tables.start_points(t, 'all'), dtype=np.float64)
    assert int(g['n_start_all']) == sp.shape[0] and str(g['sha_start_points']) == sha(sp)
    se = np.array(part_tables.start_points(t, 'edge'), dtype=np.float64)
    assert int(g['n_start_edge']) == se.shape[0] and str(g['sha_start_points_edge']) == sha(se)
    assert len(part_tables.start_points(t, 'anchor')) == 4 and len(part_tables.start_points(t, 'fixed')) == 1
    assert np.array_equal(g['pos_head'], t.sample_pos[:32]) and np.array_equal(g['normals_tail'], t.tri_normal[front][-32:])
    assert len(t.vertices_mutated) == 0        # synthetic parts never trigger the sparse-row mutation
    assert t.paint_radius == radius


@pytest.mark.parametrize('size,driver,steps,total', [(22, zigzag, 399, 320.2), (20, spiral, 323, 259.4),
                                                      (14, zigzag, 143, 115.4)])
def test_param_test_env_golden(size, driver, steps, total):
    """G1: ParamTestEnv replays the reference's trajectories and the survey's totals."""
    z = np.load(os.path.join(GOLDEN, 'g1_param_test.npz'))
    key = '%s%d' % (driver.__name__, size)
    env = ParamTestEnv(size)
    assert np.array_equal(env.reset(), z[key + '_obs'][0])
    for k, a in enumerate(z[key + '_actions']):
        o, r, d, info = env.step(int(a))
        assert np.array_equal(o, z[key + '_obs'][k + 1]) and r == z[key + '_reward'][k] and d == z[key + '_done'][k]
    n, ret, acts = driver(size)
    assert n == steps and abs(ret - total) < 1e-9 and acts == z[key + '_actions'].tolist()
    with pytest.raises(IndexError):
        env.step(7)


@pytest.mark.parametrize('mode,size', [('grid', 22), ('direct', 12), ('simple', 14), ('grid', 16)])
def test_param_test_env_observation_modes(mode, size):
    """G1 for the other OBS_MODEs (pte:17-64): Grid10Observation, DirectObservation, NoObservation -- every observation,
    reward and done flag of the reference's run under the same actions."""
    z = np.load(os.path.join(GOLDEN, 'g1_param_test_modes.npz'))
    key = '%s_zigzag%d' % (mode, size)
    try:
        ParamTestEnv.change_obs_mode(mode, size)
        env = ParamTestEnv(size)
        obs = env.reset()
        assert obs.shape == z[key + '_obs'][0].shape == ParamTestEnv.observation_space.shape
        assert np.array_equal(obs, z[key + '_obs'][0])
        for k, a in enumerate(z[key + '_actions']):
            o, r, d, info = env.step(int(a))
            assert np.array_equal(o, z[key + '_obs'][k + 1]), k
            assert r == z[key + '_reward'][k] and d == z[key + '_done'][k]
        assert d
    finally:
        ParamTestEnv.change_obs_mode('section')


def test_capi_exports_every_declared_symbol():
    """The shared library loads without a GPU and exports exactly what include/paintrl.h declares."""
    header = open(os.path.join(REPO, 'include', 'paintrl.h')).read()
    declared = set(re.findall(r'\b(prl_[a-z_0-9]+)\s*\(', header))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.prl_abi_version() == _lib.ABI_VERSION == 3
    cfg = config.make_config(obs_mode='grid', obs_grad=4)
    assert lib.prl_obs_dim(cfg) == 16
    assert lib.prl_obs_dim(config.make_config(obs_mode='discrete')) == 5
    # error paths that need no device
    import ctypes as C
    out = C.c_void_p()
    assert lib.prl_batch_create(None, 0, None, 0, cfg, C.byref(out)) < 0
    assert b'bad arguments' in lib.prl_last_error()


def test_discrete_action_table_keeps_reference_rounding():
    d1, d2, ang = config.discrete_action_table(4)
    assert d1[0] == 0.051 and d2[2] != 0.0 and abs(d1[1]) < 1e-17 and d1[1] != 0.0     # cos(pi/2)*0.051 = 3.1e-18
    assert ang[0] == 0.0 and abs(ang[1] - np.pi / 2) < 1e-15
    o1, o2, oa = oracle.paint_oracle.discrete_action_table(4)
    assert np.array_equal(o1, d1) and np.array_equal(o2, d2) and np.array_equal(oa, ang)


@pytest.mark.parametrize('name', ['door_test', 'square'])
def test_device_layout_invariants(name):
    t = synthetic_tables(name)
    d = DeviceTables(t)
    P = t.sample_pos.shape[0]
    valid = d.perm >= 0
    assert sorted(d.perm[valid].tolist()) == list(range(P)) and d.n_samples_pad % 64 == 0
    assert np.array_equal(d.perm[d.inv_perm], np.arange(P))
    assert int(np.unpackbits(d.word_valid.view(np.uint8)).sum()) == P
    o1, o2, inv, nx, ny = d.sgrid
    assert (d.sgrid_start[::nx][:ny] % 64 == 0).all()          # every cell row starts on a word
    # the samples of a word ascend on axis a1, pads (far away) last: the observation's binary search and
    # prl_part_create's validation rely on it
    xw = d.sample_xyz[t.a1].reshape(d.n_words, 64)
    assert (np.diff(xw, axis=1) >= 0).all()
    vw = valid.reshape(d.n_words, 64)
    assert (vw[:, :-1] | ~vw[:, 1:]).all()                      # no real sample after a pad
    # every sample within the paint radius of a random centre lies in the 3x3 cell block the kernel scans
    rng = np.random.RandomState(0)
    xyz = np.stack(d.sample_xyz, axis=1)
    for _ in range(200):
        c = xyz[d.inv_perm[rng.randint(P)]] + rng.normal(0, 0.02, 3)
        near = np.nonzero(((xyz - c) ** 2).sum(1) <= 0.051 ** 2)[0]
        icx, icy = int(np.floor((c[t.a1] - o1) * inv)), int(np.floor((c[t.a2] - o2) * inv))
        got = []
        for cy in range(icy - 1, icy + 2):
            if 0 <= cy < ny:
                cx0, cx1 = max(icx - 1, 0), min(icx + 1, nx - 1)
                if cx0 <= cx1:
                    got.extend(range(d.sgrid_start[cy * nx + cx0], d.sgrid_start[cy * nx + cx1 + 1]))
        assert set(near.tolist()) <= set(got)
    # grid-observation masks partition the samples
    assert int(d.obs_cell_count.sum()) == P
    total = np.zeros(d.n_words, dtype=np.uint64)
    for m in d.obs_cell_mask:
        assert not (total & m).any()
        total |= m
    assert np.array_equal(total, d.word_valid)
    # adjacency is the reference's uv_map restricted to the side, file order
    assert d.vertex_adj.max() < d.tri_records.shape[0] and (d.vertex_adj >= -1).all()
    assert sorted(d.col_rank[d.col_rank != 0x7fffffff].tolist()) == list(range(d.n_collision))
    assert d.n_collision_pad % 64 == 0 and d.n_col_chunks <= 64
    assert CELL > 0.051
    words = rng.randint(0, 2 ** 63, size=(2, d.n_words)).astype(np.uint64) & d.word_valid
    back = d.mask_to_canonical(words)
    assert back.shape == (2, P) and back.sum() == np.unpackbits(words.view(np.uint8)).sum()


def test_numpy_ray_equals_oracle_ray(door_tables):
    t = door_tables
    rng = np.random.RandomState(4)
    n = 2000
    o = np.stack([rng.uniform(-0.2, 0.3, n), rng.uniform(-0.8, 0.6, n), rng.uniform(0.1, 1.3, n)], axis=1)
    e = o + np.stack([-rng.uniform(0.2, 1.0, n), rng.normal(0, 0.2, n), rng.normal(0, 0.2, n)], axis=1)
    idx, tt, pos = geometry.ray_closest_hit(t.col_v0, t.col_e1, t.col_e2, o, e)
    oi, ot, op = oracle.Oracle(t, 1).ray_batch(o, e)
    hit = idx >= 0
    assert hit.sum() > 300 and (~hit).sum() > 100
    assert np.array_equal(idx, oi) and np.array_equal(tt[hit], ot[hit]) and np.array_equal(pos[hit], op[hit])


def test_oracle_envs_are_independent(door_tables):
    """Sharding premise: an env's trajectory does not depend on its batch neighbours."""
    sp = part_tables.start_points(door_tables, 'all')
    rng = np.random.RandomState(1)
    n, steps = 16, 15
    start = rng.randint(0, len(sp), size=n)
    acts = rng.randint(0, 4, size=(steps, n))
    full = oracle.Oracle(door_tables, n, start_points=sp)
    half = oracle.Oracle(door_tables, n // 2, start_points=sp)
    full.reset(start)
    half.reset(start[n // 2:])
    for k in range(steps):
        of, rf, df, _ = full.step(acts[k])
        oh, rh, dh, _ = half.step(acts[k][n // 2:])
        assert np.array_equal(of[n // 2:], oh) and np.array_equal(rf[n // 2:], rh) and np.array_equal(df[n // 2:], dh)


def test_tables_round_trip_on_disk(tmp_path, door_tables):
    """The on-disk table format reproduces every field the device layout and the oracle consume."""
    from paintrl_amd import preprocess
    path = str(tmp_path / 'door.npz')
    part_tables.save_tables(door_tables, path)
    t2 = part_tables.load_tables(path)
    d1, d2 = DeviceTables(door_tables), DeviceTables(t2)
    for name in ('sample_xyz', 'vertex_xyz', 'col'):
        for a, b in zip(getattr(d1, name), getattr(d2, name)):
            assert np.array_equal(a, b)
    for name in ('word_bbox', 'word_valid', 'sgrid_start', 'vertex_adj', 'tri_records', 'col_bbox', 'col_rank',
                 'start_pos', 'start_quat', 'obs_cell_mask', 'beams', 'grid_lo', 'grid_hi'):
        assert np.array_equal(getattr(d1, name), getattr(d2, name)), name
    assert part_tables.start_points(t2, 'all') == part_tables.start_points(door_tables, 'all')
    assert part_tables.start_points(t2, 'edge') == part_tables.start_points(door_tables, 'edge')
    # the command-line tool writes the same file for the synthetic sheet
    out = str(tmp_path / 'sheet.npz')
    assert preprocess.main(['--synthetic', 'square', out]) == 0
    t3 = part_tables.load_tables(out)
    assert np.array_equal(t3.sample_pos, synthetic_tables('square').sample_pos)


def test_stale_kd_tree_walk_equals_scipy_on_moved_rows():
    """Parts on which the reference moves vertex rows under its cKDTree (bpw:943-946; the synthetic coarse sheet):
    part_tables.stale_kd_query and the oracle's C statement of it give what scipy's query gives on a tree built from
    the old rows whose data was then overwritten in place -- and that is NOT always the exact nearest vertex."""
    from scipy.spatial import cKDTree
    import ctypes as C
    t = synthetic_tables('test')
    assert len(t.vertices_mutated) > 0
    side = t._side_data
    # the rows as they were when the tree was built: moved rows back to their vertex positions
    old = side.copy()
    old[t.vertices_mutated] = t.vertices[t.vertices_mutated]
    tree = cKDTree(old)
    tree.data[:] = side                                  # what bpw:943-946 does to vertices_kd_tree[side].data
    rng = np.random.RandomState(0)
    qs = t.sample_pos[rng.randint(0, len(t.sample_pos), 4000)] + rng.normal(0, 0.01, (4000, 3))
    want = tree.query(qs, k=1)[1]
    got = np.array([part_tables.stale_kd_query(t, q) for q in qs])
    assert np.array_equal(got, want)
    exact = np.argmin(((side[None, :, :] - qs[:, None, :]) ** 2).sum(-1), axis=1)
    assert 0 < (exact != want).sum() < 400               # the stale tree does change answers, rarely
    # the same walk in the oracle's C (hook point of each query: its normal identifies the vertex's triangle choice)
    orc = oracle.Oracle(t, 1)
    side_ids = np.nonzero(t.vertex_is_side)[0]
    lib = orc.lib
    lib.or_nearest_vertex.restype = C.c_int
    for q, w in zip(qs[:1500], want[:1500]):
        qq = np.ascontiguousarray(q, dtype=np.float64)
        v = lib.or_nearest_vertex(C.byref(orc.part), qq.ctypes.data_as(C.POINTER(C.c_double)))
        assert side_ids[v] == w


def test_beta_plain_is_the_reference_table():
    """part_tables.beta_plain restates rob:38-69: with the random stream in the state the reference's constructor has it
    in (six draws after random.seed(4242)) it returns the recorded table bit for bit."""
    import random
    from paintrl_amd import part_tables
    from conftest import load_episodes, synthetic_tables
    ep = load_episodes('door_hsi_cone')['g15_hsi_cone_random']
    tables = synthetic_tables('door_test')
    random.seed(4242)
    for _ in range(6):
        random.random()
    assert np.array_equal(part_tables.beta_plain(tables.density), ep['beams'])


@pytest.mark.parametrize('tag,name,part,after_reset', [('sheet', 'g2_zigzag', 'square', 'sheet_after_reset'),
                                                       ('door', 'g3_serpentine', 'door_test', 'door_after_reset'),
                                                       ('door_hsi', 'g13_hsi_serpentine', 'door_test', 'door_hsi_after_reset')])
def test_texture_image_equals_the_reference(tag, name, part, after_reset):
    """part_tables.texture_image = Part.get_texture_image() of the reference (bpw:737-738 after _label_part bpw:579-592):
    the images recorded from the imported reference after a reset and at the end of three committed episodes
    (tests/golden/textures.npz, make_golden.py --textures), reproduced from the oracle's painted bits / thickness bytes --
    irrelevant texels black, the back side (0, 255, 0), the front grey / red (the thickness byte in 'HSI'), and the
    get_texel clamp's corner texel left as the texture file had it."""
    import oracle
    from conftest import GOLDEN, env_kwargs_from_cfg, load_episodes, synthetic_tables
    from paintrl_amd import part_tables
    want = np.load(os.path.join(GOLDEN, 'textures.npz'))
    t = synthetic_tables(part)
    ep = load_episodes(tag)[name]
    cfg = ep['cfg']
    mode = cfg.get('color_mode', 'RGB')
    P = t.sample_pix.shape[0]
    fresh = part_tables.texture_image(t, painted=np.zeros(P, dtype=bool), thickness=np.full(P, 255, dtype=np.uint8), color_mode=mode)
    assert fresh.dtype == np.uint8 and np.array_equal(fresh, want[after_reset])
    back = (fresh[..., 0] == 0) & (fresh[..., 1] == 255) & (fresh[..., 2] == 0)
    assert back.sum() > 0.9 * t.back_pix.shape[0]                    # (texels in both profiles carry the front label)
    orc = oracle.Oracle(t, 1, start_points=part_tables.start_points(t, cfg['start_mode']), color_mode=mode,
                        **env_kwargs_from_cfg(cfg))
    orc.reset([int(ep['start_idx'])])
    for a in ep['actions']:
        orc.step([a])
    img = (part_tables.texture_image(t, thickness=orc.thick[0], color_mode='HSI') if mode == 'HSI'
           else part_tables.texture_image(t, painted=orc.painted_bits(0)))
    assert np.array_equal(img, want[name])
    import hashlib
    assert hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest() == str(want[name + '_sha256'])


def test_doubled_vertices_resolve_as_the_reference_tree_does():
    """Vertices written twice in the OBJ (UV seams) are equally near to every query; cKDTree.query (bpw:526) returns the one its
    own index array lists first.  part_tables._vertex_tie_rank restates that order with scipy's own tree: on the synthetic
    seam sheet every doubled vertex queried at its own position resolves as scipy resolves it, and not always to the lower
    index (the rule this project used before: the reference's episodes across the seam did not replay with it)."""
    from scipy.spatial import cKDTree
    t = synthetic_tables('door_lf')
    rows = t._side_data
    tree = cKDTree(rows)
    side = np.nonzero(t.vertex_is_side)[0]
    _, first, counts = np.unique(rows[side], axis=0, return_index=True, return_counts=True)
    doubled = side[first[counts > 1]]
    assert len(doubled) >= 40
    lower_wins = 0
    for v in doubled:
        want = int(tree.query(rows[v], k=1)[1])
        assert part_tables.nearest_side_vertex(t, rows[v]) == want
        twins = np.nonzero((rows == rows[v]).all(axis=1))[0]
        lower_wins += int(want == twins.min())
    assert 0 < lower_wins < len(doubled)                 # neither "lowest index" nor "highest index" is the rule
