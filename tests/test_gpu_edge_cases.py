"""GPU edge cases of the C ABI: odd batch sizes, masked reset, error codes, RNG reset, read-back."""
import ctypes as C

import numpy as np
import pytest

import oracle
from conftest import start_points_for, synthetic_tables

pytestmark = pytest.mark.gpu


def _env(tables, n, sp=None, **kw):
    from paintrl_amd.batched_env import BatchedPaintEnv
    from paintrl_amd.device_tables import DeviceTables
    return BatchedPaintEnv(DeviceTables(tables, obs_grad=kw.get('obs_grad', 4), start_points=sp), n, **kw)


@pytest.mark.parametrize('n', [1, 2, 3, 5, 7, 65])
def test_batch_sizes_not_multiple_of_workgroup(n):
    """4 envs share a workgroup; the last workgroup may be partly empty."""
    tables = synthetic_tables('door_test')
    env = _env(tables, n)
    orc = oracle.Oracle(tables, n)
    start = np.arange(n) % 4
    assert np.array_equal(env.reset(start_idx=start).cpu().numpy(), orc.reset(start))
    rng = np.random.RandomState(n)
    for _ in range(6):
        a = rng.randint(0, 4, size=n)
        o, r, d, i = env.step(a)
        oo, rr, dd, ii = orc.step(a)
        assert np.array_equal(o.cpu().numpy(), oo) and np.array_equal(r.cpu().numpy(), rr)
        assert np.array_equal(d.cpu().numpy(), dd) and np.array_equal(i.cpu().numpy(), ii)
    env.close()


def test_masked_reset_leaves_other_envs_untouched():
    tables = synthetic_tables('door_test')
    n = 16
    env = _env(tables, n)
    env.reset(start_idx=np.zeros(n, dtype=np.int32))
    for _ in range(3):
        env.step(np.ones(n, dtype=np.int32))
    before = env.painted_words().cpu().numpy().copy()
    st0 = env.state()
    mask = np.zeros(n, dtype=bool)
    mask[[1, 5, 6]] = True
    obs_before = env.obs.cpu().numpy().copy()
    obs = env.reset(mask=mask, start_idx=np.full(n, 2, dtype=np.int32)).cpu().numpy()
    after = env.painted_words().cpu().numpy()
    st1 = env.state()
    assert (after[mask] == 0).all() and np.array_equal(after[~mask], before[~mask])
    assert np.array_equal(obs[~mask], obs_before[~mask])              # rows of envs not reset are untouched
    assert (st1['step_counter'][mask] == 0).all() and np.array_equal(st1['step_counter'][~mask], st0['step_counter'][~mask])
    assert np.array_equal(st1['pose'][mask], np.tile(np.array(tables.anchor_points[2][0]), (3, 1)))
    assert (st1['episode'][mask] == st0['episode'][mask] + 1).all()
    env.close()


def test_library_rng_start_points_are_valid_and_seeded():
    tables = synthetic_tables('door_test')
    sp = start_points_for(tables, 'all')
    pos = np.array([p[0] for p in sp])
    a = _env(tables, 512, sp, seed=7)
    b = _env(tables, 512, sp, seed=7)
    c = _env(tables, 512, sp, seed=8)
    a.reset(), b.reset(), c.reset()
    pa, pb, pc = a.state()['pose'], b.state()['pose'], c.state()['pose']
    assert np.array_equal(pa, pb) and not np.array_equal(pa, pc)
    # every drawn pose is one of the table's start points, and many different ones are used
    d = np.abs(pa[:, None, :] - pos[None, :, :]).sum(-1).min(1)
    assert (d == 0).all() and len({tuple(p) for p in pa}) > 300
    for e in (a, b, c):
        e.close()


def test_error_codes_and_messages():
    from paintrl_amd import _lib, config
    from paintrl_amd.batched_env import BatchedPaintEnv
    from paintrl_amd.device_tables import DeviceTables
    tables = synthetic_tables('door_test')
    dt = DeviceTables(tables)
    with pytest.raises(_lib.PaintRLError, match='observation cells'):
        BatchedPaintEnv(dt, 4, obs_mode='grid', obs_grad=5)           # part packed for 4x4 cells
    with pytest.raises(_lib.PaintRLError, match='OBS_GRAD'):
        BatchedPaintEnv(dt, 4, obs_mode='section', obs_grad=99)
    with pytest.raises(_lib.PaintRLError, match='action_dim'):
        BatchedPaintEnv(dt, 4, action_mode='continuous', action_dim=3)
    with pytest.raises(_lib.PaintRLError, match='env_part_id'):
        BatchedPaintEnv(dt, 4, env_part_id=[0, 0, 1, 0])
    lib = _lib.load()
    st = dt.c_struct()
    st.n_samples_pad = dt.n_samples_pad + 1                           # not a multiple of 64
    h = C.c_void_p()
    assert lib.prl_part_create(C.byref(st), 0, C.byref(h)) == -1 and b'sample counts' in lib.prl_last_error()
    st = dt.c_struct()
    st.adj_width = 0
    assert lib.prl_part_create(C.byref(st), 0, C.byref(h)) == -3
    st = dt.c_struct()
    st.n_samples, st.n_samples_pad = 64 * 1600 + 64, 64 * 1600 + 64      # beyond what the LDS-resident masks hold
    assert lib.prl_part_create(C.byref(st), 0, C.byref(h)) == -3 and b'at most' in lib.prl_last_error()
    bad = dt.sgrid_start.copy()
    bad[3] = bad[2] - 1                                                # not monotone
    st = dt.c_struct()
    st.sgrid_start = bad.ctypes.data_as(_lib._ip)
    assert lib.prl_part_create(C.byref(st), 0, C.byref(h)) == -1 and b'non-decreasing' in lib.prl_last_error()
    shuffled = dt.sample_xyz[tables.a1].copy()                        # a word whose samples do not ascend on a1
    shuffled[[0, 1]] = shuffled[[1, 0]] if shuffled[0] != shuffled[1] else (shuffled[1] + 1.0, shuffled[1])
    st = dt.c_struct()
    st.sample_xyz[tables.a1] = shuffled.ctypes.data_as(_lib._dp)
    assert lib.prl_part_create(C.byref(st), 0, C.byref(h)) == -1 and b'do not ascend' in lib.prl_last_error()
    assert lib.prl_batch_step(None, None, None, None, None, None, None, None, None) == -1
    assert config.make_config().auto_reset == 0


def test_episode_statistics_and_returns_payload():
    """On done the kernel records return / length / coverage of the finished episode (the gather payload)."""
    tables = synthetic_tables('door_test')
    n = 64
    env = _env(tables, n, auto_reset=True, seed=3)
    orc = oracle.Oracle(tables, n)
    start = np.arange(n) % 4
    env.reset(start_idx=start)
    orc.reset(start)
    rng = np.random.RandomState(0)
    seen = np.zeros(n, dtype=bool)
    want_ret, want_len, want_cov = np.zeros(n), np.zeros(n, dtype=int), np.zeros(n, dtype=int)
    for k in range(40):
        a = rng.randint(0, 4, size=n)
        nxt = rng.randint(0, 4, size=n)
        _, _, d, _ = env.step(a, start_idx=nxt)
        _, _, dd, _ = orc.step(a)
        d = d.cpu().numpy()
        assert np.array_equal(d, dd)
        for e in np.nonzero(dd)[0]:
            st = orc.state(e)
            want_ret[e], want_len[e], want_cov[e] = st['total_return'], st['step_counter'], orc.painted_bits(e).sum()
            seen[e] = True
        if dd.any():
            orc.reset(nxt, mask=dd)
    st = env.state()
    ret = env.episode_returns().cpu().numpy()
    assert seen.sum() > 20
    assert np.array_equal(ret[seen], want_ret[seen]) and np.array_equal(st['last_episode_len'][seen], want_len[seen])
    assert np.array_equal(st['last_episode_painted'][seen], want_cov[seen])
    env.close()


def test_full_size_round_trip_properties():
    """BASELINE size (4096 envs): size-independent properties instead of an oracle replay --
    coverage only grows within an episode, reward*100 equals the coverage delta, info = (reward, penalty)."""
    tables = synthetic_tables('door_test')
    n = 4096
    env = _env(tables, n)
    start = np.arange(n) % 4
    env.reset(start_idx=start)
    rng = np.random.RandomState(1)
    cov = np.zeros(n, dtype=np.int64)
    for k in range(12):
        a = rng.randint(0, 4, size=n)
        o, r, d, i = env.step(a)
        words = env.painted_words().cpu().numpy().view(np.uint64)
        new_cov = np.unpackbits(words.view(np.uint8), axis=1).sum(1)
        i = i.cpu().numpy()
        assert (new_cov >= cov).all()
        assert np.array_equal(np.rint(i[:, 0] * 100).astype(np.int64), new_cov - cov)
        assert np.array_equal(r.cpu().numpy(), i[:, 0] - i[:, 1]) and (i[:, 1] == 0.2).all()
        o = o.cpu().numpy()
        assert (o >= 0).all() and (o <= 1).all()
        cov = new_cov
    # two envs with the same start and actions are identical (determinism, no cross-env coupling)
    env2 = _env(tables, n)
    env2.reset(start_idx=start)
    rng = np.random.RandomState(1)
    for k in range(12):
        env2.step(rng.randint(0, 4, size=n))
    assert np.array_equal(env.painted_words().cpu().numpy(), env2.painted_words().cpu().numpy())
    env.close()
    env2.close()


def test_config4_total_size_in_one_launch():
    """32 768 envs (BASELINE config 4's total) as ONE launch on one GPU -- the four-envs-per-workgroup shape
    prl_batch_step picks beyond 16 waves per CU: the size-independent properties at full size, a 4 096-env slice
    against the oracle on every output and painted bit, and the rest of the batch against that slice (env e and
    env e mod 4 096 share start and actions)."""
    tables = synthetic_tables('door_test')
    sp = start_points_for(tables, 'all')
    n, m, steps = 32768, 4096, 4
    env = _env(tables, n, sp)
    orc = oracle.Oracle(tables, m, start_points=sp, threads=8)
    rng = np.random.RandomState(99)
    start = rng.randint(0, len(sp), size=m)
    obs = env.reset(start_idx=np.tile(start, n // m)).cpu().numpy()
    assert np.array_equal(obs[:m], orc.reset(start))
    cov = np.zeros(n, dtype=np.int64)
    for k in range(steps):
        a = rng.randint(0, 4, size=m)
        o, r, d, i = env.step(np.tile(a, n // m))
        o, r, d, i = o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy(), i.cpu().numpy()
        oo, rr, dd, ii = orc.step(a)
        assert np.array_equal(o[:m], oo) and np.array_equal(r[:m], rr), 'step %d' % k
        assert np.array_equal(d[:m], dd) and np.array_equal(i[:m], ii), 'step %d' % k
        for blk in range(1, n // m):                       # every other 4 096-env block repeats the first
            sl = slice(blk * m, (blk + 1) * m)
            assert np.array_equal(o[sl], o[:m]) and np.array_equal(r[sl], r[:m]) and np.array_equal(d[sl], d[:m])
        words = env.painted_words().cpu().numpy().view(np.uint64)
        new_cov = np.unpackbits(words.view(np.uint8), axis=1).sum(1)
        assert (new_cov >= cov).all()
        assert np.array_equal(np.rint(i[:, 0] * 100).astype(np.int64), new_cov - cov)
        assert np.array_equal(r, i[:, 0] - i[:, 1]) and (i[:, 1] == 0.2).all()
        cov = new_cov
    words = env.painted_words().cpu().numpy().view(np.uint64)
    bits = env.parts[0].mask_to_canonical(words[:m])
    assert np.array_equal(bits, np.stack([orc.painted_bits(e) for e in range(m)]))
    assert np.array_equal(words.reshape(n // m, m, -1), np.broadcast_to(words[:m], (n // m, m, words.shape[1])))
    env.close()


def test_vector_env_names():
    """RLlib VectorEnv-style entry points: vector_reset / reset_at / vector_step against the oracle."""
    tables = synthetic_tables('door_test')
    sp = start_points_for(tables, 'anchor')
    n = 37
    env = _env(tables, n, sp)
    orc = oracle.Oracle(tables, n, start_points=sp)
    obs = env.vector_reset().cpu().numpy().copy()
    cand = [orc.reset(np.full(n, s, dtype=np.int32))[0] for s in range(len(sp))]
    start = np.array([next(s for s in range(len(sp)) if np.array_equal(cand[s], obs[e])) for e in range(n)])
    assert np.array_equal(orc.reset(start), obs)          # the library's own draws, identified by their observations
    rng = np.random.RandomState(3)
    for k in range(6):
        a = rng.randint(0, 4, size=n)
        o, r, d, i = env.vector_step(a)
        oo, rr, dd, ii = orc.step(a)
        assert np.array_equal(o.cpu().numpy(), oo) and np.array_equal(r.cpu().numpy(), rr)
        assert np.array_equal(d.cpu().numpy(), dd) and np.array_equal(i.cpu().numpy(), ii)
    # reset_at: one env back to a chosen start point, every other env untouched
    before = env.painted_words().cpu().numpy().copy()
    st_before = env.state()
    row = env.reset_at(5, start_idx=2).cpu().numpy()
    mask = np.zeros(n, dtype=bool)
    mask[5] = True
    want = orc.reset(np.full(n, 2, dtype=np.int32), mask=mask)
    assert np.array_equal(row, want[5])
    after = env.painted_words().cpu().numpy()
    assert not after[5].any() and np.array_equal(np.delete(after, 5, 0), np.delete(before, 5, 0))
    st = env.state()
    assert st['step_counter'][5] == 0 and np.array_equal(np.delete(st['step_counter'], 5), np.delete(st_before['step_counter'], 5))
    for k in range(4):
        a = rng.randint(0, 4, size=n)
        o, r, d, i = env.vector_step(a)
        oo, rr, dd, ii = orc.step(a)
        assert np.array_equal(o.cpu().numpy(), oo) and np.array_equal(r.cpu().numpy(), rr) and np.array_equal(d.cpu().numpy(), dd)
    env.close()


def test_full_size_batch_equals_oracle():
    """The BASELINE batch (4096 envs, 'all' start points) against the oracle for a few steps: every
    observation, reward, done flag and painted bit."""
    tables = synthetic_tables('door_test')
    sp = start_points_for(tables, 'all')
    n, steps = 4096, 4
    env = _env(tables, n, sp)
    orc = oracle.Oracle(tables, n, start_points=sp, threads=8)
    rng = np.random.RandomState(77)
    start = rng.randint(0, len(sp), size=n)
    assert np.array_equal(env.reset(start_idx=start).cpu().numpy(), orc.reset(start))
    for k in range(steps):
        a = rng.randint(0, 4, size=n)
        o, r, d, i = env.step(a)
        oo, rr, dd, ii = orc.step(a)
        assert np.array_equal(o.cpu().numpy(), oo) and np.array_equal(r.cpu().numpy(), rr), 'step %d' % k
        assert np.array_equal(d.cpu().numpy(), dd) and np.array_equal(i.cpu().numpy(), ii)
    words = env.painted_words().cpu().numpy().view(np.uint64)
    bits = env.parts[0].mask_to_canonical(words)
    want = np.stack([orc.painted_bits(e) for e in range(n)])
    assert np.array_equal(bits, want)
    env.close()


def test_hostile_continuous_actions_do_not_disturb_other_envs():
    """NaN / inf / huge continuous actions in some envs: the launch completes, and every env that was
    fed finite actions still matches the oracle exactly in done flags and painted sets (observations
    and rewards to 1e-9: continuous actions go through libm on both sides)."""
    tables = synthetic_tables('door_test')
    sp = start_points_for(tables, 'all')
    n, steps = 128, 15
    kw = dict(action_mode='continuous', action_dim=2, obs_mode='section')
    env = _env(tables, n, sp, **kw)
    orc = oracle.Oracle(tables, n, start_points=sp, threads=8, **kw)
    rng = np.random.RandomState(13)
    start = rng.randint(0, len(sp), size=n)
    env.reset(start_idx=start)
    orc.reset(start)
    bad = np.zeros(n, dtype=bool)
    bad[::7] = True
    alive = ~bad
    for k in range(steps):
        a = rng.uniform(-1.5, 1.5, size=(n, 2))
        a[bad, 0] = [np.nan, np.inf, -np.inf, 1e300][k % 4]
        o, r, d, i = env.step(a)
        oo, rr, dd, ii = orc.step(a)
        o, r, d = o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy()
        np.testing.assert_allclose(o[alive], oo[alive], rtol=0, atol=1e-9)
        np.testing.assert_allclose(r[alive], rr[alive], rtol=0, atol=1e-9)
        assert np.array_equal(d[alive], dd[alive])
    words = env.painted_words().cpu().numpy().view(np.uint64)
    bits = env.parts[0].mask_to_canonical(words)
    for e in np.nonzero(alive)[0]:
        assert np.array_equal(bits[e], orc.painted_bits(e))
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize('obs_mode', ['section', 'discrete', 'grid'])
def test_observation_at_exact_sample_coordinates(obs_mode):
    """Tools parked exactly on sample coordinates after some painting: whole texel columns / rows then
    compare EQUAL to the tool position (the 4-sector rule sends them to sector 3 and skips the sample the
    tool sits on, bpw:1034-1061) -- the case the in-lane binary search resolves with its equal-run table."""
    tables = synthetic_tables('door_test')
    n = 96
    env = _env(tables, n, obs_mode=obs_mode)
    orc = oracle.Oracle(tables, n, obs_mode=obs_mode)
    start = np.arange(n) % 4
    env.reset(start_idx=start)
    orc.reset(start)
    rng = np.random.RandomState(11)
    for _ in range(12):
        a = rng.randint(0, 4, size=n)
        env.step(a)
        orc.step(a)
    pos = tables.sample_pos
    a0 = [k for k in range(3) if k not in (tables.a1, tables.a2)][0]
    orn = [0.0, 0.0, 0.0]
    orn[a0] = -1.0
    pick = rng.randint(0, pos.shape[0], size=(n, 2))
    for i in range(n):
        p = pos[pick[i, 0]].copy()
        kind = i % 4
        if kind == 1:
            p[tables.a2] = pos[pick[i, 1], tables.a2]          # x of one sample, y of another
        elif kind == 2:
            p[tables.a1] += 1e-9                               # just off the column
        elif kind == 3:
            p[tables.a1] = pos[:, tables.a1].min() - 0.2       # left of everything, y exact
        env.set_pose(i, p, orn)
        orc.set_pose(i, p, orn)
    got, want = env.observe().cpu().numpy(), orc.observe()
    assert np.array_equal(got, want)
    assert np.array_equal(got, env.observe().cpu().numpy())   # observing changes nothing
    a = rng.randint(0, 4, size=n)                               # and the envs step on identically
    o1, r1, d1, _ = env.step(a)
    o2, r2, d2, _ = orc.step(a)
    assert np.array_equal(o1.cpu().numpy(), o2) and np.array_equal(r1.cpu().numpy(), r2)
    env.close()


@pytest.mark.parametrize('tex,kw_slots,obs_mode', [(120, 1, 'section'), (160, 2, 'section'), (120, 1, 'grid'), (160, 2, 'discrete')])
def test_small_textures_use_the_one_and_two_slot_kernels(tex, kw_slots, obs_mode):
    """The step kernel is instantiated per number of 64-word mask slots per lane (1..4); the standard door and
    sheet use 3 and 4.  A 120x120 / 160x160 texture of the same door gives 47 / 76 words: the 1- and 2-slot
    instantiations, checked against the oracle like the others (obs, reward, done, painted set, pose)."""
    from paintrl_amd import part_tables, synth_parts
    from paintrl_amd.batched_env import BatchedPaintEnv
    from paintrl_amd.device_tables import DeviceTables
    tables = part_tables.build_part_tables(mesh=synth_parts.synthetic_mesh('door_test'), tex_size=(tex, tex), name='door_test')
    dt = DeviceTables(tables)
    assert (dt.n_words + 63) // 64 == kw_slots
    n = 128
    env = BatchedPaintEnv(dt, n, obs_mode=obs_mode, overlap_penalty=True)
    orc = oracle.Oracle(tables, n, obs_mode=obs_mode, overlap_penalty=True)
    start = np.arange(n) % 4
    assert np.array_equal(env.reset(start_idx=start).cpu().numpy(), orc.reset(start))
    rng = np.random.RandomState(tex)
    for k in range(30):
        a = rng.randint(0, 4, size=n)
        o, r, d, i = env.step(a)
        oo, rr, dd, ii = orc.step(a)
        assert np.array_equal(o.cpu().numpy(), oo) and np.array_equal(r.cpu().numpy(), rr), 'step %d' % k
        assert np.array_equal(d.cpu().numpy(), dd) and np.array_equal(i.cpu().numpy(), ii), 'step %d' % k
    bits = dt.mask_to_canonical(env.painted_words().cpu().numpy().view(np.uint64))
    st = env.state()
    for e in range(n):
        assert np.array_equal(bits[e], orc.painted_bits(e))
        assert np.array_equal(st['pose'][e], orc.state(e)['pose'])
    env.close()


def test_cone_beam_step_on_a_side_stream_and_in_a_graph():
    """PAINT_METHOD 'normal' is five launches on the caller's stream handing work lists to each other (k_cone_beams.hip): a
    step issued on a non-default stream, and one captured into a graph and replayed, give what direct stepping gives."""
    import torch
    tables = synthetic_tables('door_test')
    n, steps = 96, 6
    rng = np.random.RandomState(77)
    acts = torch.from_numpy(rng.randint(0, 4, size=(steps, n)).astype(np.int32)).cuda()
    start = np.arange(n) % 4

    def run(mode):
        env = _env(tables, n, paint_method='normal', auto_reset=True, seed=11)
        env.reset(start_idx=start)
        out = []
        if mode == 'direct':
            for k in range(steps):
                env.step_raw(acts[k])
                out.append((env.obs.clone(), env.reward.clone(), env.done_u8.clone()))
        elif mode == 'stream':
            torch.cuda.synchronize()
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for k in range(steps):
                    env.step_raw(acts[k])
                    out.append((env.obs.clone(), env.reward.clone(), env.done_u8.clone()))
            s.synchronize()
        else:
            cur = torch.zeros(n, dtype=torch.int32, device='cuda')
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()                  # (a step allocates nothing: no warm-up needed)
            with torch.cuda.graph(g):
                env.step_raw(cur)
            for k in range(steps):
                cur.copy_(acts[k])
                g.replay()
                out.append((env.obs.clone(), env.reward.clone(), env.done_u8.clone()))
        torch.cuda.synchronize()
        env.close()
        return out

    ref = run('direct')
    for mode in ('stream', 'graph'):
        got = run(mode)
        for k in range(steps):
            for x, y in zip(ref[k], got[k]):
                assert torch.equal(x, y), (mode, k)


def test_cone_beams_do_not_depend_on_the_batch_they_run_in():
    """PAINT_METHOD 'normal' at 8 192 envs in one batch == the same envs as two batches of 4 096 (the work lists, their
    sub-lists and the kernels that empty them are shared by the envs of a batch; the envs themselves are independent)."""
    import torch
    tables = synthetic_tables('door_test')
    sp = start_points_for(tables, 'all')
    n, steps = 8192, 4
    rng = np.random.RandomState(404)
    start = rng.randint(0, len(sp), size=n)
    acts = rng.randint(0, 4, size=(steps, n))
    big = _env(tables, n, sp, paint_method='normal')
    halves = [_env(tables, n // 2, sp, paint_method='normal') for _ in range(2)]
    o = big.reset(start_idx=start)
    oh = [h.reset(start_idx=start[k * (n // 2):(k + 1) * (n // 2)]) for k, h in enumerate(halves)]
    assert torch.equal(o, torch.cat(oh))
    for t in range(steps):
        o, r, d, i = big.step(acts[t])
        parts = [h.step(acts[t][k * (n // 2):(k + 1) * (n // 2)]) for k, h in enumerate(halves)]
        assert torch.equal(o, torch.cat([p[0] for p in parts])), 'obs, step %d' % t
        assert torch.equal(r, torch.cat([p[1] for p in parts])) and torch.equal(d, torch.cat([p[2] for p in parts]))
    assert torch.equal(big.painted_words(), torch.cat([h.painted_words() for h in halves]))
    big.close()
    for h in halves:
        h.close()


@pytest.mark.parametrize('part,tex,grad', [('door_test', 0, 2), ('door_test', 0, 3), ('door_test', 0, 5), ('door_test', 0, 7),
                                          ('door_test', 0, 8), ('door_test', 0, 12), ('door_test', 0, 20), ('square', 0, 9),
                                          ('door_rr_big', 320, 5), ('door_rr_big', 320, 12), ('door_rr_big', 320, 24)])
def test_atan2_sectors_of_every_width(part, tex, grad):
    """OBS_GRAD != 4 (bpw:1045-1061): a sample's sector is int(angle // (2 pi / g)) of its float64 atan2 angle about the tool.
    The device counts a word whose box lies inside one wedge by popcount and lets a float atan2f decide the other samples
    wherever it safely can (prl_observe.hpp section_general_wave); per-lane packed counters up to 8 sectors, a ballot a
    sector up to 16, LDS atomics beyond -- every width against the oracle, tools also parked exactly on sample coordinates
    (angles of exactly 0, pi / 2, pi: the ends of sectors for even g)."""
    tables = synthetic_tables(part, tex_size=(tex, tex)) if tex else synthetic_tables(part)
    sp = start_points_for(tables, 'all')
    n, steps = 96, 10
    mpp = int(0.95 * tables.sample_pos.shape[0])
    kw = dict(obs_mode='section', obs_grad=grad, max_possible_point=mpp)
    env = _env(tables, n, sp, **kw)
    orc = oracle.Oracle(tables, n, start_points=sp, threads=8, **kw)
    rng = np.random.RandomState(100 + grad)
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
    pos = tables.sample_pos
    a0 = [k for k in range(3) if k not in (tables.a1, tables.a2)][0]
    orn = [0.0, 0.0, 0.0]
    orn[a0] = -1.0
    pick = rng.randint(0, pos.shape[0], size=(n, 2))
    for e in range(n):                                      # on a sample; its a1 with another's a2; a hair off
        p = pos[pick[e, 0]].copy()
        if e % 3 == 1:
            p[tables.a2] = pos[pick[e, 1], tables.a2]
        elif e % 3 == 2:
            p[tables.a1] += 1e-12
        env.set_pose(e, p, orn)
        orc.set_pose(e, p, orn)
    assert np.array_equal(env.observe().cpu().numpy(), orc.observe())
    env.close()
