"""GPU: the rollout driver (policy MLP + trajectory buffer + PPO step) around the batched env."""
import numpy as np
import pytest

from conftest import start_points_for, synthetic_tables

pytestmark = pytest.mark.gpu


def test_rollout_fragment_and_ppo_step():
    import torch
    from paintrl_amd.batched_env import BatchedPaintEnv
    from paintrl_amd.device_tables import DeviceTables
    from paintrl_amd.rollout import MLPPolicy, RolloutWorker, gae, ppo_update
    tables = synthetic_tables('door_test')
    n, T = 256, 20
    env = BatchedPaintEnv(DeviceTables(tables, start_points=start_points_for(tables, 'anchor')), n, auto_reset=True,
                          seed=11)
    torch.manual_seed(0)
    policy = MLPPolicy(env.obs_dim, 4).to(env.device)
    worker = RolloutWorker(env, policy, fragment=T, seed=1)
    batch, last_value, returns = worker.collect()
    assert batch['obs'].shape == (T, n, 6) and batch['actions'].dtype == torch.int32
    assert returns.shape == (n,) and last_value.shape == (n,)
    # bookkeeping: reward = info.reward - info.penalty; obs[t+1] == new_obs[t] unless the env was reset
    r = batch['rewards'].cpu().numpy()
    assert np.allclose(r, (batch['infos_reward'] - batch['infos_penalty']).cpu().numpy(), atol=1e-6)
    same = (~batch['dones'][:-1]).cpu().numpy()
    assert np.array_equal(batch['new_obs'][:-1].cpu().numpy()[same], batch['obs'][1:].cpu().numpy()[same])
    assert batch['dones'].any() and (batch['actions'] >= 0).all() and (batch['actions'] < 4).all()
    # logp recorded at collection time equals a fresh forward pass
    logits, value = policy(batch['obs'].reshape(-1, 6))
    logp = torch.log_softmax(logits, -1).gather(-1, batch['actions'].reshape(-1, 1).long()).squeeze(-1)
    assert torch.allclose(logp, batch['action_logp'].reshape(-1), atol=1e-5)
    adv, targets = gae(batch, last_value)
    assert torch.isfinite(adv).all() and adv.shape == (T, n)
    opt = torch.optim.Adam(policy.parameters(), lr=3e-4)
    before = [p.detach().clone() for p in policy.parameters()]
    loss = ppo_update(policy, opt, batch, last_value, epochs=1, minibatches=2)
    assert np.isfinite(loss) and any(not torch.equal(a, b) for a, b in zip(before, policy.parameters()))
    env.close()


@pytest.mark.parametrize('mode,n,T', [('two_launches', 96, 25), ('persistent', 4096, 24), ('act_step', 4096, 20)])
def test_rollout_buffers_replay_through_the_oracle(mode, n, T):
    """The kernels write straight into rows of the worker's trajectory buffers; replaying the recorded actions
    through the CPU oracle must give the recorded observations, rewards, done flags and info, row by row, until
    an env's first episode end (after it the env restarts from a start point drawn by the library's RNG).
    'persistent' / 'act_step' at 4096 envs: BASELINE config 4's per-GPU share -- the policy-driven fragment as ONE
    launch (prl_rollout_fragment with weights) and as one launch per step (prl_batch_act_step)."""
    import torch
    import oracle
    from conftest import synthetic_tables
    from paintrl_amd.batched_env import BatchedPaintEnv
    from paintrl_amd.device_tables import DeviceTables
    from paintrl_amd.rollout import MLPPolicy, RolloutWorker
    tables = synthetic_tables('door_test')
    env = BatchedPaintEnv(DeviceTables(tables), n, auto_reset=True, seed=11)
    torch.manual_seed(3)
    policy = MLPPolicy(env.obs_dim, 4).to(env.device)
    worker = RolloutWorker(env, policy, fragment=T, seed=5, persistent=mode == 'persistent', act_step=mode == 'act_step')
    start_obs = worker.raw['obs'][0].cpu().numpy().copy()
    batch, _, _ = worker.collect()
    torch.cuda.synchronize()
    b = {k: v.cpu().numpy() for k, v in batch.items()}
    raw = {k: v.cpu().numpy() for k, v in worker.raw.items()}
    raw['obs'] = np.concatenate([start_obs[None], raw['obs'][1:]])     # row 0 already holds the next fragment's start
    # which start point did the library draw for each env?  (reset observations identify it)
    orc = oracle.Oracle(tables, n, threads=8)
    n_start = orc.n_start
    cand = [orc.reset(np.full(n, s, dtype=np.int32))[0] for s in range(n_start)]
    start = np.array([next(s for s in range(n_start) if np.array_equal(cand[s], start_obs[e])) for e in range(n)])
    assert np.array_equal(orc.reset(start), start_obs)
    alive = np.ones(n, dtype=bool)
    checked = 0
    for t in range(T):
        assert np.array_equal(b['obs'][t][alive], raw['obs'][t][alive].astype(np.float32))
        o, r, d, info = orc.step(b['actions'][t])
        assert np.array_equal(raw['rewards'][t][alive], r[alive]) and np.array_equal(raw['infos'][t][alive], info[alive])
        assert np.array_equal(raw['dones'][t][alive].astype(bool), d[alive])
        cont = alive & ~d
        assert np.array_equal(raw['obs'][t + 1][cont], o[cont])               # next observation of running envs
        assert np.array_equal(raw['final_obs'][t][alive & d], o[alive & d])     # terminal observation of finished ones
        assert np.array_equal(b['new_obs'][t][alive], o[alive].astype(np.float32))
        assert np.allclose(b['rewards'][t][alive], r[alive], atol=1e-6)
        checked += int(alive.sum())
        alive = cont
    assert checked > n * 8
    env.close()


def _fragment_buffers(env, T):
    import torch
    dev, n, od = env.device, env.n_envs, env.obs_dim
    f64, f32 = dict(dtype=torch.float64, device=dev), dict(dtype=torch.float32, device=dev)
    return dict(obs=torch.zeros((T + 1, n, od), **f64), final_obs=torch.zeros((T, n, od), **f64),
                reward=torch.zeros((T, n), **f64), done=torch.zeros((T, n), dtype=torch.uint8, device=dev),
                info=torch.zeros((T, n, 2), **f64), action=torch.zeros((T, n), dtype=torch.int32, device=dev),
                logp=torch.zeros((T, n), **f32), value=torch.zeros((T, n), **f32), last_value=torch.zeros(n, **f32))


@pytest.mark.parametrize('part,obs_mode,n', [('door_test', 'section', 203), ('square', 'grid', 64), ('door_test', 'grid', 130),
                                             ('door_test', 'section/HSI', 150), ('square', 'grid/HSI', 70)])
def test_persistent_fragment_with_given_actions_equals_step_by_step(part, obs_mode, n):
    """prl_rollout_fragment without a policy (it reads the action rows): every trajectory row and the final env
    state equal T launches of prl_batch_step, including envs that finish and restart inside the fragment, batch
    sizes that are not a multiple of the workgroup's four envs, and both mask widths (3 and 4 words per lane).
    '/HSI': COLOR_MODE 'HSI' (thickness bytes; the fused kernels' thickness builds, round 5) -- the bytes too."""
    import torch
    from conftest import start_points_for, synthetic_tables
    from paintrl_amd.batched_env import BatchedPaintEnv
    from paintrl_amd.device_tables import DeviceTables
    tables = synthetic_tables(part)
    sp = start_points_for(tables, 'all')
    T = 37
    hsi = obs_mode.endswith('/HSI')
    obs_mode = obs_mode.split('/')[0]
    kw = dict(auto_reset=True, seed=21, obs_mode=obs_mode, overlap_penalty=obs_mode == 'grid',
              max_possible_point=14350 if part == 'square' else 9148, **(dict(color_mode='HSI') if hsi else {}))
    env_a = BatchedPaintEnv(DeviceTables(tables, start_points=sp), n, **kw)
    env_b = BatchedPaintEnv(DeviceTables(tables, start_points=sp), n, **kw)
    start = np.random.RandomState(1).randint(0, len(sp), size=n)
    o0 = env_a.reset(start_idx=start).clone()
    env_b.reset(start_idx=start)
    buf = _fragment_buffers(env_a, T)
    gen = torch.Generator(device=env_a.device)
    gen.manual_seed(9)
    buf['action'].copy_(torch.randint(0, 4, (T, n), generator=gen, device=env_a.device, dtype=torch.int32))
    buf['obs'][0].copy_(o0)
    env_a.rollout_fragment(T, buf['obs'], buf['final_obs'], buf['reward'], buf['done'], buf['info'], buf['action'])
    torch.cuda.synchronize()
    n_done = 0
    for t in range(T):
        o, r, d, i = env_b.step(buf['action'][t])
        assert torch.equal(buf['obs'][t + 1], o), 'obs row %d' % t
        assert torch.equal(buf['reward'][t], r) and torch.equal(buf['info'][t], i), 'reward row %d' % t
        assert torch.equal(buf['done'][t].bool(), d), 'done row %d' % t
        if bool(d.any()):
            assert torch.equal(buf['final_obs'][t][d], env_b.final_obs[d])
        n_done += int(d.sum())
    assert n_done > (0 if hsi else n // 4)
    assert torch.equal(env_a.painted_words(), env_b.painted_words())
    if hsi:
        assert np.array_equal(env_a.thickness(), env_b.thickness())
    sa, sb = env_a.state(), env_b.state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    env_a.close()
    env_b.close()


def test_persistent_fragment_with_policy_equals_two_launches_per_step():
    """RolloutWorker(persistent=True) -- policy and env step for the whole fragment in one launch -- fills the
    trajectory buffers with exactly the bits of the two-launches-per-step worker, over two fragments."""
    import torch
    from conftest import synthetic_tables
    from paintrl_amd.batched_env import BatchedPaintEnv
    from paintrl_amd.device_tables import DeviceTables
    from paintrl_amd.rollout import MLPPolicy, RolloutWorker
    tables = synthetic_tables('door_test')
    n, T = 150, 40
    envs = [BatchedPaintEnv(DeviceTables(tables), n, auto_reset=True, seed=11) for _ in range(2)]
    torch.manual_seed(3)
    policy = MLPPolicy(envs[0].obs_dim, 4).to(envs[0].device)
    workers = [RolloutWorker(envs[0], policy, fragment=T, seed=5), RolloutWorker(envs[1], policy, fragment=T, seed=5, persistent=True)]
    for frag in range(2):
        out = [w.collect() for w in workers]
        torch.cuda.synchronize()
        (b0, v0, r0), (b1, v1, r1) = out
        for k in b0:
            assert torch.equal(b0[k], b1[k]), (frag, k)
        assert torch.equal(v0, v1) and torch.equal(r0, r1)
        for k in workers[0].raw:
            assert torch.equal(workers[0].raw[k], workers[1].raw[k]), (frag, k)
    assert int(b0['dones'].sum()) > 50
    assert torch.equal(envs[0].painted_words(), envs[1].painted_words())
    for e in envs:
        e.close()


def test_act_step_launch_equals_policy_launch_plus_step_launch():
    """prl_batch_act_step -- policy and env step of one worker iteration in ONE launch -- writes exactly the rows of
    prl_policy_act followed by prl_batch_step, over two fragments (150 envs: a ragged last workgroup)."""
    import torch
    from conftest import synthetic_tables
    from paintrl_amd.batched_env import BatchedPaintEnv
    from paintrl_amd.device_tables import DeviceTables
    from paintrl_amd.rollout import MLPPolicy, RolloutWorker
    tables = synthetic_tables('door_test')
    n, T = 150, 30
    envs = [BatchedPaintEnv(DeviceTables(tables), n, auto_reset=True, seed=12) for _ in range(2)]
    torch.manual_seed(4)
    policy = MLPPolicy(envs[0].obs_dim, 4).to(envs[0].device)
    workers = [RolloutWorker(envs[0], policy, fragment=T, seed=6), RolloutWorker(envs[1], policy, fragment=T, seed=6, act_step=True)]
    for frag in range(2):
        out = [w.collect() for w in workers]
        torch.cuda.synchronize()
        (b0, v0, r0), (b1, v1, r1) = out
        for k in b0:
            assert torch.equal(b0[k], b1[k]), (frag, k)
        assert torch.equal(v0, v1) and torch.equal(r0, r1)
        for k in workers[0].raw:
            assert torch.equal(workers[0].raw[k], workers[1].raw[k]), (frag, k)
    assert torch.equal(envs[0].painted_words(), envs[1].painted_words())
    for e in envs:
        e.close()


def test_ppo_improves_the_return():
    """The rollout driver + the minimal PPO step learn (paint_ppo.py:170-195 in miniature): synthetic sheet, 1 024 envs,
    persistent 50-step fragments, fixed seeds.  A random policy leaves the part after ~24 steps (mean return ~3); within a
    dozen updates the policy has learnt to stay on it and to paint fresh area: the mean return of the episodes finishing
    in a fragment must have grown at least sixfold (measured: 3.3 -> 41), their mean length beyond 150 steps."""
    sys_path_tools = __import__('os').path.join(__import__('conftest').REPO, 'tools')
    import sys
    sys.path.insert(0, sys_path_tools)
    import ppo_learns
    hist = ppo_learns.run(part='square', n=1024, T=50, updates=13, verbose=False)
    first_ret, first_len = hist[0][0], hist[0][1]
    last_ret, last_len = hist[-1][0], hist[-1][1]
    assert first_ret < 6.0 and first_len < 40, hist[0]
    assert last_ret > 6.0 * first_ret and last_ret > 25.0 and last_len > 150, (hist[0], hist[-1])


@pytest.mark.parametrize('part,tex', [('door_test', 240), ('square', 240), ('door_rr_big', 320)])
def test_kernel_families_interleaved_keep_the_last_shot_rows(part, tex):
    """Every kernel that steps a batch writes the last-shot rows and the index of their non-zero words (StepArgs::last_nz) in
    its own way -- tracked word by word (step, act_step), whole rows marked 'all non-zero' (persistent fragments), in place in
    HBM (large parts) -- and every other one must be able to continue from it.  One batch goes through them interleaved --
    policy fragment (one persistent launch), given-action fragment, act_step launches, plain steps, a masked reset -- while a
    twin takes the same actions launch by launch (prl_policy_act + prl_batch_step); after every phase the painted rows, the RAW
    last-shot rows and the states are equal, and the index is a superset of the rows' non-zero words.  OVERLAP_PENALTY is on:
    the rewards depend on the last-shot rows."""
    import torch
    from paintrl_amd.batched_env import BatchedPaintEnv
    from paintrl_amd.device_tables import DeviceTables
    from paintrl_amd.rollout import MLPPolicy, RolloutWorker
    tables = synthetic_tables(part, tex_size=(tex, tex))
    sp = start_points_for(tables, 'all')
    n, T = 70, 9
    kw = dict(auto_reset=True, seed=33, overlap_penalty=True, max_possible_point=int(0.95 * tables.sample_pos.shape[0]))
    env_a = BatchedPaintEnv(DeviceTables(tables, start_points=sp), n, **kw)
    env_b = BatchedPaintEnv(DeviceTables(tables, start_points=sp), n, **kw)
    start = np.random.RandomState(4).randint(0, len(sp), size=n)
    env_a.reset(start_idx=start)
    env_b.reset(start_idx=start)

    def check(phase):
        torch.cuda.synchronize()
        assert torch.equal(env_a.painted_words(), env_b.painted_words()), phase
        la, nza = env_a.last_shot_words()
        lb, nzb = env_b.last_shot_words()
        assert torch.equal(la, lb), 'last-shot rows after ' + phase
        for last, nz in ((la, nza), (lb, nzb)):
            words = (last.cpu().numpy() != 0)
            bits = np.unpackbits(nz.cpu().numpy().view(np.uint8), axis=1, bitorder='little')[:, :words.shape[1]].astype(bool)
            assert not (words & ~bits).any(), 'a non-zero word the index does not name, after ' + phase
        sa, sb = env_a.state(), env_b.state()
        for k in sa:
            if k != 'facet_hint':
                assert np.array_equal(sa[k], sb[k]), (phase, k)

    torch.manual_seed(8)
    policy = MLPPolicy(env_a.obs_dim, 4).to(env_a.device)
    gen = torch.Generator(device=env_a.device)
    gen.manual_seed(10)
    # 1. the policy-driven fragment as one persistent launch / launch by launch
    wa, wb = RolloutWorker(env_a, policy, fragment=T, seed=5, persistent=True), RolloutWorker(env_b, policy, fragment=T, seed=5)
    ba, bb = wa.collect()[0], wb.collect()[0]
    assert torch.equal(ba['actions'], bb['actions']) and torch.equal(ba['rewards'], bb['rewards'])
    check('the policy fragment')
    # 2. a given-action fragment (one persistent launch) / steps
    buf = _fragment_buffers(env_a, T)
    buf['action'].copy_(torch.randint(0, 4, (T, n), generator=gen, device=env_a.device, dtype=torch.int32))
    buf['obs'][0].copy_(env_a.obs)
    env_a.rollout_fragment(T, buf['obs'], buf['final_obs'], buf['reward'], buf['done'], buf['info'], buf['action'])
    for t in range(T):
        _, r, _, _ = env_b.step(buf['action'][t])
        assert torch.equal(buf['reward'][t], r), t
    check('the given-action fragment')
    # 3. act_step launches / two launches per step (fresh sampling streams on both sides)
    # (a new worker resets its env -- the library's own start draws, the same on both sides: one more writer of the rows)
    wa2, wb2 = RolloutWorker(env_a, policy, fragment=T, seed=6, act_step=True), RolloutWorker(env_b, policy, fragment=T, seed=6)
    check('the workers\' reset')
    ba, bb = wa2.collect()[0], wb2.collect()[0]
    assert torch.equal(ba['actions'], bb['actions']) and torch.equal(ba['rewards'], bb['rewards'])
    check('the act_step launches')
    # 4. plain steps, a masked reset, plain steps
    for t in range(4):
        a = torch.randint(0, 4, (n,), generator=gen, device=env_a.device, dtype=torch.int32)
        ra, rb = env_a.step(a)[1].clone(), env_b.step(a)[1].clone()
        assert torch.equal(ra, rb)
    check('plain steps')
    mask = (np.arange(n) % 3 == 0)
    nxt = np.random.RandomState(5).randint(0, len(sp), size=n)
    assert torch.equal(env_a.reset(mask=mask, start_idx=nxt), env_b.reset(mask=mask, start_idx=nxt))
    check('a masked reset')
    # 5. ... and back into a persistent fragment
    buf['action'].copy_(torch.randint(0, 4, (T, n), generator=gen, device=env_a.device, dtype=torch.int32))
    buf['obs'][0].copy_(env_a.obs)
    env_a.rollout_fragment(T, buf['obs'], buf['final_obs'], buf['reward'], buf['done'], buf['info'], buf['action'])
    for t in range(T):
        _, r, _, _ = env_b.step(buf['action'][t])
        assert torch.equal(buf['reward'][t], r), t
    check('the second given-action fragment')
    env_a.close()
    env_b.close()
