"""GPU tests of prl_policy_act (csrc/policy_mlp.hip): the fused rollout-policy kernel against torch.

This is a floating-point kernel, so the reference is the same network in torch (float32 on the device,
and float64 as ground truth).  Tolerance: 2e-5 absolute on logits / values / log-probabilities -- the
kernel's dot products are k-ordered f32 fma chains (v_mfma_f32_16x16x4_f32), torch's use another
order, and tanhf / expf differ in the last ulp."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _policy(in_dim=6, n_actions=4, hiddens=(256, 128), seed=0, scale=1.0):
    import torch
    from paintrl_amd.rollout import MLPPolicy
    torch.manual_seed(seed)
    p = MLPPolicy(in_dim, n_actions, hiddens).to('cuda')
    with torch.no_grad():
        for q in p.parameters():
            q.mul_(scale)
    return p


@pytest.mark.parametrize('n,in_dim,n_actions,hiddens,scale', [(1000, 6, 4, (256, 128), 1.0), (4096, 6, 4, (256, 128), 4.0),
                                                              (33, 16, 8, (64, 32), 2.0), (5, 5, 3, (32, 32), 1.0),
                                                              (100, 7, 15, (48, 16), 1.5)])
def test_fused_policy_matches_torch(n, in_dim, n_actions, hiddens, scale):
    import torch
    from paintrl_amd.policy import FusedPolicy
    p = _policy(in_dim, n_actions, hiddens, seed=n, scale=scale)
    fused = FusedPolicy(p)
    g = torch.Generator(device='cuda')
    g.manual_seed(7)
    obs = torch.rand((n, in_dim), dtype=torch.float64, device='cuda', generator=g)
    u = torch.rand(n, dtype=torch.float32, device='cuda', generator=g)
    act, logp, value, logits = fused.act(obs, uniform=u, want_logits=True)
    torch.cuda.synchronize()
    with torch.no_grad():
        l32, v32 = p(obs.to(torch.float32))
        p64 = _policy(in_dim, n_actions, hiddens, seed=n, scale=scale).double()
        l64, v64 = p64(obs)
    assert (logits - l32).abs().max().item() < TOL and (value - v32).abs().max().item() < TOL
    assert (logits.double() - l64).abs().max().item() < TOL and (value.double() - v64).abs().max().item() < TOL
    # the action is the inverse-CDF draw for u under softmax(logits); allow for u within 1e-5 of a boundary
    cdf = torch.softmax(logits.double(), dim=-1).cumsum(-1).cpu().numpy()
    un, an = u.double().cpu().numpy(), act.cpu().numpy()
    want = np.minimum((un[:, None] >= cdf[:, :-1]).sum(1), n_actions - 1)
    near = (np.abs(un[:, None] - cdf[:, :-1]) < 1e-5).any(1)
    assert an.min() >= 0 and an.max() < n_actions and ((an == want) | near).all()
    lp = torch.log_softmax(logits.double(), dim=-1).gather(-1, act.long().unsqueeze(-1)).squeeze(-1)
    assert (logp.double() - lp).abs().max().item() < TOL
    # and the draw follows the distribution: empirical frequencies of many uniforms for one observation
    if n >= 1000:
        rep = obs[:1].repeat(20000, 1)
        a2, _, _ = fused.act(rep, generator=g)
        freq = np.bincount(a2.cpu().numpy(), minlength=n_actions) / 20000.0
        prob = torch.softmax(l64[0], dim=-1).cpu().numpy()
        assert np.abs(freq - prob).max() < 0.02


def test_fused_policy_follows_weight_updates_and_rejects_bad_shapes():
    import torch
    from paintrl_amd import _lib
    from paintrl_amd.policy import FusedPolicy
    p = _policy()
    fused = FusedPolicy(p)
    obs = torch.rand((64, 6), dtype=torch.float64, device='cuda')
    u = torch.full((64,), 0.5, dtype=torch.float32, device='cuda')
    _, _, v0 = fused.act(obs, uniform=u)
    with torch.no_grad():
        p.vf.bias.add_(1.0)
    _, _, v1 = fused.act(obs, uniform=u)
    assert torch.equal(v0, v1)                          # the kernel reads its own copy ...
    fused.sync()
    _, _, v2 = fused.act(obs, uniform=u)
    assert (v2 - v0 - 1.0).abs().max().item() < 1e-6    # ... until sync()
    with pytest.raises(ValueError):
        fused.act(obs.to(torch.float32), uniform=u)
    lib = _lib.load()
    w = _lib.PrlPolicyWeights()
    C.memmove(C.byref(w), C.byref(fused._w), C.sizeof(w))
    w.h1 = 100                                          # not a multiple of 16
    out = torch.empty(64, dtype=torch.int32, device='cuda')
    rc = lib.prl_policy_act(C.byref(w), 64, C.c_void_p(obs.data_ptr()), C.c_void_p(u.data_ptr()), None, 0,
                            C.c_void_p(out.data_ptr()), None, None, None, None)
    assert rc == -3 and b'multiples of 16' in lib.prl_last_error()
    assert lib.prl_policy_act(None, 64, None, None, None, 0, None, None, None, None, None) == -1
    rc = lib.prl_policy_act(C.byref(fused._w), 64, C.c_void_p(obs.data_ptr()), None, None, 0,
                            C.c_void_p(out.data_ptr()), None, None, None, None)
    assert rc == -1 and b'counter array' in lib.prl_last_error()


def test_in_kernel_sampling_stream():
    """Without uniforms or a generator the kernel draws from (seed, env, draw number): reproducible per seed,
    different across seeds, envs and draws, and distributed like the policy."""
    import torch
    from paintrl_amd.policy import FusedPolicy
    p = _policy(scale=3.0)
    obs = torch.rand((1, 6), dtype=torch.float64, device='cuda').repeat(20000, 1)
    a, b, c = FusedPolicy(p, seed=5), FusedPolicy(p, seed=5), FusedPolicy(p, seed=6)
    a1, _, _, logits = a.act(obs, want_logits=True)
    a2, _, _ = a.act(obs)
    b1, _, _ = b.act(obs)
    c1, _, _ = c.act(obs)
    assert torch.equal(a1, b1) and not torch.equal(a1, a2) and not torch.equal(a1, c1)
    prob = torch.softmax(logits[0].double(), -1).cpu().numpy()
    for draw in (a1, a2, c1):
        freq = np.bincount(draw.cpu().numpy(), minlength=4) / 20000.0
        assert np.abs(freq - prob).max() < 0.02
    assert int(a._rng_count.min()) == 2 and int(a._rng_count.max()) == 2


def test_rollout_worker_samples_with_the_fused_kernel():
    import torch
    from conftest import synthetic_tables
    from paintrl_amd.batched_env import BatchedPaintEnv
    from paintrl_amd.device_tables import DeviceTables
    from paintrl_amd.rollout import MLPPolicy, RolloutWorker, ppo_update
    env = BatchedPaintEnv(DeviceTables(synthetic_tables('door_test')), 256, auto_reset=True, seed=3)
    torch.manual_seed(0)
    policy = MLPPolicy(env.obs_dim, 4).to(env.device)
    worker = RolloutWorker(env, policy, fragment=20, seed=1)
    assert worker.fused is not None
    batch, last_value, returns = worker.collect()
    torch.cuda.synchronize()
    a = batch['actions']
    assert a.min().item() >= 0 and a.max().item() <= 3 and len(torch.unique(a)) == 4
    # the recorded log-probabilities and values are those of the torch module on the recorded observations
    with torch.no_grad():
        logits, value = policy(batch['obs'].reshape(-1, env.obs_dim))
        lp = torch.log_softmax(logits, -1).gather(-1, a.reshape(-1, 1).long()).squeeze(-1)
    assert (lp - batch['action_logp'].reshape(-1)).abs().max().item() < TOL
    assert (value - batch['vf_preds'].reshape(-1)).abs().max().item() < TOL
    opt = torch.optim.Adam(policy.parameters(), lr=3e-4)
    loss = ppo_update(policy, opt, batch, last_value, epochs=1, minibatches=2)
    assert np.isfinite(loss)
    worker.sync_policy()
    worker.collect()
    env.close()
