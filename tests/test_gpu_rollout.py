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
