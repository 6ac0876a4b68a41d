"""Rollout driver for the batched env: on-device policy MLP + trajectory buffer + a minimal PPO step.

This is the caller of the hot path in BASELINE.json configs 4-5: what one RLlib
rollout worker plus the learner of ``paint_ppo.py`` do, with the host out of the per-step loop.
Policy / value networks follow ``paint_ppo.py:170-195`` (``fcnet_hiddens: [256, 128]``, rollout
fragment ``sample_batch_size: 100``, discrete-4 actions).  Everything stays on the GPU: the env
writes float64 observations, the policy reads them as float32 on the same stream and writes int32
actions that the next ``prl_batch_step`` launch consumes.
"""
import torch
from torch import nn

from . import distributed as pdist

FRAGMENT = 100          # paint_ppo.py:190 sample_batch_size


class MLPPolicy(nn.Module):
    """obs -> 256 -> 128 -> (action logits, value); tanh like RLlib's default fcnet."""

    def __init__(self, obs_dim, n_actions, hiddens=(256, 128)):
        super().__init__()
        layers, d = [], obs_dim
        for h in hiddens:
            layers += [nn.Linear(d, h), nn.Tanh()]
            d = h
        self.body = nn.Sequential(*layers)
        self.pi = nn.Linear(d, n_actions)
        self.vf = nn.Linear(d, 1)

    def forward(self, obs):
        z = self.body(obs)
        return self.pi(z), self.vf(z).squeeze(-1)

    @torch.no_grad()
    def act(self, obs, generator=None):
        logits, value = self(obs)
        logp_all = torch.log_softmax(logits, dim=-1)
        actions = torch.multinomial(logp_all.exp(), 1, generator=generator).squeeze(-1)
        return actions.to(torch.int32), logp_all.gather(-1, actions.unsqueeze(-1)).squeeze(-1), value


class RolloutWorker(object):
    """Collects fragments of ``fragment`` steps from a BatchedPaintEnv (auto_reset=True) into
    preallocated [T, N, ...] device tensors with RLlib SampleBatch field names."""

    def __init__(self, env, policy, fragment=FRAGMENT, seed=0, fused=True):
        if not env.cfg.auto_reset:
            raise ValueError('RolloutWorker needs BatchedPaintEnv(auto_reset=True)')
        self.env, self.policy, self.T = env, policy, int(fragment)
        # sampling goes through the fused kernel (prl_policy_act); the torch module stays the learner's
        # copy -- call sync_policy() after an optimizer step.  fused=False keeps torch eager sampling.
        self.fused = None
        if fused:
            from .policy import FusedPolicy
            self.fused = FusedPolicy(policy, seed=seed)
        dev, n, od = env.device, env.n_envs, env.obs_dim
        f32 = dict(dtype=torch.float32, device=dev)
        self.buf = {
            'obs': torch.zeros((self.T, n, od), **f32), 'new_obs': torch.zeros((self.T, n, od), **f32),
            'actions': torch.zeros((self.T, n), dtype=torch.int32, device=dev),
            'rewards': torch.zeros((self.T, n), **f32), 'dones': torch.zeros((self.T, n), dtype=torch.bool, device=dev),
            'action_logp': torch.zeros((self.T, n), **f32), 'vf_preds': torch.zeros((self.T, n), **f32),
            'infos_reward': torch.zeros((self.T, n), **f32), 'infos_penalty': torch.zeros((self.T, n), **f32),
        }
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(seed)
        self.obs64 = env.reset()
        self.obs = self.obs64.to(torch.float32)
        self.steps_done = 0

    def sync_policy(self):
        if self.fused is not None:
            self.fused.sync()

    def _act(self):
        if self.fused is not None:
            return self.fused.act(self.obs64)              # in-kernel sampling stream keyed by the worker's seed
        return self.policy.act(self.obs, self.gen)

    @torch.no_grad()
    def collect(self):
        """One fragment.  Returns (batch dict, last value estimates, gathered episode returns)."""
        b, env = self.buf, self.env
        for t in range(self.T):
            actions, logp, value = self._act()
            obs64, reward, done, info = env.step(actions)
            b['obs'][t] = self.obs
            b['actions'][t] = actions
            b['action_logp'][t] = logp
            b['vf_preds'][t] = value
            b['rewards'][t] = reward
            b['dones'][t] = done
            b['infos_reward'][t] = info[:, 0]
            b['infos_penalty'][t] = info[:, 1]
            b['new_obs'][t] = torch.where(done.unsqueeze(-1), env.final_obs, obs64).to(torch.float32)
            self.obs64 = obs64
            self.obs = obs64.to(torch.float32)
        self.steps_done += self.T
        _, _, last_value = self._act()
        returns = pdist.gather_returns(env.episode_returns())       # once per fragment, RCCL when world > 1
        return b, last_value, returns


def gae(batch, last_value, gamma=0.99, lam=0.95):
    """Generalised advantage estimation over a [T, N] fragment (RLlib's compute_advantages)."""
    T = batch['rewards'].shape[0]
    adv = torch.zeros_like(batch['rewards'])
    nxt, acc = last_value, torch.zeros_like(last_value)
    for t in range(T - 1, -1, -1):
        nonterminal = (~batch['dones'][t]).to(torch.float32)
        delta = batch['rewards'][t] + gamma * nxt * nonterminal - batch['vf_preds'][t]
        acc = delta + gamma * lam * nonterminal * acc
        adv[t] = acc
        nxt = batch['vf_preds'][t]
    return adv, adv + batch['vf_preds']


def ppo_update(policy, optimizer, batch, last_value, epochs=2, minibatches=4, clip=0.2, vf_coeff=0.5,
               entropy_coeff=0.01):
    """A minimal clipped-surrogate PPO step on one fragment; returns the mean loss of the last epoch."""
    adv, targets = gae(batch, last_value)
    flat = lambda x: x.reshape(-1, *x.shape[2:])                     # noqa: E731
    obs, act, logp_old = flat(batch['obs']), flat(batch['actions']).long(), flat(batch['action_logp'])
    adv, targets = flat(adv), flat(targets)
    adv = (adv - adv.mean()) / (adv.std() + 1e-8)
    n = obs.shape[0]
    loss_val = 0.0
    for _ in range(epochs):
        perm = torch.randperm(n, device=obs.device)
        for idx in perm.chunk(minibatches):
            logits, value = policy(obs[idx])
            logp_all = torch.log_softmax(logits, dim=-1)
            logp = logp_all.gather(-1, act[idx].unsqueeze(-1)).squeeze(-1)
            ratio = (logp - logp_old[idx]).exp()
            surr = torch.min(ratio * adv[idx], ratio.clamp(1 - clip, 1 + clip) * adv[idx])
            entropy = -(logp_all.exp() * logp_all).sum(-1)
            loss = -surr.mean() + vf_coeff * (value - targets[idx]).pow(2).mean() - entropy_coeff * entropy.mean()
            optimizer.zero_grad(set_to_none=True)
            loss.backward()
            optimizer.step()
            loss_val = float(loss.detach())
    return loss_val
