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
    preallocated [T, N, ...] device tensors with RLlib SampleBatch field names.

    With the fused policy (default) a step is two launches and nothing else: ``prl_policy_act`` reads the
    observations of step t from the buffer row the env wrote them to and writes actions / log-probabilities /
    value estimates into their buffer rows; ``prl_batch_step`` reads those actions and writes the next
    observations, rewards, done flags, info and terminal observations into theirs.  The float32 views RLlib
    expects are made once per fragment."""

    def __init__(self, env, policy, fragment=FRAGMENT, seed=0, fused=True, persistent=False, act_step=False):
        """``persistent=True`` (with the fused policy): a whole fragment is ONE launch of the persistent rollout
        kernel (``prl_rollout_fragment``) instead of 2 T launches; ``act_step=True``: T launches, policy and env
        step in one (``prl_batch_act_step``).  The buffers receive the same bits either way."""
        self.persistent, self.act_step = bool(persistent), bool(act_step)
        if (self.persistent or self.act_step) and not fused:
            raise ValueError('persistent / act_step need the fused policy')
        if not env.cfg.auto_reset:
            raise ValueError('RolloutWorker needs BatchedPaintEnv(auto_reset=True)')
        self.env, self.policy, self.T = env, policy, int(fragment)
        # sampling goes through the fused kernel (prl_policy_act); the torch module stays the learner's
        # copy -- call sync_policy() after an optimizer step.  fused=False keeps torch eager sampling.
        self.fused = None
        if fused:
            from .policy import FusedPolicy
            self.fused = FusedPolicy(policy, seed=seed)
        dev, n, od, T = env.device, env.n_envs, env.obs_dim, self.T
        f32, f64 = dict(dtype=torch.float32, device=dev), dict(dtype=torch.float64, device=dev)
        # what the kernels write (float64 like the reference's observations and rewards) ...
        self.raw = {
            'obs': torch.zeros((T + 1, n, od), **f64),         # row t: observation before step t; row T: after the last
            'final_obs': torch.zeros((T, n, od), **f64), 'rewards': torch.zeros((T, n), **f64),
            'dones': torch.zeros((T, n), dtype=torch.uint8, device=dev), 'infos': torch.zeros((T, n, 2), **f64),
        }
        # ... and the SampleBatch the learner reads
        self.buf = {
            'obs': torch.zeros((T, n, od), **f32), 'new_obs': torch.zeros((T, n, od), **f32),
            'actions': torch.zeros((T, n), dtype=torch.int32, device=dev),
            'rewards': torch.zeros((T, n), **f32), 'dones': torch.zeros((T, n), dtype=torch.bool, device=dev),
            'action_logp': torch.zeros((T, n), **f32), 'vf_preds': torch.zeros((T, n), **f32),
            'infos_reward': torch.zeros((T, n), **f32), 'infos_penalty': torch.zeros((T, n), **f32),
        }
        self._last_value = torch.zeros(n, **f32)
        self._scratch_action = torch.zeros(n, dtype=torch.int32, device=dev)
        self._scratch_logp = torch.zeros(n, **f32)
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(seed)
        self.raw['obs'][0].copy_(env.reset())
        self.steps_done = 0

    def sync_policy(self):
        if self.fused is not None:
            self.fused.sync()

    @torch.no_grad()
    def collect(self):
        """One fragment.  Returns (batch dict, last value estimates, gathered episode returns)."""
        b, raw, env, T = self.buf, self.raw, self.env, self.T
        if self.persistent:
            fp = self.fused
            if fp._rng_count is None or fp._rng_count.shape[0] != env.n_envs:
                fp._rng_count = torch.zeros(env.n_envs, dtype=torch.int32, device=env.device)
            env.rollout_fragment(T, raw['obs'], raw['final_obs'], raw['rewards'], raw['dones'], raw['infos'], b['actions'],
                                 weights=fp._w, logp=b['action_logp'], value=b['vf_preds'], last_value=self._last_value,
                                 rng_count=fp._rng_count, rng_seed=fp.seed)
        elif self.act_step:
            fp = self.fused
            if fp._rng_count is None or fp._rng_count.shape[0] != env.n_envs:
                fp._rng_count = torch.zeros(env.n_envs, dtype=torch.int32, device=env.device)
            for t in range(T):
                env.act_step_into(fp._w, raw['obs'][t], fp._rng_count, fp.seed, b['actions'][t], b['action_logp'][t],
                                  b['vf_preds'][t], raw['obs'][t + 1], raw['rewards'][t], raw['dones'][t], raw['infos'][t],
                                  raw['final_obs'][t])
            self.fused.act_into(raw['obs'][T], self._scratch_action, self._scratch_logp, self._last_value)
        elif self.fused is not None:
            for t in range(T):
                self.fused.act_into(raw['obs'][t], b['actions'][t], b['action_logp'][t], b['vf_preds'][t])
                env.step_into(b['actions'][t], raw['obs'][t + 1], raw['rewards'][t], raw['dones'][t], raw['infos'][t],
                              raw['final_obs'][t])
            self.fused.act_into(raw['obs'][T], self._scratch_action, self._scratch_logp, self._last_value)
            # the extra draw advanced the sampling stream by one; harmless (it is never replayed)
        else:
            for t in range(T):
                actions, logp, value = self.policy.act(raw['obs'][t].to(torch.float32), self.gen)
                b['actions'][t], b['action_logp'][t], b['vf_preds'][t] = actions, logp, value
                env.step_into(b['actions'][t], raw['obs'][t + 1], raw['rewards'][t], raw['dones'][t], raw['infos'][t],
                              raw['final_obs'][t])
            _, _, last = self.policy.act(raw['obs'][T].to(torch.float32), self.gen)
            self._last_value.copy_(last)
        # float32 SampleBatch views, once per fragment
        b['obs'].copy_(raw['obs'][:T])
        b['dones'].copy_(raw['dones'].bool())
        b['new_obs'].copy_(torch.where(b['dones'].unsqueeze(-1), raw['final_obs'], raw['obs'][1:]))
        b['rewards'].copy_(raw['rewards'])
        b['infos_reward'].copy_(raw['infos'][..., 0])
        b['infos_penalty'].copy_(raw['infos'][..., 1])
        raw['obs'][0].copy_(raw['obs'][T])                          # the next fragment starts where this one ended
        self.steps_done += T
        returns = pdist.gather_returns(env.episode_returns())       # once per fragment, RCCL when world > 1
        return b, self._last_value.clone(), returns


class FragmentRunner(object):
    """bench.py --policy fragment: policy + env for up to ``fragment`` steps in one persistent launch, rows into a ring
    of trajectory buffers (what RolloutWorker(persistent=True) does, without the float32 SampleBatch views)."""

    def __init__(self, env, policy, fragment=FRAGMENT, seed=0, given_actions=None):
        """``given_actions``: int32 (steps, N) device tensor; the kernel then reads its action rows from it (in order,
        ``fragment`` rows per full launch) instead of running the policy."""
        from .policy import FusedPolicy
        self.env, self.T = env, int(fragment)
        self.given, self.cursor = given_actions, 0
        self.fused = FusedPolicy(policy, seed=seed)
        dev, n, od, T = env.device, env.n_envs, env.obs_dim, self.T
        f64, f32 = dict(dtype=torch.float64, device=dev), dict(dtype=torch.float32, device=dev)
        self.obs = torch.zeros((T + 1, n, od), **f64)
        self.final_obs = torch.zeros((T, n, od), **f64)
        self.reward = torch.zeros((T, n), **f64)
        self.info = torch.zeros((T, n, 2), **f64)
        self.done = torch.zeros((T, n), dtype=torch.uint8, device=dev)
        self.action = torch.zeros((T, n), dtype=torch.int32, device=dev)
        self.logp, self.value = torch.zeros((T, n), **f32), torch.zeros((T, n), **f32)
        self.last_value = torch.zeros(n, **f32)
        self.rng_count = torch.zeros(n, dtype=torch.int32, device=dev)
        self.obs[0].copy_(env.obs)

    def run(self, n_steps):
        """``n_steps`` <= fragment steps in one launch; the next call continues from the last observation."""
        n_steps = int(n_steps)
        if self.given is not None:
            rows = self.given[self.cursor:self.cursor + n_steps]
            self.cursor += n_steps
            self.env.rollout_fragment(n_steps, self.obs, self.final_obs, self.reward, self.done, self.info, rows)
        else:
            self.env.rollout_fragment(n_steps, self.obs, self.final_obs, self.reward, self.done, self.info, self.action,
                                      weights=self.fused._w, logp=self.logp, value=self.value, last_value=self.last_value,
                                      rng_count=self.rng_count, rng_seed=self.fused.seed)
        self.obs[0].copy_(self.obs[n_steps])


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
