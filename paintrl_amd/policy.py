"""Fused on-device policy step for the rollout driver: ``prl_policy_act``.

``FusedPolicy`` wraps the torch ``MLPPolicy`` of ``paintrl_amd.rollout`` (the net of
``paint_ppo.py:170-195``): the learner keeps training the torch module, the rollout worker samples
with ONE hand-written kernel per env step (``csrc/policy_mlp.hip``) that reads the env's float64
observations and writes the int32 actions the next ``prl_batch_step`` consumes.  Call ``sync()``
after every optimizer step to refresh the kernel's copy of the weights.
"""
import ctypes as C

from . import _lib


def _torch():
    import torch
    return torch


class FusedPolicy(object):
    def __init__(self, policy, seed=0):
        self.policy = policy
        self.lib = _lib.load()
        self._w = None
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self._rng_count = None                      # u32 per env, advanced by the kernel (in-kernel sampling stream)
        self.sync()

    @property
    def n_actions(self):
        return self.policy.pi.out_features

    def sync(self):
        """Copy the torch parameters into the layout the kernel reads: f32, row-major [in][out]."""
        torch = _torch()
        p = self.policy
        linears = [m for m in p.body if isinstance(m, torch.nn.Linear)]
        if len(linears) != 2:
            raise ValueError('FusedPolicy needs two hidden layers (fcnet_hiddens [h1, h2])')
        l1, l2 = linears
        with torch.no_grad():
            f = lambda t: t.detach().to(torch.float32).contiguous()          # noqa: E731
            self._tensors = dict(
                w1=f(l1.weight.t()), b1=f(l1.bias), w2=f(l2.weight.t()), b2=f(l2.bias),
                w3=f(torch.cat([p.pi.weight, p.vf.weight], dim=0).t()), b3=f(torch.cat([p.pi.bias, p.vf.bias], dim=0)))
        w = _lib.PrlPolicyWeights()
        w.in_dim, w.h1, w.h2, w.n_actions = l1.in_features, l1.out_features, l2.out_features, p.pi.out_features
        for k, t in self._tensors.items():
            if not t.is_cuda:
                raise _lib.PaintRLError('FusedPolicy: the policy must live on the GPU (no CPU fallback)')
            setattr(w, k, t.data_ptr())
        self._w = w
        self.device = self._tensors['w1'].device

    def act(self, obs, uniform=None, generator=None, want_logits=False):
        """obs: float64 (N, in_dim) device tensor -> (int32 actions, float32 log-probabilities, float32 values
        [, float32 logits]).  ``uniform``: float32 (N,) in [0, 1); if absent it is drawn from ``generator``,
        and if that is absent too the kernel draws from its own counter-based stream (seed, env, draw number)
        -- no extra launch, and replayable from a captured HIP graph."""
        torch = _torch()
        if obs.dtype != torch.float64 or not obs.is_cuda or obs.dim() != 2 or obs.shape[1] != self._w.in_dim:
            raise ValueError('obs must be a float64 (N, %d) device tensor' % self._w.in_dim)
        obs = obs.contiguous()
        n = obs.shape[0]
        if uniform is None and generator is not None:
            uniform = torch.rand(n, dtype=torch.float32, device=obs.device, generator=generator)
        if uniform is None and (self._rng_count is None or self._rng_count.shape[0] != n):
            self._rng_count = torch.zeros(n, dtype=torch.int32, device=obs.device)
        action = torch.empty(n, dtype=torch.int32, device=obs.device)
        logp = torch.empty(n, dtype=torch.float32, device=obs.device)
        value = torch.empty(n, dtype=torch.float32, device=obs.device)
        logits = torch.empty((n, self._w.n_actions), dtype=torch.float32, device=obs.device) if want_logits else None
        with torch.cuda.device(obs.device):
            stream = C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream)
            _lib.check(self.lib.prl_policy_act(C.byref(self._w), n, C.c_void_p(obs.data_ptr()),
                                               C.c_void_p(uniform.data_ptr()) if uniform is not None else None,
                                               C.c_void_p(self._rng_count.data_ptr()) if uniform is None else None,
                                               C.c_uint64(self.seed), C.c_void_p(action.data_ptr()),
                                               C.c_void_p(logp.data_ptr()), C.c_void_p(value.data_ptr()),
                                               C.c_void_p(logits.data_ptr()) if want_logits else None, stream),
                       'prl_policy_act')
        return (action, logp, value, logits) if want_logits else (action, logp, value)

    def act_into(self, obs, action, logp, value):
        """``act`` with the kernel's own sampling stream, writing into caller-owned tensors (int32 (N,), float32
        (N,), float32 (N,)) -- rows of a trajectory buffer.  No checks, no allocations: the rollout hot loop."""
        torch = _torch()
        n = obs.shape[0]
        if self._rng_count is None or self._rng_count.shape[0] != n:
            self._rng_count = torch.zeros(n, dtype=torch.int32, device=obs.device)
        rc = self.lib.prl_policy_act(C.byref(self._w), n, C.c_void_p(obs.data_ptr()), None,
                                     C.c_void_p(self._rng_count.data_ptr()), C.c_uint64(self.seed),
                                     C.c_void_p(action.data_ptr()), C.c_void_p(logp.data_ptr()),
                                     C.c_void_p(value.data_ptr()), None,
                                     C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream))
        if rc:
            _lib.check(rc, 'prl_policy_act')
