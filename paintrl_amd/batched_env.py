"""BatchedPaintEnv: N independent PaintGymEnv instances advanced by one HIP kernel per step.

This is the batched counterpart of ``PaintRLEnv/robot_gym_env.py`` PaintGymEnv
(step rge:349-368, reset rge:370-387): same configuration names, same
observation / reward / done / info semantics per env, torch tensors in and out.
PyTorch is only plumbing here (device memory + the current HIP stream); all
simulation happens in libpaintrl_hip.so through the C ABI of include/paintrl.h.
"""
import ctypes as C

import numpy as np

from . import _lib, config as _config
from .device_tables import DeviceTables


def _torch():
    import torch
    return torch


class BatchedPaintEnv(object):
    """N envs on one GPU.

    parts        : DeviceTables or list of them (<= 8); env i uses parts[env_part_id[i]]
    n_envs       : number of environments
    env_part_id  : optional int sequence (len n_envs)
    device       : torch device string / index (default current CUDA device)
    other kwargs : see paintrl_amd.config.make_config (OBS_MODE etc. of rge:127-157)
    """

    def __init__(self, parts, n_envs, env_part_id=None, device=None, **cfg_kwargs):
        torch = _torch()
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.PaintRLError('BatchedPaintEnv needs a GPU (no CPU fallback)')
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.parts = list(parts) if isinstance(parts, (list, tuple)) else [parts]
        for p in self.parts:
            if not isinstance(p, DeviceTables):
                raise TypeError('parts must be paintrl_amd.device_tables.DeviceTables')
        self.n_envs = int(n_envs)
        cfg_kwargs = dict(cfg_kwargs)
        cfg_kwargs.setdefault('paint_radius', self.parts[0].paint_radius)    # the radius the tables were built for
        self.cfg_kwargs = dict(cfg_kwargs)
        self.cfg = _config.make_config(**cfg_kwargs)
        self.obs_dim = self.lib.prl_obs_dim(C.byref(self.cfg))
        self.discrete = self.cfg.action_mode == 0
        self.action_dim = self.cfg.action_dim
        self._part_handles = []
        self._batch = C.c_void_p()
        with torch.cuda.device(self.device):
            for p in self.parts:
                h = C.c_void_p()
                st = p.c_struct()
                _lib.check(self.lib.prl_part_create(C.byref(st), self.device.index, C.byref(h)), 'prl_part_create')
                self._part_handles.append(h)
            arr = (C.c_void_p * len(self._part_handles))(*[h.value for h in self._part_handles])
            if env_part_id is None:
                ids_ptr, self.env_part_id = None, np.zeros(self.n_envs, dtype=np.int32)
            else:
                self.env_part_id = np.ascontiguousarray(env_part_id, dtype=np.int32).reshape(self.n_envs)
                ids_ptr = self.env_part_id.ctypes.data_as(_lib._ip)
            _lib.check(self.lib.prl_batch_create(arr, len(self.parts), ids_ptr, self.n_envs, C.byref(self.cfg),
                                                 C.byref(self._batch)), 'prl_batch_create')
            self.mask_stride = self.lib.prl_batch_mask_stride(self._batch)
            f64 = dict(dtype=torch.float64, device=self.device)
            self.obs = torch.zeros((self.n_envs, self.obs_dim), **f64)
            self.final_obs = torch.zeros((self.n_envs, self.obs_dim), **f64)
            self.reward = torch.zeros(self.n_envs, **f64)
            self.info = torch.zeros((self.n_envs, 2), **f64)
            self.done_u8 = torch.zeros(self.n_envs, dtype=torch.uint8, device=self.device)

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return C.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _ptr(t):
        return None if t is None else C.c_void_p(t.data_ptr())

    def _as_start_idx(self, start_idx):
        torch = _torch()
        if start_idx is None:
            return None
        t = torch.as_tensor(start_idx, device=self.device).to(torch.int32).reshape(self.n_envs).contiguous()
        return t

    # ------------------------------------------------------------------ gym-like API
    def reset(self, mask=None, start_idx=None):
        """Reset all envs (or those where ``mask`` is true).  ``start_idx`` picks start points
        explicitly; otherwise the library's counter-based RNG draws them.  Returns obs (N, obs_dim)."""
        torch = _torch()
        m = None if mask is None else torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        s = self._as_start_idx(start_idx)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.prl_batch_reset(self._batch, self._ptr(m), self._ptr(s), self._ptr(self.obs),
                                                self._stream()), 'prl_batch_reset')
        return self.obs

    def step(self, actions, start_idx=None):
        """actions: int tensor (N,) for discrete mode, float64 (N, action_dim) for continuous.
        Returns (obs, reward, done, info) tensors; info[:,0]=reward, info[:,1]=penalty (rge:368)."""
        torch = _torch()
        if self.discrete:
            a = torch.as_tensor(actions, device=self.device).to(torch.int32).reshape(self.n_envs).contiguous()
        else:
            a = torch.as_tensor(actions, device=self.device).to(torch.float64)
            a = a.reshape(self.n_envs, self.action_dim).contiguous()
        s = self._as_start_idx(start_idx)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.prl_batch_step(self._batch, self._ptr(a), self._ptr(self.obs), self._ptr(self.reward),
                                               self._ptr(self.done_u8), self._ptr(self.info),
                                               self._ptr(self.final_obs), self._ptr(s), self._stream()),
                       'prl_batch_step')
        return self.obs, self.reward, self.done_u8.bool(), self.info

    def step_raw(self, actions_i32):
        """Hot-loop variant for benchmarks: ``actions_i32`` is already an int32 (or float64) device tensor
        of the right shape; no conversions, no new tensors."""
        _lib.check(self.lib.prl_batch_step(self._batch, C.c_void_p(actions_i32.data_ptr()), self._ptr(self.obs),
                                           self._ptr(self.reward), self._ptr(self.done_u8), self._ptr(self.info),
                                           self._ptr(self.final_obs), None, self._stream()), 'prl_batch_step')

    def step_into(self, actions_i32, obs, reward, done_u8, info, final_obs):
        """Like ``step_raw`` but the kernel writes into caller-owned tensors (float64 (N, obs_dim), float64 (N,),
        uint8 (N,), float64 (N, 2), float64 (N, obs_dim)): a rollout worker passes rows of its trajectory
        buffers, so that a step costs two launches and no copies."""
        _lib.check(self.lib.prl_batch_step(self._batch, C.c_void_p(actions_i32.data_ptr()), C.c_void_p(obs.data_ptr()),
                                           C.c_void_p(reward.data_ptr()), C.c_void_p(done_u8.data_ptr()),
                                           C.c_void_p(info.data_ptr()), C.c_void_p(final_obs.data_ptr()), None,
                                           self._stream()), 'prl_batch_step')

    def act_step_into(self, weights, obs_in, rng_count, rng_seed, action, logp, value, obs, reward, done_u8, info,
                      final_obs=None):
        """Policy + env step in ONE launch (``prl_batch_act_step``): the policy (``weights``: a
        ``_lib.PrlPolicyWeights`` of device pointers) reads ``obs_in``, the sampled ``action`` (int32 (N,)), ``logp``
        and ``value`` (float32 (N,)) and the step's rows are written into the caller's tensors.  No checks, no
        allocations: the rollout hot loop.  Needs auto_reset=True, discrete actions, fast paint."""
        p = self._ptr
        rc = self.lib.prl_batch_act_step(self._batch, C.byref(weights), p(obs_in), p(rng_count), C.c_uint64(int(rng_seed)),
                                         p(action), p(logp), p(value), p(obs), p(reward), p(done_u8), p(info),
                                         p(final_obs), self._stream())
        if rc:
            _lib.check(rc, 'prl_batch_act_step')

    def rollout_fragment(self, n_steps, obs, final_obs, reward, done_u8, info, action, weights=None, logp=None,
                         value=None, last_value=None, rng_count=None, rng_seed=0):
        """``n_steps`` steps enqueued by ONE call (``prl_rollout_fragment``).  Without ``weights`` the kernel READS
        ``action`` (int32 (T, N)) and the whole fragment is one persistent launch; with ``weights`` (a
        ``_lib.PrlPolicyWeights`` holding device pointers) every step is one policy-and-step launch that WRITES
        ``action``, ``logp`` / ``value`` (float32 (T, N)), and ``last_value`` (float32 (N,)) comes from a final policy
        pass.  ``obs`` is float64 (T + 1, N, obs_dim) with row 0 = the current observations; ``final_obs``
        float64 (T, N, obs_dim) or None, ``reward`` float64 (T, N), ``done_u8`` uint8 (T, N), ``info`` float64
        (T, N, 2).  Needs auto_reset=True, discrete actions, fast paint."""
        p = self._ptr
        _lib.check(self.lib.prl_rollout_fragment(
            self._batch, C.byref(weights) if weights is not None else None, int(n_steps), p(obs), p(final_obs), p(reward),
            p(done_u8), p(info), p(action), p(logp), p(value), p(last_value), p(rng_count), C.c_uint64(int(rng_seed)),
            self._stream()), 'prl_rollout_fragment')

    # RLlib VectorEnv-style names
    def vector_reset(self):
        return self.reset()

    def reset_at(self, index, start_idx=None):
        torch = _torch()
        m = torch.zeros(self.n_envs, dtype=torch.uint8, device=self.device)
        m[index] = 1
        s = None
        if start_idx is not None:
            s = torch.zeros(self.n_envs, dtype=torch.int32, device=self.device)
            s[index] = int(start_idx)
        return self.reset(mask=m, start_idx=s)[index]

    def vector_step(self, actions):
        return self.step(actions)

    def set_pose(self, index, pose, orn):
        """Robot.reset([pose, orn]) (rob:366-372) for one env: move the tool, clear off-part bookkeeping."""
        from .part_tables import pose_orn_quaternion
        pos = (C.c_double * 3)(*[float(v) for v in pose])
        quat = (C.c_double * 4)(*pose_orn_quaternion([float(v) for v in orn]))
        _lib.check(self.lib.prl_batch_set_pose(self._batch, int(index), pos, quat), 'prl_batch_set_pose')

    def observe(self):
        """_augmented_observation (rge:306-319) of the current state of every env, without stepping."""
        torch = _torch()
        obs = torch.empty((self.n_envs, self.obs_dim), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.prl_batch_observe(self._batch, self._ptr(obs), self._stream()), 'prl_batch_observe')
        return obs

    # ------------------------------------------------------------------ read-back
    def painted_words(self):
        """int64 tensor (N, mask_stride) holding the u64 coverage words in device sample order."""
        torch = _torch()
        out = torch.zeros((self.n_envs, self.mask_stride), dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.prl_batch_get_mask(self._batch, self._ptr(out), self._stream()), 'prl_batch_get_mask')
        return out

    def last_shot_words(self):
        """(last, nonzero): the previous shot's affected set as int64 (N, mask_stride) words in device sample order
        (Part._last_painted_pixels, bpw:483) and the library's index of its non-zero words, int64 (N, nz_stride)."""
        torch = _torch()
        out = torch.zeros((self.n_envs, self.mask_stride), dtype=torch.int64, device=self.device)
        nz = torch.zeros((self.n_envs, max(4, (self.mask_stride + 63) // 64)), dtype=torch.int64, device=self.device)
        stride = C.c_int32(0)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.prl_batch_get_last_mask(self._batch, self._ptr(out), self._ptr(nz), C.byref(stride), self._stream()),
                       'prl_batch_get_last_mask')
        return out, nz.reshape(-1)[:self.n_envs * stride.value].reshape(self.n_envs, stride.value)

    def painted_bits(self, env=0):
        """bool[P] coverage of one env in canonical sample order (PartTables.sample_pix order)."""
        words = self.painted_words()[env].cpu().numpy().view(np.uint64)
        return self.parts[int(self.env_part_id[env])].mask_to_canonical(words)

    def thickness(self, env=None):
        """COLOR_MODE='HSI': uint8 thickness bytes in canonical sample order -- (N, P) array, or (P,) for one env."""
        torch = _torch()
        raw = torch.zeros((self.n_envs, self.mask_stride * 64), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.prl_batch_get_thickness(self._batch, self._ptr(raw), self._stream()),
                       'prl_batch_get_thickness')
        raw = raw.cpu().numpy()
        if env is not None:
            return raw[env][self.parts[int(self.env_part_id[env])].inv_perm]
        if len(self.parts) != 1:
            raise ValueError('thickness() of a whole mixed batch: ask per env')
        return raw[:, self.parts[0].inv_perm]

    def state(self):
        """dict of numpy arrays decoded from the per-env state records (include/paintrl.h)."""
        torch = _torch()
        raw = torch.zeros((self.n_envs, _lib.STATE_DOUBLES), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.prl_batch_get_state(self._batch, self._ptr(raw), self._stream()), 'prl_batch_get_state')
        r = raw.cpu().numpy()
        ints = r.view(np.int32).reshape(self.n_envs, -1)
        return {'pose': r[:, 0:3].copy(), 'quat': r[:, 3:7].copy(), 'last_turning_angle': r[:, 7].copy(),
                'total_reward': r[:, 8].copy(), 'total_return': r[:, 9].copy(),
                'terminate': ints[:, 20].copy(), 'terminate_counter': ints[:, 21].copy(),
                'last_on_part': ints[:, 22].copy(), 'step_counter': ints[:, 23].copy(),
                'episode': ints[:, 24].view(np.uint32).astype(np.uint64), 'facet_hint': ints[:, 25].copy(),
                'last_episode_return': r[:, 13].copy(), 'last_episode_reward': r[:, 14].copy(),
                'last_episode_len': ints[:, 30].copy(), 'last_episode_painted': ints[:, 31].copy()}

    def state_into(self, out):
        """Copy the raw per-env state records into a caller-owned float64 (N, 16) device tensor (stream-ordered)."""
        _lib.check(self.lib.prl_batch_get_state(self._batch, C.c_void_p(out.data_ptr()), self._stream()),
                   'prl_batch_get_state')

    def episode_returns(self):
        """float64 tensor (N,): return of each env's last finished episode (the RCCL gather payload)."""
        torch = _torch()
        out = torch.zeros(self.n_envs, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.prl_batch_get_returns(self._batch, self._ptr(out), self._stream()),
                       'prl_batch_get_returns')
        return out

    def ray_test_batch(self, ray_from, ray_to, part=0):
        """pybullet.rayTestBatch against a part's collision triangles -> (tri, frac, pos) tensors."""
        torch = _torch()
        f = torch.as_tensor(ray_from, dtype=torch.float64, device=self.device).reshape(-1, 3).contiguous()
        t = torch.as_tensor(ray_to, dtype=torch.float64, device=self.device).reshape(-1, 3).contiguous()
        n = f.shape[0]
        tri = torch.zeros(n, dtype=torch.int32, device=self.device)
        frac = torch.zeros(n, dtype=torch.float64, device=self.device)
        pos = torch.zeros((n, 3), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.prl_ray_batch(self._part_handles[part], n, self._ptr(f), self._ptr(t), self._ptr(tri),
                                              self._ptr(frac), self._ptr(pos), self._stream()), 'prl_ray_batch')
        return tri, frac, pos

    def timing(self, every):
        """HIP-event timing of the step kernel around every ``every``-th launch (0/False = off, 1/True = all)."""
        _lib.check(self.lib.prl_batch_timing_enable(self._batch, int(every)), 'prl_batch_timing_enable')

    def timing_read(self):
        ms, n = C.c_double(0), C.c_int64(0)
        _lib.check(self.lib.prl_batch_timing_read(self._batch, C.byref(ms), C.byref(n)), 'prl_batch_timing_read')
        return ms.value, n.value

    def step_occupancy(self):
        """How one prl_batch_step launch occupies the chip: dict(waves_per_workgroup, workgroups_per_cu, waves_per_cu,
        dynamic_lds_bytes, compute_units) from the runtime's occupancy calculator (PAINT_METHOD 'fast' kernels)."""
        out = (C.c_int32 * 4)()
        with _torch().cuda.device(self.device):
            _lib.check(self.lib.prl_batch_step_occupancy(self._batch, out), 'prl_batch_step_occupancy')
        return dict(waves_per_workgroup=out[0], workgroups_per_cu=out[1], waves_per_cu=out[0] * out[1],
                    dynamic_lds_bytes=out[2], compute_units=out[3])

    def close(self):
        if getattr(self, '_batch', None) is not None and self._batch:
            self.lib.prl_batch_destroy(self._batch)
            self._batch = C.c_void_p()
        for h in getattr(self, '_part_handles', []):
            self.lib.prl_part_destroy(h)
        self._part_handles = []

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
