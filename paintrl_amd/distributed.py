"""Multi-GPU sharding of independent environments.

Envs are independent, so the path shards with no data-path collective: global
env id e lives on rank e // envs_per_rank, every rank holds a replica of the
part tables, actions are produced rank-locally.  The only exchange is the
gather of per-env episode returns once per rollout fragment (the analogue of
RLlib's worker -> learner metrics flow, paint_ppo.py:36-61): one all_gather of
envs_per_rank float64 values over RCCL (backend "nccl" on ROCm) or gloo on CPU.
"""
import os

import torch
import torch.distributed as dist


def env_rank():
    return int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')), \
        int(os.environ.get('WORLD_SIZE', '1'))


def force_dist():
    """PAINTRL_FORCE_DIST=1: take the torch.distributed path at world size 1 too (a one-rank RCCL communicator runs
    every collective of this module -- init, the device-side all_gather on the side stream, barrier(device_ids),
    the MAX all_reduce -- on a single GPU; without it a lone rank skips torch.distributed altogether)."""
    return os.environ.get('PAINTRL_FORCE_DIST', '') not in ('', '0')


def init_process_group(backend=None):
    """Initialise torch.distributed from the torchrun environment (no-op for world size 1 unless PAINTRL_FORCE_DIST)."""
    rank, local_rank, world = env_rank()
    if (world > 1 or force_dist()) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            # PAINTRL_DIST_BACKEND=gloo is a testing aid (several ranks on one GPU; RCCL refuses that)
            backend = os.environ.get('PAINTRL_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)       # RCCL binds the communicator to the current device
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(n_total, rank, world):
    """Contiguous block of global env ids owned by ``rank`` (n_total must divide evenly)."""
    if n_total % world:
        raise ValueError('n_total=%d is not divisible by world size %d' % (n_total, world))
    per = n_total // world
    return rank * per, (rank + 1) * per


def rank_seed(base_seed, rank):
    """Decorrelate the per-rank start-point RNG streams (the kernel hashes seed ^ env ^ episode)."""
    return (int(base_seed) + 0x9E3779B97F4A7C15 * (rank + 1)) & 0xFFFFFFFFFFFFFFFF


def gather_returns(local_returns):
    """all_gather the per-env episode returns of every rank -> tensor (world * n_local,).

    ``local_returns`` is a 1-D tensor on the rank's device (cuda with RCCL, cpu with gloo)."""
    if not dist.is_initialized():
        return local_returns.clone()
    world = dist.get_world_size()
    if dist.get_backend() == 'gloo' and local_returns.is_cuda:        # gloo moves host memory
        host = local_returns.detach().cpu().contiguous()
        out = torch.empty(world * host.numel(), dtype=host.dtype)
        dist.all_gather_into_tensor(out, host)
        return out.to(local_returns.device)
    out = torch.empty(world * local_returns.numel(), dtype=local_returns.dtype, device=local_returns.device)
    dist.all_gather_into_tensor(out, local_returns.contiguous())
    return out


class ReturnsGatherer(object):
    """The per-fragment all_gather of episode returns, issued on a SIDE stream so that it overlaps the next
    fragment's step launches; ``wait()`` hands back the last gathered tensor.

    submit(local) orders the side stream after the producer of ``local`` (the caller's current stream), runs the
    collective there and returns immediately; the main stream never waits for RCCL."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.cuda = self.device.type == 'cuda'
        self.side = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.result = None
        self.count = 0

    def submit(self, local_returns):
        self.count += 1
        if not self.cuda:
            self.result = gather_returns(local_returns)
            return
        main = torch.cuda.current_stream(self.device)
        self.side.wait_stream(main)
        with torch.cuda.stream(self.side):
            local_returns.record_stream(self.side)
            self.result = gather_returns(local_returns)

    def wait(self):
        if self.cuda:
            torch.cuda.current_stream(self.device).wait_stream(self.side)
        return self.result


def max_over_ranks(value, device):
    """MAX-reduce a python float over ranks (used for the timed region of bench.py)."""
    if not dist.is_initialized():
        return float(value)
    if dist.get_backend() == 'gloo':
        device = 'cpu'
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized():
        if dist.get_backend() == 'nccl':             # name the device: RCCL otherwise guesses (and warns)
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()
