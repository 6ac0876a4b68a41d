"""World-size-2 gloo tests of the env-shard layer (no GPU): shard ranges, per-rank seeds,
the all_gather of episode returns, MAX-over-ranks timing, and shard equivalence through the oracle."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

from conftest import REPO


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from paintrl_amd import distributed as pdist, part_tables
    import oracle
    from conftest import synthetic_tables
    r, lr, w = pdist.init_process_group('gloo')
    assert (r, w) == (rank, world)
    n_total, steps = 8, 12
    lo, hi = pdist.shard_range(n_total, rank, world)
    tables = synthetic_tables('door_test')
    sp = part_tables.start_points(tables, 'all')
    rng = np.random.RandomState(3)                       # same global streams on every rank
    start = rng.randint(0, len(sp), size=n_total)
    acts = rng.randint(0, 4, size=(steps, n_total))
    env = oracle.Oracle(tables, hi - lo, start_points=sp)   # stand-in for the per-rank GPU batch
    env.reset(start[lo:hi])
    for k in range(steps):
        env.step(acts[k][lo:hi])
    local = torch.tensor([env.state(i)['total_return'] for i in range(hi - lo)], dtype=torch.float64)
    gathered = pdist.gather_returns(local)
    tmax = pdist.max_over_ranks(1.0 + rank, torch.device('cpu'))
    pdist.barrier()
    np.save(os.path.join(out_dir, 'rank%d.npy' % rank), gathered.numpy())
    assert tmax == float(world)
    assert pdist.rank_seed(5678, 0) != pdist.rank_seed(5678, 1)
    torch.distributed.destroy_process_group()


def test_two_rank_shard_equals_single_rank(tmp_path):
    import oracle
    from conftest import synthetic_tables
    from paintrl_amd import part_tables
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g0, g1 = np.load(tmp_path / 'rank0.npy'), np.load(tmp_path / 'rank1.npy')
    assert np.array_equal(g0, g1) and g0.shape == (8,)
    # the gathered per-env returns equal one rank running all 8 envs
    tables = synthetic_tables('door_test')
    sp = part_tables.start_points(tables, 'all')
    rng = np.random.RandomState(3)
    start = rng.randint(0, len(sp), size=8)
    acts = rng.randint(0, 4, size=(12, 8))
    env = oracle.Oracle(tables, 8, start_points=sp)
    env.reset(start)
    for k in range(12):
        env.step(acts[k])
    want = np.array([env.state(i)['total_return'] for i in range(8)])
    assert np.array_equal(g0, want)


def test_shard_range_validation():
    from paintrl_amd import distributed as pdist
    assert pdist.shard_range(32768, 3, 8) == (12288, 16384)
    try:
        pdist.shard_range(10, 0, 4)
    except ValueError:
        pass
    else:
        raise AssertionError('uneven shard must be rejected')
