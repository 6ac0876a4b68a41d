"""World-size-2 and -8 gloo tests of the env-shard layer (no GPU): shard ranges, per-rank seeds,
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
    n_total, steps = 16, 12
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
    assert len({pdist.rank_seed(5678, r) for r in range(world)}) == world
    torch.distributed.destroy_process_group()


import pytest


@pytest.mark.parametrize('world', [2, 8])
def test_sharded_ranks_equal_single_rank(tmp_path, world):
    """World size 2 and 8 (the node BASELINE.json's config 4 names): every rank's all_gather of the episode returns
    equals one rank running all the envs."""
    import oracle
    from conftest import synthetic_tables
    from paintrl_amd import part_tables
    oracle.build()                                        # once, before the ranks start
    synthetic_tables('door_test')
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g = [np.load(tmp_path / ('rank%d.npy' % r)) for r in range(world)]
    assert all(np.array_equal(g[0], x) for x in g[1:]) and g[0].shape == (16,)
    # the gathered per-env returns equal one rank running all 16 envs
    tables = synthetic_tables('door_test')
    sp = part_tables.start_points(tables, 'all')
    rng = np.random.RandomState(3)
    start = rng.randint(0, len(sp), size=16)
    acts = rng.randint(0, 4, size=(12, 16))
    env = oracle.Oracle(tables, 16, start_points=sp)
    env.reset(start)
    for k in range(12):
        env.step(acts[k])
    want = np.array([env.state(i)['total_return'] for i in range(16)])
    assert np.array_equal(g[0], want)


def test_shard_range_validation():
    from paintrl_amd import distributed as pdist
    assert pdist.shard_range(32768, 3, 8) == (12288, 16384)
    try:
        pdist.shard_range(10, 0, 4)
    except ValueError:
        pass
    else:
        raise AssertionError('uneven shard must be rejected')


def test_bench_launcher_names_the_failing_rank():
    """`python bench.py --gpus 8` started plainly becomes the launcher of its 8 ranks; here (no GPU) every rank fails at
    once, and the launcher must stop the others, name a failed rank and pass its status on."""
    import subprocess
    if torch.cuda.is_available():
        import pytest
        pytest.skip('needs a box without a GPU: the ranks must fail')
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '8', '--steps', '2', '--warmup', '1',
                          '--envs', '64', '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert 'of 8 exited with status' in out.stderr, out.stderr[-1500:]


def _forced_single_rank(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      PAINTRL_FORCE_DIST='1', PAINTRL_DIST_BACKEND='gloo')
    from paintrl_amd import distributed as pdist
    r, lr, w = pdist.init_process_group()
    assert torch.distributed.is_initialized() and (r, w) == (0, 1)
    local = torch.arange(5, dtype=torch.float64)
    g = pdist.ReturnsGatherer('cpu')
    g.submit(local)
    assert torch.equal(g.wait(), local) and g.count == 1
    assert pdist.max_over_ranks(2.5, torch.device('cpu')) == 2.5
    pdist.barrier()
    torch.distributed.destroy_process_group()
    open(os.path.join(out_dir, 'ok'), 'w').write('ok')


def test_force_dist_takes_the_collective_path_at_world_size_one(tmp_path):
    """PAINTRL_FORCE_DIST=1: a lone rank still initialises torch.distributed and runs the gather / barrier / MAX
    all_reduce through it (gloo here; the RCCL run of the same switch is tests/test_gpu_configs.py)."""
    mp.spawn(_forced_single_rank, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    assert (tmp_path / 'ok').read_text() == 'ok'
