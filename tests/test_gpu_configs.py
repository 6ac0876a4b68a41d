"""BASELINE.json configs 3-5 at their per-GPU size (4096 envs), against the oracle, plus the N > 1 launch path.

config 3: door, OBS_MODE='grid' + OVERLAP_PENALTY;  config 5 (one GPU's share): door / sheet alternating in one
batch with START_POINT_MODE='all';  config 4/5 multi-rank: two ranks on this one device over gloo (RCCL refuses
two ranks on one GPU), each running the real BatchedPaintEnv, gathered returns equal a single-rank run.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle
from conftest import REPO, start_points_for, synthetic_tables

pytestmark = pytest.mark.gpu


def _dt(tables, sp, obs_grad=4):
    from paintrl_amd.device_tables import DeviceTables
    return DeviceTables(tables, obs_grad=obs_grad, start_points=sp)


def test_full_size_grid_overlap_equals_oracle():
    """config 3 at N = 4096: every observation (16 cells), reward, penalty (overlap term), done flag, painted bit."""
    from paintrl_amd.batched_env import BatchedPaintEnv
    tables = synthetic_tables('door_test')
    sp = start_points_for(tables, 'all')
    n, steps = 4096, 4
    kw = dict(obs_mode='grid', obs_grad=4, overlap_penalty=True)
    env = BatchedPaintEnv(_dt(tables, sp), n, **kw)
    orc = oracle.Oracle(tables, n, start_points=sp, threads=8, **kw)
    rng = np.random.RandomState(303)
    start = rng.randint(0, len(sp), size=n)
    assert np.array_equal(env.reset(start_idx=start).cpu().numpy(), orc.reset(start))
    saw_overlap = False
    for k in range(steps):
        a = rng.randint(0, 4, size=n)
        o, r, d, i = env.step(a)
        oo, rr, dd, ii = orc.step(a)
        assert o.shape == (n, 16)
        assert np.array_equal(o.cpu().numpy(), oo) and np.array_equal(r.cpu().numpy(), rr), 'step %d' % k
        assert np.array_equal(d.cpu().numpy(), dd) and np.array_equal(i.cpu().numpy(), ii), 'step %d' % k
        saw_overlap = saw_overlap or bool(((ii[:, 1] > 0.2) & (ii[:, 1] < 0.3)).any())
    assert saw_overlap                      # the overlap term really was exercised (0.2 < penalty < 0.3)
    words = env.painted_words().cpu().numpy().view(np.uint64)
    bits = env.parts[0].mask_to_canonical(words)
    assert np.array_equal(bits, np.stack([orc.painted_bits(e) for e in range(n)]))
    env.close()


def test_full_size_mixed_parts_all_starts_equals_oracle():
    """config 5's per-GPU share at N = 4096: env i paints the door if i is even, the sheet if odd, every env
    starts from its own draw of the part's 'all' start-point table (1395 / 396 entries)."""
    from paintrl_amd.batched_env import BatchedPaintEnv
    door, sheet = synthetic_tables('door_test'), synthetic_tables('square')
    sp_d, sp_s = start_points_for(door, 'all'), start_points_for(sheet, 'all')
    assert len(sp_d) > 1000 and len(sp_s) > 300
    n, steps = 4096, 4
    ids = (np.arange(n) % 2).astype(np.int32)
    env = BatchedPaintEnv([_dt(door, sp_d), _dt(sheet, sp_s)], n, env_part_id=ids, max_possible_point=[9148, 14350])
    od = oracle.Oracle(door, n // 2, start_points=sp_d, max_possible_point=9148, threads=8)
    os_ = oracle.Oracle(sheet, n // 2, start_points=sp_s, max_possible_point=14350, threads=8)
    rng = np.random.RandomState(505)
    start = np.where(ids == 0, rng.randint(0, len(sp_d), size=n), rng.randint(0, len(sp_s), size=n))
    obs = env.reset(start_idx=start).cpu().numpy()
    assert np.array_equal(obs[0::2], od.reset(start[0::2])) and np.array_equal(obs[1::2], os_.reset(start[1::2]))
    for k in range(steps):
        a = rng.randint(0, 4, size=n)
        o, r, d, i = env.step(a)
        o, r, d, i = o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy(), i.cpu().numpy()
        o1, r1, d1, i1 = od.step(a[0::2])
        o2, r2, d2, i2 = os_.step(a[1::2])
        assert np.array_equal(o[0::2], o1) and np.array_equal(o[1::2], o2), 'obs, step %d' % k
        assert np.array_equal(r[0::2], r1) and np.array_equal(r[1::2], r2)
        assert np.array_equal(d[0::2], d1) and np.array_equal(d[1::2], d2)
        assert np.array_equal(i[0::2], i1) and np.array_equal(i[1::2], i2)
    words = env.painted_words().cpu().numpy().view(np.uint64)
    bd = env.parts[0].mask_to_canonical(words[0::2])
    bs = env.parts[1].mask_to_canonical(words[1::2])
    assert np.array_equal(bd, np.stack([od.painted_bits(e) for e in range(n // 2)]))
    assert np.array_equal(bs, np.stack([os_.painted_bits(e) for e in range(n // 2)]))
    env.close()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


_RANK_SCRIPT = r'''
import os, sys
import numpy as np
sys.path.insert(0, %(repo)r); sys.path.insert(0, os.path.join(%(repo)r, 'tests'))
import torch
from paintrl_amd import distributed as pdist, part_tables
from paintrl_amd.batched_env import BatchedPaintEnv
from paintrl_amd.device_tables import DeviceTables
from conftest import synthetic_tables
rank, local_rank, world = pdist.init_process_group('gloo')
torch.cuda.set_device(0)
n_total, steps = 256, 30
lo, hi = pdist.shard_range(n_total, rank, world)
tables = synthetic_tables('door_test')
sp = part_tables.start_points(tables, 'all')
rng = np.random.RandomState(17)                      # the same global streams on every rank
start = rng.randint(0, len(sp), size=n_total)
nxt = rng.randint(0, len(sp), size=(steps, n_total))
acts = rng.randint(0, 4, size=(steps, n_total))
env = BatchedPaintEnv(DeviceTables(tables, start_points=sp), hi - lo, device='cuda:0', auto_reset=True)
env.reset(start_idx=start[lo:hi])
g = pdist.ReturnsGatherer('cuda:0')
for k in range(steps):
    env.step(acts[k][lo:hi], start_idx=nxt[k][lo:hi])
    if (k + 1) %% 10 == 0:
        g.submit(env.episode_returns())
out = g.wait()
torch.cuda.synchronize()
pdist.barrier()
np.save(os.path.join(%(out)r, 'rank%%d.npy' %% rank), out.cpu().numpy())
np.save(os.path.join(%(out)r, 'count%%d.npy' %% rank), np.array([g.count, torch.distributed.get_world_size()]))
env.close()
torch.distributed.destroy_process_group()
'''


def test_two_ranks_on_one_device_equal_single_rank(tmp_path):
    """The N > 1 path with the real GPU env per rank (gloo, both ranks on cuda:0): the all_gather of episode
    returns, issued on the side stream every 10 steps, equals what one rank running all envs reports."""
    from paintrl_amd import part_tables
    from paintrl_amd.batched_env import BatchedPaintEnv
    script = tmp_path / 'rank.py'
    script.write_text(_RANK_SCRIPT % dict(repo=REPO, out=str(tmp_path)))
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), PAINTRL_DIST_BACKEND='gloo')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    g0, g1 = np.load(tmp_path / 'rank0.npy'), np.load(tmp_path / 'rank1.npy')
    assert np.array_equal(g0, g1) and g0.shape == (256,)
    assert list(np.load(tmp_path / 'count0.npy')) == [3, 2]
    tables = synthetic_tables('door_test')
    sp = part_tables.start_points(tables, 'all')
    rng = np.random.RandomState(17)
    start = rng.randint(0, len(sp), size=256)
    nxt = rng.randint(0, len(sp), size=(30, 256))
    acts = rng.randint(0, 4, size=(30, 256))
    env = BatchedPaintEnv(_dt(tables, sp), 256, auto_reset=True)
    env.reset(start_idx=start)
    for k in range(30):
        env.step(acts[k], start_idx=nxt[k])
    want = env.episode_returns().cpu().numpy()
    assert (want != 0).sum() > 50                         # plenty of episodes finished
    assert np.array_equal(g0, want)
    env.close()


@pytest.mark.parametrize('ranks', [2, 4])
def test_bench_self_launches_its_ranks(tmp_path, ranks):
    """`python bench.py --gpus 2` started plainly spawns its own ranks (here both on GPU 0 over gloo) and rank 0
    prints the one JSON line with the world size it really ran at."""
    env = dict(os.environ, PAINTRL_SINGLE_DEVICE='1', PAINTRL_DIST_BACKEND='gloo')
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', str(ranks), '--steps', '120', '--warmup',
                          '10', '--envs', '512', '--no-cpu-baseline'], env=env, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == ranks and 'world size %d' % ranks in rec['config']['parallelism']
    assert rec['value'] > 0 and rec['roofline']['avg_kernel_us'] <= 1e3 * rec['ms_per_step'] * 1.001


_RCCL_SCRIPT = r'''
import os, sys
import numpy as np
sys.path.insert(0, %(repo)r); sys.path.insert(0, os.path.join(%(repo)r, 'tests'))
import torch
import torch.distributed as dist
from paintrl_amd import distributed as pdist, part_tables
rank, local_rank, world = pdist.init_process_group()          # before anything else touches the GPU
assert dist.is_initialized() and dist.get_backend() == 'nccl' and world == 1, (dist.is_initialized(), world)
from paintrl_amd.batched_env import BatchedPaintEnv
from paintrl_amd.device_tables import DeviceTables
from conftest import synthetic_tables
torch.cuda.set_device(0)
tables = synthetic_tables('door_test')
sp = part_tables.start_points(tables, 'all')
n, steps = 512, 40
rng = np.random.RandomState(23)
env = BatchedPaintEnv(DeviceTables(tables, start_points=sp), n, device='cuda:0', auto_reset=True, seed=9)
env.reset(start_idx=rng.randint(0, len(sp), size=n))
g = pdist.ReturnsGatherer('cuda:0')
for k in range(steps):
    env.step(rng.randint(0, 4, size=n))
    if (k + 1) %% 10 == 0:
        g.submit(env.episode_returns())                        # device-side all_gather_into_tensor on the side stream
out = g.wait()
torch.cuda.synchronize()
local = env.episode_returns()
assert out.is_cuda and out.shape == (n,) and torch.equal(out, local) and int((local != 0).sum()) > 50
pdist.barrier()                                                # dist.barrier(device_ids=[...])
assert pdist.max_over_ranks(1.25, 'cuda:0') == 1.25            # the MAX all_reduce of the timed region
assert g.count == 4
env.close()
dist.destroy_process_group()
print('RCCL_WORLD1_OK backend=%%s' %% 'nccl')
'''


def test_rccl_communicator_runs_the_gather_on_one_gpu(tmp_path):
    """The `nccl` (= RCCL) branch of paintrl_amd/distributed.py on the one GPU there is: PAINTRL_FORCE_DIST=1 makes a
    one-rank communicator, and init, the device-side all_gather of episode returns on the side stream, the barrier with
    device_ids and the MAX all_reduce all execute (a fresh child process: the group is initialised before any GPU call)."""
    script = tmp_path / 'rccl1.py'
    script.write_text(_RCCL_SCRIPT % dict(repo=REPO))
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()),
               PAINTRL_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    env.pop('PAINTRL_DIST_BACKEND', None)
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert 'RCCL_WORLD1_OK backend=nccl' in out.stdout


def test_bench_reports_the_rccl_backend_under_force_dist():
    """`bench.py --gpus 1` with PAINTRL_FORCE_DIST=1 takes the same distributed path the N > 1 runs take (barriers, MAX over
    ranks, the per-fragment returns gather) over a one-rank RCCL communicator, and says so in its JSON line."""
    env = dict(os.environ, PAINTRL_FORCE_DIST='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'PAINTRL_DIST_BACKEND'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '1', '--steps', '60', '--warmup', '10',
                          '--envs', '1024', '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][0])
    assert rec['n_gpus'] == 1 and 'backend nccl' in rec['config']['parallelism']
    assert rec['config']['returns_gathers'] == 3 and rec['value'] > 0
