#!/usr/bin/env python3
"""Benchmark of the batched paint-coverage step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): synthetic door panel
(seed-0 generator, 9664 front samples, 742 hull facets), OBS_MODE='section',
OBS_GRAD=4, START_POINT_MODE='anchor', TERMINATION_MODE='late',
EPISODE_MAX_LENGTH=245, fast paint, 4096 envs per GPU, random discrete-4 actions
from a device generator seeded 1234, auto-reset inside the step kernel with the
library's counter-based start-point RNG (seed 5678).  One "step" = one batched
PaintGymEnv.step() of all 4096 envs of a GPU = one launch of step_kernel.

For N > 1 the driver starts one process per GPU with torch.distributed.run; envs
are sharded (weak scaling, 4096 per GPU, no data-path collective) and every 100
steps the ranks all_gather their episode returns over RCCL.

Prints ONE JSON line on rank 0 (see README / the task contract), including
"roofline" for step_kernel and "cpu_baseline" (the C oracle on the host cores).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

ENVS_PER_GPU = 4096
FRAGMENT = 100                  # rollout fragment length (paint_ppo.py:190 sample_batch_size)
TIMING_EVERY = 8                # HIP events around every 8th step launch (kernel time for the roofline)
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s


def algorithmic_bytes(dt, n_envs, obs_dim):
    """Minimal HBM bytes of one step launch (DESIGN.md "Roofline accounting")."""
    words = dt.n_words
    per_env = (2 * 2 * words * 8          # painted + last-shot masks, read and write
               + 2 * 16 * 8               # 128-byte scalar state record, read and write
               + 4                        # action (int32)
               + obs_dim * 8 + 8 + 1 + 16)  # obs, reward, done, info (float64 like the reference)
    static = (3 * dt.n_samples_pad * 8 + dt.word_bbox.nbytes + dt.word_valid.nbytes + dt.sample_rank.nbytes + dt.sgrid_start.nbytes
              + sum(a.nbytes for a in dt.vertex_xyz) + dt.vertex_rank.nbytes + dt.vertex_adj.nbytes
              + dt.vgrid_start.nbytes + dt.tri_records.nbytes
              + sum(a.nbytes for a in dt.col) + dt.col_bbox.nbytes + dt.col_rank.nbytes + dt.col_nbr.nbytes + dt.col_orient.nbytes + dt.col_chunk_bbox.nbytes + dt.grid_lo.nbytes + dt.grid_hi.nbytes
              + dt.start_pos.nbytes + dt.start_quat.nbytes
              + dt.n_samples_pad                                                # equal-run ends (u8), derived on upload
              + (dt.n_collision_pad * 96 if dt.col_convex else 0))              # hull facet records, derived on upload
    return per_env, static, per_env * n_envs + static


def usable_cores():
    """Host cores this process may really use: affinity mask and cgroup cpu quota both count."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(tables, steps_sample=60, n_envs=ENVS_PER_GPU):
    """The C oracle (a scalar float64 port of the reference step()) on all host cores."""
    import numpy as np
    import oracle
    cores = usable_cores()
    orc = oracle.Oracle(tables, n_envs, threads=cores)
    rng = np.random.RandomState(1234)
    start = rng.randint(0, 4, size=n_envs)
    orc.reset(start)
    acts = rng.randint(0, 4, size=(steps_sample, n_envs))
    t0 = time.perf_counter()
    for k in range(steps_sample):
        _, _, done, _ = orc.step(acts[k])
        if done.any():
            orc.reset(rng.randint(0, 4, size=n_envs), mask=done)
    dt = time.perf_counter() - t0
    return {'value': steps_sample / dt, 'unit': 'batched steps/s (4096 envs each)', 'cores': cores,
            'kind': 'port', 'env_steps_per_s': steps_sample * n_envs / dt,
            'sample': '%d batched steps of %d envs (same door, same action distribution, reset on done), '
                      'oracle/paint_oracle.c with OpenMP over envs, %.1f s wall' % (steps_sample, n_envs, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=200)
    ap.add_argument('--envs', type=int, default=ENVS_PER_GPU, help='envs per GPU (default: the BASELINE config)')
    ap.add_argument('--obs-mode', default='section', choices=['section', 'grid'])
    ap.add_argument('--mixed', action='store_true', help='config 5: door/sheet alternate, START_POINT_MODE all')
    ap.add_argument('--actions', default='random', choices=['random', 'sweep'],
                    help="'sweep': every env follows an on-part serpentine with a random phase (SURVEY 8d item 2), so "
                         'episodes run long instead of ending after ~17 random steps')
    ap.add_argument('--policy', default='random', choices=['random', 'mlp', 'mlp-torch'],
                    help="'mlp': actions from the 6-256-128-4 policy network on the same stream (config 4), one fused "
                         "kernel per step (prl_policy_act); 'mlp-torch': the same net in torch eager")
    ap.add_argument('--streams', type=int, default=1,
                    help='issue a batched step as S launches of envs/S envs on S streams (independent env groups: a '
                         "group's slowest wave then only delays that group); default 1 = one launch per step")
    ap.add_argument('--graph', action='store_true', help='with --policy mlp: capture policy + env step in one HIP graph')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    import numpy as np
    import torch
    from paintrl_amd import distributed as pdist
    from paintrl_amd import part_tables, synth_parts
    from paintrl_amd.batched_env import BatchedPaintEnv
    from paintrl_amd.device_tables import DeviceTables

    rank, local_rank, world = pdist.init_process_group()
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)' % (args.gpus, world))
    if os.environ.get('PAINTRL_SINGLE_DEVICE'):          # testing aid: all ranks share GPU 0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)

    tables = part_tables.build_part_tables(mesh=synth_parts.synthetic_mesh('door_test'), tex_size=(240, 240),
                                           name='door_test')
    start_mode = 'all' if args.actions == 'sweep' else 'anchor'      # the sweep starts anywhere on the part
    dt = DeviceTables(tables, obs_grad=4, start_points=part_tables.start_points(tables, start_mode))
    overlap = args.obs_mode == 'grid'
    if args.mixed:
        sheet = part_tables.build_part_tables(mesh=synth_parts.synthetic_mesh('square'), tex_size=(240, 240),
                                              name='square')
        dt = DeviceTables(tables, obs_grad=4, start_points=part_tables.start_points(tables, 'all'))
        dts = DeviceTables(sheet, obs_grad=4, start_points=part_tables.start_points(sheet, 'all'))
        env = BatchedPaintEnv([dt, dts], args.envs, env_part_id=np.arange(args.envs) % 2, device=device,
                              obs_mode=args.obs_mode, obs_grad=4, auto_reset=True, overlap_penalty=overlap,
                              seed=pdist.rank_seed(5678, rank), max_possible_point=[9148, 14350])
    else:
        if args.streams > 1 and (args.policy != 'random' or args.envs % args.streams):
            raise SystemExit('--streams needs --policy random and a divisible --envs')
        n_sub = args.envs // args.streams
        subs = [BatchedPaintEnv(dt, n_sub, device=device, obs_mode=args.obs_mode, obs_grad=4, auto_reset=True,
                                overlap_penalty=overlap, seed=pdist.rank_seed(5678 + 7919 * g, rank), max_possible_point=9148)
                for g in range(args.streams)]
        env = subs[0]
    if args.mixed:
        if args.streams > 1:
            raise SystemExit('--streams is not combined with --mixed')
        subs = [env]
    sub_streams = [torch.cuda.Stream(device=device) for _ in subs] if len(subs) > 1 else [None]
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + rank)
    total = args.steps + args.warmup
    actions = torch.randint(0, 4, (total, args.envs), generator=gen, device=device, dtype=torch.int32)
    if args.actions == 'sweep':
        pattern = torch.tensor(([1] * 12 + [0] * 2 + [3] * 12 + [0] * 2), dtype=torch.int32, device=device)
        phase = torch.randint(0, pattern.numel(), (args.envs,), generator=gen, device=device)
        steps_idx = torch.arange(total, device=device).unsqueeze(1)
        actions = pattern[(steps_idx + phase.unsqueeze(0)) % pattern.numel()].contiguous()
    for e in subs:
        e.reset()
    stream_sync = torch.cuda.synchronize
    n_sub = args.envs // len(subs)
    sub_actions = [actions] if len(subs) == 1 else [actions[:, g * n_sub:(g + 1) * n_sub].contiguous()
                                                     for g in range(len(subs))]

    policy = None
    if args.policy != 'random':
        from paintrl_amd.rollout import MLPPolicy
        torch.manual_seed(1234)
        torch_policy = MLPPolicy(env.obs_dim, 4).to(device)
        if args.policy == 'mlp':
            from paintrl_amd.policy import FusedPolicy
            fused = FusedPolicy(torch_policy)

            class _Fused(object):                   # same call shape as MLPPolicy.act, float64 observations in
                def act(self, obs32, generator=None):
                    return fused.act(env.obs)        # in-kernel sampling stream (graph-safe, no torch.rand launch)
            policy = _Fused()
        else:
            policy = torch_policy

    def obs_for_policy():
        # the torch module wants float32; the fused kernel reads the env's float64 observations itself
        return env.obs.to(torch.float32) if args.policy == 'mlp-torch' else None

    graph = None
    if policy is not None and args.graph:
        # policy forward + sampling + env step captured once; the launch-bound inner loop becomes a replay
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for _ in range(3):                      # warm the allocator and the kernels on the side stream
                act, _, _ = policy.act(obs_for_policy())
                env.step_raw(act)
        torch.cuda.current_stream(device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            act, _, _ = policy.act(obs_for_policy())
            env.step_raw(act)

    def run(k0, k1):
        for k in range(k0, k1):
            if graph is not None:
                graph.replay()
                continue
            if policy is None and len(subs) > 1:
                for g, e in enumerate(subs):
                    with torch.cuda.stream(sub_streams[g]):
                        e.step_raw(sub_actions[g][k])
            elif policy is None:
                env.step_raw(actions[k])
            else:
                act, _, _ = policy.act(obs_for_policy(), gen)
                env.step_raw(act)
            if world > 1 and (k + 1) % FRAGMENT == 0:
                if len(subs) > 1:
                    stream_sync()
                pdist.gather_returns(torch.cat([e.episode_returns() for e in subs]))

    run(0, args.warmup)
    stream_sync()
    pdist.barrier()
    stream_sync()
    for e in subs:
        e.timing(TIMING_EVERY)
    t0 = time.perf_counter()
    run(args.warmup, total)
    stream_sync()
    pdist.barrier()
    stream_sync()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = 0.0, 0
    for e in subs:
        ms_, n_ = e.timing_read()
        kernel_ms, launches = kernel_ms + ms_, launches + n_
        e.timing(0)
    elapsed = pdist.max_over_ranks(elapsed, device)

    episodes = sum(int(e.state()['episode'].sum()) for e in subs) - args.envs
    if rank == 0:
        per_env, static, per_launch = algorithmic_bytes(dt, args.envs // len(subs), env.obs_dim)   # per LAUNCH
        # inside a captured HIP graph (--graph) the per-launch events are not recorded: no kernel time then
        avg_kernel_s = kernel_ms / launches / 1e3 if launches and kernel_ms > 0 else None
        achieved = per_launch / avg_kernel_s / 1e9 if avg_kernel_s else None
        traffic = None
        tpath = os.path.join(REPO, 'profiles', 'hbm_traffic.json')
        # the committed PMC measurement is for the default workload only
        if os.path.isfile(tpath) and args.envs == ENVS_PER_GPU and not args.mixed and args.actions == 'random' and len(subs) == 1:
            with open(tpath) as f:
                traffic = json.load(f).get('bytes_per_launch_%s' % args.obs_mode)
        value = world * args.steps / elapsed
        out = {
            'metric': 'batched env steps/sec (door panel, N=4096)', 'value': value,
            'unit': 'batched steps/s (%d envs each)' % args.envs, 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'PaintGymEnv Part_NO=0 synthetic door panel, OBS_MODE=%r, %d envs per GPU, '
                                   '%s discrete-4 actions, in-kernel auto-reset' % (args.obs_mode, args.envs, ('policy-MLP (6-256-128-4, fp32, %s)' % ('fused HIP kernel' if args.policy == 'mlp' else 'torch eager')) if args.policy != 'random' else ('on-part serpentine' if args.actions == 'sweep' else 'random')),
                       'envs_per_gpu': args.envs, 'env_steps_per_s': value * args.envs,
                       'samples': int(dt.n_samples), 'collision_triangles': int(dt.n_collision),
                       'episodes_finished_rank0': episodes, 'launches_per_step': len(subs),
                       'parallelism': 'env-shard x%d' % world + (', %d independent env groups per GPU on %d streams' % (len(subs), len(subs)) if len(subs) > 1 else '')},
            'roofline': {'bound': 'hbm', 'kernel': 'step_kernel', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS if achieved else None, 'traffic': traffic,
                         'algorithmic_bytes_per_env_step': per_env, 'static_table_bytes': static,
                         'bytes_per_launch': per_launch, 'avg_kernel_us': avg_kernel_s * 1e6 if avg_kernel_s else None,
                         'launches_timed': int(launches)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(tables, n_envs=args.envs)
        print(json.dumps(out, default=lambda o: o.item() if hasattr(o, 'item') else str(o)))
    for e in subs:
        e.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
