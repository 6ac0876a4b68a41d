"""Benchmark of the batched paint-coverage step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): synthetic door panel
(seed-0 generator, 9664 front samples, 742 hull facets), OBS_MODE='section',
OBS_GRAD=4, START_POINT_MODE='anchor', TERMINATION_MODE='late',
EPISODE_MAX_LENGTH=245, fast paint, 4096 envs per GPU, random discrete-4 actions
from a device generator seeded 1234, auto-reset inside the step kernel with the
library's counter-based start-point RNG (seed 5678).  One "step" = one batched
PaintGymEnv.step() of all 4096 envs of a GPU = one launch of step_kernel.

N > 1: one process per GPU.  Under `python -m torch.distributed.run` the ranks come
from the environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*); started plainly
as `python bench.py --gpus N` the parent -- which never touches the GPU -- starts
the N ranks itself as child processes and exits with their status (the analogue of
the reference's ray.init + num_workers, paint_ppo.py:153,171).  Envs are sharded
(weak scaling, 4096 per GPU, no data-path collective) and every 100 steps the ranks
all_gather their episode returns over RCCL on a side stream.

Prints ONE JSON line on rank 0 (see README / the task contract), including
"roofline" for step_kernel and "cpu_baseline" (the C oracle on the host cores).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

ENVS_PER_GPU = 4096
FRAGMENT = 100                  # rollout fragment length (paint_ppo.py:190 sample_batch_size)
TIMING_EVERY = 8                # runs of > SAMPLE_ABOVE steps also bracket every 8th launch with its own HIP event
SAMPLE_ABOVE = 256              # pair (a pair costs ~3.5 us of stream time, so short runs are left unperturbed)
REPEATS = 5                     # the timed region (K steps) is run this many times; value = the median
PREWARM_SECONDS = 0.4           # untimed stepping of a scratch batch first: the timed region then runs at
                                # the clocks a long job holds, also at the driver's --warmup 5 --steps 20
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s
CLOCK_GHZ = 2.4                 # MI355X_MICROARCH.md: max shader clock
N_SIMD = 256 * 4                # 256 CUs x 4 SIMDs


def algorithmic_bytes(dt, n_envs, obs_dim):
    """(survey_per_env, survey_per_launch, layout_per_env, layout_static, layout_per_launch).

    `survey_*` is SURVEY.md §8(d)'s figure (what `roofline.achieved` uses): the reference's state as 1 bit per
    sample in 32-bit words of the UNPADDED sample count, 52 bytes of scalars, f32 outputs, 0.4 MB of tables.
    `layout_*` is what this implementation's own layout must move at minimum: word-aligned 64-bit mask rows, the
    128-byte record, f64 outputs, and every static table once."""
    mask32 = ((dt.n_samples + 31) // 32) * 4
    survey_env = 4 * mask32 + 2 * 52 + 4 + obs_dim * 4 + 4 + 1 + 8
    survey_launch = survey_env * n_envs + 400000
    words = dt.n_words
    per_env = (2 * 2 * words * 8          # painted + last-shot masks, read and write
               + 2 * 16 * 8               # 128-byte scalar state record, read and write
               + 4                        # action (int32)
               + obs_dim * 8 + 8 + 1 + 16)  # obs, reward, done, info (float64 like the reference)
    n_tri = dt.tri_records.shape[0]
    static = (3 * dt.n_samples_pad * 8 + dt.word_bbox.nbytes + dt.word_valid.nbytes + dt.sample_rank.nbytes + dt.sgrid_start.nbytes
              + dt.n_samples_pad * 16                                           # float copy of the samples (paint pre-filter)
              + dt.vertex_rank.shape[0] * 32 + dt.vertex_adj.nbytes             # x y z | rank vertex records
              + dt.vgrid_start.nbytes + n_tri * 24 * 8                          # triangle records incl. quaternion / centre tail
              + sum(a.nbytes for a in dt.col) + dt.col_bbox.nbytes + dt.col_rank.nbytes + dt.col_nbr.nbytes + dt.col_orient.nbytes + dt.col_chunk_bbox.nbytes + dt.grid_lo.nbytes + dt.grid_hi.nbytes
              + dt.start_pos.nbytes + dt.start_quat.nbytes
              + dt.n_samples_pad                                                # equal-run ends (u8), derived on upload
              + (dt.n_collision_pad * (96 + 12) if dt.col_convex else 0))       # hull facet records + edge neighbours
    return survey_env, survey_launch, per_env, static, per_env * n_envs + static


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


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _oracle_rate(tables, n_envs, steps_sample, threads, obs_mode='section', overlap=False):
    import numpy as np
    import oracle
    orc = oracle.Oracle(tables, n_envs, threads=threads, obs_mode=obs_mode, overlap_penalty=overlap)
    rng = np.random.RandomState(1234)
    orc.reset(rng.randint(0, 4, size=n_envs))
    acts = rng.randint(0, 4, size=(steps_sample, n_envs))
    t0 = time.perf_counter()
    for k in range(steps_sample):
        _, _, done, _ = orc.step(acts[k])
        if done.any():
            orc.reset(rng.randint(0, 4, size=n_envs), mask=done)
    return time.perf_counter() - t0


def cpu_baseline(tables, steps_sample=300, n_envs=ENVS_PER_GPU, obs_mode='section', overlap=False):
    """The C oracle (a scalar float64 port of the reference step()) on the host cores: all of them (the headline
    value) and one (SURVEY §8d asks for both); the reference's own Python step() cannot travel to this box, its
    figure is the one recorded next to the golden fixtures when they were generated from the imported reference."""
    cores = usable_cores()
    dt_all = _oracle_rate(tables, n_envs, steps_sample, cores, obs_mode, overlap)
    one_steps = max(2, steps_sample // 12)
    dt_one = _oracle_rate(tables, n_envs, one_steps, 1, obs_mode, overlap)
    out = {'value': steps_sample / dt_all, 'unit': 'batched steps/s (%d envs each)' % n_envs, 'cores': cores,
           'cpu_model': cpu_model(),
           'kind': 'port', 'env_steps_per_s': steps_sample * n_envs / dt_all,
           'sample': '%d batched steps of %d envs (same door, same action distribution, reset on done), '
                     'oracle/paint_oracle.c with OpenMP over envs, %.1f s wall' % (steps_sample, n_envs, dt_all),
           'single_thread': {'value': one_steps / dt_one, 'env_steps_per_s': one_steps * n_envs / dt_one, 'cores': 1,
                             'sample': '%d batched steps of %d envs, one thread, %.1f s wall' % (one_steps, n_envs, dt_one)}}
    try:
        with open(os.path.join(REPO, 'tests', 'golden', 'MANIFEST.json')) as f:
            man = json.load(f)
        mode = 'grid' if obs_mode == 'grid' else 'section'
        rt = man.get('reference_step_timing')
        if rt and mode in rt:
            r = rt[mode]
            out['reference_python'] = {
                'ms_per_env_step': r['ms_per_env_step'], 'ms_per_env_step_without_ray_stand_in': r['ms_per_env_step_without_ray_stand_in'],
                'ray_stand_in_ms_per_env_step': r['ray_stand_in_ms_per_env_step'],
                'env_steps_per_s_all_processes': r['env_steps_per_s_all_processes'], 'processes': r['processes'],
                'steps_per_process': r['steps_per_process'], 'cpu_model': rt.get('cpu_model'), 'cores': rt.get('cores'),
                'provenance': 'the reference PaintGymEnv.step() itself (CPython, %d processes x one env on the build container, synthetic door, '
                              'random discrete-4 actions with reset on done), timed by tests/golden/make_golden.py --timing; the stand-in '
                              "pybullet's ray time (this project's numpy ray, not Bullet) is reported separately; tests/golden/MANIFEST.json"
                              % r['processes']}
        else:
            ms = man['timings']['door_grid_ms_per_step' if obs_mode == 'grid' else 'door_section_ms_per_step']
            out['reference_python'] = {
                'ms_per_env_step': ms, 'env_steps_per_s_per_core': 1e3 / ms,
                'provenance': 'the reference PaintGymEnv.step() itself (CPython, one core of the build container, '
                              'synthetic door, ray time of the stand-in pybullet included), timed by '
                              'tests/golden/make_golden.py while it recorded the golden episodes; tests/golden/MANIFEST.json'}
    except (OSError, KeyError, ValueError):
        pass
    return out


def latest_profile(suffix):
    """profiles/rNN_<suffix> of the highest round that has one (the evidence bench.py quotes is committed, not measured
    in the run: `measured_at_commit` inside says what it was measured on)."""
    import glob
    import re
    hits = [(int(re.search(r'r(\d+)_', os.path.basename(p)).group(1)), p)
            for p in glob.glob(os.path.join(REPO, 'profiles', 'r[0-9]*_' + suffix))]
    return max(hits)[1] if hits else None


def valu_issue_bound(kernel_us, obs_mode):
    """Second roofline: the step kernel against its own vector-issue bound -- instruction counts from the committed SQ
    counters, priced with the issue costs measured on this chip (tools/valu_model.py: 2 cycles for 32-bit VOP1 / VOP2,
    4 for float64 / VOP3 / DPP / lane moves, 16 for float64 reciprocals and square roots), W waves per SIMD."""
    path = latest_profile('valu_model.json')
    if not path or not kernel_us:
        return None
    with open(path) as f:
        m = json.load(f)
    c = m.get(obs_mode)
    if not c:
        return None
    at = m.get('measured_at_commit')
    stale = None
    if at and at != 'unknown':
        # the instruction counts belong to the kernel of that commit: say so when csrc/ has changed since
        try:
            base = at.split('+')[0]
            changed = subprocess.check_output(['git', '-C', REPO, 'diff', '--name-only', base, '--', 'paintrl_amd/csrc', 'paintrl_amd/build.py'],
                                              text=True, stderr=subprocess.DEVNULL).strip()
            stale = bool(changed) or at.endswith('+dirty')
        except (subprocess.CalledProcessError, OSError):
            stale = None                  # no git here (the GPU box): unknown
    return {'bound': 'valu_issue', 'valu_per_wave': c['valu_per_wave'], 'valu_f64_per_wave': c['f64_per_wave'],
            'valu_f64_trans_per_wave': c['f64_trans_per_wave'], 'price_cycles': m['price_cycles'],
            'share_of_non_f64_at_2_cycles': c['share_of_the_rest_at_2_cycles'], 'clock_ghz': CLOCK_GHZ,
            'issue_bound_us': c['issue_bound_us'], 'frac': c['issue_bound_us'] / kernel_us,
            'sq_active_inst_valu_us': c['sq_active_inst_valu_us'], 'sq_active_inst_valu_frac': c['sq_active_inst_valu_us'] / kernel_us,
            'source': os.path.relpath(path, REPO) + ' (tools/valu_model.py: rocprofv3 --pmc counts x tools/microbench/valu_rate '
                      'issue costs, split by the static encoding mix of the kernel)', 'measured_at_commit': at,
            'kernel_sources_changed_since': stale}


def cone_second_bound(step_us):
    """PAINT_METHOD 'normal': the beams kernel (the dominant one of the step's five launches) against its vector-issue and
    texture-addresser occupancy, from the committed counters of the latest round (tools/summarise_profiles.py)."""
    path = latest_profile('cone_model.json')
    if not path:
        return None
    with open(path) as f:
        m = json.load(f)
    return {'bound': 'valu_issue', 'kernel': m['kernel'], 'frac': m['valu_busy_frac'], 'ta_busy_frac': m['ta_busy_frac'],
            'valu_per_beam_trip': m['valu_per_wave'], 'vmem_rd_per_beam_trip': m['vmem_rd_per_wave'],
            'kernel_us_when_measured': m['kernel_us_at_2p4ghz'], 'share_of_the_step': m['kernel_us_at_2p4ghz'] / step_us if step_us else None,
            'measured_at_commit': m.get('measured_at_commit'),
            'source': os.path.relpath(path, REPO) + ' (rocprofv3 --pmc: SQ_ACTIVE_INST_VALU x 4 / SIMD cycles, TA_TA_BUSY_sum / CU cycles); not measured in this run'}


# ---------------------------------------------------------------------------- self-launch (N > 1)
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv):
    """Start `n` ranks of this script as child processes and wait for them.  The parent has not imported torch
    and never touches the GPU; no process is replaced (no exec after a HIP call).  Rank 0 inherits stdout, so
    the one JSON line comes out of this process's stdout as usual."""
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        left = list(enumerate(procs))
        while left and not rc:                      # poll: the FIRST rank to fail ends the job (its peers would wait in a
            for r, p in list(left):                 # collective forever) and is named
                code = p.poll()
                if code is None:
                    continue
                left.remove((r, p))
                if code:
                    sys.stderr.write('bench.py: rank %d of %d exited with status %d; stopping the other ranks\n' % (r, n, code))
                    rc = code
                    break
            else:
                time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=200)
    ap.add_argument('--envs', type=int, default=ENVS_PER_GPU, help='envs per GPU (default: the BASELINE config)')
    ap.add_argument('--obs-mode', default='section', choices=['section', 'grid'])
    ap.add_argument('--obs-grad', type=int, default=4, help="OBS_GRAD (rge:150): 4 = the four-quadrant rule, other values the atan2 sectors")
    ap.add_argument('--mixed', action='store_true', help='config 5: door/sheet alternate, START_POINT_MODE all')
    ap.add_argument('--actions', default='random', choices=['random', 'sweep'],
                    help="'sweep': every env follows an on-part serpentine with a random phase (SURVEY 8d item 2), so "
                         'episodes run long instead of ending after ~17 random steps')
    ap.add_argument('--policy', default='random', choices=['random', 'mlp', 'mlp-torch', 'fragment', 'random-fragment'],
                    help="'mlp': actions from the 6-256-128-4 policy network on the same stream (config 4), one fused "
                         "kernel per step (prl_policy_act); 'mlp-torch': the same net in torch eager; 'fragment': "
                         'policy + env step for a whole 100-step rollout fragment in ONE persistent launch '
                         "(prl_rollout_fragment); 'random-fragment': the same kernel fed the random action rows")
    ap.add_argument('--streams', type=int, default=1,
                    help='issue a batched step as S launches of envs/S envs on S streams (independent env groups: a '
                         "group's slowest wave then only delays that group); default 1 = one launch per step")
    ap.add_argument('--graph', action='store_true', help='with --policy mlp: capture policy + env step in one HIP graph')
    ap.add_argument('--paint-method', default='fast', choices=['fast', 'normal'])
    ap.add_argument('--color-mode', default='RGB', choices=['RGB', 'HSI'], help="COLOR_MODE (rge:156): 'HSI' = thickness bytes (bpw:384-434)")
    ap.add_argument('--part', default='door_test',
                    help="synthetic part (paintrl_amd.synth_parts.PARTS): 'door_test' = the BASELINE door panel (default), 'square' "
                         "the fine sheet, 'test' the coarse sheet with the reference's stale kd-tree, 'door_rr_big' the door on a "
                         'larger texture (the large-part kernels: Part_Dict door_lf .. door_rr_big, rge:106-117)')
    ap.add_argument('--tex', type=int, default=0, help='texture edge of --part (0 = its default); sets the sample count')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-prewarm', action='store_true')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher.  Nothing above imported torch or touched HIP.
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    import numpy as np
    import torch
    from paintrl_amd import distributed as pdist
    from paintrl_amd import part_tables, synth_parts
    from paintrl_amd.batched_env import BatchedPaintEnv
    from paintrl_amd.device_tables import DeviceTables

    rank, local_rank, world = pdist.init_process_group()
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if os.environ.get('PAINTRL_SINGLE_DEVICE'):          # testing aid: all ranks share GPU 0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)

    tex = args.tex or synth_parts.TEXTURES[args.part][0][0]
    tables = part_tables.build_part_tables(mesh=synth_parts.synthetic_mesh(args.part), tex_size=(tex, tex), name=args.part)
    other_part = args.part != 'door_test' or tex != 240
    if other_part and (args.mixed or args.streams > 1):
        raise SystemExit('--part / --tex are not combined with --mixed / --streams')
    # the headline door: _max_possible_point as the reference computes it (rge:240-252, 9 148 of 9 664); other parts: 95 %
    mpp = 9148 if not other_part else int(0.95 * tables.sample_pos.shape[0])
    start_mode = 'all' if (args.actions == 'sweep' or other_part) else 'anchor'      # the sweep starts anywhere on the part
    dt = DeviceTables(tables, obs_grad=args.obs_grad, start_points=part_tables.start_points(tables, start_mode))
    overlap = args.obs_mode == 'grid'
    common = dict(device=device, obs_mode=args.obs_mode, obs_grad=args.obs_grad, auto_reset=True, overlap_penalty=overlap,
                  paint_method=args.paint_method, color_mode=args.color_mode)
    if args.mixed:
        if args.streams > 1:
            raise SystemExit('--streams is not combined with --mixed')
        sheet = part_tables.build_part_tables(mesh=synth_parts.synthetic_mesh('square'), tex_size=(240, 240),
                                              name='square')
        dt = DeviceTables(tables, obs_grad=4, start_points=part_tables.start_points(tables, 'all'))
        dts = DeviceTables(sheet, obs_grad=4, start_points=part_tables.start_points(sheet, 'all'))
        env = BatchedPaintEnv([dt, dts], args.envs, env_part_id=np.arange(args.envs) % 2,
                              seed=pdist.rank_seed(5678, rank), max_possible_point=[9148, 14350], **common)
        subs = [env]
    else:
        if args.streams > 1 and (args.policy not in ('random', 'mlp') or args.envs % args.streams):
            raise SystemExit('--streams needs --policy random or mlp and a divisible --envs')
        n_sub = args.envs // args.streams
        subs = [BatchedPaintEnv(dt, n_sub, seed=pdist.rank_seed(5678 + 7919 * g, rank), max_possible_point=mpp,
                                **common) for g in range(args.streams)]
        env = subs[0]
    main_stream = torch.cuda.current_stream(device)
    sub_streams = [torch.cuda.Stream(device=device) for _ in subs] if len(subs) > 1 else [None]
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + rank)
    total = args.warmup + REPEATS * args.steps
    actions = torch.randint(0, 4, (total, args.envs), generator=gen, device=device, dtype=torch.int32)
    if args.actions == 'sweep':
        pattern = torch.tensor(([1] * 12 + [0] * 2 + [3] * 12 + [0] * 2), dtype=torch.int32, device=device)
        phase = torch.randint(0, pattern.numel(), (args.envs,), generator=gen, device=device)
        steps_idx = torch.arange(total, device=device).unsqueeze(1)
        actions = pattern[(steps_idx + phase.unsqueeze(0)) % pattern.numel()].contiguous()

    prewarm_steps = 0
    if not args.no_prewarm and PREWARM_SECONDS > 0:
        # a scratch batch of the same shape, stepped untimed: brings the clocks up and the code / tables into cache
        scratch = BatchedPaintEnv(dt, min(args.envs, ENVS_PER_GPU), seed=1, max_possible_point=mpp, **common) \
            if not args.mixed else None
        if scratch is not None:
            scratch.reset()
            # (the timed run's own action rows, cycled: the pre-warm launches do the work the timed ones do, so that a
            # rocprofv3 average over the whole process is the timed kernels' average)
            rows = actions if scratch.n_envs == args.envs else actions[:, :scratch.n_envs].contiguous()
            t_end = time.perf_counter() + PREWARM_SECONDS
            while time.perf_counter() < t_end:
                for k in range(50):
                    scratch.step_raw(rows[(prewarm_steps + k) % total])
                prewarm_steps += 50
                torch.cuda.synchronize(device)
            scratch.close()

    for e in subs:
        e.reset()
    stream_sync = lambda: torch.cuda.synchronize(device)        # noqa: E731
    n_sub = args.envs // len(subs)
    sub_actions = [actions] if len(subs) == 1 else [actions[:, g * n_sub:(g + 1) * n_sub].contiguous()
                                                     for g in range(len(subs))]
    for s in sub_streams:                                  # resets and the action slices ran on the main stream
        if s is not None:
            s.wait_stream(main_stream)

    policy = None
    fragment_runner = None
    if args.policy in ('fragment', 'random-fragment'):
        from paintrl_amd.rollout import MLPPolicy, FragmentRunner
        torch.manual_seed(1234)
        fragment_runner = FragmentRunner(env, MLPPolicy(env.obs_dim, 4).to(device), fragment=FRAGMENT, seed=1234 + rank,
                                         given_actions=actions if args.policy == 'random-fragment' else None)
    elif args.policy != 'random':
        from paintrl_amd.rollout import MLPPolicy
        torch.manual_seed(1234)
        torch_policy = MLPPolicy(env.obs_dim, 4).to(device)
        if args.policy == 'mlp':
            from paintrl_amd.policy import FusedPolicy
            fused = FusedPolicy(torch_policy)
            group_policies = [fused] + [FusedPolicy(torch_policy, seed=g) for g in range(1, len(subs))]

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

    import torch.distributed as dist
    gatherer = pdist.ReturnsGatherer(device) if dist.is_initialized() else None      # (world 1: only under PAINTRL_FORCE_DIST)

    def gather_fragment():
        # the all_gather of the fragment's episode returns runs on a side stream, overlapped with the next steps
        if len(subs) > 1:
            for s in sub_streams:
                main_stream.wait_stream(s)
        gatherer.submit(torch.cat([e.episode_returns() for e in subs]) if len(subs) > 1 else env.episode_returns())

    def run(k0, k1):
        k = k0
        while k < k1:
            if fragment_runner is not None:
                n = min(FRAGMENT - (k % FRAGMENT), k1 - k)
                fragment_runner.run(n)              # n policy + env steps in one persistent launch
                k += n
            else:
                if graph is not None:
                    graph.replay()
                elif policy is None and len(subs) > 1:
                    for g, e in enumerate(subs):
                        with torch.cuda.stream(sub_streams[g]):
                            e.step_raw(sub_actions[g][k])
                elif policy is None:
                    env.step_raw(actions[k])
                elif len(subs) > 1:
                    # each env group alternates policy and env step on its own stream: one group's policy launch
                    # (latency bound, a few us) runs under the other group's env step
                    for g, e in enumerate(subs):
                        with torch.cuda.stream(sub_streams[g]):
                            act, _, _ = group_policies[g].act(e.obs)
                            e.step_raw(act)
                else:
                    act, _, _ = policy.act(obs_for_policy(), gen)
                    env.step_raw(act)
                k += 1
            if gatherer is not None and k % FRAGMENT == 0:
                gather_fragment()

    run(0, args.warmup)
    stream_sync()
    # kernel time for the roofline: ONE HIP event pair on the launch stream around each timed region (the
    # launches are back to back, the host runs ahead), divided by the launches -- an upper bound on the kernel's
    # duration (it includes the ~1 us dispatch gaps) that cannot exceed ms_per_step and does not perturb the run;
    # long runs additionally sample single launches with their own event pairs
    timing_every = TIMING_EVERY if args.steps > SAMPLE_ABOVE else 0
    for e in subs:
        e.timing(timing_every)
    region_ok = len(subs) == 1 and graph is None
    # the timed region -- EXACTLY --steps steps between barrier + synchronize on both sides, MAX over ranks -- REPEATS
    # times back to back; the reported value is the median region
    elapsed_all, region_ms_all = [], []
    for rep in range(REPEATS):
        k0 = args.warmup + rep * args.steps
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        pdist.barrier()
        stream_sync()
        t0 = time.perf_counter()
        ev0.record(main_stream)
        run(k0, k0 + args.steps)
        ev1.record(main_stream)
        stream_sync()
        pdist.barrier()
        stream_sync()
        elapsed_all.append(pdist.max_over_ranks(time.perf_counter() - t0, device))
        region_ms_all.append(ev0.elapsed_time(ev1))
    if gatherer is not None:
        gatherer.wait()
    kernel_ms, launches = 0.0, 0
    for e in subs:
        ms_, n_ = e.timing_read()
        kernel_ms, launches = kernel_ms + ms_, launches + n_
        e.timing(0)
    order = sorted(range(REPEATS), key=lambda r: elapsed_all[r])
    med = order[REPEATS // 2]
    elapsed = elapsed_all[med]

    episodes = sum(int(e.state()['episode'].sum()) for e in subs) - args.envs
    # which device every rank ran on (a SCALE record can be checked against it)
    rank_devices = [int(local_rank)]
    if dist.is_initialized() and dist.get_world_size() > 1:          # (the returns' own all_gather path: RCCL or gloo)
        rank_devices = [int(v) for v in pdist.gather_returns(torch.tensor([float(local_rank)], dtype=torch.float64, device=device)).tolist()]
    occupancy = None
    if args.paint_method == 'fast' and args.policy in ('random', 'mlp', 'mlp-torch'):
        occupancy = env.step_occupancy()                 # of one prl_batch_step launch
    if rank == 0:
        n_launch_envs = args.envs // len(subs)
        survey_env, survey_launch, per_env, static, per_launch = algorithmic_bytes(dt, n_launch_envs, env.obs_dim)
        ms_per_step = 1e3 * elapsed / args.steps
        sampled_us = 1e3 * kernel_ms / launches if launches and kernel_ms > 0 else None
        avg_kernel_s = None
        if region_ok and args.policy == 'random':
            avg_kernel_s = region_ms_all[med] / 1e3 / args.steps       # one step = one step_kernel launch
            assert avg_kernel_s * 1e3 <= ms_per_step * 1.001, \
                'kernel %.1f us > step %.1f us' % (avg_kernel_s * 1e6, ms_per_step * 1e3)
        elif sampled_us:
            avg_kernel_s = sampled_us / 1e6
        achieved = survey_launch / avg_kernel_s / 1e9 if avg_kernel_s else None
        traffic, traffic_at = None, None
        tpath = os.path.join(REPO, 'profiles', 'hbm_traffic.json')
        # the committed PMC measurement is for the default workload (section / grid) only
        if os.path.isfile(tpath) and args.envs == ENVS_PER_GPU and not args.mixed and args.actions == 'random' \
                and len(subs) == 1 and args.policy == 'random' and args.paint_method == 'fast' and not other_part and args.color_mode == 'RGB':
            with open(tpath) as f:
                tj = json.load(f)
            traffic = tj.get('bytes_per_launch_%s' % args.obs_mode)
            traffic_at = tj.get('measured_at_commit', 'round %s (no commit recorded)' % tj.get('round'))
        value = world * args.steps / elapsed
        if args.policy == 'random':
            act_desc = 'on-part serpentine' if args.actions == 'sweep' else 'random'
        else:
            act_desc = {'mlp': 'policy-MLP (6-256-128-4, fp32, fused HIP kernel)',
                        'mlp-torch': 'policy-MLP (6-256-128-4, fp32, torch eager)',
                        'fragment': 'policy-MLP (6-256-128-4, fp32, inside the persistent rollout-fragment kernel, %d steps '
                                    'per launch)' % FRAGMENT,
                        'random-fragment': 'random (persistent rollout-fragment kernel reading the action rows, %d steps '
                                           'per launch)' % FRAGMENT}[args.policy]
        dist_world = dist.get_world_size() if dist.is_initialized() else 1
        out = {
            'metric': 'batched env steps/sec (door panel, N=4096)', 'value': value,
            'unit': 'batched steps/s (%d envs each)' % args.envs, 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True,
            'repeats': {'n': REPEATS, 'value_is': 'median', 'values': [world * args.steps / t for t in elapsed_all],
                        'min': world * args.steps / max(elapsed_all), 'max': world * args.steps / min(elapsed_all)},
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'PaintGymEnv %s%s, OBS_MODE=%r%s, %d envs per GPU, '
                                   '%s discrete-4 actions, in-kernel auto-reset'
                                   % ('Part_NO=0 synthetic door panel' if not other_part else
                                      "synthetic part '%s' on a %d x %d texture (START_POINT_MODE all)" % (args.part, tex, tex),
                                      ' + sheet (mixed, START_POINT_MODE all)' if args.mixed else '', args.obs_mode,
                                      ' + OVERLAP_PENALTY' if overlap else '', args.envs, act_desc)
                                   + (", PAINT_METHOD='normal'" if args.paint_method == 'normal' else '')
                                   + (", COLOR_MODE='HSI'" if args.color_mode == 'HSI' else ''),
                       'envs_per_gpu': args.envs, 'env_steps_per_s': value * args.envs,
                       'part': args.part, 'texture': tex, 'samples': int(dt.n_samples), 'mask_words': int(env.mask_stride),
                       'stale_kd_nodes': len(getattr(tables, 'kd_split_dim', ())), 'collision_triangles': int(dt.n_collision),
                       'step_launch_occupancy': occupancy,
                       'rccl_ranks': (dist_world if (dist.is_initialized() and dist.get_backend() == 'nccl') else 0),
                       'rank_devices': rank_devices,
                       'episodes_finished_rank0': episodes, 'launches_per_step': len(subs),
                       'prewarm_steps_untimed_scratch_batch': prewarm_steps,
                       'returns_gathers': gatherer.count if gatherer is not None else 0,
                       'parallelism': 'env-shard x%d (torch.distributed world size %d, backend %s)'
                                      % (world, dist_world, dist.get_backend() if dist.is_initialized() else 'none')
                                      + (', %d independent env groups per GPU on %d streams' % (len(subs), len(subs))
                                         if len(subs) > 1 else '')},
            'roofline': {'bound': 'hbm', 'kernel': ('step_kernel' if env.mask_stride <= 256 else 'step_kernel_big') if args.paint_method == 'fast' else 'the five launches of a cone-beam step (path, beams, rest, far, finish)', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS if achieved else None, 'traffic': traffic,
                         'traffic_over_algorithmic': traffic / survey_launch if traffic else None,
                         'traffic_measured_at': traffic_at,
                         'traffic_from': 'profiles/hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, '
                                         'FETCH_SIZE x2 calibrated on copy_mask_kernel); not measured in this run.  About 11 MB '
                                         'of that figure are part tables fetched again in every launch only because rocprofv3 --pmc '
                                         'invalidates the L2s between dispatches (profiles/r04_l2_cold_under_pmc.txt): an artefact '
                                         'of counter collection, absent from the unprofiled back-to-back launches this line times',
                         'algorithmic_bytes_per_env_step': survey_env, 'bytes_per_launch': survey_launch,
                         'layout_bytes_per_env_step': per_env, 'layout_static_table_bytes': static,
                         'layout_bytes_per_launch': per_launch,
                         'avg_kernel_us': avg_kernel_s * 1e6 if avg_kernel_s else None,
                         'launches_timed': args.steps if (region_ok and args.policy == 'random') else int(launches),
                         'kernel_time_from': 'one HIP event pair on the launch stream around the timed region / launches '
                                             '(upper bound: includes dispatch gaps)' if (region_ok and args.policy == 'random')
                                             else 'HIP event pairs around sampled launches',
                         'avg_kernel_us_sampled': sampled_us, 'sampled_launches': int(launches),
                         'second_bound': valu_issue_bound(avg_kernel_s * 1e6 if avg_kernel_s else None, args.obs_mode)
                         if (args.envs == ENVS_PER_GPU and not args.mixed and args.policy == 'random' and args.actions == 'random'
                             and args.paint_method == 'fast' and len(subs) == 1 and not other_part and args.color_mode == 'RGB') else
                         (cone_second_bound(1e3 * ms_per_step) if (args.paint_method == 'normal' and args.envs == ENVS_PER_GPU
                                                                    and not args.mixed and len(subs) == 1) else None)},
        }
        if world == 1 and not args.no_cpu_baseline:
            # (a bounded sample: the oracle's step scales with the part's sample count)
            out['cpu_baseline'] = cpu_baseline(tables, steps_sample=max(24, int(300 * 9664 / max(9664, dt.n_samples))),
                                               n_envs=args.envs, obs_mode=args.obs_mode, overlap=overlap)
        print(json.dumps(out, default=lambda o: o.item() if hasattr(o, 'item') else str(o)))
        sys.stdout.flush()
    for e in subs:
        e.close()
    if dist.is_initialized():
        pdist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
