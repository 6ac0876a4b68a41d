#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Development container only (needs /root/reference).  The reference's Python is
imported against tests/golden/refstubs (pybullet/gym/termcolor stand-ins, see
refstubs/pybullet.py for the two calls that carry arithmetic) and driven on the
synthetic parts written by paintrl_amd.synth_parts.  What is stored is data
only: action sequences, start indices, per-step (obs, reward, done, info),
painted-texel snapshots, final pose/return, static-table digests (SURVEY.md §8c
G0-G6).  No reference source is stored.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz
"""
import contextlib
import hashlib
import io
import json
import os
import random
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import ref_import  # noqa: E402
from paintrl_amd import synth_parts  # noqa: E402

SNAP_EVERY = 25


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


class RefDriver(object):
    """One constructed reference env (START_POINT_MODE='all'), re-configured in place per episode."""

    def __init__(self, root, part_no, paint_radius=None, step_size=None, extra=None):
        ref_import.load_reference('hull')
        prof = sys.modules['bullet_paint_wrapper'].PaintToolProfile
        prof.PAINT_RADIUS = 0.051 if paint_radius is None else paint_radius
        prof.STEP_SIZE = prof.PAINT_RADIUS if step_size is None else step_size
        self.tool = (prof.PAINT_RADIUS, prof.STEP_SIZE)
        t0 = time.time()
        self.env, self.part = ref_import.make_env(root, part_no=part_no, obs_mode='section', obs_grad=4,
                                                  extra=dict({'START_POINT_MODE': 'all'}, **(extra or {})))
        self.construct_s = time.time() - t0
        self.bpw = sys.modules['bullet_paint_wrapper']
        self.rob = sys.modules['robot']
        self.rge = sys.modules['robot_gym_env']
        self.all_points = list(self.env._start_points)
        self.F = self.bpw.Side.front
        W = self.part.texture_width
        self.pix = sorted(self.part.profile[self.F], key=lambda p: p[1] * W + p[0])
        self.part_no = part_no

    def configure(self, obs_mode='section', obs_grad=4, start_mode='anchor', overlap=False, turning=False,
                  termination='late', paint_method='fast', action=('discrete', 1, 4), max_len=245, rollout=False,
                  max_points=None):
        env, part, bpw = self.env, self.part, self.bpw
        if not hasattr(self, 'part_max_points'):
            self.part_max_points = env._max_possible_point           # Part_Dict constant (rge:106-117)
        env._max_possible_point = self.part_max_points if max_points is None else max_points
        cls = self.rge.PaintGymEnv
        cls.OBS_MODE, cls.OBS_GRAD = obs_mode, obs_grad
        mode, shape, gran = action
        cls.change_action_mode(shape, mode, gran)
        if obs_mode in ('section', 'discrete'):
            part._obs_handler = bpw.SectionObservation(part, obs_grad)
        elif obs_mode == 'grid':
            part._obs_handler = bpw.GridObservation(part, obs_grad)
        else:
            part._obs_handler = bpw.NoObservation(part)
        n = {'fixed': 1, 'anchor': 4, 'all': len(self.all_points)}[start_mode]
        env._start_points = self.all_points[:n]
        env.OVERLAP_PENALTY, env.TURNING_PENALTY = overlap, turning
        env.TERMINATION_MODE = termination
        env.EPISODE_MAX_LENGTH = max_len
        env._rollout = rollout
        self.rob.Robot.PAINT_METHOD = paint_method
        self.cfg = dict(obs_mode=obs_mode, obs_grad=obs_grad, start_mode=start_mode, overlap_penalty=overlap,
                        turning_penalty=turning, termination_mode=termination, paint_method=paint_method,
                        action_mode=mode, action_dim=shape, n_discrete=gran, max_episode_len=max_len,
                        expected_episode_len=env.Expected_Episode_Length, switch_threshold=env.SWITCH_THRESHOLD,
                        max_possible_point=env._max_possible_point, part_no=self.part_no,
                        paint_radius=self.tool[0], step_size=self.tool[1])

    def reset(self, seed):
        random.seed(seed)
        obs = self.env.reset()
        pose = tuple(self.env.robot._pose)
        idx = [tuple(p[0]) for p in self.env._start_points].index(pose)
        return np.asarray(obs, dtype=np.float64), idx

    def painted_bits(self):
        return np.packbits(np.array([bool(self.part.get_pixel_status(p)) for p in self.pix]), bitorder='little')

    def episode(self, seed, policy, max_steps=400, want_idx=None, after_reset=None, snap_every=SNAP_EVERY):
        obs0, idx = self.reset(seed)
        while want_idx is not None and idx != want_idx:      # walk seeds until the wanted start comes up
            seed += 1000
            obs0, idx = self.reset(seed)
        if after_reset is not None:
            after_reset(self.env)
        rec = dict(actions=[], obs=[], reward=[], done=[], info=[], snaps=[], snap_steps=[])
        obs, done, k = obs0, False, 0
        t_step = 0.0
        with contextlib.redirect_stdout(io.StringIO()):
            while not done and k < max_steps:
                a = policy(k, obs)
                rec['actions'].append(list(a) if isinstance(a, list) else a)   # step() clips lists in place
                t0 = time.perf_counter()
                o, r, done, info = self.env.step(a)
                t_step += time.perf_counter() - t0
                obs = np.asarray(o, dtype=np.float64)
                rec['obs'].append(obs)
                rec['reward'].append(r)
                rec['done'].append(done)
                rec['info'].append([info['reward'], info['penalty']])
                k += 1
                if k % snap_every == 0 or done:
                    rec['snaps'].append(self.painted_bits())
                    rec['snap_steps'].append(k)
        if not rec['snap_steps'] or rec['snap_steps'][-1] != k:
            rec['snaps'].append(self.painted_bits())
            rec['snap_steps'].append(k)
        out = dict(start_idx=np.int32(idx), obs0=obs0, actions=np.asarray(rec['actions']),
                   obs=np.asarray(rec['obs']), reward=np.asarray(rec['reward'], dtype=np.float64),
                   done=np.asarray(rec['done'], dtype=bool), info=np.asarray(rec['info'], dtype=np.float64),
                   snaps=np.asarray(rec['snaps']), snap_steps=np.asarray(rec['snap_steps'], dtype=np.int32),
                   final_pose=np.asarray(self.env.robot._pose, dtype=np.float64),
                   final_quat=np.asarray(self.env.robot._orn, dtype=np.float64),
                   total_return=np.float64(self.env._total_return), ms_per_step=np.float64(1e3 * t_step / max(k, 1)),
                   cfg=json.dumps(self.cfg))
        return out


# ---- policies (this project's own drivers; zigzag follows the idea of zigzag.py:77-101) ----
def random_policy(seed, n=4):
    rng = np.random.RandomState(seed)
    seq = rng.randint(0, n, size=1000)
    return lambda k, obs: int(seq[k])


def zigzag_policy(pos_index=-1, cols=2):
    st = {'up': True, 'h': 0}

    def pol(k, obs):
        y = obs[pos_index]
        while True:
            if st['up']:
                if y < 0.95:
                    return 1
                if st['h'] < cols:
                    st['h'] += 1
                    return 0
                st['h'], st['up'] = 0, False
            else:
                if y > 0.05:
                    return 3
                if st['h'] < cols:
                    st['h'] += 1
                    return 0
                st['h'], st['up'] = 0, True
    return pol


def continuous_policy(seed, dim):
    rng = np.random.RandomState(seed)
    seq = rng.uniform(-1.2, 1.2, size=(1000, dim))
    if dim == 2:
        seq[5] = 0.0
    return lambda k, obs: [float(v) for v in seq[k]]


def table_digest(drv):
    """G0: digests + small slices of the reference's static tables for one part."""
    part, env, F = drv.part, drv.env, drv.F
    pos = np.array([part.profile_dicts[F][p] for p in drv.pix], dtype=np.float64)
    pix = np.array(drv.pix, dtype=np.int32)
    side_map = {drv.bpw.Side.front: 1, drv.bpw.Side.back: 2, drv.bpw.Side.other: 3}
    sides = np.array([side_map[b.get_side()] for b in part.bary_list], dtype=np.int8)
    normals = np.array([[float(c) for c in b.get_normal()] for b in part.bary_list], dtype=np.float64)
    fn = normals[sides == 1]
    lo = np.array([part.grid_dict[F][i][0] for i in range(100)], dtype=np.float64)
    hi = np.array([part.grid_dict[F][i][1] for i in range(100)], dtype=np.float64)
    sp = np.array(drv.all_points, dtype=np.float64)
    n_anchor = 4
    edge = np.array(drv.all_points[:n_anchor] + part._get_edge_start_points(drv.all_points[n_anchor:]),
                    dtype=np.float64)           # what get_start_points(mode='edge') returns (bpw:775-778)
    kd = np.asarray(part.vertices_kd_tree[F].data, dtype=np.float64)
    try:
        gp = drv.bpw.GridObservation(part, 4)._grid_pixels[F]
        cell = {}
        for i in gp:
            for j in gp[i]:
                for p in gp[i][j]:
                    cell[(int(p[0]), int(p[1]))] = i * 4 + j
        cells = np.array([cell[(int(p[0]), int(p[1]))] for p in drv.pix], dtype=np.int32)
    except KeyError:          # the reference's grid observation does not build on every part (bpw:1101)
        cells = np.zeros(0, dtype=np.int32)
    return dict(P=np.int32(len(drv.pix)), P_back=np.int32(len(part.profile[drv.bpw.Side.back])),
                T=np.int32(len(part.bary_list)), V=np.int32(len(part.vertices)),
                side_counts=np.array([(sides == k).sum() for k in (1, 2, 3)], dtype=np.int32),
                axes=np.array(list(part.principal_axes) + [part.non_principal_axis], dtype=np.int32),
                ranges=np.array(part.ranges, dtype=np.float64), lwr=np.float64(part._length_width_ratio),
                grid_lo=lo, grid_hi=hi, density=np.float64(part.get_density()),
                beams=np.array(env.robot._paint_plain, dtype=np.float64),
                n_start_all=np.int32(len(drv.all_points)), start_points_head=sp[:40], start_points_tail=sp[-40:],
                sha_pix=sha(pix), sha_pos=sha(pos), sha_sides=sha(sides), sha_front_normals=sha(fn),
                sha_start_points=sha(sp), sha_start_points_edge=sha(edge), n_start_edge=np.int32(edge.shape[0]), sha_side_vertices=sha(kd), sha_cells4=sha(cells),
                pix_head=pix[:32], pix_tail=pix[-32:], pos_head=pos[:32], pos_tail=pos[-32:],
                normals_head=fn[:32], normals_tail=fn[-32:], construct_s=np.float64(drv.construct_s))


def param_test_golden(pte, cases=((22, 'zigzag'), (20, 'spiral'), (14, 'zigzag'))):
    """G1: ParamTestEnv trajectories under this project's zigzag / spiral drivers."""
    out = {}

    def run(size, policy_name):
        env = pte.ParamTestEnv(size, train_mode=True)
        obs = env.reset()
        acts, O, R, D = [], [obs], [], []
        done = False
        st = {'up': True, 'h': 0, 'direction': 0, 'strait': size - 3, 'cur': size - 3, 'use': 3}
        while not done:
            if policy_name == 'zigzag':
                cur = round(size * obs[-1])
                a = None
                while a is None:
                    if st['up']:
                        if cur % size != size - 2:
                            a = 1
                        elif st['h'] < 1:
                            a, st['h'] = 0, st['h'] + 1
                        else:
                            st['h'], st['up'] = 0, False
                    else:
                        if cur % size != 1:
                            a = 3
                        elif st['h'] < 1:
                            a, st['h'] = 0, st['h'] + 1
                        else:
                            st['h'], st['up'] = 0, True
            else:
                st['cur'] -= 1
                a = st['direction'] % 4
            obs, r, done, info = env.step(a)
            if policy_name == 'spiral' and st['cur'] == 0:
                st['direction'] += 1
                st['use'] -= 1
                if st['use'] <= 0:
                    st['use'] = 2
                    st['strait'] -= 1
                st['cur'] = st['strait']
            acts.append(a)
            O.append(obs)
            R.append(r)
            D.append(done)
        return dict(actions=np.asarray(acts, dtype=np.int32), obs=np.asarray(O, dtype=np.float64),
                    reward=np.asarray(R, dtype=np.float64), done=np.asarray(D, dtype=bool))

    for size, pol in cases:
        r = run(size, pol)
        for k, v in r.items():
            out['%s%d_%s' % (pol, size, k)] = v
    return out


def main_param_test_modes():
    """G1 for the other OBS_MODEs of ParamTestEnv (pte:17-64, 132-139): 'grid' (Grid10Observation), 'direct'
    (DirectObservation), 'simple' (NoObservation) under the zigzag driver -> g1_param_test_modes.npz."""
    rge, bpw, rob, pte, stub = ref_import.load_reference('hull')
    out = {}
    for mode, size in (('grid', 22), ('direct', 12), ('simple', 14), ('grid', 16)):
        pte.ParamTestEnv.OBS_MODE = mode
        g = param_test_golden(pte, cases=((size, 'zigzag'),))
        for k, v in g.items():
            out['%s_%s' % (mode, k)] = v
    pte.ParamTestEnv.OBS_MODE = 'section'
    np.savez_compressed(os.path.join(HERE, 'g1_param_test_modes.npz'), **out)
    print({k: v.shape for k, v in out.items()})


def main_termination():
    """Episodes that end through each branch of _termination (rge:289-304) the other fixtures never reach: the step
    limit (``_step_counter > EPISODE_MAX_LENGTH - 1``) and ``finished`` (``_total_reward * 100 >= _max_possible_point``).
    Written to episodes_door_term.npz / episodes_sheet_term.npz; nothing else is touched."""
    root = os.path.join(HERE, '_synth_root')
    synth_parts.write_synthetic_parts(root)
    ref_import.load_reference('hull')
    door = RefDriver(root, 0)
    eps = {}
    door.configure('section', 4, 'anchor', max_len=30)
    eps['t1_max_len30'] = door.episode(7, zigzag_policy_grid(), max_steps=100, want_idx=0)
    door.configure('grid', 4, 'anchor', overlap=True, max_len=17)
    eps['t1_max_len17_grid'] = door.episode(8, zigzag_policy_grid(), max_steps=100, want_idx=0)
    door.configure('section', 4, 'anchor', max_points=600)
    eps['t2_finished600'] = door.episode(9, zigzag_policy_grid(), max_steps=245, want_idx=0)
    door.configure('grid', 4, 'anchor', overlap=True, turning=True, max_points=3000)
    eps['t2_finished3000_grid'] = door.episode(10, zigzag_policy_grid(), max_steps=245, want_idx=0)
    door.configure('section', 4, 'anchor', termination='hybrid', max_points=1500)
    eps['t2_finished1500_hybrid'] = door.episode(11, zigzag_policy_grid(), max_steps=245, want_idx=0)
    save_episodes('door_term', eps)
    for name, ep in eps.items():
        assert bool(ep['done'][-1]), name
    assert len(eps['t1_max_len30']['actions']) == 30 and len(eps['t1_max_len17_grid']['actions']) == 17
    door.env.close()
    sheet = RefDriver(root, 1)
    eps = {}
    sheet.configure('simple', 4, 'fixed', rollout=True)
    eps['t3_full245'] = sheet.episode(0, zigzag_policy(1, 1), max_steps=300)      # one column per pass: 245 steps
    sheet.configure('section', 4, 'fixed', max_points=9000)
    eps['t2_finished9000'] = sheet.episode(0, zigzag_policy(-1, 2), max_steps=300)
    save_episodes('sheet_term', eps)
    assert len(eps['t3_full245']['actions']) == 245 and bool(eps['t3_full245']['done'][-1])
    assert bool(eps['t2_finished9000']['done'][-1]) and len(eps['t2_finished9000']['actions']) < 245


def texture_of(drv):
    """Part.get_texture_image() (bpw:737-738) as the uint8 array the PIL image is made from (bpw:18-21)."""
    part = drv.part
    return np.asarray(part.texels, dtype=np.uint8).reshape(part.texture_width, part.texture_height, 3).copy()


def replay_policy(actions):
    acts = [int(a) for a in actions]
    return lambda k, obs: acts[k]


def main_textures():
    """The reference's texture image (Part.get_texture_image, bpw:737-738) after a reset and at the end of three committed
    episodes -- their recorded action lists are replayed on the reference -- : g2_zigzag (sheet), g3_serpentine (door),
    g13_hsi_serpentine (door, COLOR_MODE 'HSI').  Written to textures.npz (whole images: they compress to a few KB)."""
    root = os.path.join(HERE, '_synth_root')
    synth_parts.write_synthetic_parts(root)
    ref_import.load_reference('hull')

    def load(tag, name):
        z = np.load(os.path.join(HERE, 'episodes_%s.npz' % tag), allow_pickle=False)
        return {k.split('/', 1)[1]: z[k] for k in z.files if k.startswith(name + '/')}

    out = {}

    def record(drv, tag, name, seed, want_idx):
        ep = load(tag, name)
        got = drv.episode(seed, replay_policy(ep['actions']), max_steps=len(ep['actions']), want_idx=want_idx)
        assert np.array_equal(got['obs'], ep['obs']) and np.array_equal(got['snaps'][-1], ep['snaps'][-1]), name
        img = texture_of(drv)
        out[name] = img
        out[name + '_sha256'] = np.array(sha(img))
        print(name, img.shape, out[name + '_sha256'])

    sheet = RefDriver(root, 1)
    sheet.configure('simple', 4, 'fixed', rollout=True)
    sheet.reset(0)
    out['sheet_after_reset'] = texture_of(sheet)
    record(sheet, 'sheet', 'g2_zigzag', 0, None)
    sheet.env.close()
    door = RefDriver(root, 0)
    door.configure('section', 4, 'anchor')
    door.reset(0)
    out['door_after_reset'] = texture_of(door)
    record(door, 'door', 'g3_serpentine', 8, 0)
    door.env.close()
    hsi = RefDriver(root, 0, extra={'COLOR_MODE': 'HSI'})
    hsi.configure('section', 4, 'anchor')
    hsi.reset(0)
    out['door_hsi_after_reset'] = texture_of(hsi)
    record(hsi, 'door_hsi', 'g13_hsi_serpentine', 8, 0)
    np.savez_compressed(os.path.join(HERE, 'textures.npz'), **out)
    print('textures.npz', os.path.getsize(os.path.join(HERE, 'textures.npz')), 'bytes')


def _timing_worker(args):
    """One process = one reference env (the reference runs one env per process, paint_ppo.py:171): `steps` random discrete-4
    steps with reset on done; returns (seconds in step(), seconds of those inside the stand-in's rayTestBatch, steps, episodes)."""
    root, obs_mode, steps, seed = args
    rge, bpw, rob, pte, stub = ref_import.load_reference('hull')
    drv = RefDriver(root, 0)
    drv.configure(obs_mode, 4, 'anchor', overlap=obs_mode == 'grid')
    rng = np.random.RandomState(seed)
    random.seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        drv.env.reset()
        ray0, t_step, episodes = stub.RAY_SECONDS[0], 0.0, 0
        for k in range(steps):
            a = int(rng.randint(0, 4))
            t0 = time.perf_counter()
            _, _, done, _ = drv.env.step(a)
            t_step += time.perf_counter() - t0
            if done:
                drv.env.reset()
                episodes += 1
    return t_step, stub.RAY_SECONDS[0] - ray0, steps, episodes


def main_off_part():
    """Episodes that END through Robot._count_not_on_part's own limit (rob:292-300: more than NOT_ON_PART_TERMINATE_STEPS = 1000
    counted misses) -- the one branch of the step no other fixture reaches, because a step whose five shots all miss AND paint
    nothing ends the episode at once (rob:427-430).  The tool is put beside the sheet with Robot.reset([pose, orn]) (rob:366-372,
    what spiral.py:28-38 does), 4 / 6 cm outside its edge: every shot misses the part but its ball still reaches samples, so the
    misses are counted, five a step, until the counter passes 1000 (EPISODE_MAX_LENGTH raised to 400 for that).
    -> episodes_sheet_offpart.npz; the replay moves the tool the same way (oracle set_pose / prl_batch_set_pose)."""
    root = os.path.join(HERE, '_synth_root')
    synth_parts.write_synthetic_parts(root)
    ref_import.load_reference('hull')
    sheet = RefDriver(root, 1)
    eps = {}
    for name, off, pattern in (('t4_offpart_counter_04', 0.04, (1, 3)), ('t4_offpart_mixed_06', 0.06, (0, 2))):
        sheet.configure('section', 4, 'anchor', max_len=400)
        moved = {}

        def move(env, off=off, moved=moved):
            a1 = sheet.part.principal_axes[0]
            pose = [float(v) for v in env._start_points[0][0]]
            orn = [float(v) for v in env._start_points[0][1]]
            pose[a1] -= off
            env.robot.reset([pose, orn])
            moved['pose'], moved['orn'] = pose, orn

        ep = sheet.episode(0, lambda k, obs, pattern=pattern: pattern[k % len(pattern)], max_steps=400, want_idx=0, after_reset=move)
        ep['set_pose'] = np.asarray(moved['pose'], dtype=np.float64)
        ep['set_orn'] = np.asarray(moved['orn'], dtype=np.float64)
        ep['terminate_counter'] = np.int32(sheet.env.robot._terminate_counter)
        ep['robot_terminate'] = np.bool_(sheet.env.robot._terminate)
        eps[name] = ep
    save_episodes('sheet_offpart', eps)
    for name, ep in eps.items():
        print(name, 'steps', len(ep['actions']), 'counter', int(ep['terminate_counter']), 'terminate', bool(ep['robot_terminate']))
        assert bool(ep['done'][-1]) and int(ep['terminate_counter']) > 1000 and bool(ep['robot_terminate'])


def main_every_step():
    """Three episodes whose painted-texel set is recorded after EVERY step (the other fixtures snapshot every 25 steps and
    at the end): the headline configuration (door, section, random walk from an anchor + a serpentine), the sheet with the
    grid observation and OVERLAP_PENALTY (the last-shot set decides the penalty every step), and the door under the cone
    beams.  -> episodes_door_every_step.npz / episodes_sheet_every_step.npz; nothing else is touched."""
    root = os.path.join(HERE, '_synth_root')
    synth_parts.write_synthetic_parts(root)
    ref_import.load_reference('hull')
    door = RefDriver(root, 0)
    eps = {}
    door.configure('section', 4, 'anchor')
    for s in range(91, 140):                 # the first of these random walks that stays on the part for 30 steps or more
        ep = door.episode(611, random_policy(s), max_steps=245, snap_every=1)
        if len(ep['actions']) >= 30:
            break
    eps['e1_door_section_random'] = ep
    eps['e1_door_section_serpentine'] = door.episode(612, zigzag_policy_grid(), max_steps=90, want_idx=0, snap_every=1)
    door.configure('section', 4, 'anchor', paint_method='normal')
    eps['e3_door_cone'] = door.episode(613, zigzag_policy_grid(), max_steps=14, want_idx=0, snap_every=1)
    save_episodes('door_every_step', eps)
    door.env.close()
    sheet = RefDriver(root, 1)
    eps = {}
    sheet.configure('grid', 4, 'anchor', overlap=True)
    eps['e2_sheet_grid_overlap'] = sheet.episode(614, zigzag_policy_grid(), max_steps=80, want_idx=0, snap_every=1)
    save_episodes('sheet_every_step', eps)
    for tag in ('door_every_step', 'sheet_every_step'):
        z = np.load(os.path.join(HERE, 'episodes_%s.npz' % tag), allow_pickle=False)
        for name in json.loads(str(z['episodes'])):
            assert len(z[name + '/snap_steps']) == len(z[name + '/actions']), name
    path = os.path.join(HERE, 'MANIFEST.json')
    meta = json.load(open(path))
    meta['python'] = '%d.%d.%d' % sys.version_info[:3]        # (the tie order of equal samples is this interpreter's: part_tables)
    with open(path, 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)


def main_timing(procs=8, steps=400):
    """BASELINE.md 4.1 / SURVEY 8d: the reference's own step() on the build container's cores -- `procs` processes x one env,
    `steps` random discrete-4 steps each with reset on done, the time of the stand-in's rayTestBatch (this project's numpy
    ray, not Bullet) reported separately.  Numbers only, into MANIFEST.json."""
    import multiprocessing as mp
    root = os.path.join(HERE, '_synth_root')
    synth_parts.write_synthetic_parts(root)
    out = {}
    for mode in ('section', 'grid'):
        with mp.get_context('spawn').Pool(procs) as pool:
            t0 = time.perf_counter()
            res = pool.map(_timing_worker, [(root, mode, steps, 9000 + 17 * k) for k in range(procs)])
            wall = time.perf_counter() - t0
        t_step = sum(r[0] for r in res)
        t_ray = sum(r[1] for r in res)
        n = sum(r[2] for r in res)
        out[mode] = {'processes': procs, 'steps_per_process': steps, 'episodes': int(sum(r[3] for r in res)),
                     'ms_per_env_step': 1e3 * t_step / n, 'ms_per_env_step_without_ray_stand_in': 1e3 * (t_step - t_ray) / n,
                     'ray_stand_in_ms_per_env_step': 1e3 * t_ray / n,
                     'env_steps_per_s_all_processes': n / max(r[0] for r in res), 'wall_s_incl_construction': wall}
        print(mode, json.dumps(out[mode]))
    path = os.path.join(HERE, 'MANIFEST.json')
    meta = json.load(open(path))
    cpu = 'unknown'
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                cpu = line.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    meta['reference_step_timing'] = dict(out, cpu_model=cpu, cores=os.cpu_count(),
                                         what='the imported reference PaintGymEnv.step() on the synthetic door, OBS_MODE section / '
                                              'grid + OVERLAP_PENALTY, anchor starts, random discrete-4 actions, reset on done; one env per '
                                              'process (make_golden.py --timing)')
    with open(path, 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)


def main():
    root = os.path.join(HERE, '_synth_root')
    synth_parts.write_synthetic_parts(root)
    rge, bpw, rob, pte, stub = ref_import.load_reference('hull')
    meta = {'stub_version': stub.STUB_VERSION, 'collision_mode': stub.COLLISION_MODE,
            'numpy': np.__version__, 'python': '%d.%d.%d' % sys.version_info[:3], 'generated_by': 'tests/golden/make_golden.py'}
    np.savez_compressed(os.path.join(HERE, 'g1_param_test.npz'), **param_test_golden(pte))

    timings = {}
    # ---------------- door ----------------
    door = RefDriver(root, 0)
    np.savez_compressed(os.path.join(HERE, 'g0_tables_door.npz'), **table_digest(door))
    eps = {}
    door.configure('section', 4, 'anchor')
    for s in range(4):                       # G3: seeded random episodes, anchor starts
        eps['g3_random_%d' % s] = door.episode(100 + s, random_policy(10 + s))
    eps['g3_sweep'] = door.episode(7, zigzag_policy(-1, 2), max_steps=245, want_idx=0)
    eps['g3_serpentine'] = door.episode(8, zigzag_policy_grid(), max_steps=245, want_idx=0)
    timings['door_section_ms_per_step'] = float(eps['g3_serpentine']['ms_per_step'])
    door.configure('grid', 4, 'anchor', overlap=True)
    eps['g4_grid_overlap'] = door.episode(21, zigzag_policy_grid(), max_steps=120, want_idx=0)
    timings['door_grid_ms_per_step'] = float(eps['g4_grid_overlap']['ms_per_step'])
    door.configure('grid', 4, 'anchor', overlap=True, turning=True)
    eps['g4_grid_overlap_turning'] = door.episode(22, zigzag_policy_grid(), max_steps=60, want_idx=3)
    door.configure('section', 4, 'all')
    for s in range(6):                       # G5: 'all' starts, includes off-part terminations
        eps['g5_all_%d' % s] = door.episode(200 + s, random_policy(40 + s), max_steps=245)
    door.configure('simple', 4, 'anchor', action=('continuous', 1, 4))
    eps['g7_cont1'] = door.episode(300, continuous_policy(50, 1), max_steps=40, want_idx=0)
    door.configure('section', 4, 'anchor', action=('continuous', 2, 4), turning=True)
    eps['g7_cont2'] = door.episode(301, continuous_policy(51, 2), max_steps=40, want_idx=0)
    door.configure('discrete', 4, 'anchor', termination='early')
    eps['g8_early'] = door.episode(302, zigzag_policy_discrete(), max_steps=80, want_idx=0)
    door.configure('section', 4, 'anchor', termination='hybrid', action=('discrete', 1, 8))
    eps['g8_hybrid_gran8'] = door.episode(303, random_policy(60, 8), max_steps=80, want_idx=0)
    door.configure('section', 6, 'anchor')
    eps['g9_section6'] = door.episode(305, zigzag_policy_grid(), max_steps=40, want_idx=0)
    door.configure('discrete', 9, 'anchor', overlap=True)
    eps['g9_discrete9'] = door.episode(306, random_policy(61), max_steps=40, want_idx=3)
    door.configure('section', 4, 'anchor', paint_method='normal')
    eps['g6_normal_door'] = door.episode(304, zigzag_policy(-1, 2), max_steps=12, want_idx=0)
    save_episodes('door', eps)

    # ---------------- sheet ----------------
    door.env.close()          # one part per physics client: drop the door's collision body
    sheet = RefDriver(root, 1)
    np.savez_compressed(os.path.join(HERE, 'g0_tables_sheet.npz'), **table_digest(sheet))
    eps = {}
    sheet.configure('simple', 4, 'fixed', rollout=True)
    eps['g2_zigzag'] = sheet.episode(0, zigzag_policy(1, 2), max_steps=300)   # zigzag.py:77-101 idea
    timings['sheet_simple_ms_per_step'] = float(eps['g2_zigzag']['ms_per_step'])
    sheet.configure('section', 4, 'all')
    for s in range(2):
        eps['g5_all_%d' % s] = sheet.episode(400 + s, random_policy(70 + s), max_steps=245)
    sheet.configure('section', 4, 'fixed', paint_method='normal', rollout=True)
    eps['g6_normal'] = sheet.episode(0, zigzag_policy(-1, 2), max_steps=20)
    save_episodes('sheet', eps)

    # ---------------- sheet with a non-default tool profile (README "Change the paint tool profile") ----------------
    sheet.env.close()
    tool = RefDriver(root, 1, paint_radius=0.04, step_size=0.03)
    np.savez_compressed(os.path.join(HERE, 'g0_tables_sheet_r040.npz'), **table_digest(tool))
    eps = {}
    tool.configure('section', 4, 'anchor', overlap=True)
    eps['g10_tool_profile'] = tool.episode(500, zigzag_policy(-1, 2), max_steps=60, want_idx=0)
    tool.configure('grid', 4, 'all')
    eps['g10_tool_profile_grid'] = tool.episode(501, random_policy(90), max_steps=40)
    save_episodes('sheet_tool', eps)
    sys.modules['bullet_paint_wrapper'].PaintToolProfile.PAINT_RADIUS = 0.051
    sys.modules['bullet_paint_wrapper'].PaintToolProfile.STEP_SIZE = 0.051

    meta['timings'] = timings
    meta['ray_seconds_total'] = float(stub.RAY_SECONDS[0])
    with open(os.path.join(HERE, 'MANIFEST.json'), 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(json.dumps(meta, indent=1))


def zigzag_policy_grid():
    """Grid obs carries no pose: a fixed on-part serpentine (up 12, right 2, down 12, right 2 ...)."""
    seq = ([1] * 12 + [0] * 2 + [3] * 12 + [0] * 2) * 10
    return lambda k, obs: seq[k]


def zigzag_policy_discrete():
    seq = ([1] * 14 + [0] * 2 + [3] * 14 + [0] * 2) * 10
    return lambda k, obs: seq[k]


def save_episodes(tag, eps):
    flat = {}
    for name, ep in eps.items():
        for k, v in ep.items():
            flat['%s/%s' % (name, k)] = v
    flat['episodes'] = json.dumps(sorted(eps))
    np.savez_compressed(os.path.join(HERE, 'episodes_%s.npz' % tag), **flat)
    for name, ep in sorted(eps.items()):
        print('%-6s %-26s steps %3d  return %8.3f  done %s  painted %d' % (
            tag, name, len(ep['actions']), float(ep['total_return']), bool(ep['done'][-1]),
            int(np.unpackbits(ep['snaps'][-1], bitorder='little').sum())))


if __name__ == '__main__':
    if '--termination' in sys.argv:
        main_termination()
    elif '--param-test-modes' in sys.argv:
        main_param_test_modes()
    elif '--textures' in sys.argv:
        main_textures()
    elif '--timing' in sys.argv:
        main_timing()
    elif '--off-part' in sys.argv:
        main_off_part()
    elif '--every-step' in sys.argv:
        main_every_step()
    else:
        main()
