"""Golden data for the parts beyond door_test / square -- TEST INFRASTRUCTURE, development container only.

1. g0_reference_parts.json: static-table digests (G0) of EVERY entry of the reference's Part_Dict
   (PaintRLEnv/robot_gym_env.py:106-117), produced by importing the reference on its own meshes under
   /root/reference (construction takes the reference 15 s - 10 min per part).  Only digests are stored.
2. episodes_door_big.npz + g0_tables_door_big.npz: the synthetic door on a 480 x 480 texture ('door_rr_big',
   ~38 000 front samples: the large-part kernels), recorded from the reference like every other fixture.
3. episodes_reference_door_rr.npz: one episode on the reference's own door_rr.urdf (Part_NO 5), for the
   in-container check of the CPU oracle against the reference on a real large part (tables are rebuilt from
   /root/reference at test time; the test is skipped where the reference is absent).

    python tests/golden/make_golden_parts.py [digests] [big] [door_rr] [reference_meshes] [hsi] [hsi_cone] [sparse]
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_import  # noqa: E402
from make_golden import RefDriver, random_policy, table_digest, zigzag_policy_grid  # noqa: E402
from paintrl_amd import synth_parts  # noqa: E402

DIGEST_KEYS = ('P', 'P_back', 'T', 'V', 'side_counts', 'axes', 'ranges', 'lwr', 'density', 'n_start_all', 'n_start_edge',
               'sha_pix', 'sha_pos', 'sha_sides', 'sha_front_normals', 'sha_start_points', 'sha_start_points_edge',
               'sha_side_vertices', 'sha_cells4', 'grid_lo', 'grid_hi', 'beams', 'construct_s')


def jsonable(v):
    v = np.asarray(v)
    if v.ndim == 0:
        return v.item()
    return v.tolist()


def reference_digests():
    root = os.path.join(ref_import.REFERENCE_ROOT, 'PaintRLEnv')
    path = os.path.join(HERE, 'g0_reference_parts.json')
    out = json.load(open(path)) if os.path.isfile(path) else {}
    rge = ref_import.load_reference('hull')[0]
    for part_no in sorted(rge.Part_Dict):
        name = rge.Part_Dict[part_no][0]
        if name in out:
            continue
        # one env per process in the reference: with renders=True nothing resets the (stub) physics world, and the
        # previous part's body would intercept this part's rays (bpw:874 checks the body id of the closest hit)
        sys.modules['pybullet'].resetSimulation()
        drv = RefDriver(root, part_no)
        d = table_digest(drv)
        out[name] = {k: jsonable(d[k]) for k in DIGEST_KEYS}
        out[name]['part_no'] = part_no
        print(name, 'P', out[name]['P'], 'constructed in %.0f s' % out[name]['construct_s'], flush=True)
        with open(path, 'w') as f:                      # written part by part: the big ones take minutes
            json.dump(out, f, indent=1, sort_keys=True)


def save_episodes(path, eps):
    flat = {'episodes': json.dumps(sorted(eps))}
    for name, ep in eps.items():
        for k, v in ep.items():
            flat['%s/%s' % (name, k)] = v
    np.savez_compressed(path, **flat)


def big_synthetic():
    root = os.path.join(HERE, '_synth_root')
    synth_parts.write_synthetic_parts(root, names=('door_rr_big',))
    big = RefDriver(root, 8)
    np.savez_compressed(os.path.join(HERE, 'g0_tables_door_big.npz'), **table_digest(big))
    # Part_Dict gives Part_NO 8 a "max possible points" of 0 (rge:115), with which every episode is "finished" after
    # its first step (rge:292); the constant is the hand-entered knob of that table, set here to 95 % of the samples
    big.env._max_possible_point = 36000
    eps = {}
    big.configure('section', 4, 'all')
    for s in range(3):
        eps['g11_big_all_%d' % s] = big.episode(400 + s, random_policy(70 + s), max_steps=120)
    big.configure('grid', 4, 'anchor', overlap=True)
    eps['g11_big_grid_overlap'] = big.episode(21, zigzag_policy_grid(), max_steps=60, want_idx=0)
    big.configure('section', 6, 'anchor')
    eps['g11_big_section6'] = big.episode(410, random_policy(75), max_steps=40, want_idx=1)
    save_episodes(os.path.join(HERE, 'episodes_door_big.npz'), eps)


def reference_door_rr():
    root = os.path.join(ref_import.REFERENCE_ROOT, 'PaintRLEnv')
    drv = RefDriver(root, 5)
    eps = {}
    drv.configure('section', 4, 'all')
    for s in range(2):
        eps['g12_door_rr_%d' % s] = drv.episode(500 + s, random_policy(80 + s), max_steps=80)
    # (the reference's GridObservation raises KeyError on this part, bpw:1101: section observations only)
    drv.configure('section', 4, 'anchor', overlap=True, turning=True)
    eps['g12_door_rr_sweep'] = drv.episode(21, zigzag_policy_grid(), max_steps=50, want_idx=0)
    save_episodes(os.path.join(HERE, 'episodes_reference_door_rr.npz'), eps)


def reference_meshes(which=('door_test', 'square')):
    """Episodes on the two parts the reference's published results use, on the reference's OWN meshes (rge:106-108:
    door_test.urdf = Part_NO 0, square.urdf = Part_NO 1; zigzag.py:8, spiral.py:8, paint_ppo.py:95): section observation
    from 'all' and anchor starts, grid + both penalties, the cone beams, COLOR_MODE 'HSI', and the texture image at the
    end of one episode (the real pattern.jpg under the labels).  Replayed by the oracle on tables rebuilt from
    /root/reference at test time (tests/test_reference_parts.py, skipped where the reference is absent): the meshes do
    not travel, the recorded trajectories do."""
    from make_golden import texture_of, zigzag_policy
    root = os.path.join(ref_import.REFERENCE_ROOT, 'PaintRLEnv')
    for part_no, name in ((0, 'door_test'), (1, 'square')):
        if name not in which:
            continue
        sys.modules.get('pybullet') and sys.modules['pybullet'].resetSimulation()
        drv = RefDriver(root, part_no)
        eps = {}
        drv.configure('section', 4, 'all')
        for s in range(3):
            eps['g16_%s_all_%d' % (name, s)] = drv.episode(700 + s, random_policy(100 + s), max_steps=150)
        drv.configure('section', 4, 'anchor')
        eps['g16_%s_anchor' % name] = drv.episode(710, random_policy(110), max_steps=150)
        sweep = zigzag_policy_grid() if part_no == 0 else zigzag_policy(-1, 2)
        eps['g16_%s_sweep' % name] = drv.episode(711, sweep, max_steps=120, want_idx=0)
        eps['g16_%s_sweep' % name]['texture'] = texture_of(drv)
        drv.configure('grid', 4, 'anchor', overlap=True, turning=True)
        eps['g16_%s_grid_penalties' % name] = drv.episode(712, zigzag_policy_grid(), max_steps=80, want_idx=0)
        drv.configure('grid', 4, 'all', overlap=True)
        eps['g16_%s_grid_all' % name] = drv.episode(713, random_policy(113), max_steps=80)
        drv.configure('section', 4, 'anchor', paint_method='normal')
        eps['g16_%s_cone' % name] = drv.episode(714, zigzag_policy_grid(), max_steps=10, want_idx=0)
        drv.env.close()
        sys.modules['pybullet'].resetSimulation()
        hsi = RefDriver(root, part_no, extra={'COLOR_MODE': 'HSI'})
        hsi.configure('section', 4, 'anchor')
        n = 60
        while True:                                # a shot that hits no sample makes the reference raise: stop before it
            try:
                ep = hsi.episode(715, zigzag_policy_grid(), max_steps=n, want_idx=0)
                break
            except ValueError:
                n -= 1
        ep['final_thick'] = np.array([int(hsi.part.texels[hsi.part.get_texel(*p)]) for p in hsi.pix], dtype=np.uint8)
        ep['texture'] = texture_of(hsi)
        cfg = json.loads(str(ep['cfg']))
        cfg['color_mode'] = 'HSI'
        ep['cfg'] = json.dumps(cfg)
        eps['g16_%s_hsi' % name] = ep
        hsi.env.close()
        save_episodes(os.path.join(HERE, 'episodes_reference_%s.npz' % name), eps)
        for k, ep in sorted(eps.items()):
            print('%-28s steps %3d return %8.3f done %s' % (k, len(ep['actions']), float(ep['total_return']),
                                                            bool(ep['done'][-1])), flush=True)


def seam_synthetic():
    """The synthetic sheet with a seam of doubled vertices ('door_lf', Part_NO 2; synth_parts.seam_sheet): two vertices at one
    position are equally near to every query, and which one cKDTree.query returns decides the triangles the hook point chooses
    from.  Sweeps across the seam + random episodes, recorded from the reference -> episodes_seam.npz, g0_tables_seam.npz."""
    root = os.path.join(HERE, '_synth_root')
    synth_parts.write_synthetic_parts(root, names=('door_lf',))
    drv = RefDriver(root, 2)
    np.savez_compressed(os.path.join(HERE, 'g0_tables_seam.npz'), **table_digest(drv))
    eps = {}
    across = ([0] * 14 + [1] * 2 + [2] * 14 + [1] * 2) * 8
    drv.configure('section', 4, 'anchor', max_points=14350)
    eps['g17_seam_across'] = drv.episode(30, lambda k, obs: across[k], max_steps=200, want_idx=0)
    drv.configure('grid', 4, 'anchor', overlap=True, turning=True, max_points=14350)
    eps['g17_seam_across_grid'] = drv.episode(31, lambda k, obs: across[k], max_steps=120, want_idx=0)
    drv.configure('section', 4, 'all', max_points=14350)
    for s in range(4):
        eps['g17_seam_all_%d' % s] = drv.episode(800 + s, random_policy(120 + s), max_steps=150)
    drv.configure('section', 4, 'anchor', paint_method='normal', max_points=14350)
    eps['g17_seam_cone'] = drv.episode(32, lambda k, obs: across[k], max_steps=12, want_idx=0)
    save_episodes(os.path.join(HERE, 'episodes_seam.npz'), eps)
    for k, ep in sorted(eps.items()):
        print('%-24s steps %3d return %8.3f done %s' % (k, len(ep['actions']), float(ep['total_return']), bool(ep['done'][-1])), flush=True)


def sparse_synthetic():
    """The coarse synthetic sheet ('test', Part_NO 9): the reference moves vertex rows under its kd-tree there."""
    root = os.path.join(HERE, '_synth_root')
    synth_parts.write_synthetic_parts(root, names=('test',))
    drv = RefDriver(root, 9)
    np.savez_compressed(os.path.join(HERE, 'g0_tables_sparse.npz'), **table_digest(drv))
    eps = {}
    drv.configure('section', 4, 'all')
    for s in range(6):
        eps['g14_sparse_all_%d' % s] = drv.episode(600 + s, random_policy(90 + s), max_steps=150)
    drv.configure('grid', 4, 'anchor', overlap=True)
    eps['g14_sparse_grid'] = drv.episode(21, zigzag_policy_grid(), max_steps=120, want_idx=0)
    save_episodes(os.path.join(HERE, 'episodes_sparse.npz'), eps)


def hsi_door():
    """COLOR_MODE='HSI' (thickness bytes, bpw:384-434) on the synthetic door: on-part policies only -- a shot that
    hits no sample makes the reference raise (max of an empty array), such an episode is cut before that step."""
    root = os.path.join(HERE, '_synth_root')
    synth_parts.write_synthetic_parts(root)
    drv = RefDriver(root, 0, extra={'COLOR_MODE': 'HSI'})
    eps = {}

    def run(name, seed, policy, max_steps, want_idx):
        try:
            ep = drv.episode(seed, policy, max_steps=max_steps, want_idx=want_idx)
        except ValueError:                       # off-part shot: record the part before it
            for cut in range(max_steps - 1, 0, -1):
                try:
                    ep = drv.episode(seed, policy, max_steps=cut, want_idx=want_idx)
                    break
                except ValueError:
                    continue
        part = drv.part
        ep['final_thick'] = np.array([int(part.texels[part.get_texel(*p)]) for p in drv.pix], dtype=np.uint8)
        cfg = json.loads(str(ep['cfg']))
        cfg['color_mode'] = 'HSI'
        ep['cfg'] = json.dumps(cfg)
        eps[name] = ep

    drv.configure('section', 4, 'anchor')
    run('g13_hsi_serpentine', 8, zigzag_policy_grid(), 150, 0)
    run('g13_hsi_random', 101, random_policy(11), 30, 2)
    drv.configure('grid', 4, 'anchor', overlap=True, turning=True)
    run('g13_hsi_grid_overlap', 21, zigzag_policy_grid(), 80, 0)
    save_episodes(os.path.join(HERE, 'episodes_door_hsi.npz'), eps)


def hsi_cone_door():
    """COLOR_MODE='HSI' with PAINT_METHOD='normal' (rob:38-69 beta-profile cone + bpw:384-434, 562-566): the beam table is
    drawn with random.uniform when the env is constructed (rob:244-249), so the generator seeds `random` first and the
    table is recorded with the episodes (it is data).  A beam cone paints the sample nearest to every hit, once per hit:
    a sample under several beams receives several deposits in one shot."""
    import random
    root = os.path.join(HERE, '_synth_root')
    synth_parts.write_synthetic_parts(root)
    random.seed(4242)
    drv = RefDriver(root, 0, extra={'COLOR_MODE': 'HSI'})
    beams = np.array(drv.env.robot._paint_plain, dtype=np.float64)
    eps = {}

    def run(name, seed, policy, max_steps, want_idx):
        ep = drv.episode(seed, policy, max_steps=max_steps, want_idx=want_idx)
        part = drv.part
        ep['final_thick'] = np.array([int(part.texels[part.get_texel(*p)]) for p in drv.pix], dtype=np.uint8)
        ep['beams'] = beams
        cfg = json.loads(str(ep['cfg']))
        cfg['color_mode'] = 'HSI'
        cfg['beams_seed'] = 4242
        ep['cfg'] = json.dumps(cfg)
        eps[name] = ep

    drv.configure('section', 4, 'anchor', paint_method='normal')
    run('g15_hsi_cone_serpentine', 8, zigzag_policy_grid(), 40, 0)
    run('g15_hsi_cone_random', 101, random_policy(11), 30, 2)
    drv.configure('grid', 4, 'all', overlap=True, paint_method='normal')
    run('g15_hsi_cone_grid_overlap', 203, random_policy(43), 25, None)
    save_episodes(os.path.join(HERE, 'episodes_door_hsi_cone.npz'), eps)
    print('beams', beams.shape, 'density', drv.part.get_density())


if __name__ == '__main__':
    what = sys.argv[1:] or ['digests', 'big', 'door_rr']
    if 'seam' in what:
        seam_synthetic()
    if 'reference_meshes' in what:
        reference_meshes()
    if 'hsi_cone' in what:
        hsi_cone_door()
    if 'hsi' in what:
        hsi_door()
    if 'sparse' in what:
        sparse_synthetic()
    if 'big' in what:
        big_synthetic()
    if 'door_rr' in what:
        reference_door_rr()
    if 'digests' in what:
        reference_digests()
