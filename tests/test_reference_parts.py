"""The reference's OWN parts (PaintRLEnv/urdf/painting/*.urdf, Part_Dict rge:106-117), read from /root/reference at
test time -- these tests are skipped wherever the reference is absent (the GPU box).

* every Part_Dict entry builds, and its static tables equal the digests recorded from the imported reference
  (tests/golden/g0_reference_parts.json, written by tests/golden/make_golden_parts.py: only digests are stored);
* the CPU oracle replays episodes the reference recorded on its own door_rr.urdf (Part_NO 5: 17 891 samples), and on
  the two parts its published results use -- door_test.urdf (Part_NO 0) and square.urdf (Part_NO 1, the stale kd-tree
  case): section / grid + penalties / 'all' starts / cone beams / COLOR_MODE 'HSI', and the texture image at the end
  (tests/golden/episodes_reference_{door_test,square}.npz, make_golden_parts.py reference_meshes).
"""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, env_kwargs_from_cfg, load_episodes

REFERENCE = os.environ.get('PAINTRL_REFERENCE', '/root/reference')
PARTS_DIR = os.path.join(REFERENCE, 'PaintRLEnv', 'urdf', 'painting')
DIGESTS = os.path.join(GOLDEN, 'g0_reference_parts.json')
pytestmark = pytest.mark.skipif(not os.path.isdir(PARTS_DIR), reason='the reference is not present on this machine')

_CACHE = {}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def reference_tables(name):
    if name not in _CACHE:
        from paintrl_amd import part_tables
        _CACHE[name] = part_tables.build_part_tables(os.path.join(PARTS_DIR, name))
    return _CACHE[name]


def _recorded():
    return json.load(open(DIGESTS)) if os.path.isfile(DIGESTS) else {}


def test_part_dict_is_the_reference_table():
    from paintrl_amd.config import Part_Dict
    assert sorted(v[0] for v in Part_Dict.values()) == sorted(
        ['door_test.urdf', 'square.urdf', 'door_lf.urdf', 'door_lr.urdf', 'door_rf.urdf', 'door_rr.urdf', 'roof.urdf',
         'bonnet.urdf', 'door_rr_big.urdf', 'test.urdf'])
    for name, _ in Part_Dict.values():
        assert os.path.isfile(os.path.join(PARTS_DIR, name)), name


@pytest.mark.parametrize('name', ['door_test.urdf', 'square.urdf', 'door_lf.urdf', 'door_lr.urdf', 'door_rf.urdf',
                                  'door_rr.urdf', 'roof.urdf', 'bonnet.urdf', 'door_rr_big.urdf', 'test.urdf'])
def test_tables_of_reference_part_equal_the_reference(name):
    """G0 on the reference's own meshes: sample set and positions, sides, corrected normals, side vertices (after
    the reference's in-place row mutation), grid rows, start points, density, cone beams -- and the device layout
    packs (the seven parts beyond 16 384 samples go to the LDS-mask kernels)."""
    from paintrl_amd import part_tables
    from paintrl_amd.device_tables import DeviceTables
    rec = _recorded().get(name)
    if rec is None:
        pytest.skip('no digest recorded for %s (tests/golden/make_golden_parts.py digests)' % name)
    t = reference_tables(name)
    front = t.tri_side == 1
    assert rec['P'] == t.sample_pos.shape[0] and rec['T'] == t.tri_side.shape[0] and rec['V'] == t.vertices.shape[0]
    assert rec['side_counts'] == [int((t.tri_side == k).sum()) for k in (1, 2, 3)]
    assert rec['axes'] == [t.a1, t.a2, t.a0]
    assert np.array_equal(np.array(rec['ranges']), np.array(t.ranges)) and rec['lwr'] == t.lwr
    assert np.array_equal(np.array(rec['grid_lo']), t.grid_lo) and np.array_equal(np.array(rec['grid_hi']), t.grid_hi)
    assert rec['density'] == t.density and np.array_equal(np.array(rec['beams']).reshape(-1, 3), t.beams)
    assert rec['sha_pix'] == sha(t.sample_pix.astype(np.int32)) and rec['sha_pos'] == sha(t.sample_pos)
    assert rec['sha_sides'] == sha(t.tri_side.astype(np.int8))
    assert rec['sha_front_normals'] == sha(t.tri_normal[front])
    assert rec['sha_side_vertices'] == sha(t._side_data)
    sp = np.array(part_tables.start_points(t, 'all'), dtype=np.float64)
    assert rec['n_start_all'] == sp.shape[0] and rec['sha_start_points'] == sha(sp)
    se = np.array(part_tables.start_points(t, 'edge'), dtype=np.float64)
    assert rec['n_start_edge'] == se.shape[0] and rec['sha_start_points_edge'] == sha(se)
    d = DeviceTables(t, start_points=part_tables.start_points(t, 'anchor'))
    assert d.n_words <= 1600 and int(np.unpackbits(d.word_valid.view(np.uint8)).sum()) == t.sample_pos.shape[0]


@pytest.mark.skipif(not os.path.isfile(os.path.join(GOLDEN, 'episodes_reference_door_rr.npz')), reason='fixture not generated')
@pytest.mark.parametrize('name', ['g12_door_rr_0', 'g12_door_rr_1', 'g12_door_rr_sweep'])
def test_oracle_replays_reference_episode_on_door_rr(name):
    from paintrl_amd import part_tables
    from test_oracle_golden import replay
    ep = load_episodes('reference_door_rr')[name]
    cfg = ep['cfg']
    tables = reference_tables('door_rr.urdf')
    orc = oracle.Oracle(tables, 1, start_points=part_tables.start_points(tables, cfg['start_mode']),
                        **env_kwargs_from_cfg(cfg))

    def reset(idx):
        return orc.reset([idx])[0]

    def step(a, want_bits):
        obs, rew, done, info = orc.step([a])
        return obs[0], rew[0], done[0], info[0], orc.painted_bits(0)

    replay(step, reset, ep, exact=True)
    st = orc.state(0)
    assert np.array_equal(st['pose'], ep['final_pose']) and np.array_equal(st['quat'], ep['final_quat'])
    assert st['total_return'] == float(ep['total_return'])


def _reference_cases():
    out = []
    for part in ('door_test', 'square'):
        path = os.path.join(GOLDEN, 'episodes_reference_%s.npz' % part)
        if os.path.isfile(path):
            out += [(part, n) for n in sorted(load_episodes('reference_%s' % part))]
    return out


@pytest.mark.parametrize('part,name', _reference_cases())
def test_oracle_replays_reference_episode_on_the_published_parts(part, name):
    """SURVEY 8c "in-container-only extra check": the reference's own door_test.urdf / square.urdf (rge:106-108) -- tables
    rebuilt here from /root/reference, the recorded episode replayed by the oracle exactly (observations, rewards, done,
    painted-texel snapshots, final pose and return; COLOR_MODE 'HSI': rewards to 1e-12, thickness bytes exactly), and where
    the fixture holds one, the reference's texture image at the end (real pattern.jpg under the labels)."""
    from paintrl_amd import part_tables
    from test_oracle_golden import replay
    ep = load_episodes('reference_%s' % part)[name]
    cfg = ep['cfg']
    tables = reference_tables(part + '.urdf')
    hsi = cfg.get('color_mode', 'RGB') == 'HSI'
    orc = oracle.Oracle(tables, 1, start_points=part_tables.start_points(tables, cfg['start_mode']),
                        color_mode=cfg.get('color_mode', 'RGB'), **env_kwargs_from_cfg(cfg))

    def reset(idx):
        return orc.reset([idx])[0]

    def step(a, want_bits):
        obs, rew, done, info = orc.step([a])
        return obs[0], rew[0], done[0], info[0], orc.painted_bits(0)

    replay(step, reset, ep, exact=not hsi, atol=1e-12)
    st = orc.state(0)
    assert np.array_equal(st['pose'], ep['final_pose']) and np.array_equal(st['quat'], ep['final_quat'])
    if hsi:
        assert np.array_equal(orc.thick[0], ep['final_thick'])
    else:
        assert st['total_return'] == float(ep['total_return'])
    if 'texture' in ep:
        img = (part_tables.texture_image(tables, thickness=orc.thick[0], color_mode='HSI') if hsi
               else part_tables.texture_image(tables, painted=orc.painted_bits(0)))
        assert np.array_equal(img, ep['texture'])
