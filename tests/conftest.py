import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


_TABLES = {}


def synthetic_tables(name, paint_radius=0.051, tex_size=None):
    """Session cache of PartTables for the synthetic parts ('door_test' | 'square' | 'door_rr_big'); ``tex_size``
    overrides the part's texture size (more texels = more coverage samples on the same mesh)."""
    from paintrl_amd import part_tables, synth_parts
    tex_size = tuple(tex_size or synth_parts.TEXTURES[name][0])
    key = (name, paint_radius, tex_size)
    if key not in _TABLES:
        _TABLES[key] = part_tables.build_part_tables(mesh=synth_parts.synthetic_mesh(name), tex_size=tex_size,
                                                     name=name, paint_radius=paint_radius)
    return _TABLES[key]


@pytest.fixture(scope='session')
def door_tables():
    return synthetic_tables('door_test')


@pytest.fixture(scope='session')
def sheet_tables():
    return synthetic_tables('square')


def load_episodes(tag):
    """{episode name: {field: array}} from tests/golden/episodes_<tag>.npz."""
    z = np.load(os.path.join(GOLDEN, 'episodes_%s.npz' % tag), allow_pickle=False)
    out = {}
    for name in json.loads(str(z['episodes'])):
        ep = {k.split('/', 1)[1]: z[k] for k in z.files if k.startswith(name + '/')}
        ep['cfg'] = json.loads(str(ep['cfg']))
        out[name] = ep
    return out


def env_kwargs_from_cfg(cfg):
    """Golden cfg dict -> keyword arguments shared by oracle.Oracle and paintrl_amd.BatchedPaintEnv."""
    return dict(obs_mode=cfg['obs_mode'], obs_grad=cfg['obs_grad'], action_mode=cfg['action_mode'],
                action_dim=cfg['action_dim'], n_discrete=cfg['n_discrete'],
                termination_mode=cfg['termination_mode'], turning_penalty=cfg['turning_penalty'],
                overlap_penalty=cfg['overlap_penalty'], paint_method=cfg['paint_method'],
                max_episode_len=cfg['max_episode_len'], expected_episode_len=cfg['expected_episode_len'],
                switch_threshold=cfg['switch_threshold'], max_possible_point=cfg['max_possible_point'],
                paint_radius=cfg.get('paint_radius', 0.051), step_size=cfg.get('step_size', 0.051))


def start_points_for(tables, mode):
    from paintrl_amd import part_tables
    return part_tables.start_points(tables, mode)
