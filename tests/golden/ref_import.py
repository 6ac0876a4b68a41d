"""Import the PaintRL reference from /root/reference against tests/golden/refstubs.

TEST INFRASTRUCTURE, development container only (the reference does not exist
on the GPU box).  Used by make_golden.py and by the in-container extra checks.
"""
import contextlib
import importlib
import io
import os
import sys

REFERENCE_ROOT = os.environ.get('PAINTRL_REFERENCE', '/root/reference')
_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(os.path.dirname(_HERE))


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, 'PaintRLEnv', 'bullet_paint_wrapper.py'))


def load_reference(collision_mode='hull'):
    """Return the modules (rge, bpw, rob, pte, pybullet_stub)."""
    if not reference_available():
        raise RuntimeError('reference not present at %s' % REFERENCE_ROOT)
    for p in (os.path.join(REFERENCE_ROOT, 'PaintRLEnv'), os.path.join(_HERE, 'refstubs'), _REPO):
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
    import pybullet as stub
    stub.COLLISION_MODE = collision_mode
    bpw = importlib.import_module('bullet_paint_wrapper')
    rob = importlib.import_module('robot')
    rge = importlib.import_module('robot_gym_env')
    pte = importlib.import_module('param_test_env')
    return rge, bpw, rob, pte, stub


def make_env(urdf_root, part_no=0, obs_mode='section', obs_grad=4, extra=None, rollout=False,
             collision_mode='hull', paint_method='fast', action_mode=('discrete', 1, 4)):
    """Construct the reference PaintGymEnv (renders=True: the headless branch is broken,
    SURVEY.md §0.5) and silence its GUI-side prints afterwards (App. D)."""
    rge, bpw, rob, pte, stub = load_reference(collision_mode)
    cfg = dict(rge.PaintGymEnv.EXTRA_CONFIG)
    cfg['Part_NO'] = part_no
    if extra:
        cfg.update(extra)
    mode, shape, gran = action_mode
    rge.PaintGymEnv.change_action_mode(shape, mode, gran)
    rge.PaintGymEnv.OBS_MODE = obs_mode
    rge.PaintGymEnv.OBS_GRAD = obs_grad
    rob.Robot.PAINT_METHOD = paint_method
    with contextlib.redirect_stdout(io.StringIO()):
        env = rge.PaintGymEnv(urdf_root, with_robot=False, renders=True, render_video=False,
                              rollout=rollout, extra_config=cfg)
    env._renders = False
    part = bpw._urdf_cache[env._part_id]
    part._render = False
    return env, part
