"""Stand-in for the ~25 ``pybullet`` names PaintRL touches.

TEST INFRASTRUCTURE ONLY: lets ``tests/golden/make_golden.py`` import the
reference from /root/reference in the development container.  No reference
source is reproduced here.  Two calls carry arithmetic and are defined by THIS
project (Bullet is un-vendored and un-pinned in the reference):

* ``rayTestBatch``      -> paintrl_amd.geometry.ray_closest_hit on the collision
                           triangle set chosen by ``COLLISION_MODE``
* ``multiplyTransforms`` -> paintrl_amd.geometry.transform_point / quat_multiply

Everything else is a no-op or bookkeeping.
"""
import os as _os
import xml.etree.ElementTree as _Et

import numpy as _np

from paintrl_amd import geometry as _geo
from paintrl_amd import obj_io as _obj_io

STUB_VERSION = 'paintrl_amd-refstub-1'
COLLISION_MODE = 'hull'          # 'hull' | 'trimesh'; recorded in every fixture
RAY_SECONDS = [0.0]              # accumulated time inside rayTestBatch (for BASELINE timing)


class error(Exception):
    pass


SHARED_MEMORY, GUI, DIRECT = 3, 1, 2
URDF_ENABLE_SLEEPING, URDF_USE_SELF_COLLISION = 2048, 8
POSITION_CONTROL = 2
ER_BULLET_HARDWARE_OPENGL = 131072

_bodies = {}
_search_path = ['']
_next_id = [0]


def connect(mode, *args, **kwargs):
    return -1 if mode == SHARED_MEMORY else 0


def disconnect(*args, **kwargs):
    _bodies.clear()
    _next_id[0] = 0


def resetSimulation(*a, **k):
    _bodies.clear()
    _next_id[0] = 0


def _noop(*a, **k):
    return None


setTimeStep = setPhysicsEngineParameter = setGravity = resetDebugVisualizerCamera = _noop
changeVisualShape = changeTexture = stepSimulation = removeAllUserDebugItems = _noop
removeUserDebugItem = _noop


def setAdditionalSearchPath(path):
    _search_path[0] = path


def addUserDebugLine(*a, **k):
    return 0


def addUserDebugText(*a, **k):
    return 0


def loadTexture(path):
    return 0


def getQuaternionFromEuler(e):
    r, p, y = [0.5 * float(v) for v in e]
    cr, sr, cp, sp, cy, sy = _np.cos(r), _np.sin(r), _np.cos(p), _np.sin(p), _np.cos(y), _np.sin(y)
    return (sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy,
            cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy)


def computeViewMatrixFromYawPitchRoll(*a, **k):
    return tuple([0.0] * 16)


def loadURDF(path, basePosition=(0, 0, 0), baseOrientation=(0, 0, 0, 1), useFixedBase=False, flags=0, **k):
    body_id = _next_id[0]
    _next_id[0] += 1
    body = {'pos': tuple(float(v) for v in basePosition),
            'orn': tuple(float(v) for v in baseOrientation), 'tris': None}
    if _os.path.isfile(path):
        root = _Et.parse(path).getroot()
        mesh = root.findall('./link/collision/geometry/mesh')
        if mesh:
            obj_path = _os.path.join(_os.path.dirname(path), mesh[0].get('filename'))
            mesh_data = _obj_io.read_obj(obj_path)
            verts = mesh_data.vertices + _np.asarray(body['pos'])
            axes = _obj_io.principal_axes(verts)[0]
            tri = _geo.collision_triangles(verts, mesh_data.faces_v, COLLISION_MODE, axes)
            body['tris'] = _geo.pack_collision_triangles(tri)
    _bodies[body_id] = body
    return body_id


def getBasePositionAndOrientation(body_id):
    b = _bodies[body_id]
    return b['pos'], b['orn']


def multiplyTransforms(pa, qa, pb, qb):
    pos = _geo.transform_point(pa, qa, pb)
    return tuple(float(v) for v in pos), tuple(float(v) for v in _geo.quat_multiply(qa, qb))


def rayTestBatch(ray_from, ray_to, *a, **k):
    import time as _time
    t0 = _time.perf_counter()
    ray_from = _np.asarray(ray_from, dtype=_np.float64).reshape(-1, 3)
    ray_to = _np.asarray(ray_to, dtype=_np.float64).reshape(-1, 3)
    n = ray_from.shape[0]
    best_t = _np.full(n, _np.inf)
    best_body = _np.full(n, -1, dtype=_np.int64)
    best_pos = _np.zeros((n, 3))
    for body_id, b in _bodies.items():
        if b['tris'] is None:
            continue
        idx, t, pos = _geo.ray_closest_hit(*b['tris'], ray_from, ray_to)
        better = (idx >= 0) & (t < best_t)
        best_t[better] = t[better]
        best_body[better] = body_id
        best_pos[better] = pos[better]
    out = []
    for i in range(n):
        if best_body[i] < 0:
            out.append((-1, -1, 1.0, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0)))
        else:
            out.append((int(best_body[i]), -1, float(best_t[i]),
                        tuple(float(v) for v in best_pos[i]), (0.0, 0.0, 0.0)))
    RAY_SECONDS[0] += _time.perf_counter() - t0
    return out
