"""Host-side (setup-time) geometry used by the part-table builder.

Everything here is float64 numpy with a *fixed operation order*, because the
same arithmetic is restated on the device (``csrc/paintrl_kernels.hip``) and the
results are compared bit for bit.

Two definitions are this project's own, because the reference delegates them to
Bullet (un-vendored, un-pinned):

* ``ray_closest_hit`` -- what ``pybullet.rayTestBatch`` returns at
  ``PaintRLEnv/bullet_paint_wrapper.py:873,918``: closest two-sided
  Moller-Trumbore hit of the segment against an explicit collision triangle
  set (convex-hull facets by default).
* ``quat_rotate`` / ``transform_point`` -- ``pybullet.multiplyTransforms`` as
  used at ``PaintRLEnv/robot.py:104,267``.

``dot_fma`` reproduces ``numpy.dot`` on short float64 vectors, which the
reference calls in ``BarycentricInterpolator._get_bary_coordinate``
(``bullet_paint_wrapper.py:154-163``): OpenBLAS ddot accumulates with fused
multiply-adds, ``fma(a2,b2, fma(a1,b1, a0*b0))``.  ``numpy.vecdot`` runs the
same routine per row, which is what lets the builder vectorise it.
"""
import numpy as np

# Ray / triangle tolerances (project-defined, see DESIGN.md "Ray semantics").
RAY_EPS_DET = 1e-12    # |det| below this: segment parallel to (or triangle degenerate)
RAY_EPS_BARY = 1e-9    # barycentric slack that closes cracks between adjacent facets


def dot_fma(a, b):
    """Row-wise numpy.dot (BLAS ddot rounding) of (...,K) arrays, K small."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    if a.ndim == 1 and b.ndim == 1:
        return np.dot(a, b)
    return np.vecdot(a, b)


def cross3(a, b):
    """Plain (unfused) cross product, last axis = 3."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                     a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], axis=-1)


def dot3_plain(a, b):
    """(a0*b0 + a1*b1) + a2*b2 with no fusing -- the ray routine's dot."""
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def quat_rotate(q, v):
    """Rotate v by unit quaternion q=(x,y,z,w):  v + w*t + qv x t,  t = 2*(qv x v)."""
    q = np.asarray(q, dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    qv = q[..., :3]
    t = 2.0 * cross3(qv, v)
    return (v + q[..., 3:4] * t) + cross3(qv, t)


def transform_point(pos, quat, point):
    """multiplyTransforms(pos, quat, point, identity)[0] = pos + R(quat)*point."""
    return np.asarray(pos, dtype=np.float64) + quat_rotate(quat, point)


def quat_multiply(a, b):
    """Hamilton product of xyzw quaternions (only the stub's second return value)."""
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return (aw * bx + ax * bw + ay * bz - az * by,
            aw * by - ax * bz + ay * bw + az * bx,
            aw * bz + ax * by - ay * bx + az * bw,
            aw * bw - ax * bx - ay * by - az * bz)


def pack_collision_triangles(tri):
    """(C,3,3) corner array -> (v0, e1, e2) as the device stores them."""
    tri = np.ascontiguousarray(tri, dtype=np.float64)
    v0 = tri[:, 0, :].copy()
    e1 = tri[:, 1, :] - tri[:, 0, :]
    e2 = tri[:, 2, :] - tri[:, 0, :]
    return v0, e1, e2


def ray_closest_hit(v0, e1, e2, origins, dests, chunk=512):
    """Closest hit of each segment origin->dest against the triangle set.

    Returns (index, t, position); index = -1 and t = inf on a miss.  Among equal
    t the lowest triangle index wins.  position = origin + t*(dest-origin),
    evaluated as a multiply followed by an add.
    """
    origins = np.atleast_2d(np.asarray(origins, dtype=np.float64))
    dests = np.atleast_2d(np.asarray(dests, dtype=np.float64))
    n = origins.shape[0]
    best_t = np.full(n, np.inf)
    best_i = np.full(n, -1, dtype=np.int64)
    for s in range(0, n, chunk):
        o = origins[s:s + chunk, None, :]
        d = dests[s:s + chunk, None, :] - o
        p = cross3(d, e2[None])
        det = dot3_plain(e1[None], p)
        ok = np.abs(det) >= RAY_EPS_DET
        with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
            inv = 1.0 / det
            sv = o - v0[None]
            u = dot3_plain(sv, p) * inv
            q = cross3(sv, e1[None])
            v = dot3_plain(d, q) * inv
            t = dot3_plain(e2[None], q) * inv
            hit = ok & (u >= -RAY_EPS_BARY) & (v >= -RAY_EPS_BARY) & \
                (u + v <= 1.0 + RAY_EPS_BARY) & (t >= 0.0) & (t <= 1.0)
        t = np.where(hit, t, np.inf)
        idx = np.argmin(t, axis=1)            # first minimum = lowest index
        tt = t[np.arange(t.shape[0]), idx]
        best_t[s:s + chunk] = tt
        best_i[s:s + chunk] = np.where(np.isfinite(tt), idx, -1)
    tfin = np.where(best_i >= 0, best_t, 0.0)
    pos = origins + tfin[:, None] * (dests - origins)
    return best_i, best_t, pos


def morton2(ix, iy):
    """Interleave two 16-bit integer arrays."""
    def spread(x):
        x = x.astype(np.uint32) & 0xFFFF
        x = (x | (x << 8)) & 0x00FF00FF
        x = (x | (x << 4)) & 0x0F0F0F0F
        x = (x | (x << 2)) & 0x33333333
        x = (x | (x << 1)) & 0x55555555
        return x
    return spread(ix) | (spread(iy) << 1)


def collision_triangles(vertices, faces=None, mode='hull', principal_axes=(1, 2)):
    """The triangle set rays are tested against.

    mode='hull'    : facets of the convex hull of all vertices -- the closest
                     restatement of what Bullet builds for a URDF mesh collision
                     shape without the ``concave`` flag (door_test.urdf:14-19).
    mode='trimesh' : every triangle of the mesh.
    Triangles are ordered by the Morton code of their centroid in the principal
    plane (ties: construction order) so that neighbouring triangles are
    neighbours in memory; the order also fixes the equal-t tie break.
    """
    vertices = np.asarray(vertices, dtype=np.float64)
    if mode == 'hull':
        from scipy.spatial import ConvexHull
        simplices = ConvexHull(vertices).simplices
        tri = vertices[simplices]
    elif mode == 'trimesh':
        if faces is None:
            raise ValueError('trimesh collision mode needs the face list')
        tri = vertices[np.asarray(faces, dtype=np.int64)]
    else:
        raise ValueError('unknown collision mode %r' % (mode,))
    cen = tri.mean(axis=1)
    a1, a2 = principal_axes
    lo = vertices.min(axis=0)
    span = np.maximum(vertices.max(axis=0) - lo, 1e-12)
    q1 = np.clip(((cen[:, a1] - lo[a1]) / span[a1] * 1023.0).astype(np.int64), 0, 1023)
    q2 = np.clip(((cen[:, a2] - lo[a2]) / span[a2] * 1023.0).astype(np.int64), 0, 1023)
    order = np.argsort(morton2(q1, q2), kind='stable')
    return np.ascontiguousarray(tri[order])
