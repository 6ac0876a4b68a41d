"""Deterministic synthetic paint parts: a door-like curved panel with a window
cut-out and a quadratic sheet (BASELINE.json "synthetic
door-panel meshes").

The files are laid out as ``<root>/urdf/painting/{door_test,square}.{obj,mtl,urdf}``
plus ``pattern.jpg`` so that the reference's ``Part_Dict``
(``PaintRLEnv/robot_gym_env.py:106-117``) and this package load them unmodified.

Design targets (reference door_test.obj): ~13.3k triangles of
which ~4.9k face the front, ~2.8k front vertices, ~9.7k front texel samples on a
240x240 texture, ~740 convex-hull facets.  The last figure is met by shaping the
front as a convex polyhedral envelope on a coarse lattice with the fine mesh
dimpled slightly below it, so only lattice nodes are hull vertices.  The fine
mesh is a jittered Delaunay triangulation (no lattice ties, and every one of the
100 grid rows holds several vertices, so the reference's sparse-row vertex
mutation, SURVEY.md H4, never triggers).
"""
import os

import numpy as np

from . import obj_io


def _boundary_loop(pts, h):
    """Sample a closed polygon (list of 2-D corners) at spacing ~h, corners kept."""
    out = []
    n = len(pts)
    for k in range(n):
        a = np.asarray(pts[k], dtype=np.float64)
        b = np.asarray(pts[(k + 1) % n], dtype=np.float64)
        m = max(1, int(round(np.linalg.norm(b - a) / h)))
        for s in range(m):
            out.append(a + (b - a) * (s / m))
    return np.asarray(out)


def _inside_polygon(p, poly):
    """Even-odd test of points p (n,2) against polygon corners poly (k,2)."""
    x, y = p[:, 0], p[:, 1]
    inside = np.zeros(p.shape[0], dtype=bool)
    k = len(poly)
    for i in range(k):
        x0, y0 = poly[i]
        x1, y1 = poly[(i + 1) % k]
        cond = ((y0 > y) != (y1 > y))
        with np.errstate(divide='ignore', invalid='ignore'):
            xi = (x1 - x0) * (y - y0) / (y1 - y0) + x0
        inside ^= cond & (x < xi)
    return inside


def _dist_to_segments(p, poly):
    d = np.full(p.shape[0], np.inf)
    k = len(poly)
    for i in range(k):
        a = np.asarray(poly[i], dtype=np.float64)
        b = np.asarray(poly[(i + 1) % k], dtype=np.float64)
        ab = b - a
        tt = np.clip(((p - a) @ ab) / (ab @ ab), 0.0, 1.0)
        d = np.minimum(d, np.linalg.norm(p - (a + tt[:, None] * ab), axis=1))
    return d


def _panel(width, height, h, hole, lattice, curv, dimple, thickness, rim_layers, seed,
           uv_front, uv_back, bow=0.02, tex=240):
    """Build one closed thin panel, front facing +x.

    The mesh is generated in a parameter rectangle [0,width]x[0,height] and then
    warped so that all four edges bow outward by ``bow`` (a strictly convex
    outline).  With straight edges the hull's side walls would be planar and
    Qhull would triangulate them into facets whose projection on the principal
    plane is three collinear, non-coincident points; the reference's 2-D
    inside test (bpw:96-104) then divides by a rounding-noise denominator.
    With a bowed outline every side facet joins a front boundary node to the
    back copy of itself or its neighbour, so the projection has two coincident
    points and the denominator is exactly zero.
    """
    from scipy.spatial import Delaunay
    rng = np.random.RandomState(seed)
    outer = [(0.0, 0.0), (width, 0.0), (width, height), (0.0, height)]
    ny, nz = lattice
    ly = np.linspace(0.0, width, ny)
    lz = np.linspace(0.0, height, nz)

    # fixed points: every usable lattice node, then outer-loop and hole-loop samples that
    # are not within half a spacing of a node (nodes win, so hull vertices sit on the lattice)
    nodes = np.array([(y, z) for y in ly for z in lz])
    if hole is not None:
        nodes = nodes[~_inside_polygon(nodes, hole) & (_dist_to_segments(nodes, hole) > 0.6 * h)]
    loops = [_boundary_loop(outer, h)]
    if hole is not None:
        loops.append(_boundary_loop(hole, h))
    loop_pts = np.vstack(loops)
    from scipy.spatial import cKDTree
    loop_pts = loop_pts[cKDTree(nodes).query(loop_pts)[0] > 0.5 * h]
    fixed = np.vstack([nodes, loop_pts])

    # interior: jittered grid
    gy = np.arange(0.5 * h, width, h)
    gz = np.arange(0.5 * h, height, h)
    G = np.array([(y, z) for y in gy for z in gz])
    G = G + rng.uniform(-1.0, 1.0, size=G.shape) * np.array([0.38 * h, 0.48 * h])
    ok = (G[:, 0] > 0.45 * h) & (G[:, 0] < width - 0.45 * h) & (G[:, 1] > 0.45 * h) & (G[:, 1] < height - 0.45 * h)
    if hole is not None:
        ok &= ~_inside_polygon(G, hole) & (_dist_to_segments(G, hole) > 0.45 * h)
    G = G[ok]
    # keep away from fixed points
    dmin = cKDTree(fixed).query(G)[0]
    G = G[dmin > 0.5 * h]
    # non-node points are pulled 4e-5 (relative) toward the centre so that, after the OBJ's
    # 6-decimal rounding, boundary samples stay strictly inside the planar hull walls
    rest = np.vstack([loop_pts, G])
    ctr = np.array([0.5 * width, 0.5 * height])
    rest = ctr + (rest - ctr) * (1.0 - 4e-5)
    P2 = np.vstack([nodes, rest])

    tri = Delaunay(P2).simplices
    cen = P2[tri].mean(axis=1)
    if hole is not None:
        tri = tri[~_inside_polygon(cen, hole)]
    # drop zero-area slivers along straight boundaries, orient counter-clockwise in (y,z)
    a, b, c = P2[tri[:, 0]], P2[tri[:, 1]], P2[tri[:, 2]]
    area2 = (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])
    tri = tri[np.abs(area2) > 1e-9]
    area2 = area2[np.abs(area2) > 1e-9]
    flip = area2 < 0
    tri[flip] = tri[flip][:, [0, 2, 1]]
    used = np.unique(tri)
    remap = -np.ones(P2.shape[0], dtype=np.int64)
    remap[used] = np.arange(used.size)
    P2 = P2[used]
    tri = remap[tri]

    # height: piecewise-linear envelope of a concave quadratic on the lattice, minus a dimple
    cy, cz, x0 = curv

    def g(y, z):
        return x0 - cy * ((y - 0.5 * width) / (0.5 * width)) ** 2 - cz * ((z - 0.5 * height) / (0.5 * height)) ** 2

    def bowed(y, z):
        ps, pt = 2.0 * y / width - 1.0, 2.0 * z / height - 1.0
        return (y + bow * ps * (1.0 - pt * pt) + bow, z + bow * pt * (1.0 - ps * ps) + bow)

    def lattice_interp(p, fn):
        """Piecewise-linear interpolation (lattice cells split along the 00-11 diagonal)
        of fn evaluated at the lattice nodes; returns (values..., fy, fz)."""
        y, z = p[:, 0], p[:, 1]
        iy = np.clip(np.searchsorted(ly, y, side='right') - 1, 0, ny - 2)
        iz = np.clip(np.searchsorted(lz, z, side='right') - 1, 0, nz - 2)
        fy = (y - ly[iy]) / (ly[iy + 1] - ly[iy])
        fz = (z - lz[iz]) / (lz[iz + 1] - lz[iz])
        out = []
        for q00, q10, q01, q11 in zip(fn(ly[iy], lz[iz]), fn(ly[iy + 1], lz[iz]),
                                      fn(ly[iy], lz[iz + 1]), fn(ly[iy + 1], lz[iz + 1])):
            lower = q00 + fy * (q10 - q00) + fz * (q11 - q10)       # triangle (00,10,11)
            upper = q00 + fz * (q01 - q00) + fy * (q11 - q01)       # triangle (00,01,11)
            out.append(np.where(fy >= fz, lower, upper))
        return out, fy, fz

    (env_x, wy_, wz_), fy, fz = lattice_interp(P2, lambda y, z: (g(y, z),) + bowed(y, z))
    bump = np.sin(np.pi * fy) ** 2 + np.sin(np.pi * fz) ** 2
    xf = env_x - dimple * bump - 3e-5 * (bump > 1e-12)
    nf = P2.shape[0]
    Pw = np.stack([wy_, wz_], axis=1)
    front = np.stack([xf, Pw[:, 0], Pw[:, 1]], axis=1)
    back = np.stack([env_x - thickness, Pw[:, 0], Pw[:, 1]], axis=1)   # no dimple: walls stay planar

    # boundary edges (appear in exactly one triangle) -> rim strips
    edges = {}
    for t3 in tri:
        for k in range(3):
            e = (int(t3[k]), int(t3[(k + 1) % 3]))
            key = (min(e), max(e))
            edges.setdefault(key, []).append(e)
    bedges = [v[0] for v in edges.values() if len(v) == 1]       # oriented as in the front triangle
    bverts = sorted({v for e in bedges for v in e})
    bidx = {v: i for i, v in enumerate(bverts)}
    nb = len(bverts)
    layers = []                                                    # intermediate rim vertex layers
    for L in range(1, rim_layers):
        lay = front[bverts].copy()
        lay[:, 0] -= thickness * L / rim_layers
        layers.append(lay)
    verts = np.vstack([front, back] + layers)

    def rim_vertex(v, L):
        if L == 0:
            return v
        if L == rim_layers:
            return nf + v
        return 2 * nf + (L - 1) * nb + bidx[v]

    faces = [tuple(t3) for t3 in tri]                              # front, CCW -> +x
    faces += [(nf + t3[0], nf + t3[2], nf + t3[1]) for t3 in tri]  # back, reversed
    n_front = tri.shape[0]
    for (p, q) in bedges:
        for L in range(rim_layers):
            a0, b0 = rim_vertex(p, L), rim_vertex(q, L)
            a1, b1 = rim_vertex(p, L + 1), rim_vertex(q, L + 1)
            faces.append((a0, a1, b0))
            faces.append((b0, a1, b1))
    faces = np.asarray(faces, dtype=np.int64)

    # UVs (texture space, origin top-left; flipped to OBJ convention on write)
    (fu0, fv0, fsu, fsv) = uv_front
    (bu0, bv0, bsu, bsv) = uv_back
    wy, wz = width + 2 * bow, height + 2 * bow
    uv_f = np.stack([(fu0 + Pw[:, 0] * fsu) / tex, (fv0 + (wz - Pw[:, 1]) * fsv) / tex], axis=1)
    uv_b = np.stack([(bu0 + (wy - Pw[:, 0]) * bsu) / tex, (bv0 + (wz - Pw[:, 1]) * bsv) / tex], axis=1)
    # rim: a thin strip at the bottom of the texture, one column per boundary vertex
    uv_r = []
    for L in range(rim_layers + 1):
        for v in bverts:
            uv_r.append(((4 + 232.0 * bidx[v] / max(nb - 1, 1)) / tex, (228 + 8.0 * L / rim_layers) / tex))
    uv_r = np.asarray(uv_r)
    uvs = np.vstack([uv_f, uv_b, uv_r])

    def rim_uv(v, L):
        return 2 * nf + L * nb + bidx[v]

    faces_vt = [tuple(t3) for t3 in tri]
    faces_vt += [(nf + t3[0], nf + t3[2], nf + t3[1]) for t3 in tri]
    for (p, q) in bedges:
        for L in range(rim_layers):
            faces_vt.append((rim_uv(p, L), rim_uv(p, L + 1), rim_uv(q, L)))
            faces_vt.append((rim_uv(q, L), rim_uv(p, L + 1), rim_uv(q, L + 1)))
    faces_vt = np.asarray(faces_vt, dtype=np.int64)
    uvs_obj = uvs.copy()
    uvs_obj[:, 1] = 1.0 - uvs_obj[:, 1]
    return verts, uvs_obj, faces, faces_vt, n_front


def door_panel(seed=0):
    """1.0 x 0.9 m curved door skin with a trapezoid window, lattice 19x19."""
    hole = [(0.21, 0.49), (0.75, 0.49), (0.67, 0.78), (0.29, 0.78)]
    v, uv, f, fvt, _ = _panel(0.96, 0.86, 0.0172, hole, (19, 18), (0.045, 0.035, 0.115), 0.0016,
                              0.03, 4, seed, uv_front=(3.2, 3.3, 114.2, 114.2), uv_back=(122.2, 4.3, 112.0, 112.0))
    v = v + np.array([0.0, 0.0, 0.004])
    return v, uv, f, fvt


def quadratic_sheet(seed=1):
    """1.0 x 1.0 m paraboloid sheet, lattice 4x4 (the reference's Part_NO=1 analogue)."""
    v, uv, f, fvt, _ = _panel(0.96, 0.96, 0.027, None, (5, 5), (0.012, 0.012, 0.014), 0.0006,
                              0.002, 1, seed, uv_front=(0.3, 0.3, 121.7, 121.7), uv_back=(120.6, 0.3, 118.6, 118.6))
    v = v + np.array([0.0, 0.0, 0.001])
    return v, uv, f, fvt


def sparse_sheet(seed=2):
    """The quadratic sheet meshed four times coarser (vertex spacing ~0.11 m): most of the reference's 100 grid rows
    hold no vertex, so its _set_grid_dict moves vertex rows in place under the kd-tree (bpw:943-946) -- the case the
    reference's own coarse parts (door_lf, door_rr, ...) exercise and the fine synthetic parts never do."""
    v, uv, f, fvt, _ = _panel(0.96, 0.96, 0.11, None, (5, 5), (0.012, 0.012, 0.014), 0.0006,
                              0.002, 1, seed, uv_front=(0.3, 0.3, 121.7, 121.7), uv_back=(120.6, 0.3, 118.6, 118.6))
    v = v + np.array([0.0, 0.0, 0.001])
    return v, uv, f, fvt


def seam_sheet(seed=1):
    """The quadratic sheet with a seam: every vertex that triangles on both sides of the line y = 0.42 (local frame) share is written
    twice, the triangles beyond the line use the copy -- what OBJ exporters do along UV seams (the reference's test.obj has
    179 such vertices).  Two vertices at one position are equally near to every query: which of them cKDTree.query returns
    (bpw:526), and with it which triangles the hook point may choose from, is the tree's own business
    (part_tables._vertex_tie_rank).  Takes the Part_Dict slot of door_lf.urdf (Part_NO 2)."""
    v, uv, f, fvt = quadratic_sheet(seed)
    beyond = v[f].mean(axis=1)[:, 1] > 0.42
    used_near = np.zeros(v.shape[0], dtype=bool)
    used_far = np.zeros(v.shape[0], dtype=bool)
    used_near[f[~beyond].ravel()] = True
    used_far[f[beyond].ravel()] = True
    shared = np.nonzero(used_near & used_far)[0]
    copy_of = -np.ones(v.shape[0], dtype=np.int64)
    copy_of[shared] = v.shape[0] + np.arange(shared.size)
    v2 = np.concatenate([v, v[shared]], axis=0)
    f2 = f.copy()
    sel = f2[beyond]
    f2[beyond] = np.where(copy_of[sel] >= 0, copy_of[sel], sel)
    return v2, uv, f2, fvt


# 'door_rr_big': the door panel on a 480 x 480 texture, like the reference's Part_NO 8 (door_rr_big.urdf with
# pattern_big.jpg): ~38 000 front samples, i.e. a part whose coverage masks do not fit four words per lane.
PARTS = {'door_test': door_panel, 'square': quadratic_sheet, 'door_rr_big': door_panel, 'test': sparse_sheet, 'door_lf': seam_sheet}
TEXTURES = {'door_test': ((240, 240), 'pattern.jpg'), 'square': ((240, 240), 'pattern.jpg'),
            'door_rr_big': ((480, 480), 'pattern_big.jpg'), 'test': ((240, 240), 'pattern.jpg'),
            'door_lf': ((240, 240), 'pattern.jpg')}


def write_synthetic_parts(root, names=('door_test', 'square')):
    """Write the parts under <root>/urdf/painting/ and return {name: urdf_path}."""
    directory = os.path.join(root, 'urdf', 'painting')
    out = {}
    for name in names:
        v, uv, f, fvt = PARTS[name]()
        v = np.round(v, 6)                  # what the OBJ text carries
        tex_size, tex_name = TEXTURES[name]
        out[name] = obj_io.write_part(directory, name, v, uv, f, fvt, tex_size=tex_size, texture_name=tex_name)
    return out


def synthetic_mesh(name):
    """MeshData of a synthetic part exactly as read back from its OBJ text."""
    v, uv, f, fvt = PARTS[name]()
    v = np.array([[float('%.6f' % c) for c in row] for row in v])
    uv6 = np.array([[float('%.6f' % c) for c in row] for row in uv])
    uv_read = np.stack([uv6[:, 0], 1 - uv6[:, 1]], axis=1)
    return obj_io.MeshData(v, uv_read, f, fvt)
