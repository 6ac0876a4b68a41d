"""Mesh -> static simulator tables for one paint part.

This is the setup-time half of the simulator: everything
``bullet_paint_wrapper.load_part`` (``PaintRLEnv/bullet_paint_wrapper.py:1327-1335``)
derives once per part, restated as flat float64/int arrays that the HIP step
kernel reads.  Each stage names the reference code whose arithmetic (including
its order-dependent quirks) it follows:

  stage                          reference (bullet_paint_wrapper.py)
  -----------------------------  ------------------------------------------
  triangle records               BarycentricInterpolator.__init__ 123-146,
                                 calculate_normal_from_abc 256-263, _get_side 1219-1229
  UV rasterisation -> samples    get_uv_pixels 192-212, Part.preprocess 622-648
  per-side vertex set            Part._build_kd_tree 599-620
  ranges / anchor start points   _get_corner_points_ranges 1256-1286, set_start_points 740-747
  grid rows (ray walks)          _set_grid_dict 922-963, _get_exact_boundary 906-920
  hull normal correction         ConvHull 61-104, _correct_bary_normals_with_conv_hull 650-660
  neighbour smoothing            _smooth_bary_normals_with_neighbors 662-698
  grid-observation cells         GridObservation._set_pixels_in_grid 1072-1101
  start points 'all'/'edge'      get_start_points 749-809
  density / cone beams           get_density 834-839, robot.py:14-35

Only the painted side (front) is built.  Sample order is canonical (ascending
``j*W + i``), not CPython set order: every consumer is order independent.
"""
import math

import numpy as np

from . import geometry as geo
from . import obj_io

PAINT_RADIUS = 0.051          # PaintToolProfile.PAINT_RADIUS (bpw:42)
STEP_SIZE = PAINT_RADIUS      # PaintToolProfile.STEP_SIZE (bpw:43)
HOOK_DISTANCE_TO_PART = 0.1   # Part.HOOK_DISTANCE_TO_PART (bpw:443)
IRRELEVANT = 10.0             # Part.IRRELEVANT_POSE component (bpw:445)
GRID_GRANULARITY = 100        # Part.GRID_GRANULARITY (bpw:447)
MIN_AREA = 1e-4               # BarycentricInterpolator.MIN_AREA (bpw:121)
SIDE_FRONT, SIDE_BACK, SIDE_OTHER = 1, 2, 3


class PartTables(object):
    """Plain attribute bag; see ``build_part_tables`` for the fields."""

    def summary(self):
        return {'P': int(self.sample_pos.shape[0]), 'T': int(self.tri_side.shape[0]),
                'T_front': int((self.tri_side == SIDE_FRONT).sum()),
                'V': int(self.vertices.shape[0]), 'V_front': int(self.vertex_is_side.sum()),
                'collision_triangles': int(self.col_v0.shape[0]),
                'start_anchor': len(self.anchor_points), 'start_all': len(self.all_points),
                'density': float(self.density), 'beams': int(self.beams.shape[0])}


# ----------------------------------------------------------------------------
# small restatements of reference helpers
# ----------------------------------------------------------------------------
def normalize_tuple(v, tolerance=0.00001):
    """bpw.normalize (32-37): renormalise only when | |v|^2 - 1 | > tolerance."""
    mag2 = sum(n * n for n in v)
    if abs(mag2 - 1.0) > tolerance:
        mag = np.sqrt(mag2)
        v = tuple(n / mag for n in v)
    return v


def included_angle(a, b):
    """bpw._get_included_angle (1207-1216) without the list-identity shortcut."""
    d = np.dot(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64))
    if d > 1:
        d = 1
    elif d < -1:
        d = -1
    return np.arccos(d)


def pose_orn_quaternion(orn):
    """robot.get_pose_orn (rob:93-100): shortest-arc quaternion z -> orn, xyzw."""
    xyz = [0.0 * orn[2] - 1.0 * orn[1], 1.0 * orn[0] - 0.0 * orn[2], 0.0 * orn[1] - 0.0 * orn[0]]
    w = float(1 + (0.0 * orn[0] + 0.0 * orn[1] + 1.0 * orn[2]))
    xyz.append(w)
    return tuple(float(c) for c in normalize_tuple(xyz))


def bary_coords(P, A, V0, V1, D00, D01, D11, INV):
    """BarycentricInterpolator._get_bary_coordinate (154-163), row-wise."""
    v2 = P - A
    d20 = geo.dot_fma(v2, V0)
    d21 = geo.dot_fma(v2, V1)
    v = (D11 * d20 - D01 * d21) * INV
    w = (D00 * d21 - D01 * d20) * INV
    u = 1.0 - v - w
    deg = (INV == 0)
    if np.ndim(deg) == 0:
        if deg:
            return -1.0, -1.0, -1.0
        return u, v, w
    u = np.where(deg, -1.0, u)
    v = np.where(deg, -1.0, v)
    w = np.where(deg, -1.0, w)
    return u, v, w


def _bary_setup(a, b, c):
    v0 = b - a
    v1 = c - a
    d00 = geo.dot_fma(v0, v0)
    d01 = geo.dot_fma(v0, v1)
    d11 = geo.dot_fma(v1, v1)
    denom = d00 * d11 - d01 * d01
    with np.errstate(divide='ignore', invalid='ignore'):
        inv = np.where(denom != 0, 1.0 / np.where(denom != 0, denom, 1.0), 0.0)
    return v0, v1, d00, d01, d11, inv


def _side_of_normals(vn, a0):
    """_get_side (1219-1229) applied as set_side does: included angle of vn to +/-front <= pi/3."""
    n0 = vn[:, a0]
    lim = np.pi / 3
    with np.errstate(invalid='ignore'):
        ang_f = np.arccos(np.clip(n0, -1.0, 1.0))
        ang_b = np.arccos(np.clip(-n0, -1.0, 1.0))
        front = ang_f <= lim
        back = (~front) & (ang_b <= lim)
    side = np.full(vn.shape[0], SIDE_OTHER, dtype=np.int8)
    side[back] = SIDE_BACK
    side[front] = SIDE_FRONT
    return side


# ----------------------------------------------------------------------------
# the builder
# ----------------------------------------------------------------------------
def build_part_tables(urdf_path=None, mesh=None, tex_size=None, obs_grad=4, collision_mode='hull',
                      base_position=obj_io.PART_BASE_POSITION, name=None, verbose=False, paint_radius=PAINT_RADIUS,
                      tex_init=None):
    """Build the static tables of one part for painting its FRONT side.

    Either ``urdf_path`` (resolved like the reference does) or ``mesh``
    (an ``obj_io.MeshData``) plus ``tex_size`` must be given.  ``tex_init``: the decoded bytes of the part's texture
    file, uint8 (H, W, 3) (bpw:1316-1319) -- read from the file with ``urdf_path``, the uniform grey the synthetic
    parts are written with (obj_io.write_part) otherwise; only ``texture_image`` looks at it.
    """
    if mesh is None:
        obj_path, tex_path = obj_io.resolve_part_files(urdf_path)
        mesh = obj_io.read_obj(obj_path)
        if tex_size is None:
            tex_size = obj_io.texture_size(tex_path)
        if tex_init is None:
            tex_init = obj_io.texture_bytes(tex_path)
        if name is None:
            name = urdf_path
    W, H = int(tex_size[0]), int(tex_size[1])
    if tex_init is None:
        tex_init = np.full((H, W, 3), obj_io.SYNTHETIC_TEXTURE_GREY, dtype=np.uint8)
    tex_init = np.ascontiguousarray(tex_init, dtype=np.uint8)
    if tex_init.size != H * W * 3:
        raise ValueError('tex_init has %d bytes, the texture is %d x %d x 3' % (tex_init.size, W, H))
    t = PartTables()
    t.name = name or 'part'
    t.tex_w, t.tex_h = W, H
    t.collision_mode = collision_mode
    t.obs_grad = int(obs_grad)
    t.paint_radius = float(paint_radius)      # PaintToolProfile.PAINT_RADIUS at load time (bpw:42)
    t.tex_init = tex_init.reshape(H, W, 3)

    # -- global frame, axes (bpw:1176-1182, 1294-1300, 498-500) ----------------
    V = np.asarray(base_position, dtype=np.float64)[None, :] + mesh.vertices
    (a1, a2), a0 = obj_io.principal_axes(V)
    t.vertices = V
    t.a0, t.a1, t.a2 = a0, a1, a2
    front_normal = np.array([1.0 if k == a0 else 0.0 for k in range(3)])
    F = mesh.faces_v
    T = F.shape[0]

    # -- per-triangle records --------------------------------------------------
    A, B, C = V[F[:, 0]], V[F[:, 1]], V[F[:, 2]]
    v0, v1, d00, d01, d11, inv = _bary_setup(A, B, C)
    crs = geo.cross3(v0, v1)
    with np.errstate(divide='ignore', invalid='ignore'):
        nrm = np.sqrt(geo.dot_fma(crs, crs))
        area = nrm / 2
        vn = crs / nrm[:, None]
    center = ((A + B) + C) / 3
    side = _side_of_normals(vn, a0)
    t.tri_vidx = F.astype(np.int32)
    t.tri_side = side
    t.tri_area = area
    t.tri_area_valid = area >= MIN_AREA
    t.tri_center = center
    t.tri_a, t.tri_v0, t.tri_v1 = A, v0, v1
    t.tri_d00, t.tri_d01, t.tri_d11, t.tri_inv = d00, d01, d11, inv
    normals = [tuple(row) for row in vn]        # mutable per-triangle normals, reference style
    front_ids = np.nonzero(side == SIDE_FRONT)[0]

    # -- UV rasterisation -> samples (front) ------------------------------------
    t.sample_pix, t.sample_pos, order = _rasterise_side(mesh, V, F, front_ids, W, H, insertion_order=True)
    t.sample_tie_rank = _sample_tie_rank(t.sample_pix, t.sample_pos, order, W)
    # the back side's texels (profile[Side.back], bpw:622-645): only their label in the texture image matters (bpw:588-591)
    t.back_pix = _rasterise_side(mesh, V, F, np.nonzero(side == SIDE_BACK)[0], W, H, positions=False)
    if verbose:
        print('samples', t.sample_pix.shape[0], 'back texels', t.back_pix.shape[0])

    # -- per-side vertex set (bpw:599-620) ---------------------------------------
    has_front = np.zeros(V.shape[0], dtype=bool)
    has_front[F[front_ids].ravel()] = True
    in_face = np.zeros(V.shape[0], dtype=bool)
    in_face[F.ravel()] = True
    side_data = V.copy()
    side_data[in_face & ~has_front] = IRRELEVANT
    relevant = side_data[:, 0] != IRRELEVANT
    relevant &= in_face      # a vertex in no face has no triangle to hook onto
    t.vertex_is_side = relevant
    t.vertex_tie_rank = _vertex_tie_rank(side_data)
    # vertex -> incident front triangles in file order (uv_map restricted to the side)
    adj = [[] for _ in range(V.shape[0])]
    for ti in front_ids:
        for vi in F[ti]:
            adj[vi].append(int(ti))
    t.vertex_adj = adj

    # -- ranges, anchor start points (uncorrected normals!) ----------------------
    t.ranges = _ranges(V, a1, a2)
    corner_points = _corner_points(V, a1, a2, t.paint_radius)
    t._side_data = side_data
    t._normals = normals
    t.anchor_points = []
    for cp in corner_points:
        hp = hook_point(t, cp)
        if hp is not None:
            t.anchor_points.append(hp)

    # -- collision set + grid rows -----------------------------------------------
    tri = geo.collision_triangles(V, F, collision_mode, (a1, a2))
    t.col_v0, t.col_e1, t.col_e2 = geo.pack_collision_triangles(tri)
    t.lwr = (t.ranges[0][1] - t.ranges[0][0]) / (t.ranges[1][1] - t.ranges[1][0])
    side_before_rows = side_data.copy()
    t.grid_lo, t.grid_hi, t.vertices_mutated = _grid_rows(t)
    # The reference's _set_grid_dict moves rows of vertices_kd_tree[side].data IN PLACE on sparse grid rows
    # (bpw:943-946), after the cKDTree was built on the old positions: from then on query(k=1) (bpw:526) walks a
    # tree whose split planes describe the old rows while leaf distances use the moved ones -- it no longer returns
    # the exact nearest vertex (3.5 % of queries near the reference's door_rr differ).  Parts on which that
    # happens carry the tree, so that nearest_side_vertex / the oracle / the device can walk it the same way.
    _fill_stale_kd_tree(t, side_before_rows if t.vertices_mutated else None)
    t.grid_range = t.grid_hi - t.grid_lo
    t.max_grid_size = float(t.grid_range.max())

    # -- normal correction + smoothing -------------------------------------------
    t.n_hull_corrected = _correct_with_hull(t, front_ids, front_normal)
    t.n_smoothed = _smooth_with_neighbours(t, front_ids)
    t.tri_normal = np.array([[float(c) for c in n] for n in normals], dtype=np.float64)

    # -- observation cells, start points, density, beams --------------------------
    t.sample_cell = grid_observation_cells(t, obs_grad)
    t.all_points = _all_start_points(t, front_ids)
    t.edge_points = _edge_start_points(t, t.all_points)
    step = (t.ranges[1][1] - t.ranges[1][0]) / GRID_GRANULARITY
    area_size = 0
    for gx in t.grid_range:
        area_size += step * gx
    t.density = t.sample_pos.shape[0] / area_size
    t.beams = cone_beams(t.density)
    t.front_ids = front_ids
    return t


def _vertex_tie_rank(side_rows):
    """Who wins `vertices_kd_tree[side].query(point, k=1)` (bpw:526) between vertices at EQUAL distance -- vertices written
    twice in the OBJ (exporters double them along UV seams: 179 in the reference's test.obj), each with its own incident
    triangles, so the choice decides which triangles the hook point may take.  query.cxx keeps the first point of a leaf that
    reaches the minimum, in the order of the tree's own index array (equal points share a leaf): the rank of vertex v is its
    place in that array, of scipy's own cKDTree of the rows the reference builds it from (bpw:604-619: every vertex, those
    without a triangle of the side parked at (10, 10, 10)).  int32 [V]."""
    from scipy.spatial import cKDTree
    tree = cKDTree(side_rows)
    rank = np.empty(side_rows.shape[0], dtype=np.int32)
    rank[np.asarray(tree.indices)] = np.arange(side_rows.shape[0], dtype=np.int32)
    return rank


def tie_order_python():
    """'major.minor' of the interpreter whose set iteration order _sample_tie_rank restates by running it."""
    import sys
    return '%d.%d' % sys.version_info[:2]


def _sample_tie_rank(pix, pos, insertion, W):
    """Who wins `pixel_kd_tree.query(point, k=1)` (bpw:565) among samples at EQUAL distance -- in practice samples with
    the same 3-D position: texels of different triangles that map to one mesh vertex (109 pairs on the reference's
    door_test.obj).  query.cxx keeps the first point of a leaf that reaches the minimum (strict <), in the order of the
    tree's own index array; equal points always share a leaf.  That order comes from the tree's partitioning of the rows
    in the order the reference hands them over: `profile[side] = list(set(pixels))` (bpw:641), the iteration order of a
    CPython set of int tuples filled in rasterisation order.  Both are restated by running them: the same set here (the
    interpreter is the reference's platform), scipy's own cKDTree of the positions in that order (as the stale vertex tree
    is scipy's own object too).  Returns int32 [P]: rank of canonical sample s = its place in that index array."""
    from scipy.spatial import cKDTree
    profile = list(set(insertion))                                   # bpw:641
    lin = np.array([p[0] + p[1] * W for p in profile], dtype=np.int64)
    canon = pix[:, 0].astype(np.int64) + pix[:, 1].astype(np.int64) * W      # ascending (canonical sample order)
    to_canon = np.searchsorted(canon, lin)
    if len(profile) != len(canon) or not np.array_equal(canon[to_canon], lin):      # (not an assert: -O must not strip it)
        raise ValueError('_sample_tie_rank: the rasterised pixel set does not match the canonical sample table')
    tree = cKDTree(pos[to_canon])                                    # bpw:620
    rank = np.empty(len(canon), dtype=np.int32)
    rank[to_canon[np.asarray(tree.indices)]] = np.arange(len(canon), dtype=np.int32)
    return rank


def _rasterise_side(mesh, V, F, tri_ids, W, H, positions=True, insertion_order=False):
    """Walk the side's triangles in file order; the last triangle covering a
    pixel defines that pixel's 3-D position (bpw:622-645, 192-212).  positions=False: only the pixel set.
    insertion_order=True: also the list of (i, j) int tuples in the order the reference appends them to
    profile[side] (bpw:192-212, 633: per triangle its three corner texels, then the inside texels u-major)."""
    pos_map = np.zeros((H * W, 3), dtype=np.float64)
    have = np.zeros(H * W, dtype=bool)
    order = []
    uvs = mesh.uvs
    FT = mesh.faces_vt
    for ti in tri_ids:
        uv = uvs[FT[ti]]                                   # (3,2) float UVs, v already flipped
        pi = np.minimum(np.rint(W * uv[:, 0]).astype(np.int64), W - 1)   # round-half-even
        pj = np.minimum(np.rint(H * uv[:, 1]).astype(np.int64), H - 1)
        if pi.min() < 0 or pj.min() < 0:
            raise NotImplementedError('negative texel coordinate (UV outside [0,1]) is not supported')
        P3 = V[F[ti]]
        for k in range(3):                                 # corner pixels, c over b over a
            idx = pi[k] + pj[k] * W
            pos_map[idx] = P3[k]
            have[idx] = True
            if insertion_order:
                order.append((int(pi[k]), int(pj[k])))
        b0, b1, e00, e01, e11, einv = _bary_setup(uv[0], uv[1], uv[2])
        if einv == 0:
            continue
        us = np.arange(pi.min(), pi.max() + 1)
        vs = np.arange(pj.min(), pj.max() + 1)
        uu, vv = np.meshgrid(us, vs, indexing='ij')
        uu = uu.ravel()
        vv = vv.ravel()
        rel = np.stack([uu / W, vv / H], axis=1)
        bu, bv, bw = bary_coords(rel, uv[0][None, :], b0[None, :], b1[None, :], e00, e01, e11, einv)
        inside = (bu >= 0) & (bu <= 1) & (bv >= 0) & (bv <= 1) & (bw >= 0) & (bw <= 1)
        if not inside.any():
            continue
        idx = uu[inside] + vv[inside] * W
        have[idx] = True
        if insertion_order:
            order.extend(zip(uu[inside].tolist(), vv[inside].tolist()))
        if not positions:
            continue
        bu, bv, bw = bu[inside], bv[inside], bw[inside]
        p = (bu[:, None] * P3[0][None, :] + bv[:, None] * P3[1][None, :]) + bw[:, None] * P3[2][None, :]
        pos_map[idx] = p
    # bpw:505-506: get_texel clamps the byte index of texel (W-1, H-1) to len - 4, i.e. onto the BLUE byte of its
    # left neighbour (W-2, H-1): the corner texel keeps its painted flag in that byte.  As long as the neighbour is
    # not a sample of the same side nothing ever writes that byte except the corner's own label and paint, and the
    # corner behaves like every other sample (the reference's test.urdf).  With both in the profile, painting the
    # neighbour would clear the corner's flag in an order that depends on cKDTree internals: not restated.
    if positions and have[H * W - 1] and have[H * W - 2]:
        raise NotImplementedError('texels (W-1,H-1) and (W-2,H-1) are both samples: the reference get_texel clamp '
                                  '(bpw:505-506) aliases their state bytes')
    lin = np.nonzero(have)[0]
    pix = np.stack([lin % W, lin // W], axis=1).astype(np.int32)
    if not positions:
        return pix
    if insertion_order:
        return pix, pos_map[lin].copy(), order
    return pix, pos_map[lin].copy()


def _ranges(V, a1, a2):
    return [[float(V[:, a1].min()), float(V[:, a1].max())], [float(V[:, a2].min()), float(V[:, a2].max())]]


def _corner_points(V, a1, a2, radius=PAINT_RADIUS):
    """_get_corner_points_ranges (1256-1286): stable-sort extremes of a1+a2 and a1-a2."""
    shrink = radius / 2
    s = V[:, a1] + V[:, a2]
    d = V[:, a1] - V[:, a2]
    first_min = lambda k: int(np.argmin(k))                       # noqa: E731
    last_max = lambda k: int(len(k) - 1 - np.argmax(k[::-1]))     # noqa: E731
    out = []
    for idx, (s1, s2) in ((first_min(s), (+1, +1)), (last_max(s), (-1, -1)),
                          (first_min(d), (+1, -1)), (last_max(d), (-1, +1))):
        p = [float(c) for c in V[idx]]
        p[a1] += s1 * shrink
        p[a2] += s2 * shrink
        out.append(p)
    return out


KD_FIELDS = ('kd_split_dim', 'kd_split', 'kd_less', 'kd_greater', 'kd_start', 'kd_end', 'kd_indices', 'kd_box')


def _fill_stale_kd_tree(t, rows_at_build_time):
    """Flatten scipy's cKDTree of the side's vertex rows AS THEY WERE when the reference built it (bpw:599-620:
    every vertex, other-side rows at (10, 10, 10), default leafsize 16) into arrays; empty when no row moved."""
    if rows_at_build_time is None:
        t.kd_split_dim = np.zeros(0, dtype=np.int32)
        t.kd_split = np.zeros(0, dtype=np.float64)
        t.kd_less = t.kd_greater = t.kd_start = t.kd_end = np.zeros(0, dtype=np.int32)
        t.kd_indices = np.zeros(0, dtype=np.int32)
        t.kd_box = np.zeros((2, 3), dtype=np.float64)
        return
    from scipy.spatial import cKDTree
    tree = cKDTree(np.ascontiguousarray(rows_at_build_time))
    nodes = []

    def walk(n):
        i = len(nodes)
        nodes.append(None)
        if n.split_dim == -1:
            nodes[i] = (-1, 0.0, -1, -1, n.start_idx, n.end_idx)
        else:
            lo, hi = walk(n.lesser), walk(n.greater)
            nodes[i] = (n.split_dim, float(n.split), lo, hi, n.start_idx, n.end_idx)
        return i

    walk(tree.tree)
    t.kd_split_dim = np.array([n[0] for n in nodes], dtype=np.int32)
    t.kd_split = np.array([n[1] for n in nodes], dtype=np.float64)
    t.kd_less = np.array([n[2] for n in nodes], dtype=np.int32)
    t.kd_greater = np.array([n[3] for n in nodes], dtype=np.int32)
    t.kd_start = np.array([n[4] for n in nodes], dtype=np.int32)
    t.kd_end = np.array([n[5] for n in nodes], dtype=np.int32)
    t.kd_indices = np.asarray(tree.indices, dtype=np.int32)
    t.kd_box = np.stack([tree.mins, tree.maxes]).astype(np.float64)


def stale_kd_query(t, point):
    """scipy's cKDTree.query(point, k=1) (query.cxx, p = 2, eps = 0) on the tree of _fill_stale_kd_tree with leaf
    distances taken from the CURRENT rows (t._side_data): greedy descent to the near child, far children with a
    lower bound <= the best distance go to a priority queue, a leaf is scanned in tree order keeping strictly
    smaller distances, the search ends when the queue is empty or its nearest cell is farther than the best.
    Checked against scipy itself on 20 000 random queries per part (tests/test_reference_parts.py)."""
    import heapq
    x = [float(c) for c in point]
    side = [max(0.0, x[k] - t.kd_box[1][k], t.kd_box[0][k] - x[k]) ** 2 for k in range(3)]
    mind = (side[0] + side[1]) + side[2]
    best, dub, queue, pushed = -1, float('inf'), [], 0
    node = 0
    data = t._side_data
    while True:
        sd = int(t.kd_split_dim[node])
        if sd < 0:
            for i in range(int(t.kd_start[node]), int(t.kd_end[node])):
                v = int(t.kd_indices[i])
                d0, d1, d2 = data[v, 0] - x[0], data[v, 1] - x[1], data[v, 2] - x[2]
                d = (d0 * d0 + d1 * d1) + d2 * d2
                if d < dub:
                    dub, best = d, v
            if not queue:
                break
            mind, _, node, side = heapq.heappop(queue)
        else:
            if mind > dub:
                break
            sp = float(t.kd_split[node])
            near, far = (int(t.kd_less[node]), int(t.kd_greater[node])) if x[sd] < sp else \
                (int(t.kd_greater[node]), int(t.kd_less[node]))
            tmp = sp - x[sd]
            new = tmp * tmp
            side2 = list(side)
            mind2 = mind + (new - side2[sd])
            side2[sd] = new
            a, b = (near, mind, side), (far, mind2, side2)
            if a[1] > b[1]:
                a, b = b, a
            if b[1] <= dub:
                pushed += 1
                heapq.heappush(queue, (b[1], pushed, b[0], b[2]))
            node, mind, side = a
    return best


def nearest_side_vertex(t, point):
    """cKDTree.query(k=1) over the side's vertex set (bpw:526): the exact Euclidean nearest neighbour -- unless the
    reference moved rows under its tree (see build_part_tables), then the walk of that stale tree."""
    if len(getattr(t, 'kd_split_dim', ())):
        return stale_kd_query(t, point)
    ids = np.nonzero(t.vertex_is_side)[0]
    d = t._side_data[ids] - np.asarray(point, dtype=np.float64)[None, :]
    d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    nearest = ids[d2 == d2.min()]                      # doubled vertices: the tree's own order decides (_vertex_tie_rank)
    return int(nearest[np.argmin(t.vertex_tie_rank[nearest])])


def closest_triangle(t, point, vertex):
    """Part._get_closest_bary (508-523) over the vertex's same-side triangles."""
    best = None
    best_uvw = -1
    point = np.asarray(point, dtype=np.float64)
    for ti in t.vertex_adj[vertex]:
        u, v, w = bary_coords(point, t.tri_a[ti], t.tri_v0[ti], t.tri_v1[ti],
                              t.tri_d00[ti], t.tri_d01[ti], t.tri_d11[ti], t.tri_inv[ti])
        if 0 <= u <= 1 and 0 <= v <= 1 and 0 <= w <= 1:
            return ti
        if best is None:
            best = ti
        m = min(u, v, w)
        if m >= best_uvw:
            best_uvw = m
            best = ti
    return best


def hook_point(t, point):
    """Part._get_hook_point (525-534): [pose, orn] or None."""
    vtx = nearest_side_vertex(t, point)
    ti = closest_triangle(t, point, vtx)
    if ti is None:
        return None
    n = t._normals[ti]
    pose = [a + b for a, b in zip(point, [i * HOOK_DISTANCE_TO_PART for i in n])]
    orn = [-i for i in n]
    return [[float(c) for c in pose], [float(c) for c in orn]]


def grid_index_2(t, val_axis_2):
    """Part._get_grid_index_2 (844-851)."""
    rel = (val_axis_2 - t.ranges[1][0]) / (t.ranges[1][1] - t.ranges[1][0])
    gi = int(rel * GRID_GRANULARITY)
    if gi < 0:
        return 0
    if gi > GRID_GRANULARITY - 1:
        return GRID_GRANULARITY - 1
    return gi


def normalized_pose(t, pose):
    """Part.get_normalized_pose (965-978)."""
    r = t.paint_radius
    x1 = pose[t.a1]
    x2 = pose[t.a2]
    in2 = (x2 - t.ranges[1][0] + r) / (t.ranges[1][1] - t.ranges[1][0] + 2 * r)
    gi = grid_index_2(t, x2)
    lo, hi = t.grid_lo[gi], t.grid_hi[gi]
    if hi - lo == 0:
        in1 = 0
    else:
        in1 = (x1 - lo + r) / (hi - lo + 2 * r)
    clip = lambda v: 0.0 if v < 0 else (1.0 if v > 1 else float(v))   # noqa: E731
    return clip(in1), clip(in2)


def _exact_boundary(t, point, is_min):
    """Part._get_exact_boundary (906-920): walk outward in 1e-3 steps until the
    first ray (along the non-principal axis, growing by 1 per step at both ends) misses."""
    step = -1e-3 if is_min else 1e-3
    n_steps = int((t.ranges[0][1] - t.ranges[0][0]) / abs(step))
    base = [float(c) for c in point]
    lo_c = base[t.a0]
    hi_c = base[t.a0]
    chunk = 16
    i = 0
    while i < n_steps:
        m = min(chunk, n_steps - i)
        org = np.tile(np.asarray(base), (m, 1))
        dst = org.copy()
        cur = np.empty(m)
        for k in range(m):
            cur[k] = base[t.a1] + (i + k) * step
            lo_c -= 1
            hi_c += 1
            org[k, t.a0] = lo_c
            dst[k, t.a0] = hi_c
        org[:, t.a1] = cur
        dst[:, t.a1] = cur
        idx, _, _ = geo.ray_closest_hit(t.col_v0, t.col_e1, t.col_e2, org, dst)
        miss = np.nonzero(idx < 0)[0]
        if miss.size:
            return float(cur[miss[0]])
        i += m
        chunk = min(chunk * 2, 256)
    return None


def _grid_rows(t):
    """Part._set_grid_dict (922-963), including the in-place mutation of the
    side's vertex array on sparse rows."""
    a1, a2 = t.a1, t.a2
    data = t._side_data                     # mutated in place, like cKDTree.data in the reference
    ids = np.nonzero(data[:, 0] != IRRELEVANT)[0]
    order = list(ids[np.argsort(data[ids, a2], kind='stable')])
    axis2_range = t.ranges[1][1] - t.ranges[1][0]
    step = axis2_range / GRID_GRANULARITY
    grid = {}
    traverse = 0
    left = right = order[0]
    mutated = set()
    for i in range(GRID_GRANULARITY):
        cur = traverse
        step_max = t.ranges[1][0] + (i + 1) * step
        for index in range(cur, len(order)):
            if data[order[index], a2] >= step_max:
                if index - cur <= 1:
                    new_a2 = step_max + 0.5 * step
                    if (i - 1) not in grid:
                        new_a1 = data[order[index], a1]
                    else:
                        new_a1 = (grid[i - 1][0] + grid[i - 1][1]) / 2
                    data[left, a2] = new_a2
                    data[right, a2] = new_a2
                    data[left, a1] = new_a1
                    data[right, a1] = new_a1
                    mutated.update((int(left), int(right)))
                else:
                    target = order[cur:index]
                    keys = np.array([data[v, a1] for v in target])
                    srt = np.argsort(keys, kind='stable')
                    left = target[srt[0]]
                    right = target[srt[-1]]
                rmin = _exact_boundary(t, data[left], True)
                rmax = _exact_boundary(t, data[right], False)
                if rmin is None or rmax is None:
                    raise RuntimeError('grid row %d: boundary walk never left the part' % i)
                grid[i] = (rmin, rmax)
                traverse = index + 1
                break
        else:
            grid[i] = (0.0, 0.0)
    lo = np.array([grid[i][0] for i in range(GRID_GRANULARITY)], dtype=np.float64)
    hi = np.array([grid[i][1] for i in range(GRID_GRANULARITY)], dtype=np.float64)
    return lo, hi, sorted(mutated)


def _correct_with_hull(t, front_ids, front_normal):
    """ConvHull.separate_by_side + correct_bary_normal for the front side (bpw:61-104, 650-660)."""
    from scipy.spatial import ConvexHull
    V = t.vertices
    a0, a1, a2 = t.a0, t.a1, t.a2
    simplices = ConvexHull(V).simplices
    keep = (t._side_data[simplices][:, :, 0] != IRRELEVANT).sum(axis=1) >= 2
    S = simplices[keep]
    if S.shape[0] == 0:
        return 0
    HA, HB, HC = V[S[:, 0]], V[S[:, 1]], V[S[:, 2]]
    hcr = geo.cross3(HB - HA, HC - HA)
    with np.errstate(divide='ignore', invalid='ignore'):
        hn = hcr / np.sqrt(geo.dot_fma(hcr, hcr))[:, None]
    hside = _side_of_normals(hn, a0)
    hn = np.where((hside != SIDE_FRONT)[:, None], -hn, hn)
    ax = [a1, a2]
    h2a = HA[:, ax]
    h2v0, h2v1, h00, h01, h11, hinv = _bary_setup(HA[:, ax], HB[:, ax], HC[:, ax])
    count = 0
    normals = t._normals
    for ti in front_ids:
        c = t.tri_center[ti]
        rel = normalized_pose(t, c)
        if rel[0] <= 0.01 or rel[0] >= 0.99 or rel[1] <= 0.01 or rel[1] >= 0.99:
            continue
        p2 = np.array([c[a1], c[a2]])[None, :]
        u, v, w = bary_coords(p2, h2a, h2v0, h2v1, h00, h01, h11, hinv)
        inside = (u >= 0) & (u <= 1) & (v >= 0) & (v <= 1) & (w >= 0) & (w <= 1)
        hit = np.nonzero(inside)[0]
        if hit.size == 0:
            continue
        k = int(hit[0])
        fn = [float(x) for x in hn[k]]
        if included_angle(normals[ti], fn) > np.pi / 6:
            normals[ti] = tuple(fn)
            count += 1
    return count


def _smooth_with_neighbours(t, front_ids):
    """_smooth_bary_normals_with_neighbors / _smooth_normal (bpw:662-698):
    sequential and in place, later triangles see earlier replacements."""
    from scipy.spatial import cKDTree
    T = t.tri_side.shape[0]
    centers = np.full((T, 3), IRRELEVANT, dtype=np.float64)
    centers[front_ids] = t.tri_center[front_ids]
    tree = cKDTree(centers)
    normals = t._normals
    count = 0
    k = min(5, T)
    for ti in front_ids:
        nb = np.atleast_1d(tree.query(t.tri_center[ti], k=k)[1])
        for b in nb:
            if b == ti or b >= T:
                continue
            ang = included_angle(normals[b], normals[ti])
            if abs(ang) > np.pi / 18:
                near = tree.query_ball_point(t.tri_center[ti], t.paint_radius)
                weighted = []
                for bi in near:
                    if bi != ti:
                        weighted.append([t.tri_area[bi] * c for c in normals[bi]])
                if weighted:
                    avg = np.average(weighted, 0)
                    new = normalize_tuple(avg)
                    normals[ti] = tuple(float(c) for c in new)
                    count += 1
                break
    return count


def grid_observation_cells(t, h):
    """GridObservation cell of every sample as v_target*h + h_grid (bpw:1072-1101)."""
    a1, a2 = t.a1, t.a2
    step2 = (t.ranges[1][1] - t.ranges[1][0]) / GRID_GRANULARITY
    v_interval = int(GRID_GRANULARITY / h)
    cells = np.zeros(t.sample_pos.shape[0], dtype=np.int32)
    for s, p in enumerate(t.sample_pos):
        y_grid = min(GRID_GRANULARITY - 1, int((p[a2] - t.ranges[1][0]) / step2))
        gr = t.grid_range[y_grid]
        if gr == 0:
            x_grid = 0
        else:
            x_step = gr / h
            x_grid = min(h - 1, int((p[a1] - t.grid_lo[y_grid]) / x_step))
        v_target = y_grid // v_interval
        if not (0 <= x_grid < h and 0 <= v_target < h):
            raise KeyError('sample %d falls outside the %dx%d observation grid (reference raises KeyError)'
                           % (s, h, h))
        cells[s] = v_target * h + x_grid
    return cells


def _all_start_points(t, front_ids):
    """The 'all' candidates of Part.get_start_points (749-773), file order, corrected normals."""
    shrink = t.paint_radius / 2
    a1, a2 = t.a1, t.a2
    ax2 = [p[0][a2] for p in t.anchor_points]
    ax2_max, ax2_min = max(ax2), min(ax2)
    out = []
    for ti in front_ids:
        if not t.tri_area_valid[ti]:
            continue
        c = t.tri_center[ti]
        n = t._normals[ti]
        gi = grid_index_2(t, c[a2])
        lo, hi = t.grid_lo[gi], t.grid_hi[gi]
        if c[a1] - lo >= shrink and hi - c[a1] >= shrink and ax2_min <= c[a2] <= ax2_max:
            hook = [a + b for a, b in zip(c, [i * HOOK_DISTANCE_TO_PART for i in n])]
            out.append([[float(x) for x in hook], [float(-i) for i in n]])
    return out


def _edge_start_points(t, points):
    """Part._get_edge_start_points (785-809)."""
    if not points:
        return []
    a1, a2 = t.a1, t.a2
    rows = {}
    for p, o in points:
        rows.setdefault(grid_index_2(t, p[a2]), []).append([p, o])
    mx, mn = max(rows), min(rows)
    out = []
    for gi, lst in rows.items():
        if gi in (mx, mn):
            out.extend(lst)
        else:
            srt = sorted(lst, key=lambda v: v[0][a1])
            if (srt[0][0][a1] - t.grid_lo[gi]) / t.grid_range[gi] < 0.15:
                out.append(srt[0])
            if (t.grid_hi[gi] - srt[-1][0][a1]) / t.grid_range[gi] < 0.15:
                out.append(srt[-1])
    return out


def start_points(t, mode='anchor'):
    """Part.get_start_points (749-783): list of [pose, orn]."""
    if mode == 'fixed':
        return [t.anchor_points[0]]
    if mode == 'anchor':
        return list(t.anchor_points)
    if mode == 'edge':
        return list(t.anchor_points) + list(t.edge_points)
    if mode == 'all':
        return list(t.anchor_points) + list(t.all_points)
    raise ValueError('unknown START_POINT_MODE %r' % (mode,))


def cone_beams(density):
    """robot._get_uniformed_plain (rob:23-35): beam end points in the tool frame."""
    ratio = 0.2 / 0.5
    radius = 0.25 * ratio
    resolution = 1.8 / math.sqrt(density)
    plane = 0.2
    out = []
    i = j = -radius
    while i <= radius:
        while j <= radius:
            if math.sqrt(math.pow(i, 2) + math.pow(j, 2)) <= radius:
                out.append((i, j, plane))
            j += resolution
        i += resolution
        j = -radius
    return np.asarray(out, dtype=np.float64).reshape(-1, 3)


def beta_plain(density, beta=2, expected_points=450, uniform=None):
    """robot._get_beta_plain (rob:38-69): the beam table of COLOR_MODE 'HSI' -- rings of one beam pitch, the i-th of
    ``circles`` holding round(450 w_i / sum w) beams, w_i = (1 - (i / circles)^2)^(beta - 1), at equal angles and at a
    radius drawn with ``uniform(lower, upper)`` per beam.  The reference draws from Python's ``random`` module
    (``from random import uniform``, rob:4): with the default ``uniform`` and the same ``random.seed`` this returns the
    reference's own table; any other source gives a table of the same law (the table is data: PartTables.beams)."""
    import random
    uniform = uniform or random.uniform
    ratio = 0.2 / 0.5
    radius = 0.25 * ratio
    resolution = 1.8 / math.sqrt(density)
    plane = 0.2
    circles = math.ceil(radius / resolution)
    distribution, total = {}, 0
    for i in range(1, circles + 1):
        distribution[i] = (1 - (i / circles) ** 2) ** (beta - 1)
        total += distribution[i]
    for i in distribution:
        distribution[i] = round(expected_points * distribution[i] / total)
    out = []
    for i in range(1, circles + 1):
        lower, upper = (i - 1) * resolution, i * resolution
        angle_resolution = 2 * math.pi / distribution[i] if distribution[i] else 0
        for j in range(distribution[i]):
            r = uniform(lower, upper)
            theta = j * angle_resolution
            out.append((r * np.cos(theta), r * np.sin(theta), plane))          # rob:143-146 pol2cart
    return np.asarray(out, dtype=np.float64).reshape(-1, 3)


# ----------------------------------------------------------------------------
# the texture image (bpw:579-592 _label_part, 358-365 change_pixel, 404-406 HSI deposits, 737-738 get_texture_image)
# ----------------------------------------------------------------------------
def _set_texels(tex, W, color, pix, skip_if_first_equals=True):
    """ColorHandler.init_part / change_pixel (bpw:358-365, 377-381) for a list of pixels: the three bytes at
    get_texel(i, j) = min(3 (i + j W), len - 4) (bpw:505-506) become `color` unless the first already equals
    color[0].  Pixel (W-1, H-1) is the one the clamp moves: its bytes start on the blue byte of (W-2, H-1); those two
    are done one after the other in list order, everything else at once."""
    if len(pix) == 0:
        return
    pix = np.asarray(pix, dtype=np.int64).reshape(-1, 2)
    limit = tex.size - 4
    base = 3 * (pix[:, 0] + pix[:, 1] * W)
    special = base >= limit - 2
    b = base[~special]
    if skip_if_first_equals:
        b = b[tex[b] != color[0]]
    for k in range(3):
        tex[b + k] = color[k]
    for b in base[special]:
        b = min(int(b), limit)
        if skip_if_first_equals and tex[b] == color[0]:
            continue
        for k in range(3):
            tex[b + k] = color[k]


def label_texture(t, color_mode='RGB'):
    """Part._label_part, render branch (bpw:579-592): the texture's decoded bytes with every texel outside the
    profiles set to (0, 0, 0), the back side's to (0, 255, 0), the front side's to (191, 191, 191) -- (255, 255, 255) in
    COLOR_MODE 'HSI' -- in that order, each texel skipped when its red byte already has the target's value (`is_changed`,
    bpw:352-354: such a texel keeps the green and blue bytes of the texture file).  Returns the flat uint8 texel list
    (index (i + j W) 3, bpw:505-506) = Part.init_texture."""
    W, H = int(t.tex_w), int(t.tex_h)
    tex = np.array(t.tex_init, dtype=np.uint8).reshape(-1).copy()
    in_profile = np.zeros(H * W, dtype=bool)
    for pix in (t.sample_pix, t.back_pix):
        in_profile[pix[:, 0].astype(np.int64) + pix[:, 1].astype(np.int64) * W] = True
    ii, jj = np.meshgrid(np.arange(W), np.arange(H), indexing='ij')       # bpw:581: for i in width for j in height
    ii, jj = ii.ravel(), jj.ravel()
    irrelevant = np.stack([ii, jj], axis=1)[~in_profile[ii + jj * W]]
    front = (255, 255, 255) if color_mode == 'HSI' else (191, 191, 191)   # int(0.75 * 255), bpw:495-496
    _set_texels(tex, W, (0, 0, 0), irrelevant)
    _set_texels(tex, W, (0, 255, 0), t.back_pix)
    _set_texels(tex, W, front, t.sample_pix)
    return tex


def texture_image(t, painted=None, thickness=None, color_mode='RGB'):
    """Part.get_texture_image() (bpw:737-738, 18-21) of an env whose front samples are `painted` (bool [P], canonical
    sample order; COLOR_MODE 'RGB': painted texels are (255, 0, 0), bpw:358-365) or carry the `thickness` bytes
    (uint8 [P]; COLOR_MODE 'HSI': every deposit lowers the three bytes of a texel by the same amount in uint8
    arithmetic, bpw:404-406, so green and blue follow the red byte).  uint8 array (W, H, 3) as the reference shapes it
    (row j, column i for the square textures of every part)."""
    W, H = int(t.tex_w), int(t.tex_h)
    tex = label_texture(t, color_mode)
    pix = np.asarray(t.sample_pix, dtype=np.int64)
    if color_mode == 'HSI':
        if thickness is None:
            raise ValueError("COLOR_MODE 'HSI' needs the thickness bytes")
        base = np.minimum(3 * (pix[:, 0] + pix[:, 1] * W), tex.size - 4)
        delta = tex[base] - np.asarray(thickness, dtype=np.uint8)           # what the deposits took off so far (mod 256)
        for k in range(3):
            tex[base + k] = tex[base + k] - delta
    elif painted is not None:
        _set_texels(tex, W, (255, 0, 0), pix[np.asarray(painted, dtype=bool)])
    return tex.reshape(W, H, 3)


# ----------------------------------------------------------------------------
# on-disk table format: one .npz per part
# ----------------------------------------------------------------------------
_ARRAY_FIELDS = ['vertices', 'tri_vidx', 'tri_side', 'tri_area', 'tri_area_valid', 'tri_center', 'tri_a', 'tri_v0',
                 'tri_v1', 'tri_d00', 'tri_d01', 'tri_d11', 'tri_inv', 'tri_normal', 'sample_pix', 'sample_pos',
                 'sample_cell', 'vertex_is_side', '_side_data', 'col_v0', 'col_e1', 'col_e2', 'grid_lo', 'grid_hi',
                 'grid_range', 'beams', 'front_ids', 'back_pix', 'tex_init', 'sample_tie_rank', 'vertex_tie_rank'] + list(KD_FIELDS)
_SCALAR_FIELDS = ['name', 'tex_w', 'tex_h', 'collision_mode', 'obs_grad', 'paint_radius', 'a0', 'a1', 'a2', 'lwr', 'max_grid_size',
                  'density', 'n_hull_corrected', 'n_smoothed']
TABLE_FORMAT_VERSION = 3          # 3: back_pix, tex_init (texture_image), sample_tie_rank, vertex_tie_rank


def save_tables(t, path):
    """Write a PartTables to ``path`` (.npz).  Everything build_part_tables produced is kept, so a
    loaded table is interchangeable with a freshly built one."""
    import json
    data = {k: np.asarray(getattr(t, k)) for k in _ARRAY_FIELDS}
    meta = {k: (getattr(t, k) if not isinstance(getattr(t, k), np.generic) else getattr(t, k).item())
            for k in _SCALAR_FIELDS}
    meta.update(format_version=TABLE_FORMAT_VERSION, ranges=t.ranges, vertices_mutated=[int(v) for v in t.vertices_mutated])
    # the tie order of equal samples (_sample_tie_rank) is the iteration order of a CPython set of int tuples, which belongs
    # to the interpreter that built the tables (tuple hashing changed in CPython 3.8): recorded, checked by load_tables
    meta.update(built_with_python=tie_order_python())
    data['meta'] = np.array(json.dumps(meta))
    for key in ('anchor_points', 'all_points', 'edge_points'):
        data[key] = np.asarray(getattr(t, key), dtype=np.float64).reshape(-1, 2, 3)
    off, flat = [0], []
    for lst in t.vertex_adj:
        flat.extend(lst)
        off.append(len(flat))
    data['vertex_adj_off'] = np.asarray(off, dtype=np.int64)
    data['vertex_adj_tri'] = np.asarray(flat, dtype=np.int64)
    data['normals'] = np.asarray([[float(c) for c in n] for n in t._normals], dtype=np.float64)
    np.savez_compressed(path, **data)


def load_tables(path):
    """Inverse of save_tables."""
    import json
    z = np.load(path, allow_pickle=False)
    meta = json.loads(str(z['meta']))
    if meta.get('format_version') != TABLE_FORMAT_VERSION:
        raise ValueError('%s: table format %r, this build reads %d' % (path, meta.get('format_version'),
                                                                       TABLE_FORMAT_VERSION))
    built = meta.get('built_with_python')
    if built is not None and built != tie_order_python():
        import warnings
        warnings.warn('%s was built under Python %s, this is %s: the tie order of equally distant samples (cone-beam paint, '
                      'bpw:565 / 641) is the set iteration order of the interpreter that ran the reference; rebuild the tables '
                      'under the interpreter whose behaviour you want to match' % (path, built, tie_order_python()))
    t = PartTables()
    for k in _ARRAY_FIELDS:
        setattr(t, k, z[k])
    for k in _SCALAR_FIELDS:
        setattr(t, k, meta[k])
    t.ranges = [list(r) for r in meta['ranges']]
    t.vertices_mutated = list(meta['vertices_mutated'])
    for key in ('anchor_points', 'all_points', 'edge_points'):
        t.__dict__[key] = [[list(map(float, p[0])), list(map(float, p[1]))] for p in z[key]]
    off, flat = z['vertex_adj_off'], z['vertex_adj_tri']
    t.vertex_adj = [[int(v) for v in flat[off[i]:off[i + 1]]] for i in range(len(off) - 1)]
    t._normals = [tuple(row) for row in z['normals']]
    return t
