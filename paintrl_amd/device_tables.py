"""PartTables (host, reference order) -> the device layout of include/paintrl.h PrlPartTables.

Layout decisions (DESIGN.md "Data layout in HBM"):
  * samples and same-side vertices are sorted by the cell of a uniform grid over
    the principal plane (cell edge 0.052 > paint radius 0.051), x-fastest, so the
    3x3 neighbourhood of a shot is three contiguous index ranges;
  * a sample's coverage bit is its position in that order; per 64-sample word the
    table keeps a principal-plane bounding box and a valid-bit mask;
  * collision triangles are SoA (v0, e1, e2 components) with a float32 box each.
"""
import numpy as np

from . import _lib
from . import part_tables as pt

CELL = 0.052             # > PAINT_RADIUS, so a ball overlaps at most 3x3 cells
NBR_WIDTH = 32           # facets sharing a vertex with a hull facet, itself included (door: <= 26)
FAR = 1.0e15             # coordinate of padding samples


class DeviceTables(object):
    """Numpy arrays in device layout + the ctypes struct that points at them."""

    def __init__(self, tables, obs_grad=4, start_points=None, beams=None):
        t = tables
        a1, a2 = t.a1, t.a2
        self.tables = t
        P = t.sample_pos.shape[0]
        # ---- samples sorted by grid cell; every cell ROW starts on a 64-sample word boundary, so no
        # word mixes the end of one row with the start of the next (such words would straddle the
        # tool in every observation)
        o1, o2 = float(t.sample_pos[:, a1].min()), float(t.sample_pos[:, a2].min())
        self.paint_radius = float(getattr(t, 'paint_radius', pt.PAINT_RADIUS))
        self.cell = max(CELL, 1.02 * self.paint_radius)      # > radius, so a ball overlaps at most 3x3 cells
        inv = 1.0 / self.cell
        cx = np.floor((t.sample_pos[:, a1] - o1) * inv).astype(np.int64)
        cy = np.floor((t.sample_pos[:, a2] - o2) * inv).astype(np.int64)
        nx, ny = int(cx.max()) + 1, int(cy.max()) + 1
        cell = cy * nx + cx
        # within a cell row the samples ascend on axis a1 (which also orders them by cell): the samples
        # left / right of a vertical section line are then a prefix / suffix of every word of the row,
        # found by binary search instead of a per-sample pass (observation of the 4-sector rule)
        order = np.lexsort((np.arange(P), t.sample_pos[:, a1], cy))
        per_cell = np.bincount(cell, minlength=nx * ny)
        starts = np.zeros(nx * ny + 1, dtype=np.int64)
        pos = 0
        for row in range(ny):
            pos = ((pos + 63) // 64) * 64
            for col in range(nx):
                starts[row * nx + col] = pos
                pos += per_cell[row * nx + col]
        starts[nx * ny] = pos
        n_pad = ((pos + 63) // 64) * 64
        n_words = n_pad // 64
        # device position of each sorted sample; cells of one row are contiguous, rows are word aligned
        rank_in_cell = np.arange(P) - np.searchsorted(cell[order], cell[order], side='left')
        dev_pos = starts[cell[order]] + rank_in_cell
        self.perm = -np.ones(n_pad, dtype=np.int64)       # device position -> canonical sample index (-1 = pad)
        self.perm[dev_pos] = order
        self.inv_perm = np.empty(P, dtype=np.int64)        # canonical sample index -> device position
        self.inv_perm[order] = dev_pos
        valid = self.perm >= 0
        xyz = np.full((3, n_pad), FAR, dtype=np.float64)
        xyz[:, valid] = t.sample_pos[self.perm[valid]].T
        self.sample_xyz = [np.ascontiguousarray(xyz[k]) for k in range(3)]
        # a row's range for cells [cx0, cx1] is [start[row, cx0], start[row, cx1 + 1]); at the end of a row
        # that takes in the alignment pads, which are far away and never valid
        self.sgrid_start = starts.astype(np.int32)
        self.word_valid = np.packbits(valid, bitorder='little').view(np.uint64).copy()
        # who wins the nearest-sample query among equally distant samples (samples sharing one position): the reference
        # tree's own order, part_tables._sample_tie_rank (bpw:565)
        tie = np.asarray(t.sample_tie_rank, dtype=np.int64)
        self.sample_rank = np.where(valid, tie[np.maximum(self.perm, 0)], 0x7fffffff).astype(np.int32)
        bbox = np.empty((n_words, 4), dtype=np.float64)
        c1 = np.where(valid, xyz[a1], np.nan).reshape(n_words, 64)
        c2 = np.where(valid, xyz[a2], np.nan).reshape(n_words, 64)
        with np.errstate(all='ignore'):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                bbox[:, 0], bbox[:, 1] = np.nanmin(c1, axis=1), np.nanmax(c1, axis=1)
                bbox[:, 2], bbox[:, 3] = np.nanmin(c2, axis=1), np.nanmax(c2, axis=1)
        empty = ~valid.reshape(n_words, 64).any(axis=1)
        bbox[empty] = (np.inf, -np.inf, np.inf, -np.inf)
        self.word_bbox = bbox
        self.n_samples, self.n_samples_pad, self.n_words = P, n_pad, n_words
        self.sgrid = (o1, o2, inv, nx, ny)
        # ---- grid-observation cell masks (bpw GridObservation)
        self.obs_grad = int(obs_grad)
        # (a granularity whose 100/h rows do not tile the part has no grid observation in the reference
        # either -- it raises KeyError; section / discrete / simple modes do not need the cells)
        try:
            cells = t.sample_cell if obs_grad == t.obs_grad else pt.grid_observation_cells(t, obs_grad)
            n_cells = self.obs_grad ** 2
        except KeyError:
            cells, n_cells = np.zeros(P, dtype=np.int32), 0
        self.n_obs_cells = n_cells
        onehot = np.zeros((max(n_cells, 1), n_pad), dtype=bool)
        if n_cells:
            onehot[cells, self.inv_perm] = True
        self.obs_cell_mask = np.packbits(onehot, axis=1, bitorder='little').view(np.uint64).reshape(
            max(n_cells, 1), n_words).copy()
        self.obs_cell_count = np.bincount(cells, minlength=max(n_cells, 1)).astype(np.int32) if n_cells else \
            np.zeros(1, dtype=np.int32)
        # ---- same-side vertices sorted by grid cell, CSR adjacency to compact triangle ids
        side_ids = np.nonzero(t.vertex_is_side)[0]
        vpos = t._side_data[side_ids]
        vo1, vo2 = float(vpos[:, a1].min()), float(vpos[:, a2].min())
        # The vertex grid has its own cell parameter but uses the sample cell: the query points are ray hits
        # on the collision set (in hull mode up to centimetres above the surface), and the ring search
        # accepts a block only when the 3-D distance of the best vertex is within ring * 0.99 * cell.  A
        # cell sized to the vertex density alone (0.03 on the door) sends most searches to rings 2-3
        # (measured: 58.6 -> 85.6 us per step).
        self.vcell = self.cell
        inv = 1.0 / self.vcell
        vcx = np.floor((vpos[:, a1] - vo1) * inv).astype(np.int64)
        vcy = np.floor((vpos[:, a2] - vo2) * inv).astype(np.int64)
        vnx, vny = int(vcx.max()) + 1, int(vcy.max()) + 1
        vcell = vcy * vnx + vcx
        vorder = np.argsort(vcell, kind='stable')
        self.vertex_xyz = [np.ascontiguousarray(vpos[vorder, k]) for k in range(3)]
        # equally distant vertices (doubled in the OBJ): the order of the reference's own tree, part_tables._vertex_tie_rank
        self.vertex_rank = np.asarray(t.vertex_tie_rank, dtype=np.int32)[side_ids][vorder]
        self.vgrid_start = np.searchsorted(vcell[vorder], np.arange(vnx * vny + 1)).astype(np.int32)
        self.vgrid = (vo1, vo2, inv, vnx, vny)
        self.vgrid_accept = 0.99 * self.vcell
        # the reference's stale vertex kd-tree (part_tables._fill_stale_kd_tree), leaf points as device positions
        n_kd = len(getattr(t, 'kd_split_dim', ()))
        self.kd_node = np.zeros((max(n_kd, 1), 4), dtype=np.int32)
        self.kd_split = np.zeros(max(n_kd, 1), dtype=np.float64)
        self.kd_points = np.zeros(1, dtype=np.int32)
        self.kd_box = np.zeros(6, dtype=np.float64)
        self.n_kd_nodes = n_kd
        if n_kd:
            dev_of_full = -np.ones(t.vertices.shape[0], dtype=np.int64)
            dev_of_full[side_ids[vorder]] = np.arange(vorder.size)
            leaf = np.asarray(t.kd_split_dim) < 0
            self.kd_node[:, 0] = t.kd_split_dim
            self.kd_node[:, 1] = np.where(leaf, t.kd_start, t.kd_less)
            self.kd_node[:, 2] = np.where(leaf, t.kd_end, t.kd_greater)
            self.kd_split = np.ascontiguousarray(t.kd_split, dtype=np.float64)
            self.kd_points = dev_of_full[np.asarray(t.kd_indices, dtype=np.int64)].astype(np.int32)
            self.kd_box = np.ascontiguousarray(np.asarray(t.kd_box, dtype=np.float64).reshape(6))
        front_ids = np.nonzero(t.tri_side == pt.SIDE_FRONT)[0]
        compact = -np.ones(t.tri_side.shape[0], dtype=np.int64)
        compact[front_ids] = np.arange(front_ids.size)
        lists = [[int(compact[ti]) for ti in t.vertex_adj[v]] for v in side_ids[vorder]]
        self.adj_width = max(1, max(len(l) for l in lists))
        if self.adj_width > 64:
            raise NotImplementedError('a vertex has %d incident same-side triangles (max 64)' % self.adj_width)
        adj = -np.ones((len(lists), self.adj_width), dtype=np.int32)
        for i, l in enumerate(lists):
            adj[i, :len(l)] = l
        self.vertex_adj = adj
        # ---- triangle records: a v0 v1 d00 d01 d11 inv normal
        rec = np.empty((front_ids.size, 16), dtype=np.float64)
        rec[:, 0:3], rec[:, 3:6], rec[:, 6:9] = t.tri_a[front_ids], t.tri_v0[front_ids], t.tri_v1[front_ids]
        rec[:, 9], rec[:, 10], rec[:, 11] = t.tri_d00[front_ids], t.tri_d01[front_ids], t.tri_d11[front_ids]
        rec[:, 12] = t.tri_inv[front_ids]
        rec[:, 13:16] = t.tri_normal[front_ids]
        self.tri_records = rec
        # ---- collision triangles: SoA + float32 3-D boxes (outward by 1e-6 and one float ulp).
        # ``col_rank`` keeps the reference order for the equal-t tie break.
        C0 = t.col_v0.shape[0]
        corners = np.stack([t.col_v0, t.col_v0 + t.col_e1, t.col_v0 + t.col_e2], axis=1)
        lo, hi = corners.min(axis=1), corners.max(axis=1)
        nrm = np.cross(t.col_e1, t.col_e2)
        with np.errstate(all='ignore'):
            facing = nrm[:, t.a0] / np.linalg.norm(nrm, axis=1)
        large = ((hi[:, a1] - lo[:, a1]) > 4 * self.cell) | ((hi[:, a2] - lo[:, a2]) > 4 * self.cell)
        facing = np.nan_to_num(facing)
        # chunks of 64: small facets bucketed by a coarse K x K grid over the principal plane (compact
        # chunk boxes), then large front-facing, large back-facing and large other facets, each group
        # padded to a multiple of 64
        small_ids = np.nonzero(~large)[0]
        K = max(1, int(round(np.sqrt(max(small_ids.size, 1) / 48.0))))
        cen = corners.mean(axis=1)
        span1 = max(float(hi[:, a1].max() - lo[:, a1].min()), 1e-12)
        span2 = max(float(hi[:, a2].max() - lo[:, a2].min()), 1e-12)
        k1 = np.clip(((cen[:, a1] - lo[:, a1].min()) / span1 * K).astype(np.int64), 0, K - 1)
        k2 = np.clip(((cen[:, a2] - lo[:, a2].min()) / span2 * K).astype(np.int64), 0, K - 1)
        buckets = [small_ids[(k2[small_ids] * K + k1[small_ids]) == c] for c in range(K * K)]
        big = [np.nonzero(large & (facing > 0.5))[0], np.nonzero(large & (facing < -0.5))[0],
               np.nonzero(large & (np.abs(facing) <= 0.5))[0]]
        # (their boxes span the part whichever way they face: every general search visits all three chunks, each a dependent
        # round trip -- as one chunk where they fit one: the door's hull has 6 + 9 + 30 of them)
        buckets += [np.concatenate(big)] if sum(b.size for b in big) <= 64 else big
        slots = []
        for ids in buckets:
            if ids.size:
                slots.extend(ids.tolist())
                slots.extend([-1] * ((-ids.size) % 64))
        slots = np.asarray(slots, dtype=np.int64)
        c_pad = slots.size
        real = slots >= 0
        corder = slots[real]
        col = np.zeros((9, c_pad), dtype=np.float64)
        col[0:3, real], col[3:6, real], col[6:9, real] = t.col_v0[corder].T, t.col_e1[corder].T, t.col_e2[corder].T
        self.col = [np.ascontiguousarray(col[k]) for k in range(9)]
        self.col_rank = np.full(c_pad, 0x7fffffff, dtype=np.int32)
        self.col_rank[real] = corder
        box = np.empty((c_pad, 8), dtype=np.float32)
        box[:, 0::2], box[:, 1::2] = np.inf, -np.inf
        for k, ax in enumerate((a1, a2, t.a0)):
            box[real, 2 * k] = np.nextafter((lo[corder, ax] - 1e-6).astype(np.float32), np.float32(-np.inf))
            box[real, 2 * k + 1] = np.nextafter((hi[corder, ax] + 1e-6).astype(np.float32), np.float32(np.inf))
        box[:, 6], box[:, 7] = 0.0, 0.0
        self.col_bbox = box
        n_chunks = c_pad // 64
        cb = box.reshape(n_chunks, 64, 8)
        chunk = np.empty((((n_chunks + 63) // 64) * 64, 8), dtype=np.float32)
        chunk[:, 0::2], chunk[:, 1::2] = np.inf, -np.inf
        for k in range(3):
            chunk[:n_chunks, 2 * k] = cb[:, :, 2 * k].min(1)
            chunk[:n_chunks, 2 * k + 1] = cb[:, :, 2 * k + 1].max(1)
        chunk[:, 6], chunk[:, 7] = 0.0, 0.0
        self.col_chunk_bbox, self.n_col_chunks = chunk, n_chunks
        self.n_collision, self.n_collision_pad = C0, c_pad
        # convex-hull fast path: per facet, the facets sharing a vertex with it (itself first) and the
        # sign that makes `orient * det > 0` mean "the segment enters the hull through this facet"
        self.nbr_width = NBR_WIDTH
        self.col_nbr = -np.ones((c_pad, NBR_WIDTH), dtype=np.int32)
        self.col_orient = np.zeros(c_pad, dtype=np.int32)
        self.col_convex = 0
        if t.collision_mode == 'hull':
            dev_of = -np.ones(C0, dtype=np.int64)
            dev_of[corder] = np.nonzero(real)[0]
            flat = corners.reshape(-1, 3)
            _, vid = np.unique(flat, axis=0, return_inverse=True)
            tri_v = vid.reshape(C0, 3)
            by_vertex = {}
            for ti, vs in enumerate(tri_v):
                for v in vs:
                    by_vertex.setdefault(int(v), []).append(ti)
            centroid = flat.mean(axis=0)
            ok = True
            for ti in range(C0):
                nb = sorted(set(by_vertex[int(tri_v[ti, 0])]) | set(by_vertex[int(tri_v[ti, 1])]) |
                            set(by_vertex[int(tri_v[ti, 2])]))
                nb.remove(ti)
                if len(nb) + 1 > NBR_WIDTH:
                    ok = False
                    continue                       # no list: the kernel falls back to the full search here
                self.col_nbr[dev_of[ti], 0] = dev_of[ti]
                self.col_nbr[dev_of[ti], 1:1 + len(nb)] = dev_of[nb]
            outward = np.einsum('ij,ij->i', nrm, cen - centroid[None, :])
            self.col_orient[dev_of] = np.where(outward > 0, 1, -1)
            self.col_convex = 1
            self.col_nbr_complete = ok
        # ---- rows, start points, beams
        self.grid_lo = np.ascontiguousarray(t.grid_lo, dtype=np.float64)
        self.grid_hi = np.ascontiguousarray(t.grid_hi, dtype=np.float64)
        if start_points is None:
            start_points = t.anchor_points
        self.start_pos = np.asarray([p[0] for p in start_points], dtype=np.float64).reshape(-1, 3)
        self.start_quat = np.asarray([pt.pose_orn_quaternion(p[1]) for p in start_points],
                                     dtype=np.float64).reshape(-1, 4)
        # cone beams of PAINT_METHOD 'normal': the part's uniform lattice (rob:23-35), or the caller's table -- COLOR_MODE
        # 'HSI' draws its own (part_tables.beta_plain, rob:38-69)
        self.beams = np.ascontiguousarray(t.beams if beams is None else beams, dtype=np.float64).reshape(-1, 3)

    # ------------------------------------------------------------------
    def c_struct(self):
        """PrlPartTables pointing at this object's arrays (keep ``self`` alive while it is used)."""
        t = self.tables
        s = _lib.PrlPartTables()

        def dp(a):
            return a.ctypes.data_as(_lib._dp)

        def ip(a):
            return a.ctypes.data_as(_lib._ip)

        s.n_samples, s.n_samples_pad = self.n_samples, self.n_samples_pad
        for k in range(3):
            s.sample_xyz[k] = dp(self.sample_xyz[k])
            s.vertex_xyz[k] = dp(self.vertex_xyz[k])
        s.word_bbox = dp(self.word_bbox)
        s.word_valid = self.word_valid.ctypes.data_as(_lib._up)
        s.sample_rank = ip(self.sample_rank)
        s.sgrid_origin[0], s.sgrid_origin[1], s.sgrid_inv_cell, s.sgrid_nx, s.sgrid_ny = self.sgrid
        s.sgrid_start = ip(self.sgrid_start)
        s.n_obs_cells = self.n_obs_cells
        s.obs_cell_mask = self.obs_cell_mask.ctypes.data_as(_lib._up)
        s.obs_cell_count = ip(self.obs_cell_count)
        s.n_vertices = self.vertex_rank.shape[0]
        s.vertex_rank = ip(self.vertex_rank)
        s.adj_width, s.vertex_adj = self.adj_width, ip(self.vertex_adj)
        s.vgrid_origin[0], s.vgrid_origin[1], s.vgrid_inv_cell, s.vgrid_nx, s.vgrid_ny = self.vgrid
        s.vgrid_accept = self.vgrid_accept
        s.vgrid_start = ip(self.vgrid_start)
        s.n_kd_nodes, s.kd_node, s.kd_split = self.n_kd_nodes, ip(self.kd_node), dp(self.kd_split)
        s.n_kd_points, s.kd_points = int(self.kd_points.shape[0]), ip(self.kd_points)
        for k in range(6):
            s.kd_box[k] = float(self.kd_box[k])
        s.n_triangles = self.tri_records.shape[0]
        s.tri_records = dp(self.tri_records)
        s.n_collision, s.n_collision_pad = self.n_collision, self.n_collision_pad
        for k in range(9):
            s.col_v0e1e2[k] = dp(self.col[k])
        s.col_bbox = self.col_bbox.ctypes.data_as(_lib._fp)
        s.col_rank = ip(self.col_rank)
        s.col_convex, s.nbr_width = self.col_convex, self.nbr_width
        s.col_nbr, s.col_orient = ip(self.col_nbr), ip(self.col_orient)
        s.n_col_chunks = self.n_col_chunks
        s.col_chunk_bbox = self.col_chunk_bbox.ctypes.data_as(_lib._fp)
        s.grid_lo, s.grid_hi = dp(self.grid_lo), dp(self.grid_hi)
        s.range1[0], s.range1[1] = t.ranges[0]
        s.range2[0], s.range2[1] = t.ranges[1]
        s.length_width_ratio = t.lwr
        s.axis0, s.axis1, s.axis2 = t.a0, t.a1, t.a2
        s.n_start = self.start_pos.shape[0]
        s.start_pos, s.start_quat = dp(self.start_pos), dp(self.start_quat)
        s.n_beams = self.beams.shape[0]
        s.beams = dp(self.beams)
        return s

    def mask_to_canonical(self, words):
        """u64[..., n_words] device-order coverage words -> bool[..., P] in canonical sample order
        (ascending j*W+i, the order of PartTables.sample_pix)."""
        words = np.ascontiguousarray(words, dtype=np.uint64)
        bits = np.unpackbits(words[..., :self.n_words].view(np.uint8), axis=-1, bitorder='little')
        return bits[..., self.inv_perm].astype(bool)
