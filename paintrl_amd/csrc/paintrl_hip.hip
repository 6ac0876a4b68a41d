// paintrl_hip.hip -- MI355X (gfx950) batched paint-coverage simulator: the host side of the C ABI (include/paintrl.h).
//
// Table upload (prl_part_create: the device layouts and everything derived from the host tables), batch state, and the
// entry points that launch the kernels.  The kernels live in their own translation units, compiled side by side
// (prl_launch.hpp): k_step.hip (one launch per batched step), k_cone.hip (PAINT_METHOD 'normal'), k_rollout.hip
// (policy + step, persistent fragments), k_big.hip (parts beyond 16 384 samples), policy_mlp.hip (the policy alone).
// The device code is in the prl_*.hpp headers next to this file.  There is no CPU fallback anywhere in the library.
#include "prl_all.hpp"
#include "prl_kargs.hpp"
#define PRL_HAVE_F32X4
#include "prl_policy.hpp"

#include <algorithm>
#include <array>
#include <cstdarg>
#include <limits>
#include <map>
#include <new>
#include <vector>

#ifndef PRL_HG_CELL
#define PRL_HG_CELL 0.25           // edge of a front-facet grid cell in root mean facet areas: the walk's starting facets (0.7: 3.7 loop trips a beam trip, 0.2: 3.0)
#endif
#ifndef PRL_FINE_CELL
#define PRL_FINE_CELL 1.6          // edge of a fine sample-grid cell in mean sample spacings (prl_cone.hpp; 1.5 / 1.6 / 1.75 / 2.0: beams + far kernel 137 / 134 / 139 / 147 us)
#endif

namespace {

// ---------------------------------------------------------------- rayTestBatch drop-in: one wave per ray
__global__ __launch_bounds__(256) void ray_batch_kernel(const PartDev *part, int n, const double *from,
                                                        const double *to, int *tri, double *frac, double *pos) {
    const int lane = threadIdx.x & 63;
    const int r = rfl(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (r >= n) return;
    const double o[3] = {from[3 * r], from[3 * r + 1], from[3 * r + 2]};
    const double e[3] = {to[3 * r], to[3 * r + 1], to[3 * r + 2]};
    double t, hit[3] = {0, 0, 0};
    int hint = -1;
    const int facet = ray_closest_wave(*(const PartDev CAS *)part, o, e, lane, t, hit, hint, wave_lds<false>().cand);
    if (lane == 0) {
        tri[r] = facet >= 0 ? part->col_rank[facet] : -1;          // the reference's triangle index
        frac[r] = t;
        pos[3 * r] = hit[0];
        pos[3 * r + 1] = hit[1];
        pos[3 * r + 2] = hit[2];
    }
}

// Copies the coverage words out (prl_batch_get_mask).  Same access shape as the step kernel's mask
// traffic (8 bytes per lane, coalesced), so tools/hbm_calibration.py uses it to calibrate FETCH_SIZE.
__global__ void copy_mask_kernel(const uint64_t *src, uint64_t *dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

__global__ void gather_state_kernel(const double *state, int n, int field, double *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = state[(size_t)i * PRL_STATE_DOUBLES + field];
}

// ================================================================= host side
thread_local char g_error[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(PRL_E_HIP, "%s: %s", #expr, hipGetErrorString(e_));     \
    } while (0)

}  // namespace

struct PrlPart {
    int device = 0;
    PartDev dev{};                 // device pointers inside
    std::vector<void *> allocs;
    int max_adj = 0;
    PartDev *dev_copy = nullptr;   // single-part device copy (prl_ray_batch)
};

struct PrlBatch {
    int device = 0, n_envs = 0, n_parts = 0, mask_stride = 0, kw = 0;
    bool kd = false;               // some part carries the reference's stale vertex kd-tree
    int max_beams = 0;             // largest cone-beam count of the parts (PAINT_METHOD 'normal')
    PrlConfig cfg{};
    PartDev *parts_dev = nullptr;
    CfgDev *cfg_dev = nullptr;
    int *env_part_dev = nullptr;
    int *slot_env_dev = nullptr;          // StepArgs::slot_env (batches of several parts)
    uint64_t *painted = nullptr, *last = nullptr;
    uint64_t *last_nz = nullptr;          // StepArgs::last_nz
    int nz_stride = 0;                    // words of it per env: KW_MAX (register-resident masks), (mask_stride + 63) / 64 (large parts)
    uint8_t *thick = nullptr;      // COLOR_MODE 'HSI' only
    double *cone_shots = nullptr, *cone_aux = nullptr;      // PAINT_METHOD 'normal' only (StepArgs)
    int *cone_hits = nullptr, *cone_work = nullptr;
    double *cone_far = nullptr;
    int32_t *scratch_action = nullptr;    // prl_rollout_fragment's launch-by-launch path: the bootstrap pass's discarded draw
    int cone_nb = 0;
    std::vector<double *> reset_obs;   // per part: [n_start][obs_dim], see PartDev::reset_obs
    int resident_envs = 0;             // envs whose waves are all resident at once (16 per CU): see STEP_WAVES_WIDE
    double *state = nullptr;
    int timing_every = 0;          // 0 = off; k = HIP events around every k-th step launch
    long long launch_no = 0;
    std::vector<hipEvent_t> ev_start, ev_stop;
    size_t ev_used = 0;
};

namespace {

template <typename T>
int upload(PrlPart *p, const T *host, size_t count, const T GAS **out) {
    *out = nullptr;
    if (count == 0) return PRL_OK;
    if (!host) return fail(PRL_E_INVALID, "null table pointer");
    void *d = nullptr;
    HIP_TRY(hipMalloc(&d, count * sizeof(T)));
    p->allocs.push_back(d);
    HIP_TRY(hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T GAS *)(d);
    return PRL_OK;
}

#define UP(field, host, count)                                        \
    do {                                                              \
        int rc_ = upload(p, host, (size_t)(count), &p->dev.field);    \
        if (rc_) return rc_;                                          \
    } while (0)

// grid-cell start tables index the sorted arrays: a malformed one would send device loads out of bounds
bool monotone_starts(const int32_t *start, size_t n, int limit) {
    if (!start || n == 0 || start[0] < 0) return false;
    for (size_t k = 1; k < n; ++k)
        if (start[k] < start[k - 1]) return false;
    return start[n - 1] <= limit;
}

int part_fill(PrlPart *p, const PrlPartTables *t) {
    PartDev &d = p->dev;
    if (t->n_samples <= 0 || t->n_samples_pad % 64 || t->n_samples_pad < t->n_samples)
        return fail(PRL_E_INVALID, "bad sample counts %d/%d", t->n_samples, t->n_samples_pad);
    d.n_samples = t->n_samples;
    d.n_samples_pad = t->n_samples_pad;
    d.n_words = t->n_samples_pad / 64;
    if (d.n_words > BIG_MAX_WORDS)
        return fail(PRL_E_UNSUPPORTED, "part has %d samples; at most %d fit the LDS-resident masks of the large-part kernels",
                    t->n_samples, 64 * BIG_MAX_WORDS);
    for (int k = 0; k < 3; ++k) UP(samp[k], t->sample_xyz[k], t->n_samples_pad);
    {   // float copy of the sample positions for the conservative paint pre-filter (prl_paint.hpp)
        if (!t->word_valid) return fail(PRL_E_INVALID, "null table pointer");
        std::vector<float> f4((size_t)t->n_samples_pad * 4, 0.0f);
        double amax = 0;
        for (int i = 0; i < t->n_samples_pad; ++i) {
            const bool real = (t->word_valid[i >> 6] >> (i & 63)) & 1;
            for (int k = 0; k < 3; ++k) {
                const double v = t->sample_xyz[k][i];
                f4[(size_t)i * 4 + k] = (float)v;
                if (real && std::fabs(v) > amax) amax = std::fabs(v);
            }
        }
        if (!(amax < 1.0e6)) return fail(PRL_E_UNSUPPORTED, "sample coordinates beyond 1e6 (%g)", amax);
        d.samp_absmax = amax;
        UP(samp_f32, f4.data(), f4.size());
    }
    UP(word_bbox, t->word_bbox, (size_t)d.n_words * 4);
    UP(word_valid, t->word_valid, d.n_words);
    UP(samp_rank, t->sample_rank, t->n_samples_pad);
    d.sg_o1 = t->sgrid_origin[0];
    d.sg_o2 = t->sgrid_origin[1];
    d.sg_inv = t->sgrid_inv_cell;
    d.sg_nx = t->sgrid_nx;
    d.sg_ny = t->sgrid_ny;
    if (d.sg_nx <= 0 || d.sg_ny <= 0) return fail(PRL_E_INVALID, "empty sample grid");
    UP(sg_start, t->sgrid_start, (size_t)d.sg_nx * d.sg_ny + 1);
    if (!monotone_starts(t->sgrid_start, (size_t)d.sg_nx * d.sg_ny + 1, t->n_samples_pad))
        return fail(PRL_E_INVALID, "sample grid starts must be non-decreasing and within the padded sample count");
    d.n_obs_cells = t->n_obs_cells;
    if (d.n_obs_cells > 0) {
        UP(cell_mask, t->obs_cell_mask, (size_t)d.n_obs_cells * d.n_words);
        UP(cell_count, t->obs_cell_count, d.n_obs_cells);
        // which cells a word's samples lie in (derived): nearly every word lies in ONE cell of the grid observation, whose count
        // is then the popcount of its painted word -- no mask read; the words on a cell boundary name their two to four cells
        std::vector<int32_t> wc((size_t)d.n_words);
        for (int w = 0; w < d.n_words; ++w) {
            uint32_t packed = 0xffffffffu;
            int n = 0;
            for (int c = 0; c < d.n_obs_cells && n <= 4; ++c)
                if (t->obs_cell_mask[(size_t)c * d.n_words + w] != 0) {
                    if (n < 4 && c < 255) packed = (packed & ~(0xffu << (8 * n))) | ((uint32_t)c << (8 * n));
                    else n = 4;                                   // a fifth cell, or an index the byte cannot hold
                    ++n;
                }
            wc[(size_t)w] = (int32_t)(n > 4 ? 0xfffffffeu : packed);
        }
        UP(word_cells, wc.data(), wc.size());
    }
    d.n_vertices = t->n_vertices;
    if (d.n_vertices <= 0) return fail(PRL_E_INVALID, "part has no same-side vertices");
    {
        for (int k = 0; k < 3; ++k)
            if (!t->vertex_xyz[k]) return fail(PRL_E_INVALID, "null table pointer");
        if (!t->vertex_rank) return fail(PRL_E_INVALID, "null table pointer");
        std::vector<double> v4((size_t)d.n_vertices * 4, 0.0);
        for (int i = 0; i < d.n_vertices; ++i) {
            for (int k = 0; k < 3; ++k) v4[(size_t)i * 4 + k] = t->vertex_xyz[k][i];
            const int32_t pair[2] = {t->vertex_rank[i], 0};
            std::memcpy(&v4[(size_t)i * 4 + 3], pair, sizeof pair);
        }
        UP(vert4, v4.data(), v4.size());
    }
    d.n_triangles = t->n_triangles;
    d.adj_width = t->adj_width;
    if (d.adj_width < 1 || d.adj_width > 64)
        return fail(PRL_E_UNSUPPORTED, "adjacency width %d (a vertex may have at most 64 incident triangles)", d.adj_width);
    if (!t->vertex_adj) return fail(PRL_E_INVALID, "null adjacency table");
    for (size_t k = 0; k < (size_t)d.n_vertices * d.adj_width; ++k)
        if (t->vertex_adj[k] < -1 || t->vertex_adj[k] >= d.n_triangles)
            return fail(PRL_E_INVALID, "adjacency entry %zu out of range", k);
    UP(vadj, t->vertex_adj, (size_t)d.n_vertices * d.adj_width);
    d.vg_o1 = t->vgrid_origin[0];
    d.vg_o2 = t->vgrid_origin[1];
    d.vg_inv = t->vgrid_inv_cell;
    d.vg_accept = t->vgrid_accept;
    d.vg_nx = t->vgrid_nx;
    d.vg_ny = t->vgrid_ny;
    if (d.vg_nx <= 0 || d.vg_ny <= 0) return fail(PRL_E_INVALID, "empty vertex grid");
    UP(vg_start, t->vgrid_start, (size_t)d.vg_nx * d.vg_ny + 1);
    if (!monotone_starts(t->vgrid_start, (size_t)d.vg_nx * d.vg_ny + 1, d.n_vertices) ||
        t->vgrid_start[(size_t)d.vg_nx * d.vg_ny] != d.n_vertices)
        return fail(PRL_E_INVALID, "vertex grid starts must be non-decreasing and end at n_vertices");
    d.n_kd_nodes = t->n_kd_nodes;
    if (d.n_kd_nodes > 0) {       // the reference's stale vertex kd-tree: every index is checked, the device walks it blindly
        if (!t->kd_node || !t->kd_split || !t->kd_points || t->n_kd_points < 1) return fail(PRL_E_INVALID, "null kd-tree table");
        for (int i = 0; i < d.n_kd_nodes; ++i) {
            const int32_t *nd = t->kd_node + 4 * (size_t)i;
            const bool leaf = nd[0] < 0;
            if (nd[0] > 2) return fail(PRL_E_INVALID, "kd node %d: split dimension %d", i, nd[0]);
            if (leaf ? (nd[1] < 0 || nd[2] < nd[1] || nd[2] > t->n_kd_points)
                     : (nd[1] <= i || nd[2] <= i || nd[1] >= d.n_kd_nodes || nd[2] >= d.n_kd_nodes))
                return fail(PRL_E_INVALID, "kd node %d: children / point range out of order", i);   // children follow
        }                                                                                           // their parent: no cycles
        for (int i = 0; i < t->n_kd_points; ++i)
            if (t->kd_points[i] < -1 || t->kd_points[i] >= d.n_vertices) return fail(PRL_E_INVALID, "kd point %d out of range", i);
        UP(kd_node, t->kd_node, (size_t)d.n_kd_nodes * 4);
        UP(kd_split, t->kd_split, d.n_kd_nodes);
        UP(kd_points, t->kd_points, t->n_kd_points);
        {   // the leaves' vertex records in tree order (derived): a leaf scan reads its points' x y z | {vertex, 0} in ONE round
            // trip instead of two (the point's vertex id, then that vertex's record); a parked row (-1) keeps id -1
            std::vector<int32_t> leaf_of((size_t)t->n_kd_points, 0);         // the leaf (node index) each tree-order point belongs to
            for (int i = 0; i < d.n_kd_nodes; ++i) {
                const int32_t *nd = t->kd_node + 4 * (size_t)i;
                if (nd[0] < 0)
                    for (int p = nd[1]; p < nd[2]; ++p) leaf_of[(size_t)p] = i;
            }
            std::vector<double> kr((size_t)t->n_kd_points * 4, 0.0);
            for (int i = 0; i < t->n_kd_points; ++i) {
                const int v = t->kd_points[i];
                if (v >= 0)
                    for (int k = 0; k < 3; ++k) kr[(size_t)i * 4 + k] = t->vertex_xyz[k][v];
                const int32_t pair[2] = {v, leaf_of[(size_t)i]};
                std::memcpy(&kr[(size_t)i * 4 + 3], pair, sizeof pair);
            }
            UP(kd_rec, kr.data(), kr.size());
            d.n_kd_points = t->n_kd_points;
        }
        {   // the same tree for the lane-parallel query (prl_search.hpp nearest_vertex_kd_lanes): one lane per node, the leaves'
            // points in rows of sixteen.  Children follow their parents (checked above), so one pass in index order sees a
            // node's parent first; anything that is not a tree of small leaves keeps the general walk.
            const int n = d.n_kd_nodes;
            bool ok = n <= 64;
            std::vector<int> parent((size_t)n, 0), depth((size_t)n, 0), lesser((size_t)n, 0), ordinal((size_t)n, -1), seen((size_t)n, 0);
            int n_leaves = 0, deepest = 0;
            seen[0] = 1;
            for (int i = 0; ok && i < n; ++i) {
                const int32_t *nd = t->kd_node + 4 * (size_t)i;
                if (!seen[(size_t)i]) ok = false;
                else if (nd[0] < 0) {
                    int live = 0;                                    // (rows parked at (10, 10, 10) are never the nearest: left out --
                    for (int pt = nd[1]; pt < nd[2]; ++pt) live += t->kd_points[pt] >= 0;     // scipy keeps them all in ONE leaf)
                    ok = live <= 16;
                    ordinal[(size_t)i] = n_leaves++;
                } else
                    for (int k = 1; k <= 2 && ok; ++k) {
                        const int c = nd[k];
                        if (seen[(size_t)c]) ok = false;
                        seen[(size_t)c] = 1;
                        parent[(size_t)c] = i;
                        depth[(size_t)c] = depth[(size_t)i] + 1;
                        lesser[(size_t)c] = k == 1;
                        deepest = std::max(deepest, depth[(size_t)c]);
                    }
            }
            ok = ok && n_leaves <= 32 && d.n_vertices < (1 << 28);
            if (ok) {
                std::vector<int32_t> kl((size_t)n * 4, 0);
                // (padding: points at +inf with vertex 0 -- their distance is +inf, never below the best; whole trips of 64 slots)
                const size_t slots16 = ((size_t)n_leaves * 16 + 63) / 64 * 64;
                std::vector<double> k16(slots16 * 4, std::numeric_limits<double>::infinity());
                for (size_t q = 0; q < slots16; ++q) k16[q * 4 + 3] = 0.0;
                for (int i = 0; i < n; ++i) {
                    const int32_t *nd = t->kd_node + 4 * (size_t)i;
                    const bool leaf = nd[0] < 0;
                    const int pa = parent[(size_t)i], psd = i ? t->kd_node[4 * (size_t)pa] : 0;
                    kl[4 * (size_t)i] = (leaf ? 3 : nd[0]) | (psd << 2) | (lesser[(size_t)i] << 4) | (depth[(size_t)i] << 5) | (pa << 11) |
                                        ((leaf ? 0 : nd[1]) << 17) | ((leaf ? 0 : nd[2]) << 23);
                    kl[4 * (size_t)i + 1] = ordinal[(size_t)i];
                    const double psp = i ? t->kd_split[pa] : 0.0;
                    std::memcpy(&kl[4 * (size_t)i + 2], &psp, sizeof psp);
                    if (leaf) {
                        int j = 0;
                        for (int pt = nd[1]; pt < nd[2]; ++pt) {     // the live points in tree order, then padding
                            const int v = t->kd_points[pt];
                            if (v < 0) continue;
                            double *r = &k16[((size_t)ordinal[(size_t)i] * 16 + (size_t)j++) * 4];
                            for (int k = 0; k < 3; ++k) r[k] = t->vertex_xyz[k][v];
                            const int32_t pair[2] = {v, i};
                            std::memcpy(r + 3, pair, sizeof pair);
                        }
                    }
                }
                std::vector<uint64_t> anc((size_t)n, 0);
                for (int i = 1; i < n; ++i) {
                    const int dim = t->kd_node[4 * (size_t)parent[(size_t)i]];
                    for (int c = parent[(size_t)i]; c != 0; c = parent[(size_t)c])
                        if (t->kd_node[4 * (size_t)parent[(size_t)c]] == dim) anc[(size_t)i] |= 1ull << c;
                }
                UP(kd_anc, anc.data(), anc.size());
                UP(kd_lane, kl.data(), kl.size());
                UP(kd_rec16, k16.data(), k16.size());
                d.n_kd_leaves = n_leaves;
                d.kd_depth = deepest;
            }
        }
        for (int k = 0; k < 6; ++k) d.kd_box[k] = t->kd_box[k];
    }
    {   // device triangle records: the host's 16 doubles + the quaternion and shot-centre offset a hit on the
        // triangle produces (prl_device.hpp TRI_REC), so that the step kernel reads them instead of running a
        // square root, four divisions and a rotation per sub-shot
        if (!t->tri_records || d.n_triangles <= 0) return fail(PRL_E_INVALID, "no triangle records");
        std::vector<double> rec((size_t)d.n_triangles * TRI_REC, 0.0);
        for (int i = 0; i < d.n_triangles; ++i) {
            const double *src = t->tri_records + (size_t)i * 16;
            double *r = rec.data() + (size_t)i * TRI_REC;
            for (int k = 0; k < 16; ++k) r[k] = src[k];
            const double orn[3] = {-src[13], -src[14], -src[15]};
            double q[4], off[3];
            pose_orn_quat(orn, q);
            quat_rotate(q, 0.0, 0.0, SHOT_CENTRE_OFFSET, off);
            for (int k = 0; k < 4; ++k) r[16 + k] = q[k];
            for (int k = 0; k < 3; ++k) r[20 + k] = off[k];
        }
        UP(tri_rec, rec.data(), rec.size());
    }
    d.n_col = t->n_collision;
    d.n_col_pad = t->n_collision_pad;
    if (d.n_col <= 0 || d.n_col_pad % 64 || d.n_col_pad < d.n_col) return fail(PRL_E_INVALID, "bad collision counts");
    for (int k = 0; k < 9; ++k) UP(col[k], t->col_v0e1e2[k], d.n_col_pad);
    UP(col_bbox, t->col_bbox, (size_t)d.n_col_pad * 8);
    UP(col_rank, t->col_rank, d.n_col_pad);
    d.col_convex = t->col_convex ? 1 : 0;
    d.nbr_width = t->nbr_width;
    if (d.col_convex) {
        if (d.nbr_width < 1 || d.nbr_width > 64) return fail(PRL_E_INVALID, "nbr_width must be 1..64");
        for (size_t k = 0; k < (size_t)d.n_col_pad * d.nbr_width; ++k)
            if (t->col_nbr[k] < -1 || t->col_nbr[k] >= d.n_col_pad) return fail(PRL_E_INVALID, "col_nbr entry out of range");
        UP(col_nbr, t->col_nbr, (size_t)d.n_col_pad * d.nbr_width);
        UP(col_orient, t->col_orient, d.n_col_pad);
        // per-facet record of the single-facet fast path (ray_closest_wave): geometry, the barycentric
        // margin that keeps a hit FACET_EDGE_MARGIN away from every edge, |e1 x e2|^2, orientation
        std::vector<double> rec((size_t)d.n_col_pad * 12, 0.0);
        for (int i = 0; i < d.n_col_pad; ++i) {
            double *r = rec.data() + (size_t)i * 12;
            for (int k = 0; k < 9; ++k) r[k] = t->col_v0e1e2[k][i];
            const double *e1 = r + 3, *e2 = r + 6;
            const double n0 = e1[1] * e2[2] - e1[2] * e2[1], n1 = e1[2] * e2[0] - e1[0] * e2[2],
                         n2 = e1[0] * e2[1] - e1[1] * e2[0];
            const double nn = n0 * n0 + n1 * n1 + n2 * n2;
            const double l1 = e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2];
            const double l2 = e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2];
            const double f0 = e2[0] - e1[0], f1 = e2[1] - e1[1], f2 = e2[2] - e1[2];
            const double l3 = f0 * f0 + f1 * f1 + f2 * f2;
            const double lmax = std::max(l1, std::max(l2, l3));
            const double hmin = lmax > 0 ? std::sqrt(nn / lmax) : 0.0;          // smallest height of the facet
            r[9] = hmin > 0 ? FACET_EDGE_MARGIN / hmin : INFINITY;              // never met by a sliver
            r[10] = nn;
            r[11] = (double)t->col_orient[i];                                   // 0 for pads: never entered
        }
        UP(col_rec, rec.data(), rec.size());
        // edge neighbours for the cone beams' surface walk (prl_cone.hpp): the corners v0, v0 + e1, v0 + e2 of facets
        // that share a vertex agree to rounding, so corners are matched on a 1e-9 m lattice; an edge that does not
        // end up with exactly two facets (a non-manifold soup, or a corner that straddles a lattice line) gets no
        // neighbour and the walk falls back to the wave-wide search there -- a speed matter only
        {
            std::map<std::array<long long, 3>, int> vid;
            std::vector<std::array<int, 3>> corner((size_t)d.n_col_pad, std::array<int, 3>{-1, -1, -1});
            for (int i = 0; i < d.n_col_pad; ++i) {
                if (t->col_orient[i] == 0) continue;                           // pad
                const double *r = rec.data() + (size_t)i * 12;
                const double c[3][3] = {{r[0], r[1], r[2]},
                                        {r[0] + r[3], r[1] + r[4], r[2] + r[5]},
                                        {r[0] + r[6], r[1] + r[7], r[2] + r[8]}};
                for (int k = 0; k < 3; ++k) {
                    const std::array<long long, 3> q = {std::llround(c[k][0] * 1e9), std::llround(c[k][1] * 1e9),
                                                        std::llround(c[k][2] * 1e9)};
                    corner[i][k] = vid.emplace(q, (int)vid.size()).first->second;
                }
            }
            std::map<std::pair<int, int>, std::vector<int>> edge_facets;
            auto key = [](int a, int b) { return a < b ? std::make_pair(a, b) : std::make_pair(b, a); };
            // edge e of a facet with corners (c0, c1 = c0 + e1, c2 = c0 + e2):  0: u = 0 -> (c0, c2);  1: v = 0 -> (c0, c1);
            // 2: u + v = 1 -> (c1, c2)
            static const int ea[3] = {0, 0, 1}, eb[3] = {2, 1, 2};
            for (int i = 0; i < d.n_col_pad; ++i)
                if (corner[i][0] >= 0)
                    for (int e = 0; e < 3; ++e) edge_facets[key(corner[i][ea[e]], corner[i][eb[e]])].push_back(i);
            std::vector<int32_t> enbr((size_t)d.n_col_pad * 4, -1);       // three neighbours | the facet's reference index
            for (int i = 0; i < d.n_col_pad; ++i) {
                enbr[(size_t)i * 4 + 3] = t->col_rank[i];
                if (corner[i][0] >= 0)
                    for (int e = 0; e < 3; ++e) {
                        const std::vector<int> &fs = edge_facets[key(corner[i][ea[e]], corner[i][eb[e]])];
                        if (fs.size() == 2) enbr[(size_t)i * 4 + e] = fs[0] == i ? fs[1] : fs[0];
                    }
            }
            UP(col_enbr, enbr.data(), enbr.size());
        }
        // front-facet grid for the cone beams' walks (prl_cone.hpp): for the centre of every cell of a grid over the set's
        // extent in the principal plane, the facet a line along axis a0 meets first coming from the tool's side; cells the
        // set does not cover take the facet of the nearest covered cell.  Only a starting point: any facet is correct.
        {
            const int a0 = t->axis0, a1 = t->axis1, a2 = t->axis2;
            if (a0 < 0 || a0 > 2 || a1 < 0 || a1 > 2 || a2 < 0 || a2 > 2) return fail(PRL_E_INVALID, "axes must be a permutation of 0,1,2");
            double lo1 = INFINITY, hi1 = -INFINITY, lo2 = INFINITY, hi2 = -INFINITY, zsum = 0;
            int n_real = 0;
            for (int i = 0; i < d.n_col_pad; ++i) {
                if (t->col_orient[i] == 0) continue;
                const double *r = rec.data() + (size_t)i * 12;
                for (int c = 0; c < 3; ++c) {
                    double v[3];
                    for (int k = 0; k < 3; ++k) v[k] = r[k] + (c == 1 ? r[3 + k] : (c == 2 ? r[6 + k] : 0.0));
                    lo1 = std::fmin(lo1, v[a1]), hi1 = std::fmax(hi1, v[a1]);
                    lo2 = std::fmin(lo2, v[a2]), hi2 = std::fmax(hi2, v[a2]);
                    zsum += v[a0];
                }
                ++n_real;
            }
            d.hg_nx = d.hg_ny = 0;
            if (n_real > 0 && hi1 > lo1 && hi2 > lo2 && t->start_pos && t->n_start > 0) {
                const bool from_above = t->start_pos[a0] >= zsum / (3.0 * n_real);          // the tool's side of the part
                double cell = PRL_HG_CELL * std::sqrt((hi1 - lo1) * (hi2 - lo2) / (double)n_real);
                while ((hi1 - lo1) / cell > 768 || (hi2 - lo2) / cell > 768) cell *= 1.5;
                const int nx = (int)std::floor((hi1 - lo1) / cell) + 1, ny = (int)std::floor((hi2 - lo2) / cell) + 1;
                std::vector<int32_t> g((size_t)nx * ny, -1);
                for (int i = 0; i < d.n_col_pad; ++i) {
                    if (t->col_orient[i] == 0) continue;
                    const double *r = rec.data() + (size_t)i * 12;
                    // the facet in the principal plane: p = q0 + u f1 + v f2; cells whose centre it covers
                    const double q1 = r[a1], q2 = r[a2], f11 = r[3 + a1], f12 = r[3 + a2], f21 = r[6 + a1], f22 = r[6 + a2];
                    const double den = f11 * f22 - f12 * f21;
                    if (!(std::fabs(den) > 1e-14)) continue;                                 // edge-on
                    const double x0 = std::fmin(q1, std::fmin(q1 + f11, q1 + f21)), x1 = std::fmax(q1, std::fmax(q1 + f11, q1 + f21));
                    const double y0 = std::fmin(q2, std::fmin(q2 + f12, q2 + f22)), y1 = std::fmax(q2, std::fmax(q2 + f12, q2 + f22));
                    const int cx0 = std::max(0, (int)std::floor((x0 - lo1) / cell - 0.5)), cx1 = std::min(nx - 1, (int)std::ceil((x1 - lo1) / cell));
                    const int cy0 = std::max(0, (int)std::floor((y0 - lo2) / cell - 0.5)), cy1 = std::min(ny - 1, (int)std::ceil((y1 - lo2) / cell));
                    for (int cy = cy0; cy <= cy1; ++cy)
                        for (int cx = cx0; cx <= cx1; ++cx) {
                            const double px = lo1 + (cx + 0.5) * cell - q1, py = lo2 + (cy + 0.5) * cell - q2;
                            const double u = (px * f22 - py * f21) / den, v = (py * f11 - px * f12) / den;
                            if (u < -1e-9 || v < -1e-9 || u + v > 1 + 1e-9) continue;
                            const double z = r[a0] + u * r[3 + a0] + v * r[6 + a0];
                            int32_t &slot = g[(size_t)cy * nx + cx];
                            if (slot >= 0) {
                                const double *ro = rec.data() + (size_t)slot * 12;
                                const double qo1 = ro[a1], qo2 = ro[a2], o11 = ro[3 + a1], o12 = ro[3 + a2], o21 = ro[6 + a1], o22 = ro[6 + a2];
                                const double deno = o11 * o22 - o12 * o21;
                                const double pxo = lo1 + (cx + 0.5) * cell - qo1, pyo = lo2 + (cy + 0.5) * cell - qo2;
                                const double uo = (pxo * o22 - pyo * o21) / deno, vo = (pyo * o11 - pxo * o12) / deno;
                                const double zo = ro[a0] + uo * ro[3 + a0] + vo * ro[6 + a0];
                                if (from_above ? !(z > zo) : !(z < zo)) continue;
                            }
                            slot = i;
                        }
                }
                // cells beside the set: the facet of the nearest covered cell (breadth-first from the covered ones)
                std::vector<int> queue;
                for (int c = 0; c < nx * ny; ++c)
                    if (g[c] >= 0) queue.push_back(c);
                for (size_t h = 0; h < queue.size(); ++h) {
                    const int c = queue[h], cx = c % nx, cy = c / nx;
                    static const int dx[4] = {1, -1, 0, 0}, dy[4] = {0, 0, 1, -1};
                    for (int k = 0; k < 4; ++k) {
                        const int ex = cx + dx[k], ey = cy + dy[k];
                        if (ex < 0 || ex >= nx || ey < 0 || ey >= ny || g[(size_t)ey * nx + ex] >= 0) continue;
                        g[(size_t)ey * nx + ex] = g[c];
                        queue.push_back(ey * nx + ex);
                    }
                }
                if (!queue.empty()) {
                    d.hg_o1 = lo1;
                    d.hg_o2 = lo2;
                    d.hg_inv = 1.0 / cell;
                    d.hg_nx = nx;
                    d.hg_ny = ny;
                    UP(hg_facet, g.data(), g.size());
                }
            }
        }
    }
    d.n_col_chunks = t->n_col_chunks;
    if (d.n_col_chunks != d.n_col_pad / 64) return fail(PRL_E_INVALID, "n_col_chunks must be n_collision_pad / 64");
    UP(col_chunk_bbox, t->col_chunk_bbox, (size_t)((d.n_col_chunks + 63) / 64) * 64 * 8);
    UP(grid_lo, t->grid_lo, GRID_GRANULARITY);
    UP(grid_hi, t->grid_hi, GRID_GRANULARITY);
    d.r1min = t->range1[0];
    d.r1max = t->range1[1];
    d.r2min = t->range2[0];
    d.r2max = t->range2[1];
    d.lwr = t->length_width_ratio;
    d.a0 = t->axis0;
    d.a1 = t->axis1;
    d.a2 = t->axis2;
    if (d.a0 < 0 || d.a0 > 2 || d.a1 < 0 || d.a1 > 2 || d.a2 < 0 || d.a2 > 2 || d.a0 == d.a1 || d.a1 == d.a2 ||
        d.a0 == d.a2)
        return fail(PRL_E_INVALID, "axes must be a permutation of 0,1,2");
    d.samp_a1 = d.samp[d.a1];
    d.samp_a2 = d.samp[d.a2];
    {   // the samples of a word ascend on axis a1 (paintrl.h); derive the equal-run ends the observation uses
        const double *x = t->sample_xyz[d.a1];
        std::vector<uint8_t> ub((size_t)t->n_samples_pad);
        std::vector<double> pivot((size_t)d.n_words * 8);
        for (int w = 0; w < d.n_words; ++w) {
            const double *xw = x + (size_t)w * 64;
            for (int j = 0; j < 8; ++j) pivot[(size_t)w * 8 + j] = xw[8 * j + 7];
            for (int j = 1; j < 64; ++j)
                if (!(xw[j - 1] <= xw[j]))
                    return fail(PRL_E_INVALID, "samples of word %d do not ascend on axis %d", w, d.a1);
            int end = 64;
            for (int j = 63; j >= 0; --j) {
                if (j < 63 && xw[j] != xw[j + 1]) end = j + 1;
                ub[(size_t)w * 64 + j] = (uint8_t)end;
            }
        }
        UP(samp_ub, ub.data(), ub.size());
        UP(word_pivot, pivot.data(), pivot.size());
        // float copy of the samples' axis-a2 coordinate (nearest float): the observation's pass over the words that straddle
        // the tool's horizontal line only (prl_observe.hpp section4_accumulate, pass 3a)
        const double *y = t->sample_xyz[d.a2];
        std::vector<float> yf((size_t)t->n_samples_pad);
        for (int i = 0; i < t->n_samples_pad; ++i) yf[(size_t)i] = (float)y[i];
        UP(samp_a2_f32, yf.data(), yf.size());
        // per word: its cell row and number of valid samples, and a float interval around its a1 range (section4_big).  Cell
        // rows must start on word boundaries and hold exactly the samples whose a2 coordinate falls into them by the device's
        // own cell arithmetic: the observation of a large part classifies whole rows against the tool's row.
        std::vector<int32_t> info((size_t)d.n_words, 0);
        std::vector<float> x32((size_t)d.n_words * 2);
        std::vector<int> row_of((size_t)d.n_words, -1);
        for (int r = 0; r < d.sg_ny; ++r) {
            const int b = t->sgrid_start[(size_t)r * d.sg_nx], e = t->sgrid_start[(size_t)(r + 1) * d.sg_nx];
            if (b % 64 && e > b) return fail(PRL_E_INVALID, "sample grid row %d does not start on a 64-sample boundary", r);
            for (int w = b / 64; w < (e + 63) / 64 && w < d.n_words; ++w) row_of[(size_t)w] = r;
        }
        if (d.sg_ny > 0xffff) return fail(PRL_E_UNSUPPORTED, "sample grid of %d rows", d.sg_ny);
        for (int w = 0; w < d.n_words; ++w) {
            const uint64_t vw = t->word_valid[w];
            int nv = 0;
            double lo = INFINITY, hi = -INFINITY;
            for (int j = 0; j < 64; ++j)
                if ((vw >> j) & 1) {
                    ++nv;
                    const size_t i = (size_t)w * 64 + j;
                    lo = std::min(lo, x[i]);
                    hi = std::max(hi, x[i]);
                    if (row_of[(size_t)w] < 0 || cell_coord(y[i], d.sg_o2, d.sg_inv, d.sg_ny) != row_of[(size_t)w])
                        return fail(PRL_E_INVALID, "sample %zu does not lie in the cell row its word belongs to", i);
                }
            info[(size_t)w] = (int32_t)(((uint32_t)nv << 16) | (uint32_t)(row_of[(size_t)w] < 0 ? 0 : row_of[(size_t)w]));
            float flo = (float)lo, fhi = (float)hi;
            if ((double)flo > lo) flo = std::nextafterf(flo, -INFINITY);
            if ((double)fhi < hi) fhi = std::nextafterf(fhi, INFINITY);
            x32[(size_t)w * 2] = flo;
            x32[(size_t)w * 2 + 1] = fhi;
        }
        UP(word_info, info.data(), info.size());
        UP(word_x32, x32.data(), x32.size());
#ifdef PRL_OBS_YSORT
        if (true) {                     // (A/B: the small parts' observation too)
#else
        if (d.n_samples > 16384) {      // (observation_big's parts) the words' samples by ascending a2, and the suffix masks of that order
#endif
            std::vector<double> ysort((size_t)d.n_words * 64, INFINITY);
            std::vector<uint64_t> ymask((size_t)d.n_words * 65, 0);
            for (int w = 0; w < d.n_words; ++w) {
                const uint64_t vw = t->word_valid[w];
                int order[64], nv = 0;
                for (int j = 0; j < 64; ++j)
                    if ((vw >> j) & 1) order[nv++] = j;
                std::stable_sort(order, order + nv, [&](int a, int b) { return y[(size_t)w * 64 + a] < y[(size_t)w * 64 + b]; });
                uint64_t m = 0;
                for (int r = nv - 1; r >= 0; --r) {
                    m |= 1ull << order[r];
                    ymask[(size_t)w * 65 + r] = m;
                    ysort[(size_t)w * 64 + r] = y[(size_t)w * 64 + order[r]];
                }
            }
            std::vector<double> ypivot((size_t)d.n_words * 8);
            for (int w = 0; w < d.n_words; ++w)
                for (int j = 0; j < 8; ++j) ypivot[(size_t)w * 8 + j] = ysort[(size_t)w * 64 + 8 * j + 7];
            UP(word_ypivot, ypivot.data(), ypivot.size());
            UP(word_ysort, ysort.data(), ysort.size());
            UP(word_ymask, ymask.data(), ymask.size());
        }
    }
    {   // fine grid over the real samples for the lane-parallel nearest-sample query (prl_cone.hpp): ~2.6 samples a cell
        const double *x1 = t->sample_xyz[d.a1], *x2 = t->sample_xyz[d.a2];
        double lo1 = INFINITY, hi1 = -INFINITY, lo2 = INFINITY, hi2 = -INFINITY;
        std::vector<int> real;
        for (int i = 0; i < t->n_samples_pad; ++i)
            if ((t->word_valid[i >> 6] >> (i & 63)) & 1) {
                real.push_back(i);
                lo1 = std::fmin(lo1, x1[i]);
                hi1 = std::fmax(hi1, x1[i]);
                lo2 = std::fmin(lo2, x2[i]);
                hi2 = std::fmax(hi2, x2[i]);
            }
        if ((int)real.size() != t->n_samples) return fail(PRL_E_INVALID, "word_valid marks %zu samples, n_samples is %d", real.size(), t->n_samples);
        double cell = PRL_FINE_CELL * std::sqrt(std::fmax((hi1 - lo1) * (hi2 - lo2), 1e-12) / (double)real.size());
        cell = std::fmax(cell, 1e-6);
        while (((hi1 - lo1) / cell + 1) * ((hi2 - lo2) / cell + 1) > 4.0e6) cell *= 2;
        const double inv = 1.0 / cell;
        const int nx = (int)std::floor((hi1 - lo1) * inv) + 1, ny = (int)std::floor((hi2 - lo2) * inv) + 1;
        std::vector<int> cell_of(real.size()), start((size_t)nx * ny + 1, 0);
        for (size_t j = 0; j < real.size(); ++j) {
            int cx = (int)std::floor((x1[real[j]] - lo1) * inv), cy = (int)std::floor((x2[real[j]] - lo2) * inv);
            cx = cx < 0 ? 0 : (cx > nx - 1 ? nx - 1 : cx);
            cy = cy < 0 ? 0 : (cy > ny - 1 ? ny - 1 : cy);
            cell_of[j] = cy * nx + cx;
            ++start[(size_t)cell_of[j] + 1];
        }
        for (size_t c = 0; c < (size_t)nx * ny; ++c) start[c + 1] += start[c];
        std::vector<int> fill(start.begin(), start.end() - 1);
        std::vector<double> rec(real.size() * 4);
        for (size_t j = 0; j < real.size(); ++j) {
            const int i = real[j], slot = fill[cell_of[j]]++;
            for (int k = 0; k < 3; ++k) rec[(size_t)slot * 4 + k] = t->sample_xyz[k][i];
            const int32_t pair[2] = {t->sample_rank[i], (int32_t)i};
            std::memcpy(&rec[(size_t)slot * 4 + 3], pair, sizeof pair);
        }
        d.fg_o1 = lo1;
        d.fg_o2 = lo2;
        d.fg_inv = inv;
        d.fg_accept = 0.99 * cell;
        d.fg_nx = nx;
        d.fg_ny = ny;
        UP(fg_start, start.data(), start.size());
        {   // fg_seed: for every cell the record of the sample nearest to the cell's centre in the principal plane -- of the cell
            // itself, or of the nearest cell with samples (two raster sweeps that hand the nearest source cell from neighbour
            // to neighbour: nearest but for rare ties in the sweep order; any sample is a correct first bound of the far
            // kernel's search, a near one a tight one)
            std::vector<int> src((size_t)nx * ny, -1);
            for (int c = 0; c < nx * ny; ++c)
                if (start[c + 1] > start[c]) src[c] = c;
            auto dist2 = [&](int c, int s) {
                const long long dx = c % nx - s % nx, dy = c / nx - s / nx;
                return dx * dx + dy * dy;
            };
            auto relax = [&](int c, int x, int y) {
                if (x < 0 || x >= nx || y < 0 || y >= ny) return;
                const int s = src[y * nx + x];
                if (s >= 0 && (src[c] < 0 || dist2(c, s) < dist2(c, src[c]))) src[c] = s;
            };
            for (int y = 0; y < ny; ++y) {
                for (int x = 0; x < nx; ++x) relax(y * nx + x, x - 1, y), relax(y * nx + x, x, y - 1), relax(y * nx + x, x - 1, y - 1), relax(y * nx + x, x + 1, y - 1);
                for (int x = nx - 1; x >= 0; --x) relax(y * nx + x, x + 1, y);
            }
            for (int y = ny - 1; y >= 0; --y) {
                for (int x = nx - 1; x >= 0; --x) relax(y * nx + x, x + 1, y), relax(y * nx + x, x, y + 1), relax(y * nx + x, x + 1, y + 1), relax(y * nx + x, x - 1, y + 1);
                for (int x = 0; x < nx; ++x) relax(y * nx + x, x - 1, y);
            }
            std::vector<int> seed((size_t)nx * ny, -1);
            for (int c = 0; c < nx * ny; ++c) {
                if (src[c] < 0) continue;                                          // (no sample at all)
                const double m1 = lo1 + (c % nx + 0.5) * cell, m2 = lo2 + (c / nx + 0.5) * cell;
                double best = INFINITY;
                int pick = start[src[c]];
                for (int i = start[src[c]]; i < start[src[c] + 1]; ++i) {
                    const double d1 = rec[(size_t)i * 4 + d.a1] - m1, d2 = rec[(size_t)i * 4 + d.a2] - m2;
                    if (d1 * d1 + d2 * d2 < best) best = d1 * d1 + d2 * d2, pick = i;
                }
                seed[c] = pick;
            }
            UP(fg_seed, seed.data(), seed.size());
        }
        UP(fg_rec, rec.data(), rec.size());
        {
            std::vector<float> rec32((real.size() + 4) * 4, 0.0f);       // (four records of padding: the scan reads ahead)
            for (size_t j = 0; j < real.size(); ++j) {
                for (int k = 0; k < 3; ++k) rec32[j * 4 + k] = (float)rec[j * 4 + k];
                int32_t pair[2];
                std::memcpy(pair, &rec[j * 4 + 3], sizeof pair);
                std::memcpy(&rec32[j * 4 + 3], &pair[1], sizeof(int32_t));
            }
            UP(fg_rec32, rec32.data(), rec32.size());
        }
        {   // box pyramid (PartDev::py_*): bounding boxes of the samples of every cell, then of 2 x 2 nodes, up to one node
            d.py_levels = 0;
            std::vector<float> box;
            std::vector<int> lnx, lny, loff;
            int cnx = nx, cny = ny;
            for (int l = 0; l < PY_MAX_LEVELS; ++l) {
                lnx.push_back(cnx), lny.push_back(cny), loff.push_back((int)(box.size() / 8));
                box.resize(box.size() + (size_t)cnx * cny * 8);
                float *lv = box.data() + (size_t)loff[l] * 8;
                for (int c = 0; c < cnx * cny; ++c) {
                    float *b = lv + (size_t)c * 8;
                    b[0] = b[1] = b[2] = INFINITY, b[4] = b[5] = b[6] = -INFINITY, b[3] = b[7] = 0.0f;
                    if (l == 0) {                    // the spare floats of a cell's box: its record range (int bits)
                        std::memcpy(&b[3], &start[c], sizeof(int32_t));
                        std::memcpy(&b[7], &start[c + 1], sizeof(int32_t));
                    }
                }
                if (l == 0) {
                    for (size_t j = 0; j < real.size(); ++j) {
                        float *b = lv + (size_t)cell_of[j] * 8;
                        for (int k = 0; k < 3; ++k) {
                            const double v = t->sample_xyz[k][real[j]];
                            b[k] = std::fmin(b[k], std::nextafterf((float)v, -INFINITY));
                            b[4 + k] = std::fmax(b[4 + k], std::nextafterf((float)v, INFINITY));
                        }
                    }
                } else {
                    const float *pv = box.data() + (size_t)loff[l - 1] * 8;
                    const int pnx = lnx[l - 1], pny = lny[l - 1];
                    for (int cy = 0; cy < cny; ++cy)
                        for (int cx = 0; cx < cnx; ++cx) {
                            float *b = lv + ((size_t)cy * cnx + cx) * 8;
                            for (int sy = 0; sy < 2; ++sy)
                                for (int sx = 0; sx < 2; ++sx) {
                                    const int px = 2 * cx + sx, py = 2 * cy + sy;
                                    if (px >= pnx || py >= pny) continue;
                                    const float *q = pv + ((size_t)py * pnx + px) * 8;
                                    for (int k = 0; k < 3; ++k) b[k] = std::fmin(b[k], q[k]), b[4 + k] = std::fmax(b[4 + k], q[4 + k]);
                                }
                        }
                }
                d.py_levels = l + 1;
                if (cnx == 1 && cny == 1) break;
                cnx = (cnx + 1) / 2, cny = (cny + 1) / 2;
            }
            if (lnx.back() != 1 || lny.back() != 1) d.py_levels = 0;      // (a grid beyond 4096 cells a side: no pyramid)
            for (int l = 0; l < PY_MAX_LEVELS; ++l) {
                d.py_off[l] = l < d.py_levels ? loff[l] : 0;
                d.py_nx[l] = l < d.py_levels ? lnx[l] : 0;
                d.py_ny[l] = l < d.py_levels ? lny[l] : 0;
            }
            UP(py_box, box.data(), box.size());
        }
    }
    {   // outline of the collision set in the principal plane (Andrew's monotone chain over the projected corners)
        std::vector<std::pair<double, double>> pts;
        double zlo = INFINITY, zhi = -INFINITY;
        for (int i = 0; i < t->n_collision_pad; ++i) {
            if (t->col_rank[i] == 0x7fffffff) continue;                 // pad
            for (int c = 0; c < 3; ++c) {
                double v[3];
                for (int k = 0; k < 3; ++k)
                    v[k] = t->col_v0e1e2[k][i] + (c == 1 ? t->col_v0e1e2[3 + k][i] : (c == 2 ? t->col_v0e1e2[6 + k][i] : 0.0));
                pts.emplace_back(v[d.a1], v[d.a2]);
                zlo = std::fmin(zlo, v[d.a0]);
                zhi = std::fmax(zhi, v[d.a0]);
            }
        }
        std::sort(pts.begin(), pts.end());
        pts.erase(std::unique(pts.begin(), pts.end()), pts.end());
        std::vector<std::pair<double, double>> hull(2 * pts.size() + 2);
        size_t k = 0;
        auto cross = [](const std::pair<double, double> &o, const std::pair<double, double> &a, const std::pair<double, double> &b) {
            return (a.first - o.first) * (b.second - o.second) - (a.second - o.second) * (b.first - o.first);
        };
        for (size_t i = 0; i < pts.size(); ++i) {
            while (k >= 2 && cross(hull[k - 2], hull[k - 1], pts[i]) <= 0) --k;
            hull[k++] = pts[i];
        }
        for (size_t i = pts.size() - 1, lower = k + 1; i-- > 0;) {
            while (k >= lower && cross(hull[k - 2], hull[k - 1], pts[i]) <= 0) --k;
            hull[k++] = pts[i];
        }
        const size_t nh = k > 1 ? k - 1 : 0;                            // counter-clockwise, closed
        std::vector<double> out;
        int n_edges = 0;
        if (nh >= 3 && nh <= 1024) {
            for (size_t i = 0; i < nh; ++i) {
                const auto &p0 = hull[i], &p1 = hull[(i + 1) % nh];
                const double ex = p1.first - p0.first, ey = p1.second - p0.second, len = std::sqrt(ex * ex + ey * ey);
                if (!(len > 0)) continue;
                out.insert(out.end(), {p0.first, p0.second, ey / len, -ex / len});      // outward of a CCW polygon
                ++n_edges;
            }
            // every projected corner must lie on the inner side (the test below trusts that)
            for (const auto &q : pts)
                for (int e = 0; e < n_edges; ++e)
                    if (out[4 * e + 2] * (q.first - out[4 * e]) + out[4 * e + 3] * (q.second - out[4 * e + 1]) > 1e-9) n_edges = -1 - n_edges;
            if (n_edges < 0) n_edges = 0;
        }
        d.n_outline = n_edges;
        d.slab_lo = zlo;
        d.slab_hi = zhi;
        out.resize((size_t)((n_edges + 63) / 64 + 1) * 64 * 4, 0.0);    // pads: zero normal, never separating
        UP(outline, out.data(), out.size());
    }
    d.n_start = t->n_start;
    if (d.n_start <= 0) return fail(PRL_E_INVALID, "part has no start points");
    UP(start_pos, t->start_pos, (size_t)d.n_start * 3);
    UP(start_quat, t->start_quat, (size_t)d.n_start * 4);
    d.n_beams = t->n_beams;
    if (d.n_beams > 0) UP(beams, t->beams, (size_t)d.n_beams * 3);
    return PRL_OK;
}

int check_config(const PrlConfig *c) {
    if (c->obs_mode < 0 || c->obs_mode > 3) return fail(PRL_E_INVALID, "obs_mode %d", c->obs_mode);
    if ((c->obs_mode == PRL_OBS_SECTION || c->obs_mode == PRL_OBS_DISCRETE) && (c->obs_grad < 1 || c->obs_grad > 62))
        return fail(PRL_E_UNSUPPORTED, "section/discrete observation: OBS_GRAD must be 1..62");
    if (c->obs_mode == PRL_OBS_GRID && (c->obs_grad < 1 || c->obs_grad > 16))
        return fail(PRL_E_UNSUPPORTED, "grid observation: OBS_GRAD must be 1..16");
    if (c->action_mode == PRL_ACT_DISCRETE) {
        if (c->n_discrete < 1 || c->n_discrete > PRL_MAX_DISCRETE) return fail(PRL_E_INVALID, "n_discrete %d", c->n_discrete);
    } else if (c->action_mode == PRL_ACT_CONTINUOUS) {
        if (c->action_dim < 1 || c->action_dim > 2) return fail(PRL_E_INVALID, "action_dim %d", c->action_dim);
    } else {
        return fail(PRL_E_INVALID, "action_mode %d", c->action_mode);
    }
    if (c->termination_mode < 0 || c->termination_mode > 2) return fail(PRL_E_INVALID, "termination_mode");
    if (c->paint_method != PRL_PAINT_FAST && c->paint_method != PRL_PAINT_NORMAL) return fail(PRL_E_INVALID, "paint_method");
    if (c->color_mode != PRL_COLOR_RGB && c->color_mode != PRL_COLOR_HSI) return fail(PRL_E_INVALID, "color_mode");
    if (c->max_episode_len < 1 || c->expected_episode_len < 1) return fail(PRL_E_INVALID, "episode lengths");
    if (!(c->paint_radius > 0) || !(c->step_size > 0)) return fail(PRL_E_INVALID, "paint_radius and step_size must be positive");
    return PRL_OK;
}

bool general_section(const PrlConfig &c) {
    return (c.obs_mode == PRL_OBS_SECTION || c.obs_mode == PRL_OBS_DISCRETE) && c.obs_grad != 4;
}

// Launches go to the CALLER's current device and stream (one process per GPU): a batch used while another device is
// current would launch there on pointers of this one.  Cheap to check (thread-local), so every entry point does.
int check_device(const PrlBatch *b) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != b->device)
        return fail(PRL_E_INVALID, "the batch lives on device %d but the calling thread's current device is %d "
                                   "(hipSetDevice / torch.cuda.set_device first)", b->device, cur);
    return PRL_OK;
}

StepArgs base_args(PrlBatch *b) {
    StepArgs a{};
    a.parts = b->parts_dev;
    a.cfg = b->cfg_dev;
    a.env_part = b->env_part_dev;
    a.slot_env = b->slot_env_dev;
    a.n_envs = b->n_envs;
    a.mask_stride = b->mask_stride;
    a.painted = b->painted;
    a.last = b->last;
    a.last_nz = b->last_nz;
    a.nz_stride = b->nz_stride;
    a.thick = b->thick;
    a.state = b->state;
    a.cone_shots = b->cone_shots;
    a.cone_aux = b->cone_aux;
    a.cone_hits = b->cone_hits;
    a.cone_work = b->cone_work;
    a.cone_far = b->cone_far;
    a.cone_nb = b->cone_nb;
    return a;
}

// kernel-unit dispatch by mask width (prl_launch.hpp): kw = 1..4 register-resident masks, anything larger the LDS-mask unit
#define PRL_KW_SWITCH(kw, call_suffix)                  \
    ((kw) == 1   ? prl_k1_##call_suffix                  \
     : (kw) == 2 ? prl_k2_##call_suffix                  \
     : (kw) == 3 ? prl_k3_##call_suffix                  \
     : (kw) == 4 ? prl_k4_##call_suffix                  \
                 : prl_k0_##call_suffix)

int launch_failed(int hip_error, const char *what) {
    return fail(PRL_E_HIP, "%s: %s", what, hipGetErrorString(static_cast<hipError_t>(hip_error)));
}

PrlStepSel step_sel(const PrlBatch *b) {
    PrlStepSel sel{};
    sel.gensec = general_section(b->cfg) ? 1 : 0;
    sel.hsi = b->cfg.color_mode == PRL_COLOR_HSI ? 1 : 0;
    sel.kd = b->kd ? 1 : 0;
    sel.wide = b->n_envs <= b->resident_envs ? 1 : 0;
    sel.grid = b->cfg.obs_mode == PRL_OBS_GRID ? 1 : 0;
    return sel;
}

}  // namespace

// ================================================================= C ABI
// error slot shared with the other translation units of the library (policy_mlp.hip); not exported
extern "C" __attribute__((visibility("hidden"))) int prl_set_error_(int code, const char *msg) { return fail(code, "%s", msg); }

extern "C" {

int prl_abi_version(void) { return PRL_ABI_VERSION; }
const char *prl_last_error(void) { return g_error; }

int prl_obs_dim(const PrlConfig *cfg) {
    if (!cfg) return fail(PRL_E_INVALID, "null config");
    return obs_dim_of(cfg->obs_mode, cfg->obs_grad);
}

int prl_struct_sizes(int *config_bytes, int *part_tables_bytes) {
    if (config_bytes) *config_bytes = (int)sizeof(PrlConfig);
    if (part_tables_bytes) *part_tables_bytes = (int)sizeof(PrlPartTables);
    return PRL_OK;
}

int prl_part_create(const PrlPartTables *t, int device, PrlPart **out) {
    if (!t || !out) return fail(PRL_E_INVALID, "null argument");
    *out = nullptr;
    HIP_TRY(hipSetDevice(device));
    PrlPart *p = new (std::nothrow) PrlPart();
    if (!p) return fail(PRL_E_NOMEM, "out of host memory");
    p->device = device;
    int rc = part_fill(p, t);
    if (rc == PRL_OK) {
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p->dev_copy), sizeof(PartDev));
        if (e == hipSuccess) e = hipMemcpy(p->dev_copy, &p->dev, sizeof(PartDev), hipMemcpyHostToDevice);
        if (e != hipSuccess) rc = fail(PRL_E_HIP, "part descriptor upload: %s", hipGetErrorString(e));
    }
    if (rc != PRL_OK) {
        prl_part_destroy(p);
        return rc;
    }
    *out = p;
    return PRL_OK;
}

void prl_part_destroy(PrlPart *p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    for (void *d : p->allocs) (void)hipFree(d);
    if (p->dev_copy) (void)hipFree(p->dev_copy);
    delete p;
}

int prl_part_mask_words(const PrlPart *p) { return p ? p->dev.n_words : fail(PRL_E_INVALID, "null part"); }

int prl_batch_create(PrlPart *const *parts, int n_parts, const int32_t *env_part_id, int n_envs, const PrlConfig *cfg,
                     PrlBatch **out) {
    if (!parts || !cfg || !out || n_parts < 1 || n_parts > 8 || n_envs < 1)
        return fail(PRL_E_INVALID, "bad arguments (n_parts 1..8, n_envs >= 1)");
    *out = nullptr;
    int rc = check_config(cfg);
    if (rc) return rc;
    for (int i = 0; i < n_parts; ++i) {
        if (!parts[i]) return fail(PRL_E_INVALID, "null part %d", i);
        if (parts[i]->device != parts[0]->device) return fail(PRL_E_INVALID, "parts live on different devices");
        if (cfg->paint_radius * parts[i]->dev.sg_inv >= 1.0)
            return fail(PRL_E_INVALID, "part %d: sample grid cell %.4f does not exceed the paint radius %.4f", i,
                        1.0 / parts[i]->dev.sg_inv, cfg->paint_radius);
        if (cfg->paint_method == PRL_PAINT_NORMAL && parts[i]->dev.n_beams <= 0)
            return fail(PRL_E_INVALID, "part %d has no cone beams but PAINT_METHOD='normal' was requested", i);
        if (cfg->obs_mode == PRL_OBS_GRID && parts[i]->dev.n_obs_cells != cfg->obs_grad * cfg->obs_grad)
            return fail(PRL_E_INVALID, "part %d was packed for %d observation cells, config wants %d", i,
                        parts[i]->dev.n_obs_cells, cfg->obs_grad * cfg->obs_grad);
    }
    if (env_part_id)
        for (int i = 0; i < n_envs; ++i)
            if (env_part_id[i] < 0 || env_part_id[i] >= n_parts) return fail(PRL_E_INVALID, "env_part_id[%d] out of range", i);
    PrlBatch *b = new (std::nothrow) PrlBatch();
    if (!b) return fail(PRL_E_NOMEM, "out of host memory");
    b->device = parts[0]->device;
    b->n_envs = n_envs;
    b->n_parts = n_parts;
    b->cfg = *cfg;
    for (int i = 0; i < n_parts; ++i)
        if (parts[i]->dev.n_words > b->mask_stride) b->mask_stride = parts[i]->dev.n_words;
    b->kw = (b->mask_stride + 63) / 64;
    for (int i = 0; i < n_parts; ++i) b->kd = b->kd || parts[i]->dev.n_kd_nodes > 0;
    for (int i = 0; i < n_parts; ++i) b->max_beams = std::max(b->max_beams, parts[i]->dev.n_beams);
    hipError_t e = hipSetDevice(b->device);
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, b->device) == hipSuccess) b->resident_envs = 16 * cus;
    }
    std::vector<PartDev> pd(n_parts);
    const int od = obs_dim_of(cfg->obs_mode, cfg->obs_grad);
    // (the rollout entry points allocate nothing when they run: a fragment can be captured into a graph)
    if (e == hipSuccess && cfg->auto_reset && cfg->action_mode == PRL_ACT_DISCRETE)
        e = hipMalloc(reinterpret_cast<void **>(&b->scratch_action), sizeof(int32_t) * (size_t)n_envs);
    b->reset_obs.assign(n_parts, nullptr);
    for (int i = 0; i < n_parts; ++i) {
        pd[i] = parts[i]->dev;
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->reset_obs[i]), sizeof(double) * (size_t)pd[i].n_start * od);
        pd[i].reset_obs = (gdouble_p)b->reset_obs[i];
    }
    const size_t mask_bytes = (size_t)n_envs * b->mask_stride * sizeof(uint64_t);
    const size_t state_bytes = (size_t)n_envs * PRL_STATE_DOUBLES * sizeof(double);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->parts_dev), sizeof(PartDev) * n_parts);
    if (e == hipSuccess) e = hipMemcpy(b->parts_dev, pd.data(), sizeof(PartDev) * n_parts, hipMemcpyHostToDevice);
    CfgDev cd{};
    static_cast<PrlConfig &>(cd) = *cfg;
    for (int k = 0; k < PRL_MAX_DISCRETE; ++k) {
        cd.act_d1[k] = cfg->act_delta1[k] / PAINT_PER_ACTION;
        cd.act_d2[k] = cfg->act_delta2[k] / PAINT_PER_ACTION;
    }
    for (int k = 0; k < 8; ++k) {
        cd.expected_reward[k] = cfg->max_possible_point[k] / (cfg->expected_episode_len * 100);
        cd.switch_points[k] = cfg->switch_threshold * cfg->max_possible_point[k] / 100;
    }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->cfg_dev), sizeof(CfgDev));
    if (e == hipSuccess) e = hipMemcpy(b->cfg_dev, &cd, sizeof(CfgDev), hipMemcpyHostToDevice);
    if (e == hipSuccess && env_part_id) {
        e = hipMalloc(reinterpret_cast<void **>(&b->env_part_dev), sizeof(int) * n_envs);
        if (e == hipSuccess) e = hipMemcpy(b->env_part_dev, env_part_id, sizeof(int) * n_envs, hipMemcpyHostToDevice);
#ifndef PRL_NO_XCD_PARTS                             // (A/B switch)
        if (e == hipSuccess && n_parts > 1) {
            // XCD-aware placement of a mixed batch.  Workgroup b of a launch runs on XCD b % 8, and every XCD has its own 4 MB L2:
            // with the envs of all parts spread evenly over the workgroups each L2 has to hold every part's tables (door + sheet:
            // 5 MB) and the waves' dependent look-ups miss.  The wave slots of the workgroups of one XCD are therefore handed
            // consecutive envs of the batch SORTED BY PART: an XCD then works on one part (two at a seam).  Results do not depend
            // on which wave steps an env.
            constexpr int N_XCD = 8;
            const int waves = n_envs <= b->resident_envs ? STEP_WAVES_WIDE : STEP_WAVES_NARROW;      // (step_sel: the launch shape of this batch)
            const int n_wg = (n_envs + waves - 1) / waves;
            std::vector<int> sorted((size_t)n_envs), slot_env((size_t)n_envs, 0);
            for (int i = 0; i < n_envs; ++i) sorted[(size_t)i] = i;
            std::stable_sort(sorted.begin(), sorted.end(), [&](int x, int y) { return env_part_id[x] < env_part_id[y]; });
            size_t next = 0;
            for (int x = 0; x < N_XCD; ++x)
                for (int wg = x; wg < n_wg; wg += N_XCD)
                    for (int w = 0; w < waves; ++w) {
                        const int slot = wg * waves + w;
                        if (slot < n_envs) slot_env[(size_t)slot] = sorted[next++];
                    }
            e = hipMalloc(reinterpret_cast<void **>(&b->slot_env_dev), sizeof(int) * n_envs);
            if (e == hipSuccess) e = hipMemcpy(b->slot_env_dev, slot_env.data(), sizeof(int) * n_envs, hipMemcpyHostToDevice);
        }
#endif
    }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->painted), mask_bytes);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->last), mask_bytes);
    b->nz_stride = b->kw <= KW_MAX ? KW_MAX : (b->mask_stride + 63) / 64;
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->last_nz), sizeof(uint64_t) * b->nz_stride * (size_t)n_envs);
    if (e == hipSuccess) e = hipMemset(b->last_nz, 0, sizeof(uint64_t) * b->nz_stride * (size_t)n_envs);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->state), state_bytes);
    if (e == hipSuccess && cfg->color_mode == PRL_COLOR_HSI) {
        e = hipMalloc(reinterpret_cast<void **>(&b->thick), mask_bytes * 8);        // one byte per sample
        if (e == hipSuccess) e = hipMemset(b->thick, 255, mask_bytes * 8);
    }
    if (e == hipSuccess && cfg->paint_method == PRL_PAINT_NORMAL) {
        // what the cone-beam kernels of a step hand to each other (StepArgs, k_cone_beams.hip)
        b->cone_nb = ((b->max_beams + 63) / 64) * 64;
        const size_t items = (size_t)n_envs * PAINT_PER_ACTION * (b->cone_nb / 64);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->cone_shots), sizeof(double) * 8 * PAINT_PER_ACTION * n_envs);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->cone_aux), sizeof(double) * 2 * n_envs);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->cone_hits), sizeof(int) * PAINT_PER_ACTION * (size_t)b->cone_nb * n_envs);
        // counters, the trip list (every trip fits), the ray sub-lists and the sub-lists' counters: k_cone_beams.hip
        if (items >= ((size_t)1 << 25)) {
            rc = fail(PRL_E_INVALID, "%zu beam trips per step: more than the cone-beam work lists index", items);
            prl_batch_destroy(b);
            return rc;
        }
        const size_t work_ints = 4 + items + 2 * (size_t)PRL_CONE_WORK_LISTS * prl_cone_ray_sub_cap((int)items) + 2 * 16 * PRL_CONE_WORK_LISTS;
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->cone_work), sizeof(int) * work_ints);
        if (e == hipSuccess) e = hipMemset(b->cone_work, 0, sizeof(int) * work_ints);
        // hit points handed to the far search, per env and step over all sub-lists: a quarter of the step's beams, at least 128
        // (the door: 33 of 520 beams on average; a 70 654-sample part casts 3 860 beams a step and hands over 132 -- with 128 a
        // step the sub-lists were full every step and whole trips went through the general code: 3.2 ms a step, round 5), a
        // sub-list at least one trip's worth
        const size_t far_per_env = std::max<size_t>(128, (size_t)PAINT_PER_ACTION * b->cone_nb / 4);
        const int far_cap = (int)std::min<size_t>(std::max<size_t>((size_t)n_envs * far_per_env / PRL_CONE_WORK_LISTS, 64), (size_t)1 << 18);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->cone_far), sizeof(double) * 4 * (size_t)far_cap * PRL_CONE_WORK_LISTS);
        if (e == hipSuccess) e = hipMemcpy(b->cone_work + 2, &far_cap, sizeof(int), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMemset(b->painted, 0, mask_bytes);
    if (e == hipSuccess) e = hipMemset(b->last, 0, mask_bytes);
    if (e == hipSuccess) e = hipMemset(b->state, 0, state_bytes);
    // the observation a reset to each start point returns (PartDev::reset_obs), once per part
    for (int i = 0; i < n_parts && e == hipSuccess; ++i) {
        const int words = pd[i].n_words, kw_i = words > 64 * KW_MAX ? 0 : (words + 63) / 64;
        e = static_cast<hipError_t>(PRL_KW_SWITCH(kw_i, reset_obs)(
            b->parts_dev + i, b->cfg_dev, b->reset_obs[i], pd[i].n_start, words, general_section(*cfg) ? 1 : 0));
        if (e == hipSuccess) e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        rc = fail(PRL_E_HIP, "batch allocation: %s", hipGetErrorString(e));
        prl_batch_destroy(b);
        return rc;
    }
    *out = b;
    return PRL_OK;
}

void prl_batch_destroy(PrlBatch *b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    for (hipEvent_t ev : b->ev_start) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : b->ev_stop) (void)hipEventDestroy(ev);
    (void)hipFree(b->parts_dev);
    (void)hipFree(b->cfg_dev);
    (void)hipFree(b->env_part_dev);
    (void)hipFree(b->slot_env_dev);
    (void)hipFree(b->painted);
    (void)hipFree(b->last);
    (void)hipFree(b->last_nz);
    (void)hipFree(b->thick);
    (void)hipFree(b->cone_shots);
    (void)hipFree(b->cone_aux);
    (void)hipFree(b->cone_hits);
    (void)hipFree(b->cone_work);
    (void)hipFree(b->cone_far);
    (void)hipFree(b->scratch_action);
    for (double *p : b->reset_obs) (void)hipFree(p);
    (void)hipFree(b->state);
    delete b;
}

int prl_batch_mask_stride(const PrlBatch *b) { return b ? b->mask_stride : fail(PRL_E_INVALID, "null batch"); }

int prl_batch_reset(PrlBatch *b, const uint8_t *reset_mask, const int32_t *start_idx, double *obs, void *stream) {
    if (!b) return fail(PRL_E_INVALID, "null batch");
    if (int rc = check_device(b)) return rc;
    StepArgs a = base_args(b);
    a.reset_mask = reset_mask;
    a.start_idx = start_idx;
    a.obs = obs;
    if (int e = PRL_KW_SWITCH(b->kw, reset)(&a, general_section(b->cfg) ? 1 : 0, stream)) return launch_failed(e, "prl_batch_reset");
    return PRL_OK;
}

int prl_batch_observe(PrlBatch *b, double *obs, void *stream) {
    if (!b || !obs) return fail(PRL_E_INVALID, "null argument");
    if (int rc = check_device(b)) return rc;
    StepArgs a = base_args(b);
    a.obs = obs;
    if (int e = PRL_KW_SWITCH(b->kw, observe)(&a, general_section(b->cfg) ? 1 : 0, stream)) return launch_failed(e, "prl_batch_observe");
    return PRL_OK;
}

int prl_batch_step(PrlBatch *b, const void *actions, double *obs, double *reward, uint8_t *done, double *info,
                   double *final_obs, const int32_t *start_idx, void *stream) {
    if (!b || !actions || !obs || !reward || !done || !info) return fail(PRL_E_INVALID, "null argument");
    if (int rc = check_device(b)) return rc;
    StepArgs a = base_args(b);
    a.actions = actions;
    a.obs = obs;
    a.reward = reward;
    a.done = done;
    a.info = info;
    a.final_obs = final_obs;
    a.start_idx = start_idx;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool normal = b->cfg.paint_method == PRL_PAINT_NORMAL;
    const bool timed = b->timing_every > 0 && (b->launch_no++ % b->timing_every) == 0;
    if (timed) {
        if (b->ev_used == b->ev_start.size()) {
            hipEvent_t e0, e1;
            HIP_TRY(hipEventCreate(&e0));
            HIP_TRY(hipEventCreate(&e1));
            b->ev_start.push_back(e0);
            b->ev_stop.push_back(e1);
        }
        HIP_TRY(hipEventRecord(b->ev_start[b->ev_used], s));
    }
    const PrlStepSel sel = step_sel(b);
    int e;
    if (normal) {                                  // tool path, beams, the beams' leftovers, fold + finish (k_cone_beams.hip)
        e = prl_kc_path(&a, sel.kd, sel.wide, stream);
        if (!e) e = prl_kc_beams(&a, stream);
        if (!e) e = PRL_KW_SWITCH(b->kw, cone)(&a, &sel, stream);
    } else {
        e = PRL_KW_SWITCH(b->kw, step)(&a, &sel, stream);
    }
    if (e) return launch_failed(e, "prl_batch_step");
    if (timed) {
        HIP_TRY(hipEventRecord(b->ev_stop[b->ev_used], s));
        b->ev_used += 1;
    }
    return PRL_OK;
}

int prl_batch_step_occupancy(PrlBatch *b, int32_t *out) {
    if (!b || !out) return fail(PRL_E_INVALID, "null argument");
    if (int rc = check_device(b)) return rc;
    const StepArgs a = base_args(b);
    const PrlStepSel sel = step_sel(b);
    int o[3] = {0, 0, 0};
    if (int e = PRL_KW_SWITCH(b->kw, step_occupancy)(&a, &sel, o)) return launch_failed(e, "prl_batch_step_occupancy");
    int cus = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, b->device));
    out[0] = o[0];
    out[1] = o[1];
    out[2] = o[2];
    out[3] = cus;
    return PRL_OK;
}

int prl_batch_set_pose(PrlBatch *b, int env_index, const double *pos, const double *quat) {
    if (!b || !pos || !quat || env_index < 0 || env_index >= b->n_envs) return fail(PRL_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipDeviceSynchronize());
    EnvState S;
    double *rec = b->state + (size_t)env_index * PRL_STATE_DOUBLES;
    HIP_TRY(hipMemcpy(&S, rec, sizeof S, hipMemcpyDeviceToHost));
    for (int k = 0; k < 3; ++k) S.pose[k] = pos[k];
    for (int k = 0; k < 4; ++k) S.quat[k] = quat[k];
    S.terminate = 0;
    S.terminate_counter = 0;
    S.last_on_part = 1;
    S.last_angle = 0;
    S.facet_hint = -1;
    HIP_TRY(hipMemcpy(rec, &S, sizeof S, hipMemcpyHostToDevice));
    return PRL_OK;
}

int prl_batch_get_mask(PrlBatch *b, uint64_t *painted, void *stream) {
    if (!b || !painted) return fail(PRL_E_INVALID, "null argument");
    const size_t n = (size_t)b->n_envs * b->mask_stride;
    hipLaunchKernelGGL(copy_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), b->painted, painted, n);
    HIP_TRY(hipGetLastError());
    return PRL_OK;
}

int prl_batch_get_last_mask(PrlBatch *b, uint64_t *last, uint64_t *nonzero_words, int32_t *nonzero_stride, void *stream) {
    if (!b || !last) return fail(PRL_E_INVALID, "null argument");
    const size_t n = (size_t)b->n_envs * b->mask_stride;
    hipLaunchKernelGGL(copy_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), b->last, last, n);
    HIP_TRY(hipGetLastError());
    if (nonzero_words)
        HIP_TRY(hipMemcpyAsync(nonzero_words, b->last_nz, sizeof(uint64_t) * (size_t)b->n_envs * b->nz_stride, hipMemcpyDeviceToDevice,
                               static_cast<hipStream_t>(stream)));
    if (nonzero_stride) *nonzero_stride = b->nz_stride;
    return PRL_OK;
}

int prl_batch_get_thickness(PrlBatch *b, uint8_t *thick, void *stream) {
    if (!b || !thick) return fail(PRL_E_INVALID, "null argument");
    if (!b->thick) return fail(PRL_E_INVALID, "prl_batch_get_thickness: the batch was not created with COLOR_MODE 'HSI'");
    HIP_TRY(hipMemcpyAsync(thick, b->thick, (size_t)b->n_envs * b->mask_stride * 64, hipMemcpyDeviceToDevice,
                           static_cast<hipStream_t>(stream)));
    return PRL_OK;
}

int prl_batch_get_state(PrlBatch *b, double *state, void *stream) {
    if (!b || !state) return fail(PRL_E_INVALID, "null argument");
    HIP_TRY(hipMemcpyAsync(state, b->state, (size_t)b->n_envs * PRL_STATE_DOUBLES * sizeof(double),
                           hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    return PRL_OK;
}

int prl_batch_get_returns(PrlBatch *b, double *episode_return, void *stream) {
    if (!b || !episode_return) return fail(PRL_E_INVALID, "null argument");
    hipLaunchKernelGGL(gather_state_kernel, dim3((b->n_envs + 255) / 256), dim3(256), 0,
                       static_cast<hipStream_t>(stream), b->state, b->n_envs, 13, episode_return);
    HIP_TRY(hipGetLastError());
    return PRL_OK;
}

int prl_ray_batch(PrlPart *p, int n, const double *from, const double *to, int32_t *tri, double *frac, double *pos,
                  void *stream) {
    if (!p || n < 0 || !from || !to || !tri || !frac || !pos) return fail(PRL_E_INVALID, "bad argument");
    if (n == 0) return PRL_OK;
    hipLaunchKernelGGL(ray_batch_kernel, dim3((n + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream),
                       p->dev_copy, n, from, to, tri, frac, pos);
    HIP_TRY(hipGetLastError());
    return PRL_OK;
}

// what both rollout entry points ask of the batch (the fast-path instantiation of the step: prl_step.hpp)
static int check_rollout_batch(PrlBatch *b, const char *who) {
    if (int rc = check_device(b)) return rc;
    const PrlConfig &c = b->cfg;
    if (!c.auto_reset) return fail(PRL_E_INVALID, "%s: the batch must be created with auto_reset", who);
    if (c.action_mode != PRL_ACT_DISCRETE) return fail(PRL_E_UNSUPPORTED, "%s: discrete actions", who);
    return PRL_OK;
}

// The fused rollout kernels (k_rollout.hip) are built around the ball painter's step (masks in registers, or for parts beyond
// 16 384 samples in HBM; COLOR_MODE 'RGB' or 'HSI') and the 4-sector / grid observation; the other configurations (cone beams,
// atan2 sectors) take the same entry points as their definition reads: prl_policy_act + prl_batch_step, launch by launch.
static bool fused_rollout(const PrlBatch *b) {
    const PrlConfig &c = b->cfg;
#ifdef PRL_ROLLOUT_NO_HSI                  // (variant libraries built without the thickness rollout kernels)
    if (c.color_mode != PRL_COLOR_RGB) return false;
#endif
    return c.paint_method == PRL_PAINT_FAST && !general_section(c);
}

// which build of a fused rollout kernel a batch takes (k_rollout.hip): bit 0 the stale kd-tree, bit 1 OBS_MODE 'grid',
// bit 2 COLOR_MODE 'HSI'
static int rollout_flags(const PrlBatch *b) {
    return (b->kd ? 1 : 0) | (b->cfg.obs_mode == PRL_OBS_GRID ? 2 : 0) | (b->cfg.color_mode == PRL_COLOR_HSI ? 4 : 0);
}

static int check_policy(const PrlBatch *b, const PrlPolicyWeights *w, const char *who) {
    const PrlConfig &c = b->cfg;
    if (!w->w1 || !w->b1 || !w->w2 || !w->b2 || !w->w3 || !w->b3) return fail(PRL_E_INVALID, "%s: null weights", who);
    if (w->in_dim != obs_dim_of(c.obs_mode, c.obs_grad))
        return fail(PRL_E_INVALID, "%s: policy input %d, observation %d", who, w->in_dim, obs_dim_of(c.obs_mode, c.obs_grad));
    if (w->h1 < 16 || w->h1 % 16 || w->h2 < 16 || w->h2 % 16 || w->n_actions != c.n_discrete || w->n_actions > 15)
        return fail(PRL_E_UNSUPPORTED, "%s: hidden sizes multiples of 16, n_actions = n_discrete <= 15", who);
    if (sizeof(float) * (size_t)policy_lds_layout(*w).floats > 120 * 1024)
        return fail(PRL_E_UNSUPPORTED, "%s: layer sizes need more than 120 KB of LDS per 16 envs", who);
    if (reinterpret_cast<uintptr_t>(w->w2) % 16) return fail(PRL_E_INVALID, "%s: w2 must be 16-byte aligned", who);
    return PRL_OK;
}

int prl_batch_act_step(PrlBatch *b, const PrlPolicyWeights *w, const double *obs_in, uint32_t *rng_count, uint64_t rng_seed,
                       int32_t *action, float *logp, float *value, double *obs, double *reward, uint8_t *done, double *info,
                       double *final_obs, void *stream) {
    if (!b || !w || !obs_in || !rng_count || !action || !logp || !value || !obs || !reward || !done || !info)
        return fail(PRL_E_INVALID, "prl_batch_act_step: null argument");
    if (int rc = check_rollout_batch(b, "prl_batch_act_step")) return rc;
    if (int rc = check_policy(b, w, "prl_batch_act_step")) return rc;
    if (!fused_rollout(b)) {
        if (int rc = prl_policy_act(w, b->n_envs, obs_in, nullptr, rng_count, rng_seed, action, logp, value, nullptr, stream)) return rc;
        return prl_batch_step(b, action, obs, reward, done, info, final_obs, nullptr, stream);
    }
    ActStepArgs f{};
    f.s = base_args(b);
    f.s.obs = obs;
    f.s.reward = reward;
    f.s.done = done;
    f.s.info = info;
    f.s.final_obs = final_obs;
    f.w = *w;
    f.obs_in = obs_in;
    f.action = action;
    f.logp = logp;
    f.value = value;
    f.rng_count = rng_count;
    f.rng_seed = rng_seed;
    const size_t lds = sizeof(float) * (size_t)policy_lds_layout(*w).floats;
    if (int e = PRL_KW_SWITCH(b->kw, act_step)(&f, lds, rollout_flags(b), stream)) return launch_failed(e, "prl_batch_act_step");
    return PRL_OK;
}

int prl_rollout_fragment(PrlBatch *b, const PrlPolicyWeights *w, int n_steps, double *obs, double *final_obs,
                         double *reward, uint8_t *done, double *info, int32_t *action, float *logp, float *value,
                         float *last_value, uint32_t *rng_count, uint64_t rng_seed, void *stream) {
    if (!b || !obs || !reward || !done || !info || !action || n_steps < 1)
        return fail(PRL_E_INVALID, "prl_rollout_fragment: null argument or n_steps < 1");
    if (int rc = check_rollout_batch(b, "prl_rollout_fragment")) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (w && (!logp || !value || !last_value || !rng_count))
        return fail(PRL_E_INVALID, "prl_rollout_fragment: the policy needs logp, value, last_value and rng_count");
    if (w)
        if (int rc = check_policy(b, w, "prl_rollout_fragment")) return rc;
    if (!fused_rollout(b)) {
        // launch by launch (see fused_rollout): the rows are the same by definition
        const size_t n = (size_t)b->n_envs, od = (size_t)obs_dim_of(b->cfg.obs_mode, b->cfg.obs_grad);
        for (int t = 0; t < n_steps; ++t) {
            if (w)
                if (int rc = prl_policy_act(w, b->n_envs, obs + t * n * od, nullptr, rng_count, rng_seed, action + t * n, logp + t * n,
                                            value + t * n, nullptr, stream))
                    return rc;
            if (int rc = prl_batch_step(b, action + t * n, obs + (t + 1) * n * od, reward + t * n, done + t * n, info + 2 * t * n,
                                        final_obs ? final_obs + t * n * od : nullptr, nullptr, stream))
                return rc;
        }
        if (w) {                                   // the bootstrap value; its draw is discarded
            if (!b->scratch_action) return fail(PRL_E_INVALID, "prl_rollout_fragment: batch without the rollout scratch row");
            if (int rc = prl_policy_act(w, b->n_envs, obs + (size_t)n_steps * n * od, nullptr, rng_count, rng_seed, b->scratch_action, nullptr,
                                        last_value, nullptr, stream))
                return rc;
        }
        return PRL_OK;
    }
    if (w) {
        // policy: ONE persistent launch too (rollout_policy_kernel): the sixteen envs of a workgroup alternate policy and
        // step; after the last step the policy runs once more for the bootstrap value (its draw is discarded)
        if (!logp || !value || !last_value || !rng_count)
            return fail(PRL_E_INVALID, "prl_rollout_fragment: the policy needs logp, value, last_value and rng_count");
        if (int rc = check_policy(b, w, "prl_rollout_fragment")) return rc;
        {
            PolicyFragmentArgs g{};
            g.f.s = base_args(b);
            g.f.T = n_steps;
            g.f.obs = obs;
            g.f.final_obs = final_obs;
            g.f.reward = reward;
            g.f.done = done;
            g.f.info = info;
            g.f.action = action;
            g.w = *w;
            g.logp = logp;
            g.value = value;
            g.last_value = last_value;
            g.rng_count = rng_count;
            g.rng_seed = rng_seed;
            const size_t lds = sizeof(float) * (size_t)policy_lds_layout(*w).floats;
            if (int e = PRL_KW_SWITCH(b->kw, rollout_policy)(&g, lds, rollout_flags(b), stream)) return launch_failed(e, "prl_rollout_fragment");
            return PRL_OK;
        }
    }
    // given actions: ONE persistent launch, the waves never meet (rollout_fragment_kernel)
    FragmentArgs f{};
    f.s = base_args(b);
    f.T = n_steps;
    f.obs = obs;
    f.final_obs = final_obs;
    f.reward = reward;
    f.done = done;
    f.info = info;
    f.action = action;
    const size_t lds = b->kw > KW_MAX ? 0 : (size_t)POLICY_WAVES * 2 * b->mask_stride * sizeof(uint64_t);
    if (lds > 120 * 1024) return fail(PRL_E_UNSUPPORTED, "prl_rollout_fragment: %zu bytes of LDS per workgroup", lds);
    if (int e = PRL_KW_SWITCH(b->kw, rollout_fragment)(&f, rollout_flags(b), stream)) return launch_failed(e, "prl_rollout_fragment");
    return PRL_OK;
}

int prl_batch_timing_enable(PrlBatch *b, int enable) {
    if (!b) return fail(PRL_E_INVALID, "null batch");
    b->timing_every = enable < 0 ? 0 : enable;
    b->launch_no = 0;
    b->ev_used = 0;
    return PRL_OK;
}

int prl_batch_timing_read(PrlBatch *b, double *total_ms, int64_t *launches) {
    if (!b || !total_ms || !launches) return fail(PRL_E_INVALID, "null argument");
    double sum = 0;
    for (size_t i = 0; i < b->ev_used; ++i) {
        HIP_TRY(hipEventSynchronize(b->ev_stop[i]));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, b->ev_start[i], b->ev_stop[i]));
        sum += ms;
    }
    *total_ms = sum;
    *launches = (int64_t)b->ev_used;
    b->ev_used = 0;
    return PRL_OK;
}

}  // extern "C"
