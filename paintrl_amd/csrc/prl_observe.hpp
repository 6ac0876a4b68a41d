// prl_observe.hpp -- observations (rge:306-319, bpw:965-1139).
// Part of the single translation unit paintrl_hip.hip (device code, anonymous namespace); see that
// file for the overall design.  Compile with -ffp-contract=off.
#pragma once

namespace {

// ---------------------------------------------------------------- observation (rge:306-319)
__device__ __forceinline__ int grid_index_2(PartRef P, double val) {
    const double rel = (val - P.r2min) / (P.r2max - P.r2min);
    const double g = rel * GRID_GRANULARITY;
    int gi;
    if (!(g > -2147483648.0 && g < 2147483648.0)) gi = g > 0 ? GRID_GRANULARITY - 1 : 0;
    else gi = (int)g;
    return gi < 0 ? 0 : (gi > GRID_GRANULARITY - 1 ? GRID_GRANULARITY - 1 : gi);
}

__device__ __forceinline__ double clip01(double v) { return v < 0 ? 0.0 : (v > 1 ? 1.0 : v); }

__device__ __forceinline__ int handle_pos(double v) {        // rge:92-98
    if (v == 0) return 0;
    if (v == 1) return 21;
    return (int)(v * 20) + 1;
}

// CPython float_floor_div, the `//` of bpw:1030 (exact floor of the true quotient)
__device__ __forceinline__ double py_floor_div(double vx, double wx) {
    const double mod = fmod(vx, wx);
    double div = (vx - mod) / wx;
    if (mod != 0.0 && ((wx < 0) != (mod < 0))) div -= 1.0;
    if (div != 0.0) {
        double fl = floor(div);
        if (div - fl > 0.5) fl += 1.0;
        return fl;
    }
    return copysign(0.0, vx / wx);
}

// bpw:1026-1031, 1045-1061 with section != 4: every sample's sector is int(angle // (2 pi / g)) of its float64 atan2 angle about
// the tool.  Until round 5 exactly that per sample (atan2, Python's floor division, two LDS atomics on g counters: 284 us a
// step on the door, 1.9 ms at 70 654 samples -- first measured then).  Now a float atan2f decides wherever it can: its angle is
// within 1e-6 rad of the true one (inputs rounded to float, a few units in the last place of the function), a sector is at least
// 2 pi / 64 wide, so a sample whose float position inside its sector is more than SECTOR_BAND from both ends IS in that sector,
// whatever the float64 arithmetic rounds to; the others (one in five thousand: a word in eighty has one) take the reference's
// own expression.  The counters: one ballot a sector and word (g <= GENSEC_BALLOT_MAX), lane j keeps sector j's.
// `lds_painted`: the mask row of a large part (LDS copy or the env's row in HBM), or nullptr (then the register slots `painted` are used).
#define SECTOR_BAND 1.0e-4f
constexpr int GENSEC_BALLOT_MAX = 16;
template <int KW, typename PW>
__device__ void section_general_wave(PartRef P, int g, double x1, double x2, const uint64_t painted[KW_MAX],
                                     PW lds_painted, int lane, int *cnt /* LDS: [2][64] for this wave */,
                                     double *out) {
    gdouble_p sx = P.samp_a1, sy = P.samp_a2;
    const bool by_ballot = g <= GENSEC_BALLOT_MAX;
    if (!by_ballot) {
        cnt[lane] = 0;
        cnt[64 + lane] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // zeroes before the atomics of other lanes
        __builtin_amdgcn_wave_barrier();
    }
    const double two_pi = 2 * PI, basis = two_pi / g;
    const float two_pi_f = 6.2831855f, inv_basis_f = (float)g / 6.2831855f;
    int tot_j = 0, und_j = 0;                                       // lane j: sector j's counts (by_ballot)
    // g <= 8: this lane's own counts, sector i in the 16-bit field i & 3 of word i >> 2 (at most n_words <= 4 096 a field and lane),
    // summed over the wave at the end -- ten instructions a word instead of a ballot a sector
    const bool packed = g <= 8;
    uint64_t tot_p[2] = {0, 0}, und_p[2] = {0, 0};
    // the float sector of a point relative to the tool, or -1 where the float cannot be trusted (within SECTOR_BAND of a sector's end)
    auto sector_f = [&](float fx, float fy) -> int {
        float af = atan2f(fy, fx);
        if (af < 0) af = two_pi_f + af;
        const float pf = af * inv_basis_f, fl = floorf(pf), fr = pf - fl;
        const int i = (int)fl;
        return ((fr > SECTOR_BAND) & (fr < 1.0f - SECTOR_BAND) & (i >= 0) & (i < g)) ? i : -1;
    };
    const f64x4 GAS *wbox = reinterpret_cast<const f64x4 GAS *>(P.word_bbox);
    for (int w0 = 0; w0 < P.n_words; w0 += 64) {
        // One word per lane first: a word whose box lies inside ONE wedge (its four corners safely in the same sector; wedges of
        // g >= 2 sectors are convex, and the tool is not in the box then) is counted whole by popcount -- three words in four on a
        // 70 654-sample part, whose rays cross ~ 240 of 1 100 words.
        const int wl = w0 + lane;
        const bool inw = wl < P.n_words;
        const int wc = inw ? wl : w0;
        const f64x4 bb = ldg(wbox, wc);
        const uint64_t v_l = inw ? ldg(P.word_valid, wc) : 0;
        uint64_t p_l = 0;
        if (lds_painted) {
            p_l = lds_painted[wc];
        } else {
#pragma unroll
            for (int k = 0; k < KW; ++k)
                if (k == (w0 >> 6)) p_l = painted[k];
        }
        const float lx = (float)(bb.x - x1), hx = (float)(bb.y - x1), ly = (float)(bb.z - x2), hy = (float)(bb.w - x2);
        const int c0 = sector_f(lx, ly), c1 = sector_f(hx, ly), c2 = sector_f(lx, hy), c3 = sector_f(hx, hy);
        const bool whole = (g >= 2) & (c0 >= 0) & (c0 == c1) & (c0 == c2) & (c0 == c3) & (v_l != 0);
        if (whole) {
            const uint64_t nt = (uint64_t)__popcll(v_l), nu = (uint64_t)__popcll(v_l & ~p_l);
            if (packed) {
                const int sh = 16 * (c0 & 3);
                tot_p[c0 >> 2] += nt << sh;
                und_p[c0 >> 2] += nu << sh;
            } else if (!by_ballot) {
                atomicAdd(&cnt[c0], (int)nt);
                atomicAdd(&cnt[64 + c0], (int)nu);
            }
        }
        if (by_ballot && !packed) {                                   // (g = 9 .. 16: lane j keeps sector j's counts)
            for (int j = 0; j < g; ++j) {
                uint64_t m = ballot64(whole & (c0 == j));
                int st = 0, su = 0;
                while (m) {
                    const int L = __builtin_ctzll(m);
                    m &= m - 1;
                    const uint64_t vv = bcast_u64(v_l, L), pp = bcast_u64(p_l, L);
                    st += (int)__popcll(vv);
                    su += (int)__popcll(vv & ~pp);
                }
                if (lane == j) {
                    tot_j += st;
                    und_j += su;
                }
            }
        }
        // the other words sample by sample
        uint64_t mixed = ballot64(!whole & (v_l != 0));
        while (mixed) {
            const int L = __builtin_ctzll(mixed);
            mixed &= mixed - 1;
            const int w = w0 + L;
            const uint64_t vw = bcast_u64(v_l, L), pw = bcast_u64(p_l, L);
            const int s = (w << 6) + lane;
            const double rx = ldg(sx, s) - x1, ry = ldg(sy, s) - x2;
            const bool counted = ((vw >> lane) & 1) && !(rx == 0 && ry == 0);      // (bpw:1032: the tool's own sample is skipped)
            int idx = sector_f((float)rx, (float)ry);
            const bool unsure = counted & (idx < 0);
            if (ballot64(unsure) != 0) {
                if (unsure) {
                    double ang = atan2(ry, rx);
                    if (ang < 0) ang = two_pi + ang;
                    idx = (int)py_floor_div(ang, basis);
                    idx = idx > g - 1 ? g - 1 : (idx < 0 ? 0 : idx);
                }
            }
            if (packed) {
                const uint64_t one = counted ? 1ull << (16 * (idx & 3)) : 0, uno = ((pw >> lane) & 1) ? 0 : one;
                const bool hi = idx >= 4;
                tot_p[0] += hi ? 0 : one;
                tot_p[1] += hi ? one : 0;
                und_p[0] += hi ? 0 : uno;
                und_p[1] += hi ? uno : 0;
            } else if (by_ballot) {
                for (int j = 0; j < g; ++j) {
                    const uint64_t m = ballot64(counted & (idx == j));
                    if (lane == j) {
                        tot_j += (int)__popcll(m);
                        und_j += (int)__popcll(m & ~pw);
                    }
                }
            } else if (counted) {
                atomicAdd(&cnt[idx], 1);
                if (!((pw >> lane) & 1)) atomicAdd(&cnt[64 + idx], 1);
            }
        }
    }
    if (packed) {
        for (int j = 0; j < g; ++j) {
            const uint32_t t = wave_sum_u32((uint32_t)(tot_p[j >> 2] >> (16 * (j & 3))) & 0xffffu);
            const uint32_t u = wave_sum_u32((uint32_t)(und_p[j >> 2] >> (16 * (j & 3))) & 0xffffu);
            if (lane == j) {
                tot_j = (int)t;
                und_j = (int)u;
            }
        }
    }
    if (by_ballot) {
        if (lane < g) out[lane] = tot_j == 0 ? 0.0 : (double)und_j / (double)tot_j;
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // all atomics before the read-out
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < g) {
        const int t = cnt[lane], u = cnt[64 + lane];
        out[lane] = t == 0 ? 0.0 : (double)u / (double)t;
    }
}

// bpw:965-978 get_normalized_pose.  The two quotients are formed in two lanes at once (wave-uniform inputs: lane 0 the
// position along axis a1 within its grid row, lane 1 along axis a2) and handed round: one division sequence, not two.
__device__ __forceinline__ void normalized_pose(PartRef P, CfgRef C, const double pose[3], double &x1, double &x2,
                                                double &np0, double &np1) {
    const double r = C.paint_radius;
    x1 = sel3(pose[0], pose[1], pose[2], P.a1);
    x2 = sel3(pose[0], pose[1], pose[2], P.a2);
    const int gi = grid_index_2(P, x2);
    const double lo = P.grid_lo[gi], hi = P.grid_hi[gi];
    const bool second = (threadIdx.x & 63) == 1;
    const double num = second ? x2 - P.r2min + r : x1 - lo + r;
    const double den = second ? P.r2max - P.r2min + 2 * r : hi - lo + 2 * r;
    double q = num / den;
    if (!second && hi - lo == 0) q = 0;
    q = clip01(q);
    np0 = bcast_d(q, 0);
    np1 = bcast_d(q, 1);
}

// bpw:1126-1139 grid observation, 16 cells starting at c0: adds popcount(painted word & cell mask) of word w into
// four packed accumulators (4 x 16-bit per u64; a lane adds at most 64 per word)
__device__ __forceinline__ void grid_accumulate(PartRef P, int c0, int cells, uint64_t pw, int w, uint64_t acc[4]) {
#ifndef PRL_GRID_ALL_MASKS                           // (A/B and parity switch: every cell's mask tried for every word)
    // Round 5: the cells a word's samples lie in are known (PartDev::word_cells): a word inside ONE cell -- four words in five
    // -- adds the popcount of its painted word (painted bits are valid samples) with no mask read, a word on a boundary reads
    // the masks of its two to four cells: 0.4 mask words a word instead of 16 (142 KB an env-step at 70 654 samples).
    const uint32_t wc = (uint32_t)ldg(P.word_cells, w);
    if (wc != 0xfffffffeu) {
        const bool single = (wc >> 8) == 0xffffffu;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = (int)((wc >> (8 * q)) & 0xffu), j = c - c0;
            if (c != 0xff && j >= 0 && j < 16) {
                const uint64_t m = single ? ~0ull : ldg(P.cell_mask, c * P.n_words + w);
                const uint64_t v = (uint64_t)__popcll(pw & m) << (16 * (j & 3));
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] += (j >> 2) == g ? v : 0;
            }
        }
        return;
    }
#endif
#pragma unroll
    for (int j = 0; j < 16; ++j)
        if (c0 + j < cells)
            acc[j >> 2] += (uint64_t)__popcll(pw & ldg(P.cell_mask, (c0 + j) * P.n_words + w)) << (16 * (j & 3));
}

// 4-sector rule bpw:1034-1043 for the words  lane + 64 (slot0 + k), k < KW,  whose painted bits are painted[k]:
// adds the per-sector totals / unpainted counts of those words into tot_l / und_l (4 x 16-bit fields per lane).
template <int KW>
__device__ __forceinline__ void section4_accumulate(PartRef P, double x1, double x2, const uint64_t painted[KW_MAX],
                                                    int slot0, int lane, uint64_t &tot_l, uint64_t &und_l) {
    gdouble_p sx = P.samp_a1, sy = P.samp_a2;
    // Pass 1, one word per lane and slot: a word whose box lies in one sector is counted whole.  A word
    // that straddles only the vertical line x1 (its row is clear of x2) is resolved by its own lane
    // below; only the rest -- the words of the row that x2 crosses -- is classified sample by sample.
    bool vline[KW_MAX] = {false, false, false, false}, above[KW_MAX] = {false, false, false, false};
    uint64_t valid[KW_MAX] = {0, 0, 0, 0}, smask[KW_MAX] = {0, 0, 0, 0};
#ifdef PRL_OBS_YPASS
    uint64_t ymask[KW_MAX] = {0, 0, 0, 0}, xgmask[KW_MAX] = {0, 0, 0, 0};      // pass 3a: straddles x2 only | of those: right of x1
#endif
#ifdef PRL_OBS_YSORT
    uint32_t yo_bits = 0, xr_bits = 0;             // bit k: this lane's word of slot k straddles x2 only | lies right of x1
#endif
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        const int w = lane + 64 * (slot0 + k);
        bool straddle = false;
#if defined(PRL_OBS_YPASS) || defined(PRL_OBS_YSORT)
        bool yonly = false, xright = false;
#endif
        if constexpr (KW < 4) {
            // (straight-line: a lane beyond the part's words reads word 0 and carries an empty `valid`)
            const bool inw = w < P.n_words;
            const int wc = inw ? w : 0;
            const f64x4 bb = ldg(reinterpret_cast<const f64x4 GAS *>(P.word_bbox), wc);
            const uint64_t vw = ldg(P.word_valid, wc);
            valid[k] = inw ? vw : 0;
            const bool xg = bb.x > x1, xl = bb.y < x1, yg = bb.z > x2, yl = bb.w < x2;
            const bool whole = (xg | xl) & (yg | yl);        // (bitwise on purpose, here and below: `&&` / `||` on lane predicates become branches)
            const int idx = (xg & yg) ? 0 : ((xl & yg) ? 1 : ((xl & yl) ? 2 : 3));
            const uint64_t cv = whole ? valid[k] : 0;
            tot_l += (uint64_t)__popcll(cv) << (16 * idx);
            und_l += (uint64_t)__popcll(cv & ~painted[k]) << (16 * idx);
            const bool rest = !whole & (valid[k] != 0);
            vline[k] = rest & (yg | yl);
            above[k] = yg;
            straddle = rest & !(yg | yl);
#if defined(PRL_OBS_YPASS) || defined(PRL_OBS_YSORT)
            yonly = straddle & (xg | xl);
            xright = xg;
#endif
        } else if (w < P.n_words) {                  // (four slots: the straight-line form costs the KW = 4 kernels two spilled registers)
            const f64x4 bb = ldg(reinterpret_cast<const f64x4 GAS *>(P.word_bbox), w);
            valid[k] = ldg(P.word_valid, w);
            const bool xg = bb.x > x1, xl = bb.y < x1, yg = bb.z > x2, yl = bb.w < x2;
            if ((xg | xl) & (yg | yl)) {
                const int idx = (xg & yg) ? 0 : ((xl & yg) ? 1 : ((xl & yl) ? 2 : 3));
                tot_l += (uint64_t)__popcll(valid[k]) << (16 * idx);
                und_l += (uint64_t)__popcll(valid[k] & ~painted[k]) << (16 * idx);
            } else if (valid[k] != 0) {
                vline[k] = yg | yl;
                above[k] = yg;
                straddle = !vline[k];
#if defined(PRL_OBS_YPASS) || defined(PRL_OBS_YSORT)
                yonly = straddle & (xg | xl);
                xright = xg;
#endif
            }
        }
#if defined(PRL_OBS_YSORT)
        yo_bits |= (yonly ? 1u : 0u) << k;
        xr_bits |= (xright ? 1u : 0u) << k;
        smask[k] = ballot64(straddle & !yonly);
#elif defined(PRL_OBS_YPASS)
        ymask[k] = ballot64(yonly);
        xgmask[k] = ballot64(yonly & xright);
        smask[k] = ballot64(straddle & !yonly);
#else
        smask[k] = ballot64(straddle);
#endif
    }
    // Pass 2: the samples of a word ascend on axis a1 (device_tables), so { xs < x1 } is a prefix and
    // { xs > x1 } a suffix of the word.  The owning lane finds the prefix length in two round trips, all its rows at
    // once: the word's eight pivots (samples 7, 15, .. 63) name the group of eight that holds the boundary, the
    // group itself gives the position.
    // Above the line the rule reads  > -> 0, < -> 1, == -> 3;  below it  < -> 2, else 3  (bpw:1034-1043).
    // (four slots' probes at once are 64 vector registers of pivots: KW = 4 takes them two slots at a time -- one round trip
    // more for the reference's 14 482-sample sheet, and no spilled registers)
    constexpr int G = KW == 4 ? 2 : KW;
#if defined(PRL_OBS_CUT) && PRL_OBS_CUT >= 2           // (timing builds, wrong results: the observation's passes cut away one by one)
    if (false)
#endif
#pragma unroll
    for (int k0 = 0; k0 < KW; k0 += G) {
        bool any = false;
#pragma unroll
        for (int k = k0; k < k0 + G; ++k) any = any || vline[k];
        if (ballot64(any)) {
            const f64x4 GAS *pv = reinterpret_cast<const f64x4 GAS *>(P.word_pivot);
            const f64x4 GAS *sx4 = reinterpret_cast<const f64x4 GAS *>(sx);
            int grp[G];
            {
                f64x4 pa[G], pb[G];
#pragma unroll
                for (int k = 0; k < G; ++k) {
                    const int w = vline[k0 + k] ? lane + 64 * (slot0 + k0 + k) : 0;    // other lanes probe word 0: harmless, in bounds
                    pa[k] = ldg(pv, 2 * w);
                    pb[k] = ldg(pv, 2 * w + 1);
                }
                __builtin_amdgcn_sched_barrier(0);      // the slots' probes travel together: one round trip
#pragma unroll
                for (int k = 0; k < G; ++k)
                    grp[k] = (pa[k].x < x1) + (pa[k].y < x1) + (pa[k].z < x1) + (pa[k].w < x1) + (pb[k].x < x1) + (pb[k].y < x1) +
                             (pb[k].z < x1) + (pb[k].w < x1);
            }
            int pos[G];
            bool eq[G];
            {
                f64x4 ga[G], gb[G];
#pragma unroll
                for (int k = 0; k < G; ++k) {
                    const int w = vline[k0 + k] ? lane + 64 * (slot0 + k0 + k) : 0;
                    const int g8 = grp[k] < 8 ? grp[k] : 7;                  // grp = 8: the whole word lies left of x1
                    ga[k] = ldg(sx4, 16 * w + 2 * g8);
                    gb[k] = ldg(sx4, 16 * w + 2 * g8 + 1);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < G; ++k) {
                    const int in = (ga[k].x < x1) + (ga[k].y < x1) + (ga[k].z < x1) + (ga[k].w < x1) + (gb[k].x < x1) + (gb[k].y < x1) +
                                   (gb[k].z < x1) + (gb[k].w < x1);
                    pos[k] = grp[k] < 8 ? 8 * grp[k] + in : 64;
                    // the first sample not left of x1 lies in this group: it equals x1 iff some sample of the group does
                    eq[k] = (grp[k] < 8) & ((ga[k].x == x1) | (ga[k].y == x1) | (ga[k].z == x1) | (ga[k].w == x1) | (gb[k].x == x1) |
                                         (gb[k].y == x1) | (gb[k].z == x1) | (gb[k].w == x1));
                }
            }
#pragma unroll
            for (int kk = 0; kk < G; ++kk) {
                const int k = k0 + kk;
                int ub = pos[kk];
                if (ballot64(eq[kk] & vline[k])) {                 // a sample exactly on the line: the run of equals ends at samp_ub
                    const int at = ((vline[k] ? lane + 64 * (slot0 + k) : 0) << 6) + (pos[kk] < 64 ? pos[kk] : 63);
                    if (eq[kk]) ub = (int)ldg(P.samp_ub, at);
                }
                if (x1 != x1) ub = 64;                              // NaN: nothing is greater either
                if (vline[k]) {
                    const uint64_t lt = pos[kk] >= 64 ? ~0ull : ((1ull << pos[kk]) - 1);
                    const uint64_t ng = ub >= 64 ? ~0ull : ((1ull << ub) - 1);
                    const uint64_t v = valid[k], u = valid[k] & ~painted[k];
                    const uint64_t vg = __popcll(v & ~ng), vl = __popcll(v & lt), ve = __popcll(v) - vg - vl;
                    const uint64_t ug = __popcll(u & ~ng), ul = __popcll(u & lt), ue = __popcll(u) - ug - ul;
                    tot_l += above[k] ? (vg | (vl << 16) | (ve << 48)) : ((vl << 32) | ((vg + ve) << 48));
                    und_l += above[k] ? (ug | (ul << 16) | (ue << 48)) : ((ul << 32) | ((ug + ue) << 48));
                }
            }
        }
    }
#ifdef PRL_OBS_YSORT
    // [A/B switch] the words of the tool's row that lie wholly left or right of it, each by its own lane (section4_big's way):
    // two binary searches in the word's a2 values in ascending order, two suffix masks, popcounts
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        const bool mine = (yo_bits >> k) & 1;
        if (ballot64(mine)) {
            const int wc = mine ? lane + 64 * (slot0 + k) : 0;
            gdouble_p ys_w = P.word_ysort + (size_t)wc * 64;
            int lo_lt = 0, hi_lt = 64, lo_le = 0, hi_le = 64;
#pragma unroll
            for (int it = 0; it < 7; ++it) {
                const int m_lt = (lo_lt + hi_lt) >> 1, m_le = (lo_le + hi_le) >> 1;
                const double y_lt = ldg(ys_w, m_lt < 64 ? m_lt : 63), y_le = m_le == m_lt ? y_lt : ldg(ys_w, m_le < 64 ? m_le : 63);
                if (lo_lt < hi_lt) {
                    if (y_lt < x2) lo_lt = m_lt + 1;
                    else hi_lt = m_lt;
                }
                if (lo_le < hi_le) {
                    if (y_le <= x2) lo_le = m_le + 1;
                    else hi_le = m_le;
                }
            }
            const uint64_t ge = ldg(P.word_ymask, (size_t)wc * 65 + lo_lt), gt = ldg(P.word_ymask, (size_t)wc * 65 + lo_le);
            if (mine) {
                const uint64_t v_i = valid[k], u_i = v_i & ~painted[k], lt = v_i & ~ge, eq = ge & ~gt;
                if ((xr_bits >> k) & 1) {
                    tot_l += (uint64_t)__popcll(gt) | ((uint64_t)__popcll(v_i & ~gt) << 48);
                    und_l += (uint64_t)__popcll(gt & u_i) | ((uint64_t)__popcll(u_i & ~gt) << 48);
                } else {
                    tot_l += ((uint64_t)__popcll(gt) << 16) | ((uint64_t)__popcll(lt) << 32) | ((uint64_t)__popcll(eq) << 48);
                    und_l += ((uint64_t)__popcll(gt & u_i) << 16) | ((uint64_t)__popcll(lt & u_i) << 32) | ((uint64_t)__popcll(eq & u_i) << 48);
                }
            }
        }
    }
#endif
    // Pass 3: the words that straddle the tool on both axes or on x2 only, one sample per lane, four words per trip
    // (their loads travel together).  32-bit work only: the uniform valid / painted words become lane predicates
    // (inverse ballot) and the counters are four 8-bit fields (a lane sees at most 64 straddling words per call).
    // (-DPRL_OBS_SCALAR_MASKS, A/B: the six compares of a word as lane masks and the sector rule + eight population counts
    // as scalar arithmetic on them -- ~10 vector instructions a word instead of ~35, ~55 scalar ones more: 39.2 against
    // 37.6 us, profiles/r04_ab_log.txt: the scalar instructions are not free, a wave issues them in its own order.)
#ifdef PRL_OBS_SCALAR_MASKS
    uint64_t tot_sc = 0, und_sc = 0;
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        uint64_t sm = smask[k];
        while (sm) {                                // wave-uniform loop over the words that straddle the tool
            WCNT(6, 1);
            int L[4];
            uint64_t vs[4];
            double xs[4], ys[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool has = sm != 0;
                L[q] = has ? __builtin_ctzll(sm) : 0;
                sm &= sm - 1;                       // (0 stays 0)
                const int w2 = L[q] + 64 * (slot0 + k);
                xs[q] = ldg(sx, (w2 << 6) + lane);
                ys[q] = ldg(sy, (w2 << 6) + lane);
                vs[q] = has ? P.word_valid[w2] : 0;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint64_t pw = bcast_u64(painted[k], L[q]);
                const uint64_t gx = ballot64(xs[q] > x1), lx = ballot64(xs[q] < x1), gy = ballot64(ys[q] > x2), ly = ballot64(ys[q] < x2);
                const uint64_t on_tool = ballot64(xs[q] == x1 && ys[q] == x2);            // bpw:1032: the tool's own sample is skipped
                const uint64_t c = vs[q] & ~on_tool, u = c & ~pw;
                const uint64_t s0 = gx & gy, s1 = lx & gy, s2 = lx & ly, s3 = ~(s0 | s1 | s2);
                tot_sc += (uint64_t)__popcll(c & s0) | ((uint64_t)__popcll(c & s1) << 16) | ((uint64_t)__popcll(c & s2) << 32) |
                          ((uint64_t)__popcll(c & s3) << 48);
                und_sc += (uint64_t)__popcll(u & s0) | ((uint64_t)__popcll(u & s1) << 16) | ((uint64_t)__popcll(u & s2) << 32) |
                          ((uint64_t)__popcll(u & s3) << 48);
            }
        }
    }
    if (lane == (slot0 & 63)) {
        tot_l += tot_sc;
        und_l += und_sc;
    }
#else
    uint32_t tot_s = 0, und_s = 0;                 // 4 x 8-bit counters per lane for the straddling words
#if defined(PRL_OBS_CUT) && PRL_OBS_CUT >= 1
#pragma unroll
    for (int k = 0; k < KW; ++k) smask[k] = 0;
#endif
#ifdef PRL_OBS_YPASS
    // [-DPRL_OBS_YPASS, A/B switch, OFF: bit-equal results, the door's step 36.4 against 36.1 us without it (two more lane masks
    // per slot held across pass 2: profiles/r05_ab_log.txt); the large parts' own pass 3 (section4_big) is built this way.]
    // Pass 3a: most of those words -- every word of the row x2 crosses but the one x1 crosses too -- lie wholly left or right
    // of the tool (their box says which): only the a2 coordinate is needed, and a float copy of it decides unless it comes
    // within a float's spacing of x2 (nearest-float rounding is monotone: f(ys) > the float at or above x2 implies ys > x2,
    // f(ys) < the float at or below x2 implies ys < x2); a word with such a sample is redone in float64.  4 bytes a sample
    // instead of 16, a dozen vector instructions a word instead of 35 -- on a 70 654-sample part the row is 65 words.
    // The tool's own sample (bpw:1032) cannot be here: a word that does not straddle x1 holds no sample with xs == x1.
    {
        const float y_lo = f32_at_or_below(x2), y_hi = f32_at_or_above(x2);
        gfloat_p yf32 = P.samp_a2_f32;
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            uint64_t sm = ymask[k];
            const uint64_t xgm = xgmask[k];
            while (sm) {
                WCNT(6, 1);
                int L[4];
                uint64_t vs[4];
                float yf[4];
                bool amb = false;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool has = sm != 0;
                    L[q] = has ? __builtin_ctzll(sm) : 0;
                    sm &= sm - 1;
                    const int w2 = L[q] + 64 * (slot0 + k);
                    yf[q] = ldg(yf32, (w2 << 6) + lane);
                    vs[q] = has ? P.word_valid[w2] : 0;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) amb |= (yf[q] >= y_lo) & (yf[q] <= y_hi) & __builtin_amdgcn_inverse_ballot_w64(vs[q]);
                const bool redo = ballot64(amb) != 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    bool gy = yf[q] > y_hi, ly = yf[q] < y_lo;
                    if (redo) {                                     // (rare: a sample within a float's spacing of the line)
                        const double yd = ldg(sy, ((L[q] + 64 * (slot0 + k)) << 6) + lane);
                        gy = yd > x2;
                        ly = yd < x2;
                    }
                    const uint64_t pw = bcast_u64(painted[k], L[q]);
                    const bool right = (xgm >> L[q]) & 1;           // wave-uniform
                    // bpw:1034-1043:  (>, >) -> 0, (<, >) -> 1, (<, <) -> 2, else 3
                    const uint32_t up = right ? 0u : 8u, down = right ? 24u : 16u;
                    const uint32_t sh = gy ? up : (ly ? down : 24u);
                    const uint32_t one = __builtin_amdgcn_inverse_ballot_w64(vs[q]) ? (1u << sh) : 0u;
                    tot_s += one;
                    und_s += __builtin_amdgcn_inverse_ballot_w64(pw) ? 0u : one;
                }
            }
        }
    }
#endif
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        uint64_t sm = smask[k];
        while (sm) {                                // wave-uniform loop over the words that straddle the tool
            WCNT(6, 1);
            int L[4];
            uint64_t vs[4];
            double xs[4], ys[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool has = sm != 0;
                L[q] = has ? __builtin_ctzll(sm) : 0;
                sm &= sm - 1;                       // (0 stays 0)
                const int w2 = L[q] + 64 * (slot0 + k);
                xs[q] = ldg(sx, (w2 << 6) + lane);
                ys[q] = ldg(sy, (w2 << 6) + lane);
                vs[q] = has ? P.word_valid[w2] : 0;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint64_t pw = bcast_u64(painted[k], L[q]);
                const bool cnt = __builtin_amdgcn_inverse_ballot_w64(vs[q]) & !((xs[q] == x1) & (ys[q] == x2));
                const bool gy = ys[q] > x2, lx = xs[q] < x1;
                const uint32_t sh = ((xs[q] > x1) & gy) ? 0u : ((lx & gy) ? 8u : ((lx & (ys[q] < x2)) ? 16u : 24u));
                const uint32_t one = cnt ? (1u << sh) : 0u;
                tot_s += one;
                und_s += __builtin_amdgcn_inverse_ballot_w64(pw) ? 0u : one;
            }
        }
    }
    // widen the 8-bit straddle counters into the 16-bit fields
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        tot_l += (uint64_t)((tot_s >> (8 * q)) & 0xffu) << (16 * q);
        und_l += (uint64_t)((und_s >> (8 * q)) & 0xffu) << (16 * q);
    }
#endif
}

// section / discrete tail: out[0..g-1] are written by the caller; the pose part follows
__device__ __forceinline__ void section_pose_tail(int mode, int g, double np0, double np1, double *out) {
    if (mode == PRL_OBS_SECTION) {
        out[g] = np0;
        out[g + 1] = np1;
    } else {
        const int position = (handle_pos(np0) + 1) * 22 + handle_pos(np1);     // rge:101-103
        out[g] = 1.0 / (double)position;
    }
}

template <typename PW>
__device__ void section4_big(PartRef P, double x1, double x2, PW painted, int lane, int *list, uint64_t &tot_l, uint64_t &und_l);

// GENSEC selects the atan2-sector variant at compile time so that the default kernel carries none of
// its registers or code.  Masks in registers (parts with at most 16 384 samples).
// OBSM: which observation modes the instantiation carries -- -1: all of them, chosen at run time (rollout kernels, cone finish,
// reset / observe); 0: every mode but 'grid'; 1: 'grid' only.  The per-step kernel is built once for each (round 5): with both
// in one kernel the grid code's registers cost the section path 0.4 us (two spilled vector registers at the 128 ceiling),
// profiles/r05_ab_log.txt.
template <int KW, bool GENSEC, int OBSM = -1>
__device__ void observation_wave(PartRef P, CfgRef C, const double pose[3],
                                 const uint64_t painted[KW_MAX], int lane, double *out, int *cnt_lds) {
    double x1, x2, np0, np1;
    normalized_pose(P, C, pose, x1, x2, np0, np1);
    const int mode = OBSM == 1 ? PRL_OBS_GRID : C.obs_mode;
    if (OBSM != 1 && mode == PRL_OBS_SIMPLE) {
        if (lane == 0) {
            out[0] = np0;
            out[1] = np1;
        }
        return;
    }
    if (OBSM != 0 && mode == PRL_OBS_GRID) {       // bpw:1126-1139: 1 - painted/num per cell
        // 16 cells per pass: four packed accumulators, four DPP sums, then lane j finishes cell j (one division
        // per lane, one coalesced store)
        const int cells = P.n_obs_cells;
        for (int c0 = 0; c0 < cells; c0 += 16) {
            uint64_t acc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                const int w = lane + 64 * k;
                if (w < P.n_words) grid_accumulate(P, c0, cells, painted[k], w, acc);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = wave_sum_u64(acc[g]);
            const int cell = c0 + lane;
            if (lane < 16 && cell < cells) {
                uint64_t a4 = acc[0];
#pragma unroll
                for (int g = 1; g < 4; ++g) a4 = (lane >> 2) == g ? acc[g] : a4;
                const int dn = (int)((a4 >> (16 * (lane & 3))) & 0xffff);
                const int num = ldg(P.cell_count, cell);
                out[cell] = num == 0 ? 0.0 : 1.0 - (double)dn / (double)num;
            }
        }
        return;
    }
    if constexpr (OBSM == 1) {
        return;                                    // (a grid-only instantiation has nothing below)
    } else if constexpr (GENSEC) {                 // section / discrete with atan2 sectors (OBS_GRAD != 4)
        section_general_wave<KW>(P, C.obs_grad, x1, x2, painted, static_cast<const uint64_t *>(nullptr), lane, cnt_lds, out);
        if (lane == 0) section_pose_tail(mode, C.obs_grad, np0, np1, out);
        return;
    } else {
        uint64_t tot_l = 0, und_l = 0;             // 4 x 16-bit counters per lane (total / unpainted per sector)
#ifdef PRL_OBS_ROW                                  // (A/B switch: the large parts' three passes on an LDS copy of the painted row)
        {
            __shared__ uint64_t s_obsrow[MAX_WAVES_PER_WG][64 * KW];
            __shared__ int s_obslist[MAX_WAVES_PER_WG][64];
            const int wv = rfl((int)(threadIdx.x >> 6));
#pragma unroll
            for (int k = 0; k < KW; ++k) s_obsrow[wv][lane + 64 * k] = painted[k];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            section4_big(P, x1, x2, static_cast<const uint64_t *>(s_obsrow[wv]), lane, s_obslist[wv], tot_l, und_l);
        }
#else
        section4_accumulate<KW>(P, x1, x2, painted, 0, lane, tot_l, und_l);
#endif
        tot_l = wave_sum_u64(tot_l);
        und_l = wave_sum_u64(und_l);
        if (lane < 4) {                             // the four ratios in four lanes: one division sequence
            const uint32_t t = (uint32_t)((tot_l >> (16 * lane)) & 0xffff);
            const uint32_t u = (uint32_t)((und_l >> (16 * lane)) & 0xffff);
            out[lane] = t == 0 ? 0.0 : (double)u / (double)t;
        }
        if (lane == 0) section_pose_tail(mode, 4, np0, np1, out);
    }
}

// ---------------------------------------------------------------- 4-sector rule on a LARGE part's row of words
// A 70 654-sample part (door_rr_big, rge:116) has 1 100 mask words.  Walking them three lane slots at a time through
// section4_accumulate cost 57 us of a 105 us step: 19 us of it the first pass (48 bytes of box / valid / painted a word, the
// box and valid tables thrashing the L1 of sixteen envs a CU), 16 us the second (six rounds of two dependent probes for ~18
// words), 21 us the third.  Here:
//   1. one word per lane and slot, 20 bytes a word: a float interval of its a1 range (rounded outward: a word it places left
//      or right of x1 is there; the others are looked at exactly below), the word's cell ROW and its number of valid samples
//      in one int (PartDev::word_info), its painted word.  The samples of cell row r are those with cell_coord(a2) == r, and
//      cell_coord is monotone: a row above the tool's row holds only samples with a2 > x2, a row below only a2 < x2 -- no box
//      needed on that axis.  Pads are never painted, so the unpainted count of a whole word is valid - popcount(painted).
//   2. the words that straddle x1 outside the tool's row (one a row) are collected in this wave's LDS list and resolved
//      TOGETHER, one word per lane: the two probes of section4_accumulate's pass 2, once instead of once per slot group.
//   3. the words of the tool's own row, in order, four a trip: those the float interval places left or right of x1 need the
//      a2 coordinate only (float copy, float64 where a sample comes within a float's spacing of x2: section4_accumulate
//      pass 3a), the one or two that straddle x1 as well are classified on both float64 coordinates.
// Same counts as section4_accumulate bit for bit (tests/test_gpu_big_parts.py, test_gpu_full_size.py against the oracle).
// tot / und: per-lane 4 x 16-bit fields (at most 64 x 25 a lane), as there.
template <typename PW>
__device__ void section4_big(PartRef P, double x1, double x2, PW painted, int lane, int *list /* LDS, 64 ints */, uint64_t &tot_l,
                             uint64_t &und_l) {
    gdouble_p sx = P.samp_a1, sy = P.samp_a2;
    const float x1_lo = f32_at_or_below(x1), x1_hi = f32_at_or_above(x1);
    const int cy2 = cell_coord(x2, P.sg_o2, P.sg_inv, P.sg_ny);
    const bool x2ok = x2 == x2;                        // (NaN: no row is above or below, every word is looked at per sample -> sector 3)
    const f32x2 GAS *xbox = reinterpret_cast<const f32x2 GAS *>(P.word_x32);
    const int n_slots = (P.n_words + 63) >> 6;
    int n_list = 0;
    // pass 2 for the words collected so far: lane i takes entry i = word | above << 30
    auto resolve = [&](int n) {
        const bool act = lane < n;
        const int e = act ? list[lane] : 0;
        const int w = e & 0x3fffffff;
        const bool above = (e >> 30) & 1;
        const f64x4 GAS *pv = reinterpret_cast<const f64x4 GAS *>(P.word_pivot);
        const f64x4 GAS *sx4 = reinterpret_cast<const f64x4 GAS *>(sx);
        const f64x4 pa = ldg(pv, 2 * w), pb = ldg(pv, 2 * w + 1);
        const uint64_t v = act ? ldg(P.word_valid, w) : 0, pw = painted[w];
        const int grp = (pa.x < x1) + (pa.y < x1) + (pa.z < x1) + (pa.w < x1) + (pb.x < x1) + (pb.y < x1) + (pb.z < x1) + (pb.w < x1);
        const int g8 = grp < 8 ? grp : 7;                            // grp = 8: the whole word lies left of x1
        const f64x4 ga = ldg(sx4, 16 * w + 2 * g8), gb = ldg(sx4, 16 * w + 2 * g8 + 1);
        const int in = (ga.x < x1) + (ga.y < x1) + (ga.z < x1) + (ga.w < x1) + (gb.x < x1) + (gb.y < x1) + (gb.z < x1) + (gb.w < x1);
        const int pos = grp < 8 ? 8 * grp + in : 64;
        const bool eq = act & (grp < 8) & ((ga.x == x1) | (ga.y == x1) | (ga.z == x1) | (ga.w == x1) | (gb.x == x1) | (gb.y == x1) |
                                           (gb.z == x1) | (gb.w == x1));
        int ub = pos;
        if (ballot64(eq)) {                                          // a sample exactly on the line: the run of equals ends at samp_ub
            const int at = (w << 6) + (pos < 64 ? pos : 63);
            if (eq) ub = (int)ldg(P.samp_ub, at);
        }
        if (x1 != x1) ub = 64;                                       // NaN: nothing is greater either
        const uint64_t lt = pos >= 64 ? ~0ull : ((1ull << pos) - 1);
        const uint64_t ng = ub >= 64 ? ~0ull : ((1ull << ub) - 1);
        const uint64_t u = v & ~pw;
        const uint64_t vg = __popcll(v & ~ng), vl = __popcll(v & lt), ve = __popcll(v) - vg - vl;
        const uint64_t ug = __popcll(u & ~ng), ul = __popcll(u & lt), ue = __popcll(u) - ug - ul;
        // above the line  > -> 0, < -> 1, == -> 3;  below it  < -> 2, else 3  (bpw:1034-1043)
        tot_l += above ? (vg | (vl << 16) | (ve << 48)) : ((vl << 32) | ((vg + ve) << 48));
        und_l += above ? (ug | (ul << 16) | (ue << 48)) : ((ul << 32) | ((ug + ue) << 48));
    };
    // (three slots a trip: their nine loads are issued together, one round trip)
#ifndef PRL_BIG_OBS_G1
#define PRL_BIG_OBS_G1 3                  // (A/B: 2 / 4 / 6 slots a trip)
#endif
    constexpr int G1 = PRL_BIG_OBS_G1;
    for (int k0 = 0; k0 < n_slots; k0 += G1) {
        uint32_t info[G1];
        f32x2 xb[G1];
        uint64_t pw[G1];
        bool inw[G1];
#pragma unroll
        for (int g = 0; g < G1; ++g) {
            const int w = lane + 64 * (k0 + g);
            inw[g] = w < P.n_words;
            const int wc = inw[g] ? w : 0;
            info[g] = ldg(P.word_info, wc);
            xb[g] = ldg(xbox, wc);
            pw[g] = painted[wc];
        }
#pragma unroll
        for (int g = 0; g < G1; ++g) {
            const int w = lane + 64 * (k0 + g);
            const int nv = inw[g] ? (int)(info[g] >> 16) : 0, row = (int)(info[g] & 0xffffu);
            const bool xg = xb[g].x > x1_hi, xl = xb[g].y < x1_lo, yg = (row > cy2) & x2ok, yl = row < cy2;
            const bool whole = (xg | xl) & (yg | yl);
            const int idx = (xg & yg) ? 0 : ((xl & yg) ? 1 : ((xl & yl) ? 2 : 3));
            const uint64_t c = whole ? (uint64_t)nv : 0, pc = (whole & inw[g]) ? (uint64_t)__popcll(pw[g]) : 0;
            tot_l += c << (16 * idx);
            und_l += (c - pc) << (16 * idx);
            const bool vline = !whole & (nv > 0) & (yg | yl);
            const uint64_t vm = ballot64(vline);
            if (vm) {
                const int np = __popcll(vm);
                if (n_list + np > 64) {                               // (more rows than a part has: kept for generality)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    resolve(n_list);
                    __builtin_amdgcn_wave_barrier();
                    n_list = 0;
                }
                if (vline)
                    list[n_list + __builtin_amdgcn_mbcnt_hi((uint32_t)(vm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vm, 0))] = w | (yg ? 1 << 30 : 0);
                n_list += np;
            }
        }
    }
#if defined(PRL_OBS_CUT) && PRL_OBS_CUT >= 2           // (timing builds, wrong results)
    n_list = 0;
#endif
    if (n_list) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");        // the lanes that found the words wrote the list, lanes 0.. read it
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        resolve(n_list);
    }
    // pass 3: the tool's own row of cells.  One word per lane first -- its float interval, valid and painted words in ONE round
    // trip for the whole row (65 words at 70 654 samples) -- which leaves a single load per word for the loop below: its float
    // a2 coordinates, eight words a trip.  (Word by word with the interval, valid and painted words read where they were
    // needed the loop was a chain of ~70 dependent round trips: 28 of the step's 95 us.)
#if defined(PRL_OBS_CUT) && PRL_OBS_CUT >= 1
    if (false)
#endif
    if (cy2 >= 0 && cy2 < P.sg_ny) {
        const int ws0 = P.sg_start[cy2 * P.sg_nx] >> 6, ws1 = (P.sg_start[(cy2 + 1) * P.sg_nx] + 63) >> 6;
        const float y_lo = f32_at_or_below(x2), y_hi = f32_at_or_above(x2);
        gfloat_p yf32 = P.samp_a2_f32;
        uint32_t tot_s = 0, und_s = 0;                 // 4 x 8-bit counters per lane; widened before they can overflow
        int since = 0;
        auto widen = [&]() {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                tot_l += (uint64_t)((tot_s >> (8 * q)) & 0xffu) << (16 * q);
                und_l += (uint64_t)((und_s >> (8 * q)) & 0xffu) << (16 * q);
            }
            tot_s = und_s = 0;
            since = 0;
        };
        for (int c0 = ws0; c0 < ws1; c0 += 64) {
            const int wi = c0 + lane;
            const bool inr = wi < ws1;
            const int wc = inr ? wi : c0;
            const f32x2 xb = ldg(xbox, wc);
            const uint64_t v_i = inr ? ldg(P.word_valid, wc) : 0, p_i = painted[wc];
            const bool right_i = xb.x > x1_hi, side_i = right_i | (xb.y < x1_lo);
            const uint64_t have = ballot64(v_i != 0), rightm = ballot64(right_i);
            uint64_t ym = ballot64(side_i) & have, bm = have & ~ym;
            constexpr int G3 = 8;
#ifndef PRL_OBS_ROW_BY_SAMPLES                      // (A/B and parity switch: the per-sample loop below for every word)
            if (P.word_ysort && ym) {
                // Wholly left or right of the tool (all but one or two words of the row): the a2 coordinate decides, sample by
                // sample -- 65 words of 64 samples at 70 654 samples, ~1 300 instructions a step in the loop below.  Each lane
                // takes ITS word instead: the number of its samples below / not above the line by two binary searches in the
                // word's a2 values in ascending order (PartDev::word_ysort, seven probes, all lanes' in flight together), the
                // samples above / not below the line as ONE precomputed mask each (word_ymask), the sectors' counts by popcount.
                const bool mine = (ym >> lane) & 1;
                // (two probes, as for a1 in pass 2: the word's eight pivots name the group of eight that holds the line, the group
                // gives the position -- seven dependent probes of a binary search were most of this pass's time)
                const f64x4 GAS *yp4 = reinterpret_cast<const f64x4 GAS *>(P.word_ypivot);
                const f64x4 GAS *ys4 = reinterpret_cast<const f64x4 GAS *>(P.word_ysort);
                const f64x4 pa = ldg(yp4, 2 * wc), pb = ldg(yp4, 2 * wc + 1);
                const int grp_lt = (pa.x < x2) + (pa.y < x2) + (pa.z < x2) + (pa.w < x2) + (pb.x < x2) + (pb.y < x2) + (pb.z < x2) + (pb.w < x2);
                const int grp_le = (pa.x <= x2) + (pa.y <= x2) + (pa.z <= x2) + (pa.w <= x2) + (pb.x <= x2) + (pb.y <= x2) + (pb.z <= x2) +
                                   (pb.w <= x2);
                const int g_lt = grp_lt < 8 ? grp_lt : 7, g_le = grp_le < 8 ? grp_le : 7;
                const f64x4 la = ldg(ys4, 16 * wc + 2 * g_lt), lb = ldg(ys4, 16 * wc + 2 * g_lt + 1);
                f64x4 ea = la, eb = lb;
                if (ballot64(g_le != g_lt) != 0) {                       // (values equal to the line across a group's end: rare)
                    ea = ldg(ys4, 16 * wc + 2 * g_le);
                    eb = ldg(ys4, 16 * wc + 2 * g_le + 1);
                }
                const int in_lt = (la.x < x2) + (la.y < x2) + (la.z < x2) + (la.w < x2) + (lb.x < x2) + (lb.y < x2) + (lb.z < x2) + (lb.w < x2);
                const int in_le = (ea.x <= x2) + (ea.y <= x2) + (ea.z <= x2) + (ea.w <= x2) + (eb.x <= x2) + (eb.y <= x2) + (eb.z <= x2) +
                                  (eb.w <= x2);
                const int lo_lt = grp_lt < 8 ? 8 * grp_lt + in_lt : 64, lo_le = grp_le < 8 ? 8 * grp_le + in_le : 64;
                const uint64_t ge = ldg(P.word_ymask, (size_t)wc * 65 + lo_lt), gt = ldg(P.word_ymask, (size_t)wc * 65 + lo_le);
                if (mine) {
                    const uint64_t u_i = v_i & ~p_i, lt = v_i & ~ge, eq = ge & ~gt;
                    // bpw:1034-1043:  (>, >) -> 0, (<, >) -> 1, (<, <) -> 2, else 3
                    if (right_i) {
                        tot_l += (uint64_t)__popcll(gt) | ((uint64_t)__popcll(v_i & ~gt) << 48);
                        und_l += (uint64_t)__popcll(gt & u_i) | ((uint64_t)__popcll(u_i & ~gt) << 48);
                    } else {
                        tot_l += ((uint64_t)__popcll(gt) << 16) | ((uint64_t)__popcll(lt) << 32) | ((uint64_t)__popcll(eq) << 48);
                        und_l += ((uint64_t)__popcll(gt & u_i) << 16) | ((uint64_t)__popcll(lt & u_i) << 32) | ((uint64_t)__popcll(eq & u_i) << 48);
                    }
                }
                ym = 0;
            }
#endif
            while (ym) {                                              // wholly left or right of the tool: the a2 coordinate decides
                int L[G3];
                float yf[G3];
                bool has[G3];
#pragma unroll
                for (int q = 0; q < G3; ++q) {
                    has[q] = ym != 0;
                    L[q] = has[q] ? __builtin_ctzll(ym) : 0;
                    ym &= ym - 1;                                     // (0 stays 0)
                    yf[q] = ldg(yf32, ((c0 + L[q]) << 6) + lane);
                }
                uint64_t vs[G3];
                bool amb = false;
#pragma unroll
                for (int q = 0; q < G3; ++q) {
                    vs[q] = has[q] ? bcast_u64(v_i, L[q]) : 0;
                    amb |= (yf[q] >= y_lo) & (yf[q] <= y_hi) & __builtin_amdgcn_inverse_ballot_w64(vs[q]);
                }
                const bool redo = ballot64(amb) != 0;
#pragma unroll
                for (int q = 0; q < G3; ++q) {
                    bool gy = yf[q] > y_hi, ly = yf[q] < y_lo;
                    if (redo) {                                       // (rare: a sample within a float's spacing of the line)
                        const double yd = ldg(sy, ((c0 + L[q]) << 6) + lane);
                        gy = yd > x2;
                        ly = yd < x2;
                    }
                    const uint64_t pw = bcast_u64(p_i, L[q]);
                    const bool right = (rightm >> L[q]) & 1;          // wave-uniform
                    // bpw:1034-1043:  (>, >) -> 0, (<, >) -> 1, (<, <) -> 2, else 3
                    const uint32_t up = right ? 0u : 8u, down = right ? 24u : 16u;
                    const uint32_t sh = gy ? up : (ly ? down : 24u);
                    const uint32_t one = __builtin_amdgcn_inverse_ballot_w64(vs[q]) ? (1u << sh) : 0u;
                    tot_s += one;
                    und_s += __builtin_amdgcn_inverse_ballot_w64(pw) ? 0u : one;
                }
                since += G3;
                if (since > 255 - G3) widen();
            }
            while (bm) {                                              // straddles both lines (one or two words): both coordinates, float64
                const int Lb = __builtin_ctzll(bm);
                bm &= bm - 1;
                const int w = c0 + Lb;
                const double xs = ldg(sx, (w << 6) + lane), ys = ldg(sy, (w << 6) + lane);
                const uint64_t vw = bcast_u64(v_i, Lb), pw = bcast_u64(p_i, Lb);
                const bool cnt = __builtin_amdgcn_inverse_ballot_w64(vw) & !((xs == x1) & (ys == x2));      // bpw:1032: the tool's own sample is skipped
                const bool gy = ys > x2, lx = xs < x1;
                const uint32_t sh = ((xs > x1) & gy) ? 0u : ((lx & gy) ? 8u : ((lx & (ys < x2)) ? 16u : 24u));
                const uint32_t one = cnt ? (1u << sh) : 0u;
                tot_s += one;
                und_s += __builtin_amdgcn_inverse_ballot_w64(pw) ? 0u : one;
                since += 1;
                if (since > 255 - G3) widen();
            }
        }
        widen();
    }
}

// The same observation for a part with more than 16 384 samples: the painted mask is a row of n_words words -- an LDS copy
// (`const uint64_t *`) or the env's row in HBM (`uint64_t GAS *`: coalesced 8-byte loads, 8.9 KB per env-step at 71 000
// samples) -- walked 64 words per lane slot; counts can exceed 16 bits, so the per-lane 16-bit fields (at most 64 x 25 per
// lane) are split into 32-bit halves before the wave sums.
template <bool GENSEC, int OBSM = -1, typename PW>
__device__ void observation_big(PartRef P, CfgRef C, const double pose[3], PW painted, int lane,
                                double *out, int *cnt_lds, int *list_lds /* this wave's 64-int row (WaveLds::cand) */) {
    double x1, x2, np0, np1;
    normalized_pose(P, C, pose, x1, x2, np0, np1);
    const int mode = OBSM == 1 ? PRL_OBS_GRID : C.obs_mode;       // (OBSM: observation_wave)
    if (OBSM != 1 && mode == PRL_OBS_SIMPLE) {
        if (lane == 0) {
            out[0] = np0;
            out[1] = np1;
        }
        return;
    }
    const int n_slots = (P.n_words + 63) >> 6;
    if (OBSM != 0 && mode == PRL_OBS_GRID) {
        const int cells = P.n_obs_cells;
        for (int c0 = 0; c0 < cells; c0 += 16) {
            uint64_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};       // 2 x 32-bit per u64: fields 0,1 and 2,3 of acc
            for (int k = 0; k < n_slots; ++k) {
                uint64_t acc[4] = {0, 0, 0, 0};
                const int w = lane + 64 * k;
                if (w < P.n_words) grid_accumulate(P, c0, cells, painted[w], w, acc);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    lo[g] += (acc[g] & 0xffffull) | ((acc[g] & 0xffff0000ull) << 16);
                    hi[g] += ((acc[g] >> 32) & 0xffffull) | ((acc[g] >> 48) << 32);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                lo[g] = wave_sum_u64(lo[g]);
                hi[g] = wave_sum_u64(hi[g]);
            }
            const int cell = c0 + lane;
            if (lane < 16 && cell < cells) {
                uint64_t l4 = lo[0], h4 = hi[0];
#pragma unroll
                for (int g = 1; g < 4; ++g) {
                    l4 = (lane >> 2) == g ? lo[g] : l4;
                    h4 = (lane >> 2) == g ? hi[g] : h4;
                }
                const uint64_t pair = (lane & 2) ? h4 : l4;
                const int dn = (int)((lane & 1) ? (pair >> 32) : (pair & 0xffffffffull));
                const int num = ldg(P.cell_count, cell);
                out[cell] = num == 0 ? 0.0 : 1.0 - (double)dn / (double)num;
            }
        }
        return;
    }
    if constexpr (OBSM == 1) {
        return;
    } else if constexpr (GENSEC) {
        const uint64_t none[KW_MAX] = {0, 0, 0, 0};
        section_general_wave<1>(P, C.obs_grad, x1, x2, none, painted, lane, cnt_lds, out);
        if (lane == 0) section_pose_tail(mode, C.obs_grad, np0, np1, out);
        return;
    } else {
        uint64_t tot_l = 0, und_l = 0;
#ifdef PRL_BIG_OBS_BY_SLOTS                          // (A/B and parity switch: the small parts' passes, three lane slots a trip)
        constexpr int G = 3;
        for (int k = 0; k < n_slots; k += G) {
            uint64_t pk[KW_MAX] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const int w = lane + 64 * (k + j);
                pk[j] = w < P.n_words ? painted[w] : 0;
            }
            section4_accumulate<G>(P, x1, x2, pk, k, lane, tot_l, und_l);
        }
#else
        section4_big(P, x1, x2, painted, lane, list_lds, tot_l, und_l);
#endif
        uint64_t t01 = (tot_l & 0xffffull) | ((tot_l & 0xffff0000ull) << 16), t23 = ((tot_l >> 32) & 0xffffull) | ((tot_l >> 48) << 32);
        uint64_t u01 = (und_l & 0xffffull) | ((und_l & 0xffff0000ull) << 16), u23 = ((und_l >> 32) & 0xffffull) | ((und_l >> 48) << 32);
        t01 = wave_sum_u64(t01);
        t23 = wave_sum_u64(t23);
        u01 = wave_sum_u64(u01);
        u23 = wave_sum_u64(u23);
        if (lane < 4) {
            const uint64_t tp = lane < 2 ? t01 : t23, up = lane < 2 ? u01 : u23;
            const uint32_t t = (uint32_t)((lane & 1) ? tp >> 32 : tp), u = (uint32_t)((lane & 1) ? up >> 32 : up);
            out[lane] = t == 0 ? 0.0 : (double)u / (double)t;
        }
        if (lane == 0) section_pose_tail(mode, 4, np0, np1, out);
    }
}

}  // namespace
