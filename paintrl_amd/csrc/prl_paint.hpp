// prl_paint.hpp -- ball-query painting: the five shots of a step together (bpw:568-577).
// Part of the single translation unit paintrl_hip.hip (device code, anonymous namespace); see that
// file for the overall design.  Compile with -ffp-contract=off.
#pragma once

namespace {

// ---------------------------------------------------------------- bit of one sample in a register-resident mask (cone-beam paint)
template <int KW>
__device__ __forceinline__ void set_word(uint64_t cur[KW_MAX], int w, uint64_t b, int lane) {
    const int owner = w & 63, slot = w >> 6;
#pragma unroll
    for (int k = 0; k < KW; ++k)
        if (k == slot && lane == owner) cur[k] |= b;
}

// ---------------------------------------------------------------- the five shots of one step, painted together
// The five shot centres of a step are 0.0102 apart, so their 3x3 neighbourhoods overlap almost
// entirely.  Each candidate sample is loaded once and tested against all five centres; the per-word
// hit ballots b_0..b_4 are wave-uniform, so the reference's shot-by-shot bookkeeping (bpw:572-577:
// count newly painted, paint, valid = affected minus last shot, last = affected) runs on the scalar
// unit for that word and is written back to the lane that owns the word.  Words outside the
// neighbourhood have no hits in any shot: painted is unchanged and their last-shot bits become 0.
// The centres are written to LDS by the shot loop (a register array indexed by the runtime shot
// number would live in scratch) and read back wave-uniformly here.
//
// Conservative float pre-filter.  The distance test `d^2 <= r^2` of bpw:569 is float64; here every candidate is
// first tested in float32 on a float copy of the sample table (one 16-byte load per sample instead of three
// 8-byte loads, and float32 vector operations issue at twice the float64 rate).  With M the largest |coordinate|
// of a real sample, a centre within reach of a sample has |c| <= M + r, so rounding sample and centre to float
// moves each coordinate by at most 2^-24 (M + r), each difference by at most 2^-22 (M + r) including its own
// rounding, and the float d^2 -- three squares and two sums, each rounded -- differs from the exact one by less
// than  band = 16 r 2^-23 (M + r) + 2^-20 r^2  for d near r (a factor > 4 above the worst case).  A sample
// whose float d^2 is <= r^2 - band is inside, one above r^2 + band is outside, bit for bit as in float64; a word
// with any sample in between (about one shot in a hundred) is recomputed in float64, the original code.
// -DPRL_FORCE_F64_PAINT sends every word through the float64 branch; -DPRL_WIDE_PAINT_BAND widens the band
// 4096-fold so that the mixed path runs constantly: both builds must reproduce the product's results exactly.

// How the painter reads and writes one 64-sample word of the env's masks.  Small parts (<= 16 384 samples) keep
// the masks in registers: word w lives in lane w & 63, slot w >> 6 ...
template <int KW>
struct RegWords {
    static constexpr bool HBM = false;
    uint64_t *painted;            // [KW_MAX] register arrays of the caller
    const uint64_t *last;
    uint64_t *new_last;
    int lane;
    __device__ __forceinline__ void get(int w, uint64_t &pw, uint64_t &lw) const {
        const int owner = w & 63, slot = w >> 6;
        pw = 0;
        lw = 0;
#pragma unroll
        for (int k = 0; k < KW; ++k)
            if (k == slot) {
                pw = bcast_u64(painted[k], owner);
                lw = bcast_u64(last[k], owner);
            }
    }
    __device__ __forceinline__ void put(int w, uint64_t pw, uint64_t lw, uint64_t, uint64_t) const {
        const int owner = w & 63, slot = w >> 6;
#pragma unroll
        for (int k = 0; k < KW; ++k)
            if (k == slot && lane == owner) {
                painted[k] = pw;
                new_last[k] = lw;
            }
    }
};

// The same with the last-shot mask and its successor in this wave's LDS rows (step_kernel): sixteen vector registers less
// while the painter runs -- the widest masks (four words per lane: the reference's sheet) spilled 11 - 63 without -- and an LDS
// read instead of a lane broadcast per word.  `new_last` must be zero on entry.
template <int KW>
struct RowWords {
    static constexpr bool HBM = false;
    uint64_t *painted;            // [KW_MAX] registers of the caller
    const uint64_t *last;         // LDS, word w at [w]
    uint64_t *new_last;           // LDS
    int lane;
    __device__ __forceinline__ void get(int w, uint64_t &pw, uint64_t &lw) const {
        const int owner = w & 63, slot = w >> 6;
        pw = 0;
#pragma unroll
        for (int k = 0; k < KW; ++k)
            if (k == slot) pw = bcast_u64(painted[k], owner);
        lw = last[w];
    }
    __device__ __forceinline__ void put(int w, uint64_t pw, uint64_t lw, uint64_t, uint64_t) const {
        const int owner = w & 63, slot = w >> 6;
#pragma unroll
        for (int k = 0; k < KW; ++k)
            if (k == slot && lane == owner) painted[k] = pw;
        if (lane == 0) new_last[w] = lw;
    }
};

// ... larger ones (the 480 x 480 textures: up to ~70 000 samples) keep them in LDS for the length of the kernel.
struct LdsWords {
    static constexpr bool HBM = false;
    uint64_t *painted;            // [n_words] in LDS
    const uint64_t *last;
    uint64_t *new_last;           // zero on entry
    int lane;
    __device__ __forceinline__ void get(int w, uint64_t &pw, uint64_t &lw) const {
        pw = painted[w];
        lw = last[w];
    }
    __device__ __forceinline__ void put(int w, uint64_t pw, uint64_t lw, uint64_t, uint64_t) const {
        if (lane == 0) {
            painted[w] = pw;
            new_last[w] = lw;
        }
    }
};

// ... or leave them where they are: the env's rows in HBM / L2 (step_kernel_big, round 5).  A step paints a few dozen of
// a large part's hundreds to 1 100 words: the painter reads the painted and the last-shot word of each word it visits (one load
// each, every lane the same address: a single request, issued with the word's sample records) and lane 0 writes back what
// changed.  The last-shot row of a step is zero outside the words its last shot touched; which words are not zero is kept as a
// bit set per env (StepArgs::last_nz, one bit per word: lane k of the wave holds words 64 k .. 64 k + 63), so that the words a
// step does NOT visit are cleared one by one instead of the row being rewritten (HbmMasks::paint_done, prl_step.hpp).
//   vis: words this step's painter visited;  nzn: those of them whose new last-shot word is not zero.
struct HbmWords {
    static constexpr bool HBM = true;
    gu64_rw_p painted, last;             // rows of this env
    int lane;
    uint64_t *vis, *nzn;                 // this lane's tracking words (registers of the caller)
    // (the loads: every lane the same address, one request; the words are moved to scalar registers where they are used
    // (do_word) so that the shot-by-shot bookkeeping runs on the scalar unit as for the register-resident masks)
    __device__ __forceinline__ void get(int w, uint64_t &pw, uint64_t &lw) const {
        pw = painted[w];
        lw = last[w];
    }
    __device__ __forceinline__ void put(int w, uint64_t pw, uint64_t lw, uint64_t pw_old, uint64_t lw_old) const {
        if (lane == 0) {
            if (pw != pw_old) painted[w] = pw;
            if (lw != lw_old) last[w] = lw;
        }
        const uint64_t bit = lane == (w >> 6) ? 1ull << (w & 63) : 0;
        *vis |= bit;
        *nzn |= lw != 0 ? bit : 0;
    }
};

// Works for any spread of the five centres: the rows cy_lo-1 .. cy_hi+1 of the sample grid are walked four at a
// time (one trip in practice: the centres of a step are 10 mm apart), columns cx_lo-1 .. cx_hi+1.  A sample outside
// a shot's own 3 x 3 cell block is more than a cell edge (> radius) away from that centre, so testing every sample
// of the bounding block against all five centres gives each shot exactly its own ball query.
template <typename Words>
__device__ void paint_shots_union(PartRef P, double radius, const double *cen_lds, int lane, const Words &words,
                                  int &succeeded, int &pixel_counter, const int *sg_lds = nullptr) {
    const double r2 = radius * radius;
    // the float64 centres are only needed for the cell ranges and by the rare float64 branch, which reads them
    // from LDS again: fifteen doubles held across the word loop would be thirty vector registers
    float cf[PAINT_PER_ACTION][3];
    // cell ranges of the five centres: the cell coordinate is monotone in the position, so the range of the cells
    // is the cells of the range (four conversions instead of ten)
    double lo1 = INFINITY, hi1 = -INFINITY, lo2 = INFINITY, hi2 = -INFINITY;
#pragma unroll
    for (int k = 0; k < PAINT_PER_ACTION; ++k) {
        const double c0 = cen_lds[3 * k], c1 = cen_lds[3 * k + 1], c2 = cen_lds[3 * k + 2];
        cf[k][0] = (float)c0;
        cf[k][1] = (float)c1;
        cf[k][2] = (float)c2;
        const double h1 = sel3(c0, c1, c2, P.a1), h2 = sel3(c0, c1, c2, P.a2);
        lo1 = fmin(lo1, h1);
        hi1 = fmax(hi1, h1);
        lo2 = fmin(lo2, h2);
        hi2 = fmax(hi2, h2);
    }
    // (a NaN centre hits nothing and is ignored by fmin / fmax; five of them leave the inverted range: cell -2 as for a single NaN)
    const bool none = !(lo1 <= hi1);
    // HbmWords: how far a word's box may lie from the centres' box and still hold a sample within the radius (a hair more)
    const double reach_r = radius * (1.0 + 1.0e-9) + 1.0e-12;
    const double reach_lo1 = lo1 - reach_r, reach_hi1 = hi1 + reach_r, reach_lo2 = lo2 - reach_r, reach_hi2 = hi2 + reach_r;
    const int cx_lo = none ? -2 : cell_coord(lo1, P.sg_o1, P.sg_inv, P.sg_nx), cx_hi = none ? -2 : cell_coord(hi1, P.sg_o1, P.sg_inv, P.sg_nx);
    const int cy_lo = none ? -2 : cell_coord(lo2, P.sg_o2, P.sg_inv, P.sg_ny), cy_hi = none ? -2 : cell_coord(hi2, P.sg_o2, P.sg_inv, P.sg_ny);
    double band = 16.0 * radius * 1.1920928955078125e-07 * (P.samp_absmax + radius) + 9.5367431640625e-07 * r2;
#ifdef PRL_WIDE_PAINT_BAND
    band *= 4096.0;
#endif
    // thresholds rounded outward, so that the float comparisons are at least as cautious as the double ones
    const float r2_in = f32_at_or_below(r2 - band), r2_out = f32_at_or_above(r2 + band);
    const f32x4 GAS *s4 = reinterpret_cast<const f32x4 GAS *>(P.samp_f32);
    const int cx0 = cx_lo - 1 < 0 ? 0 : cx_lo - 1, cx1 = cx_hi + 1 > P.sg_nx - 1 ? P.sg_nx - 1 : cx_hi + 1;
    const int row_lo = rfl(cy_lo - 1 < 0 ? 0 : cy_lo - 1), row_hi = rfl(cy_hi + 1 > P.sg_ny - 1 ? P.sg_ny - 1 : cy_hi + 1);
    int done_w = -1;                                 // a word shared by two rows' ranges is handled once
#ifdef PRL_PAINT_ONE_ROW_PER_TRIP                   // diagnostic build: the multi-trip path in every parity test
    constexpr int TRIP = 1;
#else
    constexpr int TRIP = 4;
#endif
    for (int r0 = row_lo; r0 <= row_hi; r0 += TRIP) {
    // TRIP rows per trip: lanes 0 .. 2 TRIP - 1 fetch the range bounds
    const int rcy = r0 + (lane >> 1);
    const bool ok = lane < 2 * TRIP && rcy <= row_hi && cx0 <= cx1;
#ifdef PRL_GRID_LDS                                  // (A/B switch, k_step.hip: the table's LDS copy where the kernel has one)
    const int bidx = ok ? rcy * P.sg_nx + ((lane & 1) ? cx1 + 1 : cx0) : 0;
    const int bound = ok ? (sg_lds ? sg_lds[bidx] : ldg(P.sg_start, bidx)) : 0;
#else
    (void)sg_lds;
    const int bound = ok ? ldg(P.sg_start, rcy * P.sg_nx + ((lane & 1) ? cx1 + 1 : cx0)) : 0;
#endif
    int rb[TRIP], re[TRIP];
#pragma unroll
    for (int r = 0; r < TRIP; ++r) {
        rb[r] = __builtin_amdgcn_readlane(bound, 2 * r);
        re[r] = __builtin_amdgcn_readlane(bound, 2 * r + 1);
    }
    // One word of samples per trip of the loop below: its 64 float records, the five distance tests, the fold into the masks.
    // [-DPRL_PAINT_PIPELINE] The words of these rows are known before the first is fetched, so the NEXT word's records can be
    // requested before the current word is worked on (a word's load is otherwise waited for where it is issued: ~7 exposed
    // round trips a step).  Measured slower -- the step is bound by instruction issue, not by these waits -- and off.
    // (pw_in, lw_in: the word's painted / last-shot words where the caller has fetched them with the records -- HbmWords)
    auto do_word = [&](int w, const f32x4 pf, int lo, int hi, uint64_t pw_in, uint64_t lw_in) {
        WCNT(5, 1);
        uint64_t pw = pw_in, lw = lw_in;
        if constexpr (Words::HBM) {                  // (loaded by every lane from one address: wave-uniform, to scalar registers)
            pw = uni_u64(pw);
            lw = uni_u64(lw);
        }
        const int s = (w << 6) + lane;
        // every cell row starts on a word boundary (device_tables), so a word holds samples of one row only: [lo, hi)
        const bool in = (s >= lo) & (s < hi);
        // (a lane outside the row's range is moved far away instead of masked in each of the ten comparisons below: one select,
        // ten scalar ANDs fewer per word)
        const float pfx = in ? pf.x : 3.0e18f;
        uint64_t b[PAINT_PER_ACTION];
        uint64_t any = 0, unsure = 0;
        // the five float distances, two shots per instruction (v_pk_add / v_pk_mul / v_pk_fma_f32: the same IEEE operations
        // in the same order as one shot at a time, 18 instructions instead of 30)
        float dd5[PAINT_PER_ACTION];
        static_assert(PAINT_PER_ACTION == 5, "shot pairs below");
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f32x2 cx = {cf[2 * q][0], cf[2 * q + 1][0]}, cy = {cf[2 * q][1], cf[2 * q + 1][1]}, cz = {cf[2 * q][2], cf[2 * q + 1][2]};
            const f32x2 px = {pfx, pfx}, py = {pf.y, pf.y}, pz = {pf.z, pf.z};
            const f32x2 dx = px - cx, dy = py - cy, dz = pz - cz;
            const f32x2 dd = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
            dd5[2 * q] = dd.x;
            dd5[2 * q + 1] = dd.y;
        }
        {
            const float dx = pfx - cf[4][0], dy = pf.y - cf[4][1], dz = pf.z - cf[4][2];
            dd5[4] = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
        }
#pragma unroll
        for (int k = 0; k < PAINT_PER_ACTION; ++k) {
            b[k] = ballot64(dd5[k] <= r2_in);
            unsure |= b[k] ^ ballot64(dd5[k] <= r2_out);
            any |= b[k];
        }
#ifdef PRL_FORCE_F64_PAINT
        unsure = 1;
#endif
        if (unsure) {                           // some sample within rounding reach of the sphere: float64 decides
            WCNT(6, 1);
            const double xr = ldg(P.samp[0], s), y = ldg(P.samp[1], s), z = ldg(P.samp[2], s);
            const double x = in ? xr : 1.0e150;
            any = 0;
#pragma unroll
            for (int k = 0; k < PAINT_PER_ACTION; ++k) {
                const double dx = x - cen_lds[3 * k], dy = y - cen_lds[3 * k + 1], dz = z - cen_lds[3 * k + 2];
                const double dd = (dx * dx + dy * dy) + dz * dz;
                b[k] = ballot64(dd <= r2);
                any |= b[k];
            }
        }
        if constexpr (!Words::HBM) words.get(w, pw, lw);
        if (any == 0 && lw == 0) return;         // nothing to record for this word
        const uint64_t pw_old = pw, lw_old = lw;
        // bpw:572-577 shot by shot (count newly painted, paint, valid = affected minus last shot, last =
        // affected), folded: the newly painted samples of the five shots are the union minus what was
        // painted before, and each shot's valid set only looks one shot back
        succeeded += __popcll(any & ~pw);
        pw |= any;
        uint64_t uw = b[0] & ~lw;
#pragma unroll
        for (int k = 1; k < PAINT_PER_ACTION; ++k) uw |= b[k] & ~b[k - 1];
        lw = b[PAINT_PER_ACTION - 1];
        pixel_counter += __popcll(uw);
        words.put(w, pw, lw, pw_old, lw_old);
    };
    // the words of the trip's rows as a bit set relative to the first (rows are consecutive in memory: a few dozen words)
    int base = -1;
    uint64_t wm = 0;
    bool flat = true;
#pragma unroll
    for (int r = 0; r < TRIP; ++r) {
        if (re[r] <= rb[r]) continue;
        const int w0 = (rb[r] >> 6) > done_w ? (rb[r] >> 6) : done_w + 1, w1 = (re[r] - 1) >> 6;
        if (w1 < w0) continue;
        if (base < 0) base = w0;
        if (w0 < base || w1 - base >= 64) flat = false;
        else wm |= (w1 - w0 == 63 ? ~0ull : ((1ull << (w1 - w0 + 1)) - 1)) << (w0 - base);
    }
#ifndef PRL_PAINT_PIPELINE                           // A/B switch; OFF: the prefetching loop hides ~7 round trips a step and adds
    flat = false;                                    // ~150 scalar / vector instructions: 41.5 against 40.2 us (profiles/r04_ab_log.txt)
#endif
    if (flat) {
        if (wm) {
            int w_cur = base + __builtin_ctzll(wm);
            wm &= wm - 1;
            f32x4 pf = ldg(s4, (w_cur << 6) + lane);
            for (;;) {
                const bool more = wm != 0;
                int w_nxt = w_cur;
                f32x4 pn = pf;
                if (more) {
                    w_nxt = base + __builtin_ctzll(wm);
                    wm &= wm - 1;
                    pn = ldg(s4, (w_nxt << 6) + lane);
                }
                int lo = 0, hi = 0;                  // the row this word belongs to (scalar selects)
#pragma unroll
                for (int r = 0; r < TRIP; ++r)
                    if (re[r] > rb[r] && w_cur >= (rb[r] >> 6) && w_cur <= ((re[r] - 1) >> 6)) lo = rb[r], hi = re[r];
                do_word(w_cur, pf, lo, hi, 0, 0);
                done_w = w_cur;
                if (!more) break;
                w_cur = w_nxt;
                pf = pn;
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < TRIP; ++r) {
            if (re[r] <= rb[r]) continue;
            const int wlast = (re[r] - 1) >> 6;
            const int wfirst = (rb[r] >> 6) > done_w ? (rb[r] >> 6) : done_w + 1;
#ifdef PRL_PAINT_PREPASS                              // (A/B switch: the word-box pre-pass for the register-resident masks too)
            constexpr bool prepass = true;
#else
            constexpr bool prepass = Words::HBM;
#endif
            if constexpr (prepass) {
                // Large parts: a row's stretch of the cell block is dozens of words (70 654 samples: ~17 a row, ~70 a step) of
                // which the five balls reach a third.  One word per lane first: a word whose box (principal plane) lies more
                // than the radius beyond the centres' own box holds no sample within the radius of any centre -- |dx| > r gives
                // dx dx > r r whatever dy, dz add -- so its hit sets are empty and the painter would leave it as it found it,
                // except for clearing its last-shot word: HbmMasks::paint does that for every word not visited.
                const f64x4 GAS *wb4 = reinterpret_cast<const f64x4 GAS *>(P.word_bbox);
                for (int wb = wfirst; wb <= wlast; wb += 64) {
                    const int wi = wb + lane;
                    const bool inr = wi <= wlast;
                    const f64x4 bb = ldg(wb4, inr ? wi : wb);
                    // ... and, inside the centres' box, a word no single ball reaches: the distance from a centre to the word's
                    // box in the principal plane (the third axis can only add) against the radius, a hair more
                    bool near = false;
#pragma unroll
                    for (int k = 0; k < PAINT_PER_ACTION; ++k) {
                        const double c0 = cen_lds[3 * k], c1 = cen_lds[3 * k + 1], c2 = cen_lds[3 * k + 2];
                        const double h1 = sel3(c0, c1, c2, P.a1), h2 = sel3(c0, c1, c2, P.a2);
                        const double ex = fmax(fmax(bb.x - h1, h1 - bb.y), 0.0), ey = fmax(fmax(bb.z - h2, h2 - bb.w), 0.0);
                        near |= ex * ex + ey * ey <= reach_r * reach_r;
                    }
                    const bool reach = inr & near & !((bb.x > reach_hi1) | (bb.y < reach_lo1) | (bb.z > reach_hi2) | (bb.w < reach_lo2));
                    uint64_t m = ballot64(reach);
#ifdef PRL_BIG_PREFETCH                              // (A/B switch: the next word's loads one word ahead; see profiles/r05_ab_log.txt)
                    if (m) {
                        int w_cur = wb + __builtin_ctzll(m);
                        m &= m - 1;
                        f32x4 pf = ldg(s4, (w_cur << 6) + lane);
                        uint64_t pw_c, lw_c;
                        words.get(w_cur, pw_c, lw_c);
                        for (;;) {
                            const bool more = m != 0;
                            const int w_nxt = more ? wb + __builtin_ctzll(m) : w_cur;
                            m &= m - 1;                                   // (0 stays 0)
                            const f32x4 pn = ldg(s4, (w_nxt << 6) + lane);       // (the last trip reads its own word again: harmless)
                            uint64_t pw_n, lw_n;
                            words.get(w_nxt, pw_n, lw_n);
                            do_word(w_cur, pf, rb[r], re[r], pw_c, lw_c);
                            if (!more) break;
                            w_cur = w_nxt;
                            pf = pn;
                            pw_c = pw_n;
                            lw_c = lw_n;
                        }
                    }
                }
#else
                    // (GP words a trip with their records and mask words requested together: 2 spilled ten vector registers, 4 fifty-seven --
                    // 18 k samples 43.3 -> 48.4 us, 70 k 84.2 -> 81.7: one at a time)
                    while (m) {
#ifdef PRL_BIG_GP
                        constexpr int GP = PRL_BIG_GP;
#else
                        constexpr int GP = 1;
#endif
                        int wq[GP];
                        bool hq[GP];
                        f32x4 pfq[GP];
                        uint64_t pwq[GP], lwq[GP];
#pragma unroll
                        for (int q = 0; q < GP; ++q) {
                            hq[q] = m != 0;
                            wq[q] = wb + (hq[q] ? __builtin_ctzll(m) : 0);
                            m &= m - 1;                                   // (0 stays 0)
                            pfq[q] = ldg(s4, (wq[q] << 6) + lane);
                            words.get(wq[q], pwq[q], lwq[q]);
                        }
#pragma unroll
                        for (int q = 0; q < GP; ++q)
                            if (hq[q]) do_word(wq[q], pfq[q], rb[r], re[r], pwq[q], lwq[q]);
                    }
                }
#endif
                done_w = wlast > done_w ? wlast : done_w;
            } else {
                for (int w = wfirst; w <= wlast; ++w) {
                    do_word(w, ldg(s4, (w << 6) + lane), rb[r], re[r], 0, 0);
                    done_w = w;
                }
            }
        }
    }
    }
}

// ---------------------------------------------------------------- COLOR_MODE = 'HSI': thickness bytes (bpw:384-434)
// As the reference behaves (oracle/paint_oracle.c apply_paint_hsi states it in full): each front texel carries a
// uint8, 255 after reset; shot by shot every hit texel whose byte is not 0 loses
//     quantity = int(25 (1 - (d / r)^2)) + 1,     d = distance to the shot centre, r = the shot's largest d,
// in wrapping uint8 arithmetic, and contributes quantity / 255 to the shot's "succeed counter".  The status bit the
// observation reads stays "byte == 255" (bpw:723-725).  `thick` is this env's row of bytes in HBM (device sample order); only
// the words a shot touches are read and written.  The float sum is reduced in lane order: tests allow 1e-12 on rewards.
//
// Until round 5 the five shots ran one after the other, two passes over a shot's 3 x 3 cell block each (76 us a step on the
// door, every word read up to ten times), and for parts beyond 16 384 samples on four LDS copies of the mask rows with row-wide
// passes between the shots (one wave a SIMD: 755 us a step at 70 654 samples).  But a shot's largest distance depends on the
// geometry alone, and everything else is local to a word once the five r's are known.  So: pass 0 finds the five r's, pass 1
// visits each word of the shots' bounding cell block ONCE -- a sample's byte takes the five deposits in shot order in a
// register, the status bit is the byte's last state, and as in paint_shots_union the valid sets only look one shot back:
//     union = (b0 & ~last) | (b1 & ~b0) | ... ,  new last = b4.
// A sample outside a shot's own 3 x 3 cell block is more than the radius away from its centre (paint_shots_union), so inside the
// bounding block the distance alone decides -- oracle/paint_oracle.c ball_query.  Words the balls cannot reach are left out by the
// word-box test of paint_shots_union's pre-pass; words not visited keep their bytes and painted bits and lose their last-shot
// word (HbmMasks::paint_hsi clears what was set before and not visited now; register masks: the new last-shot words start at zero).
// Words: RegWords (masks in registers) or HbmWords (rows in HBM).  61 us on the door, 140 us at 70 411 samples.
template <int KW>
__device__ void paint_shots_hsi(PartRef P, double radius, const double *cen_lds, int lane, uint64_t painted[KW_MAX],
                                uint64_t last[KW_MAX], uint8_t *thick, double &succeeded, int &pixel_counter, double *scratch);

template <typename Words>
__device__ void paint_shots_hsi_words(PartRef P, double radius, const double *cen_lds, int lane, const Words &words, uint8_t *thick,
                                      double &succeeded, int &pixel_counter, double *scratch) {      // scratch: 10 doubles of this wave's LDS
    const double r2 = radius * radius;
    double lo1 = INFINITY, hi1 = -INFINITY, lo2 = INFINITY, hi2 = -INFINITY;
#pragma unroll
    for (int k = 0; k < PAINT_PER_ACTION; ++k) {
        const double c0 = cen_lds[3 * k], c1 = cen_lds[3 * k + 1], c2 = cen_lds[3 * k + 2];
        const double h1 = sel3(c0, c1, c2, P.a1), h2 = sel3(c0, c1, c2, P.a2);
        lo1 = fmin(lo1, h1);
        hi1 = fmax(hi1, h1);
        lo2 = fmin(lo2, h2);
        hi2 = fmax(hi2, h2);
    }
    const bool none = !(lo1 <= hi1);                 // (NaN centres: they hit nothing)
    const double reach_r = radius * (1.0 + 1.0e-9) + 1.0e-12;
    const double reach_lo1 = lo1 - reach_r, reach_hi1 = hi1 + reach_r, reach_lo2 = lo2 - reach_r, reach_hi2 = hi2 + reach_r;
    const int cx_lo = none ? -2 : cell_coord(lo1, P.sg_o1, P.sg_inv, P.sg_nx), cx_hi = none ? -2 : cell_coord(hi1, P.sg_o1, P.sg_inv, P.sg_nx);
    const int cy_lo = none ? -2 : cell_coord(lo2, P.sg_o2, P.sg_inv, P.sg_ny), cy_hi = none ? -2 : cell_coord(hi2, P.sg_o2, P.sg_inv, P.sg_ny);
    const int cx0 = cx_lo - 1 < 0 ? 0 : cx_lo - 1, cx1 = cx_hi + 1 > P.sg_nx - 1 ? P.sg_nx - 1 : cx_hi + 1;
    const int row_lo = rfl(cy_lo - 1 < 0 ? 0 : cy_lo - 1), row_hi = rfl(cy_hi + 1 > P.sg_ny - 1 ? P.sg_ny - 1 : cy_hi + 1);
    const f64x4 GAS *wb4 = reinterpret_cast<const f64x4 GAS *>(P.word_bbox);
    // every word of the bounding block that a ball may reach, once, in ascending order: fn(word)
    auto for_words = [&](auto &&fn) {
        int done_w = -1;
        for (int r0 = row_lo; r0 <= row_hi; r0 += 32) {
            const int rcy = r0 + (lane >> 1);
            const bool ok = rcy <= row_hi && cx0 <= cx1;
            const int bound = ok ? ldg(P.sg_start, rcy * P.sg_nx + ((lane & 1) ? cx1 + 1 : cx0)) : 0;
            const int n_rows = row_hi - r0 + 1 < 32 ? row_hi - r0 + 1 : 32;
            for (int r = 0; r < n_rows; ++r) {
                const int rb = __builtin_amdgcn_readlane(bound, 2 * r), re = __builtin_amdgcn_readlane(bound, 2 * r + 1);
                if (re <= rb) continue;
                const int wlast = (re - 1) >> 6, wfirst = (rb >> 6) > done_w ? (rb >> 6) : done_w + 1;
                for (int wb = wfirst; wb <= wlast; wb += 64) {
                    const int wi = wb + lane;
                    const bool inr = wi <= wlast;
                    const f64x4 bb = ldg(wb4, inr ? wi : wb);
                    bool near = false;
#pragma unroll
                    for (int k = 0; k < PAINT_PER_ACTION; ++k) {
                        const double c0 = cen_lds[3 * k], c1 = cen_lds[3 * k + 1], c2 = cen_lds[3 * k + 2];
                        const double h1 = sel3(c0, c1, c2, P.a1), h2 = sel3(c0, c1, c2, P.a2);
                        const double ex = fmax(fmax(bb.x - h1, h1 - bb.y), 0.0), ey = fmax(fmax(bb.z - h2, h2 - bb.w), 0.0);
                        near |= ex * ex + ey * ey <= reach_r * reach_r;
                    }
                    const bool reach = inr & near & !((bb.x > reach_hi1) | (bb.y < reach_lo1) | (bb.z > reach_hi2) | (bb.w < reach_lo2));
                    uint64_t m = ballot64(reach);
                    while (m) {
                        const int w = wb + __builtin_ctzll(m);
                        m &= m - 1;
                        fn(w);
                    }
                }
                done_w = wlast > done_w ? wlast : done_w;
            }
        }
    };
    // the five squared distances of this lane's sample of word w (the rows' alignment pads lie 1e15 away: device_tables.FAR)
    auto distances = [&](int w, double dd[PAINT_PER_ACTION]) {
        const int s = (w << 6) + lane;
        const double x = ldg(P.samp[0], s), y = ldg(P.samp[1], s), z = ldg(P.samp[2], s);
#pragma unroll
        for (int k = 0; k < PAINT_PER_ACTION; ++k) {
            const double dx = cen_lds[3 * k] - x, dy = cen_lds[3 * k + 1] - y, dz = cen_lds[3 * k + 2] - z;
            dd[k] = (dx * dx + dy * dy) + dz * dz;
        }
    };
    // pass 0: each shot's largest distance
    double dmax_l[PAINT_PER_ACTION] = {-1.0, -1.0, -1.0, -1.0, -1.0};
    for_words([&](int w) {
        double dd[PAINT_PER_ACTION];
        distances(w, dd);
#pragma unroll
        for (int k = 0; k < PAINT_PER_ACTION; ++k) dmax_l[k] = (dd[k] <= r2) & (dd[k] > dmax_l[k]) ? dd[k] : dmax_l[k];
    });
    // (the five r's and 1 / r^2's wait in LDS: twenty scalar registers across the word loop were spilled to vector lanes)
#pragma unroll
    for (int k = 0; k < PAINT_PER_ACTION; ++k) {
        const double r = sqrt(wave_max_d(dmax_l[k]));             // (no hit: NaN, and no deposit either)
        if (lane == 0) {
            scratch[k] = r;
            scratch[PAINT_PER_ACTION + k] = 1.0 / (r * r);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // pass 1: the deposits, word by word
    double succ_l = 0.0;
    int pix = 0;
    for_words([&](int w) {
        double dd[PAINT_PER_ACTION];
        distances(w, dd);
        uint64_t b[PAINT_PER_ACTION], any = 0;
#pragma unroll
        for (int k = 0; k < PAINT_PER_ACTION; ++k) {
            b[k] = ballot64(dd[k] <= r2);
            any |= b[k];
        }
        if (any == 0) return;                        // (its last-shot word, if set, is cleared with the words not visited)
        uint64_t pw_old, lw_old;
        words.get(w, pw_old, lw_old);
        if constexpr (Words::HBM) {                  // (loaded by every lane from one address: wave-uniform, to scalar registers)
            pw_old = uni_u64(pw_old);
            lw_old = uni_u64(lw_old);
        }
        const int s = (w << 6) + lane;
        const bool mine = (any >> lane) & 1;
        const uint8_t v_old = mine ? thick[s] : (uint8_t)0;
        uint8_t v = v_old;
        uint64_t pw = pw_old, uw = 0, prev = lw_old;
#pragma unroll
        for (int k = 0; k < PAINT_PER_ACTION; ++k) {
            if (b[k]) {
                const bool hit = (b[k] >> lane) & 1;
                // quantity = int(25 (1 - (d / r)^2)) + 1 with d = sqrt(dd): a square root and a division per hit.  25 (1 - dd / r^2)
                // with the shot's 1 / r^2 is the same number to a few units in the last place (< 1e-13 absolute); its integer part
                // can only differ where it lies within that of an integer -- those lanes (the farthest sample, a sample on the
                // centre, a NaN) take the reference's own expression
                const double t = 25 * (1 - dd[k] * scratch[PAINT_PER_ACTION + k]);
                const double fr = t - floor(t);
                int quantity = (int)t + 1;
                const bool exact = hit && !((fr > 1.0e-9) & (fr < 1.0 - 1.0e-9));
                if (ballot64(exact) != 0) {
                    if (exact) {
                        const double q = sqrt(dd[k]) / scratch[k];
                        quantity = (int)(25 * (1 - q * q)) + 1;
                    }
                }
                if (hit && v != 0) {
                    v = (uint8_t)(v - quantity);
                    succ_l += quantity / 255.0;
                }
                pw = (pw & ~b[k]) | ballot64(hit && v == 255);
            }
            uw |= b[k] & ~prev;
            prev = b[k];
        }
        if (mine && v != v_old) thick[s] = v;
        pix += (int)__popcll(uw);
        words.put(w, pw, prev, pw_old, lw_old);
    });
    pixel_counter = pix;
    succeeded = wave_sum_d(succ_l);
}

// masks in registers (parts of up to 16 384 samples)
template <int KW>
__device__ void paint_shots_hsi(PartRef P, double radius, const double *cen_lds, int lane, uint64_t painted[KW_MAX],
                                uint64_t last[KW_MAX], uint8_t *thick, double &succeeded, int &pixel_counter, double *scratch) {
    uint64_t new_last[KW_MAX] = {0, 0, 0, 0};
    paint_shots_hsi_words(P, radius, cen_lds, lane, RegWords<KW>{painted, last, new_last, lane}, thick, succeeded, pixel_counter, scratch);
#pragma unroll
    for (int k = 0; k < KW; ++k) last[k] = new_last[k];
}

// COLOR_MODE 'HSI' under the cone beams (bpw:419-434; oracle/paint_oracle.c apply_paint_hsi_list): the deposits of ONE shot's
// hit list -- every ENTRY deposits on its sample, a sample under k beams receives k deposits of the same quantity, each unless
// the byte is 0 at that moment.  `list`: the shot's hit list in LDS (n_beams entries, -1 = no hit), up to 64 trips of 64.
// Until round 5 the first entry of a sample counted its multiplicity by reading the whole list: quadratic in the beams (a
// 70 654-sample part casts 772 a shot: 22.8 ms a step).  Here in rounds: the pending entries race for a bit of a hashed set
// (16 384 bits, cleared per round); a winner makes ONE deposit and retires, a loser -- an entry of the same sample, or of
// another that hashes alike -- waits for the next round.  A sample gets at most one deposit a round and as many rounds as it
// has entries: the same bytes.  Rounds = the largest multiplicity (+ the odd collision); used from HSI_ROUNDS_FROM beams a shot.  Then the status bits (byte == 255)
// through each sample's first entry, found with the hit row's own atomicOr.  `row`, `stat`: the shot's hit / status rows in LDS.
constexpr int HSI_HASH_WORDS = 256;
constexpr int HSI_ROUNDS_FROM = 256;      // beams a shot from which the rounds pay (the door's 104: 253 us a step counting, 404 in rounds --
                                          // few samples under many beams: ten rounds; 772 beams at 70 654 samples: 22.8 -> 6.6 ms)
__device__ __forceinline__ void hsi_list_deposits(PartRef P, const int *list, int n_beams, int lane, const double c[3], double rmax,
                                                  uint8_t *thick, uint64_t *row, uint64_t *stat, uint64_t *hash, double &succ_l) {
    auto sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    uint64_t pending = 0, rep = 0;
    {
        int t = 0;
        for (int b0 = 0; b0 < n_beams; b0 += 64, ++t) {
            const int sidx = list[b0 + lane];
            if (sidx >= 0) {
                pending |= 1ull << t;
                const unsigned long long bit = 1ull << (sidx & 63);
                const unsigned long long old = atomicOr(reinterpret_cast<unsigned long long *>(&row[sidx >> 6]), bit);
                if (!(old & bit)) rep |= 1ull << t;
            }
        }
    }
    while (ballot64(pending != 0) != 0) {
        for (int i = lane; i < HSI_HASH_WORDS; i += 64) hash[i] = 0;
        sync();
        int t = 0;
        for (int b0 = 0; b0 < n_beams; b0 += 64, ++t) {
            const bool mine = (pending >> t) & 1;
            if (ballot64(mine) == 0) continue;
            const int sidx = list[b0 + lane];
            if (mine) {
                const uint32_t h = ((uint32_t)sidx * 2654435761u) >> 18;      // 14 bits
                const unsigned long long bit = 1ull << (h & 63);
                const unsigned long long old = atomicOr(reinterpret_cast<unsigned long long *>(&hash[h >> 6]), bit);
                if (!(old & bit)) {
                    pending &= ~(1ull << t);
                    const double dx = c[0] - ldg(P.samp[0], sidx), dy = c[1] - ldg(P.samp[1], sidx), dz = c[2] - ldg(P.samp[2], sidx);
                    const double dd = (dx * dx + dy * dy) + dz * dz;
                    const double q = sqrt(dd) / rmax;
                    const int quantity = (int)(25 * (1 - q * q)) + 1;
                    const uint8_t v = thick[sidx];
                    if (v != 0) {
                        thick[sidx] = (uint8_t)(v - quantity);
                        succ_l += quantity / 255.0;
                    }
                }
            }
        }
        sync();                                      // (a sample's next deposit reads this one's byte: same wave, in order)
    }
    int t = 0;
    for (int b0 = 0; b0 < n_beams; b0 += 64, ++t) {
        if ((rep >> t) & 1) {
            const int sidx = list[b0 + lane];
            if (thick[sidx] == 255) atomicOr(reinterpret_cast<unsigned long long *>(&stat[sidx >> 6]), 1ull << (sidx & 63));
        }
    }
}

}  // namespace
