// prl_paint.hpp -- ball-query painting: one shot, and the five shots of a step together (bpw:568-577).
// Part of the single translation unit paintrl_hip.hip (device code, anonymous namespace); see that
// file for the overall design.  Compile with -ffp-contract=off.
#pragma once

namespace {

// ---------------------------------------------------------------- bpw:568-570 fast_paint (ball query)
template <int KW>
__device__ __forceinline__ void set_word(uint64_t cur[KW_MAX], int w, uint64_t b, int lane) {
    const int owner = w & 63, slot = w >> 6;
#pragma unroll
    for (int k = 0; k < KW; ++k)
        if (k == slot && lane == owner) cur[k] |= b;
}

template <int KW>
__device__ void ball_query_wave(PartRef P, double radius, const double c[3], int lane,
                                uint64_t cur[KW_MAX]) {
    const double r2 = radius * radius;
    const double c1 = sel3(c[0], c[1], c[2], P.a1), c2 = sel3(c[0], c[1], c[2], P.a2);
    const int icx = cell_coord(c1, P.sg_o1, P.sg_inv, P.sg_nx), icy = cell_coord(c2, P.sg_o2, P.sg_inv, P.sg_ny);
    const Rows3 R = grid_rows3(P.sg_start, P.sg_nx, P.sg_ny, icx, icy, lane);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int begin = R.begin[r], end = begin + R.count[r];
        if (R.count[r] <= 0) continue;
        const int wlast = (end - 1) >> 6;
        for (int w = begin >> 6; w <= wlast; w += 2) {            // two words per trip: six loads in flight
            const int s0 = (w << 6) + lane, s1 = s0 + 64;
            const bool two = w + 1 <= wlast;
            const double x0 = ldg(P.samp[0], s0), y0 = ldg(P.samp[1], s0), z0 = ldg(P.samp[2], s0);
            double x1 = 0, y1 = 0, z1 = 0;
            if (two) {
                x1 = ldg(P.samp[0], s1);
                y1 = ldg(P.samp[1], s1);
                z1 = ldg(P.samp[2], s1);
            }
            {
                const double dx = x0 - c[0], dy = y0 - c[1], dz = z0 - c[2];
                const double dd = (dx * dx + dy * dy) + dz * dz;
                const uint64_t b = __ballot(s0 >= begin && s0 < end && dd <= r2);
                if (b) set_word<KW>(cur, w, b, lane);
            }
            if (two) {
                const double dx = x1 - c[0], dy = y1 - c[1], dz = z1 - c[2];
                const double dd = (dx * dx + dy * dy) + dz * dz;
                const uint64_t b = __ballot(s1 >= begin && s1 < end && dd <= r2);
                if (b) set_word<KW>(cur, w + 1, b, lane);
            }
        }
    }
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

template <int KW>
__device__ bool paint_shots_union(PartRef P, double radius, const double *cen_lds, int lane,
                                  uint64_t painted[KW_MAX],
                                  const uint64_t last[KW_MAX], uint64_t new_last[KW_MAX], int &succeeded,
                                  int &pixel_counter) {
    const double r2 = radius * radius;
    // the float64 centres are only needed for the cell ranges and by the rare float64 branch, which reads them
    // from LDS again: fifteen doubles held across the word loop would be thirty vector registers
    float cf[PAINT_PER_ACTION][3];
    int cx_lo = 0x7fffffff, cx_hi = -0x7fffffff, cy_lo = 0x7fffffff, cy_hi = -0x7fffffff;
#pragma unroll
    for (int k = 0; k < PAINT_PER_ACTION; ++k) {
        const double c0 = cen_lds[3 * k], c1 = cen_lds[3 * k + 1], c2 = cen_lds[3 * k + 2];
        cf[k][0] = (float)c0;
        cf[k][1] = (float)c1;
        cf[k][2] = (float)c2;
        const int icx = cell_coord(sel3(c0, c1, c2, P.a1), P.sg_o1, P.sg_inv, P.sg_nx);
        const int icy = cell_coord(sel3(c0, c1, c2, P.a2), P.sg_o2, P.sg_inv, P.sg_ny);
        cx_lo = icx < cx_lo ? icx : cx_lo;
        cx_hi = icx > cx_hi ? icx : cx_hi;
        cy_lo = icy < cy_lo ? icy : cy_lo;
        cy_hi = icy > cy_hi ? icy : cy_hi;
    }
    double band = 16.0 * radius * 1.1920928955078125e-07 * (P.samp_absmax + radius) + 9.5367431640625e-07 * r2;
#ifdef PRL_WIDE_PAINT_BAND
    band *= 4096.0;
#endif
    // thresholds rounded outward, so that the float comparisons are at least as cautious as the double ones
    const float r2_in = nextafterf((float)(r2 - band), -INFINITY), r2_out = nextafterf((float)(r2 + band), INFINITY);
    const f32x4 GAS *s4 = reinterpret_cast<const f32x4 GAS *>(P.samp_f32);
#ifdef PRL_FORCE_PER_SHOT_PAINT                     // diagnostic build: exercise the general path in the parity tests
    return false;
#endif
    if (cy_hi - cy_lo > 1) return false;            // centres spread over > 2 cell rows: caller paints shot by shot
    // rows cy_lo-1 .. cy_hi+1 (<= 4), columns cx_lo-1 .. cx_hi+1: lanes 0..7 fetch the range bounds
    const int cx0 = cx_lo - 1 < 0 ? 0 : cx_lo - 1, cx1 = cx_hi + 1 > P.sg_nx - 1 ? P.sg_nx - 1 : cx_hi + 1;
    const int rcy = cy_lo - 1 + (lane >> 1);
    const bool ok = lane < 8 && rcy <= cy_hi + 1 && rcy >= 0 && rcy < P.sg_ny && cx0 <= cx1;
    const int bound = ok ? ldg(P.sg_start, rcy * P.sg_nx + ((lane & 1) ? cx1 + 1 : cx0)) : 0;
    int rb[4], re[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        rb[r] = __builtin_amdgcn_readlane(bound, 2 * r);
        re[r] = __builtin_amdgcn_readlane(bound, 2 * r + 1);
    }
    int done_w = -1;                                 // a word shared by two rows' ranges is handled once
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (re[r] <= rb[r]) continue;
        const int wlast = (re[r] - 1) >> 6;
        for (int w = (rb[r] >> 6) > done_w ? (rb[r] >> 6) : done_w + 1; w <= wlast; ++w) {
            WCNT(5, 1);
            const int s = (w << 6) + lane;
            const f32x4 pf = ldg(s4, s);
            // every cell row starts on a word boundary (device_tables), so a word holds samples of one row only
            const bool in = s >= rb[r] && s < re[r];
            uint64_t b[PAINT_PER_ACTION];
            uint64_t any = 0, unsure = 0;
#pragma unroll
            for (int k = 0; k < PAINT_PER_ACTION; ++k) {
                const float dx = pf.x - cf[k][0], dy = pf.y - cf[k][1], dz = pf.z - cf[k][2];
                const float dd = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                b[k] = __ballot(in && dd <= r2_in);
                unsure |= b[k] ^ __ballot(in && dd <= r2_out);
                any |= b[k];
            }
#ifdef PRL_FORCE_F64_PAINT
            unsure = 1;
#endif
            if (unsure) {                           // some sample within rounding reach of the sphere: float64 decides
                WCNT(6, 1);
                const double x = ldg(P.samp[0], s), y = ldg(P.samp[1], s), z = ldg(P.samp[2], s);
                any = 0;
#pragma unroll
                for (int k = 0; k < PAINT_PER_ACTION; ++k) {
                    const double dx = x - cen_lds[3 * k], dy = y - cen_lds[3 * k + 1], dz = z - cen_lds[3 * k + 2];
                    const double dd = (dx * dx + dy * dy) + dz * dz;
                    b[k] = __ballot(in && dd <= r2);
                    any |= b[k];
                }
            }
            done_w = w;
            const int owner = w & 63, slot = w >> 6;
            uint64_t pw = 0, lw = 0;
#pragma unroll
            for (int k = 0; k < KW; ++k)
                if (k == slot) {
                    pw = bcast_u64(painted[k], owner);
                    lw = bcast_u64(last[k], owner);
                }
            if (any == 0 && lw == 0) continue;       // nothing to record for this word
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
#pragma unroll
            for (int k = 0; k < KW; ++k)
                if (k == slot && lane == owner) {
                    painted[k] = pw;
                    new_last[k] = lw;
                }
        }
    }
    return true;
}

}  // namespace
