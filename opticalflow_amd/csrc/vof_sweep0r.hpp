// vof_sweep0r.hpp - k_sweep0r: the level-0 smoother pass with the vectors in REGISTERS (round 3).
//
// k_sweep0m (vof_device.hpp) shares an LDS ring of x rows between the four waves of a workgroup: every point update reads its
// 22 neighbour values and 9 image values from LDS and the waves meet at one workgroup barrier per two rows.  Measured in
// round 2: 3.0 TB/s on the bytes the pass has to move, neither the vector issue, nor the LDS reads, nor the load latency,
// nor the halo alone explain it - what is left is the step structure (2 waves per SIMD, each alternating between a
// dependent LDS round trip and a barrier).  k_sweep0r removes the structure instead of tuning it:
//
//  * ONE WAVE owns a strip of 128 columns and runs all 2 NS colour stages of a step itself, in program order.  Lane i holds
//    the column pair (2 i, 2 i + 1) of every live row IN REGISTERS (double2 per field); the left / right neighbours of a
//    point are the lane's own other column or the neighbouring lane's, fetched with a DPP lane shift (v_mov_b32_dpp
//    wave_shr:1 / wave_shl:1, two per double) - no LDS traffic for x, no barrier anywhere, waves are independent.
//  * Because the stages of a step run in program order, sweep k + 1 follows sweep k TWO rows behind (k_sweep0m: six, the
//    price of a barrier between dependent stages): E_k works on row e - 2 k, O_k on row e - 2 k - 1, rows e - 2 NS and
//    e - 2 NS + 1 are final and written out.  The register window is rows e - 2 NS .. e + 3 (8 rows for two sweeps per pass).
//  * Rows e + 2, e + 3 are requested at the START of step e straight into the top of the window and first touched by the
//    rotation at its END (a full step of latency slack); the b rows of next step's stages are requested as soon as this
//    step's stage has consumed its own (same registers).  In the steady-state steps every load is unconditional, so the
//    compiler's vmcnt waits are exact.
//  * The image rows go through a small wave-private LDS ring (parity-split halves, 6.3 KB per wave): a lane-shifted LDS
//    address IS the lateral shift, and the image needs three columns per point.
//  * Same update function (gs0_point), same colour order, same dependency cone (one column per colour and side) as every
//    other level-0 smoother: results are bit-identical to the per-colour kernels (the tests compare bits).
//
// Variants as in k_sweep0m: FROM_ZERO (first pre-smoothing pass), EC (x_in + P e: the coarse-grid correction interpolated
// into the rows as they enter the window), TRAIL = 1 (v = A x_out with the Krylov dot products from the rows that have just
// become final; needs one more live row, not four).  Beyond k_sweep0m (second half of round 3):
//  * the diagonal blocks of the first sweep's rows go through a second wave-private LDS ring to the second sweep's stages and to
//    the trailing stage (S0R::DC, gs0_point<.., DC>): a third of an update's FP64 instructions is not repeated;
//  * TRAIL = 2: the trailing stage forms the residual b - A x_out and restricts it - the coarse right-hand side leaves the
//    pre-smoothing pass, the residual + restriction kernel's pass over x, b and the image is gone;
//  * BF: the pass from zero forms its own right-hand side (s = r - alpha v with (s, s), or p = r + beta (p - omega v): the
//    BiCGStab updates that would have written it) or reads it once, and hands the rows from the first sweep's stages to the
//    second's in registers.
#pragma once
#include "vof_device.hpp"

namespace vof {

// value of lane - 1 (lane 0: 0) / lane + 1 (lane 63: 0).  mov_dpp with bound_ctrl: lanes without a source read 0, and the
// destination needs no initial value (update_dpp with old = 0 costs a v_mov per half).
__device__ __forceinline__ double lane_shr1(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x138, 0xF, 0xF, true);   // wave_shr:1
    hi = __builtin_amdgcn_mov_dpp(hi, 0x138, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_shl1(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x130, 0xF, 0xF, true);   // wave_shl:1
    hi = __builtin_amdgcn_mov_dpp(hi, 0x130, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

#ifndef VOF_S0R_DCACHE
#define VOF_S0R_DCACHE 1
#endif
#ifndef VOF_S0R_BRING
#define VOF_S0R_BRING 1         // post-smoothing pass: b handed from sweep to sweep through LDS (S0R::BL); 0: read once per sweep
#endif
#ifndef VOF_S0R_BL_DC
#define VOF_S0R_BL_DC 1         // ... and a four-row diagonal-block ring beside it (0: no diagonal-block ring in that pass)
#endif
#ifndef VOF_S0R_BCARRY
#define VOF_S0R_BCARRY 1        // two-sweep pass from zero: b is read once and handed on in registers (BF = 3); 0: read once per sweep
                                // (the passes with one wave per SIMD lose 5 % with it - measured -, so only that one)
#endif
#ifndef VOF_S0R_FZ_WAVES
#define VOF_S0R_FZ_WAVES 2      // waves per SIMD of the pass from zero
#endif
template <int NS, int TRAIL = 0> struct S0R {
    static constexpr int EXT = TRAIL ? 1 : 0;
    static constexpr int LO = 2 * NS + EXT;          // lowest live row of step e: e - LO (TRAIL: the row above the last final one)
    static constexpr int NRW = LO + 4;               // register window: rows e - LO .. e + 3 (the top two in flight)
    static constexpr int HALO = 4 * NS + 2 * EXT, OUT = S0_W - 2 * HALO;
    static constexpr int IPW = 66;                   // doubles per parity half of an image row (65 used: full columns 0 .. 129)
    static constexpr int IRB = 2 * IPW * 8;          // bytes per image ring row
    static constexpr int NRI = LO + 2;               // image ring rows: e - LO .. e + 1; rows e + 2, e + 3 replace the two oldest
    static constexpr int LDS_BYTES = NRI * IRB;      // the image ring (all k_sweep0p uses)
    // the diagonal blocks of the first sweep's rows, handed on to the second sweep (and the trailing product) through LDS: rows
    // e - 2 NS + 1 - EXT .. e = 2 NS + EXT slots (slot = row modulo their number); a row = 4 values x 64 lanes x 16 B
    // BL (post-smoothing pass, TRAIL = 1): the rows of b go from the first sweep's stages to the second's through LDS instead of
    // being read again (two steps x two row parities = four slots of 3 x 64 x 16 B).  With the full diagonal-block ring beside it
    // (40 160 B per wave) only three waves fit a CU and the pass loses 8 % (measured); so there the ring holds four rows - enough
    // for the second sweep's stages - and the trailing stage computes its diagonal blocks itself (DCT)
    static constexpr bool BL = (VOF_S0R_BRING != 0) && NS == 2 && TRAIL == 1;
    static constexpr bool DC = (VOF_S0R_DCACHE != 0) && NS == 2 && (!BL || VOF_S0R_BL_DC);
    static constexpr bool DCT = DC && !BL;            // the trailing stage takes the diagonal blocks out of the ring too
    static constexpr int ND = DC ? 2 * NS + (DCT ? EXT : 0) : 0;
    static constexpr int DRB = 4 * 64 * 16;
    static constexpr int NBL = BL ? 4 : 0, BRB = 3 * 64 * 16;
    static constexpr int LDS_TOTAL = LDS_BYTES + ND * DRB + NBL * BRB;
};

struct S0RRow { double2 u, w, g; };   // one x row of the strip: .x = column 2 lane, .y = column 2 lane + 1

// PO: 0 = forward colour order 0, 1, 2, 3; 1 = reverse order (rows shifted by one, odd columns first) - a template parameter
// so that the column parity of a phase is a compile-time constant (one code path per phase)
// QK: the reference's 'dy' == 'dx' quirk (OF.py:698-699) as a compile-time constant (the select costs four instructions per point)
// BF: the right-hand side of a pass from zero is FORMED here instead of read - the BiCGStab vector update that would have written
// it is folded into the pass (S0BSrc): 1: s = r - alpha v (+ the block partial sums of (s, s)), 2: p = r + beta (p_old - omega v).
// Stages 0 / 1 (first sweep) read the operand rows, form b, store the owned part and hand the row on to stages 2 / 3 (second
// sweep, one step later) in registers, so b is neither written and re-read nor read twice.  One wave per SIMD (register budget).
template <int NS, bool EC, bool FROM_ZERO, int TRAIL, typename ET, int PO, int QK = 1, int BF = 0>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu((FROM_ZERO && (BF == 0 || BF == 3) && TRAIL != 2) ? VOF_S0R_FZ_WAVES : 1, (FROM_ZERO && (BF == 0 || BF == 3) && TRAIL != 2) ? VOF_S0R_FZ_WAVES : 1))) void k_sweep0r(
    Fine0 pol, int ni, int nj, int TI, int /*po*/, int nx, int ny, int nz, const double* __restrict__ x_in,
    double* __restrict__ x_out, const double* __restrict__ b, const int* __restrict__ active,
    const ET* __restrict__ ecoarse, int nci, int ncj, S0Trail tr, int skip_first = 0, int skip_count = 0,
    S0BSrc bsrc = S0BSrc{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}) {
    static_assert(BF == 0 || NS == 2, "b is handed from the first sweep's stages to the second's");
    static_assert(BF == 0 || BF == 3 || (FROM_ZERO && !EC && TRAIL != 1 && PO == 0), "the folded vector update belongs to the first pre-smoothing pass");
    // TRAIL = 2: the trailing stage forms the RESIDUAL b - A x_out of the rows that have just become final and restricts it (full
    // weighting, R = P^T / 4, as k_stream_resrestrict0) - the coarse right-hand side comes out of the pre-smoothing pass and the
    // residual + restriction kernel's pass over x, b and the image is gone.  ET = type of the coarse right-hand side, written to tr.v.
    static_assert(TRAIL != 2 || (FROM_ZERO && !EC && PO == 0 && BF != 0), "the residual + restriction stage belongs to the pre-smoothing pass");
    constexpr bool RR = TRAIL == 2;
    // (skip_first, skip_count: the strips [skip_first, skip_first + skip_count) belong to another launch - k_sweep0p takes the
    // interior strips in its mode -; nx counts the strips of THIS launch)
    typedef S0R<NS, TRAIL> G;
    constexpr int W = S0_W, LO = G::LO, NRW = G::NRW, NRI = G::NRI, IRB = G::IRB, IHB = G::IPW * 8, EXT = G::EXT;
    constexpr int NST = 2 * NS, po = PO;                                  // stages per step: E_0, O_0, E_1, O_1, ...: stage st works on row e - st
    constexpr bool DC = G::DC;
    constexpr int ND = G::ND, DRB = G::DRB;
    extern __shared__ double sw_lds[];
    char* iring = reinterpret_cast<char*>(sw_lds);
    char* dring = iring + G::LDS_BYTES;
    char* bring = dring + ND * DRB;
    constexpr bool BL = G::BL && BF == 0;
    int bph = 0;                                                            // BL: this step's slot pair (toggles every step)
    const unsigned nblocks = (unsigned)nx * ny * nz;
    unsigned lb = blockIdx.x;
    if ((nblocks & 7u) == 0) lb = (lb & 7u) * (nblocks >> 3) + (lb >> 3);   // XCD-aware remap, see k_sweep
    const int bxl = lb % nx, by = (lb / nx) % ny;
    const int bx = bxl < skip_first ? bxl : bxl + skip_count;
    const int pair = lb / (nx * ny);
    if (active && !active[pair]) return;
    const int lane = threadIdx.x;
    const int p0 = by * TI - po;                 // true row of relative row 0 (reverse order: rows shifted by one)
    const int qs = bx * G::OUT - G::HALO;        // true column of local column 0 (even: 16-byte aligned pairs)
    const size_t npts = (size_t)ni * nj, off = (size_t)pair * 3 * npts;
    const double* xin = FROM_ZERO ? nullptr : x_in + off;
    double* xout = x_out + off;
    const double* bp = b + off;
    const size_t ncpts = (size_t)nci * ncj;
    const ET* ec = EC ? ecoarse + (size_t)pair * 3 * ncpts : nullptr;
    double alpha = pol.alpha, beta = pol.beta;
    int fidx = pair;
    if (pol.pp) { alpha = pol.pp[pair].alpha; beta = pol.pp[pair].beta; fidx = pol.pp[pair].frame; }
    const double* img = pol.frames + (size_t)fidx * pol.frame_stride;
    const int Nj = pol.Nj;
    constexpr int quirks = QK;
    const double inv_g = 1.0 / (-1 - 4 * beta);

    // lane <-> column pair (2 lane, 2 lane + 1); pair validity is all-or-nothing (qs and nj are even)
    const int qpair = qs + 2 * lane;
    const bool pair_ok = qpair >= 0 && qpair + 1 < nj;
    const size_t qg = pair_ok ? (size_t)qpair : 0;
    const bool ipair_ok = qpair >= 0 && qpair + 1 <= nj + 1;               // image columns qs + 2 lane, + 1 (full image)
    const size_t iqg = ipair_ok ? (size_t)qpair : 0;
    const int xq = qs + 2 * 64;                                            // image pair 64 (strip columns 128, 129): lane 0 only
    const bool xpair_ok = lane == 0 && xq >= 0 && xq + 1 <= nj + 1;
    const bool st_ok = pair_ok && lane >= G::HALO / 2 && lane < (W - G::HALO) / 2;   // owned column pairs
    // ghost columns (mirror boundary rows, OF.py:1053-1070): column -1 mirrors column 1, column n_j mirrors n_j - 2; only an
    // even column can be the first and (n_j even) only an odd one the last
    const bool glE = qpair == 0, grO = qpair + 1 == nj - 1;
    const bool border_strip = qs < 0 || qs + W > nj;                      // wave-uniform: strips away from the image's sides skip the selects
    const int cqs = qs >> 1;                                              // coarse column of lane 0 (EC)

    // ---- the register window: X[j] = row e - LO + j
    S0RRow X[NRW];
#pragma unroll
    for (int j = 0; j < NRW; ++j) X[j].u = X[j].w = X[j].g = double2{0.0, 0.0};
    double2 B[NST][3];                                                     // b of the stage's row (requested one step ahead)
    double2 BV[2][3], BQ[2][3];                                            // BF: the v (and p_old) rows of stages 0, 1
    double2 BC[2][2][3];                                                   // BF: b rows handed on: [this / next step's][stage 0 / 1][field]
    double bss = 0.0, bcA = 0.0, bcB = 0.0;
    const double* bfr = nullptr; const double* bfv = nullptr; const double* bfq = nullptr; double* bfo = nullptr;
    if constexpr (BF == 3) bfr = bp;     // 3: b itself, read once (no update folded in) and handed on like the formed rows
    if constexpr (BF != 0) {
        if (BF != 3) { bfr = bsrc.r + off; bfv = bsrc.v + off; bfo = bsrc.out + off; }
        if (BF == 1) bcA = bsrc.sc[pair].alpha;
        if (BF == 2) { bfq = bsrc.p_old + off; bcA = bsrc.sc[pair].beta; bcB = bsrc.sc[pair].omega; }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int f = 0; f < 3; ++f) { BV[a][f] = BQ[a][f] = BC[0][a][f] = BC[1][a][f] = double2{0.0, 0.0}; }
    }
    double2 li[2] = {{0, 0}, {0, 0}}, lix[2] = {{0, 0}, {0, 0}};           // image rows e + 2, e + 3 in flight
    ET crv[3] = {0, 0, 0};                                                 // EC: the coarse row in flight
    double CR[2][3];                                                       // EC: coarse rows (cp0, cp0 + 1) of the rows entering the window
    double2 tn[2][3];                                                      // TRAIL: dot partner of the next step's two rows
    double2 BCo[3];                                                        // RR: b of row e - 4 (stage 2's of the previous step)
    double hprev[3] = {0.0, 0.0, 0.0};                                     // RR: column-restricted residual of row e - 5
    double ts0 = 0.0, ts1 = 0.0;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int f = 0; f < 3; ++f) { CR[r][f] = 0.0; tn[r][f] = double2{0.0, 0.0}; BCo[f] = double2{0.0, 0.0}; }

    // row ranges of the stages (as k_sweep0m): E_k rows [-2 m, TI + 2 m], O_k rows [-2 m + 1, TI + 2 m - 1], m = NS - 1 - k + EXT
    auto st_lo = [](int st) { const int k = st >> 1, odd = st & 1; return -2 * (NS - 1 - k + EXT) + odd; };
    auto st_hi = [TI](int st) { const int k = st >> 1, odd = st & 1; return TI + 2 * (NS - 1 - k + EXT) - odd; };

    // first step: e = -2 (NS + EXT) - 2 (requests rows -2 (NS + EXT), + 1); last step: writes out row TI - 1 = e - 2 NS + 1
    const int s_first = -(NS + EXT) - 1, s_last = (TI + 2 * NS - 2) / 2;
    const int e0 = 2 * s_first;
    // steady-state steps: every row any part of the step touches exists, is an interior row and lies in its stage's range
    int e_lo = 2 * NS, e_hi = TI + 2 * (NS + EXT) - 4;                      // write-out rows >= 0; rows e + 2, e + 3 are loaded
    e_lo = max(e_lo, 2 * NS - p0);  e_hi = min(e_hi, ni - 4 - p0);         // stores / loads inside the image
    e_hi = min(e_hi, TI + 2 * NS - 2);                                      // write-out rows < TI
#pragma unroll
    for (int st = 0; st < NST; ++st) {
        e_lo = max(e_lo, max(st_lo(st) + st, 1 - p0 + st));
        e_hi = min(e_hi, min(st_hi(st) + st - 2, ni - 4 - p0 + st));       // (- 2: the b row of the next step is requested too)
    }
    if (TRAIL) {   // the rows the operator is applied to are interior rows; the dot partner's rows two further exist
        e_lo = max(e_lo, 2 * NS + 1 - p0);
        e_hi = min(e_hi, min(TI + 2 * NS - 4, ni - 5 - p0 + 2 * NS));
    }
    if (EC) { e_lo = max(e_lo, -p0); e_hi = min(e_hi, 2 * nci - 8 - p0); }

    int islot = 0;                                                          // image ring slot of row e - LO
    auto irow = [&](int j) { int s = islot + j; if (s >= NRI) s -= NRI; return iring + s * IRB; };   // ring row of window row j (j < NRI)
    int dslot = 0;                                                          // diagonal-block ring slot of row e
    auto drow = [&](int k) { int s = dslot - k; if (s < 0) s += ND; return dring + s * DRB + lane * 16; };   // ... of row e - k

    // ---- requests: rows e + 2, e + 3 of x and of the image, the coarse row they need next
    auto request_rows = [&](auto edge_tag, int e) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        const bool do_load = EDGE ? (e + 3 <= TI + 2 * (NS + EXT) - 1) : true;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int pL = p0 + e + 2 + r;
            if (FROM_ZERO) X[LO + 2 + r].u = X[LO + 2 + r].w = X[LO + 2 + r].g = double2{0.0, 0.0};
            if (!FROM_ZERO) {
                const bool rowok = EDGE ? (do_load && pL >= 0 && pL < ni) : true;
                S0RRow& d = X[LO + 2 + r];
                if (EDGE) d.u = d.w = d.g = double2{0.0, 0.0};
                if (EDGE ? (rowok && pair_ok) : true) {   // (steady state = interior strip: every lane's pair exists)
                    const double* src = xin + (size_t)pL * nj + qg;
                    d.u = *reinterpret_cast<const double2*>(src);
                    d.w = *reinterpret_cast<const double2*>(src + npts);
                    d.g = *reinterpret_cast<const double2*>(src + 2 * npts);
                }
            }
            {
                const int pI = pL + 1;                                      // full-image row of interior row pL
                const bool rowok = EDGE ? (do_load && pI >= 0 && pI <= ni + 1) : true;
                if (EDGE) { li[r] = double2{0.0, 0.0}; lix[r] = double2{0.0, 0.0}; }
                if (rowok) {
                    const double* frow = img + (size_t)pI * Nj;
                    if (EDGE ? ipair_ok : true) li[r] = *reinterpret_cast<const double2*>(frow + iqg);
                    if (EDGE ? xpair_ok : true) lix[r] = *reinterpret_cast<const double2*>(frow + xq);   // (steady state: every lane reads the same pair)
                }
            }
        }
        if constexpr (EC) {   // coarse rows of the fine rows e + 2, e + 3: (pL >> 1), + 1; the upper one is new
            const int knew = ((p0 + e + 2) >> 1) + 2;   // (this step's rows use (pL0 >> 1), + 1: in CR already)
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                crv[f] = 0;
                const int c0 = cqs + lane;
                if (EDGE ? (knew >= 0 && knew < nci && c0 >= 0 && c0 < ncj) : true) crv[f] = ec[(size_t)f * ncpts + (size_t)knew * ncj + c0];
            }
        }
    };

    // b of stage st for the row it handles at step e_next (requested after the stage has consumed this step's)
    auto request_b = [&](auto edge_tag, auto st_tag, int e_next) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        constexpr int st = decltype(st_tag)::value;
        const int rr = e_next - st, p = p0 + rr;
        const bool rowok = EDGE ? (rr >= st_lo(st) && rr <= st_hi(st) && p >= 0 && p < ni) : true;
        if constexpr (BF != 0) {
            if constexpr (st < 2) {   // operand rows of the folded update (stages 2, 3 get their b handed on)
                if (EDGE) {
#pragma unroll
                    for (int f = 0; f < 3; ++f) B[st][f] = BV[st][f] = BQ[st][f] = double2{0.0, 0.0};
                }
                if (EDGE ? (rowok && pair_ok) : true) {
                    const size_t ro = (size_t)p * nj + qg;
#pragma unroll
                    for (int f = 0; f < 3; ++f) {
                        B[st][f] = *reinterpret_cast<const double2*>(bfr + ro + f * npts);
                        if (BF != 3) BV[st][f] = *reinterpret_cast<const double2*>(bfv + ro + f * npts);
                        if (BF == 2) BQ[st][f] = *reinterpret_cast<const double2*>(bfq + ro + f * npts);
                    }
                }
            }
        } else if constexpr (!(BL && st >= 2)) {   // (BL: stages 2, 3 take their rows out of LDS)
            if (EDGE) B[st][0] = B[st][1] = B[st][2] = double2{0.0, 0.0};
            if (EDGE ? (rowok && pair_ok) : true) {
                const double* brow = bp + (size_t)p * nj + qg;
                B[st][0] = *reinterpret_cast<const double2*>(brow);
                B[st][1] = *reinterpret_cast<const double2*>(brow + npts);
                B[st][2] = *reinterpret_cast<const double2*>(brow + 2 * npts);
            }
        }
    };

    // ---- one colour stage: the two colours of window row jc (relative row rr), first the columns of true parity po
    auto stage = [&](auto edge_tag, auto border_tag, auto jc_tag, int rr, const double2 (&bs)[3]) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        constexpr bool BORDER = decltype(border_tag)::value;   // the strip touches the left / right side of the image
        constexpr int jc = decltype(jc_tag)::value;
        const int p = p0 + rr;
        const bool oU = EDGE && p - 1 < 0, oD = EDGE && p + 1 >= ni;
        // ghost row -1 mirrors row 1, ghost row n_i mirrors row n_i - 2 (the image has real border rows: no folding there)
        S0RRow RU = X[jc - 1], RD = X[jc + 1];
        if (EDGE) {
            if (oU) RU = X[jc + 1];
            if (oD) RD = X[jc - 1];
        }
        const double2 bs0 = bs[0], bs1 = bs[1], bs2 = bs[2];
        const char* iu = irow(jc - 1);
        const char* ic = irow(jc);
        const char* id = irow(jc + 1);
        auto LD = [](const char* r, int o) { return *reinterpret_cast<const double*>(r + o); };
        // image columns (full image) 2 lane .. 2 lane + 3 of the three rows: E[lane], O[lane], E[lane + 1], O[lane + 1]
        const int ie = lane * 8, io = IHB + lane * 8;
        const double iuA = LD(iu, ie), iuB = LD(iu, io), iuC = LD(iu, ie + 8), iuD = LD(iu, io + 8);
        const double icA = LD(ic, ie), icB = LD(ic, io), icC = LD(ic, ie + 8), icD = LD(ic, io + 8);
        const double idA = LD(id, ie), idB = LD(id, io), idC = LD(id, ie + 8), idD = LD(id, io + 8);
        // steady state: the differences below - above of the lane's two columns once per stage (the rows do not change during it);
        // a point's own column gives du71 / dw71, the neighbouring columns' differences - lane-shifted - give U4 / W4
        constexpr int VPRE = EDGE ? 0 : 1;
        double2 dU = {0.0, 0.0}, dW = {0.0, 0.0};
        if (VPRE && !(FROM_ZERO && LO - jc == 0)) {   // (the first stage of a pass from zero has zero rows above and below)
            dU.x = RD.u.x - RU.u.x; dU.y = RD.u.y - RU.u.y;
            dW.x = RD.w.x - RU.w.x; dW.y = RD.w.y - RU.w.y;
        }
        // diagonal blocks of the row's two points: the first sweep's stages store them, the later ones take them over
        constexpr int stg_dc = LO - jc;                                     // stage number
        constexpr int DCM = !DC ? 0 : (stg_dc < 2 ? 1 : 2);
        double2 dq[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};                   // axx, ayy, c, inv; .x / .y = even / odd column
        if constexpr (DCM == 2) {
            const char* dr = drow(stg_dc);
#pragma unroll
            for (int v = 0; v < 4; ++v) dq[v] = *reinterpret_cast<const double2*>(dr + v * 1024);
        }
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            constexpr int par_of_ph[2] = {po, 1 - po};
            const int par = par_of_ph[ph];            // column parity of this phase (compile time after unrolling)
            // a pass FROM ZERO knows which neighbours its first sweep can meet (gs0_point's ZERO): stage 0 = first row colour pair,
            // stage 1 = second; phase 0 / 1 = first / second colour of the row
            constexpr int stg = LO - jc;
            const int zero_of[2][2] = {{1, 2}, {3, 0}};
            const int ZM = (FROM_ZERO && stg < 2) ? zero_of[stg][ph] : 0;
            const bool needUD = ZM == 0 || ZM == 3, needLR = ZM == 0 || ZM == 2;
            S0RRow& RC = X[jc];
            Nbr n = {};       // (entries known to be zero are neither fetched nor read)
            double imv[9];
            bool gl = false, gr = false;
            VDiff0 vd = {0.0, 0.0, 0.0, 0.0};
            if (par == 0) {   // even columns: left neighbour = odd column of lane - 1, right neighbour = the lane's odd column
                if (needUD) {
                    n.u[1] = RU.u.x; n.w[1] = RU.w.x; n.g[1] = RU.g.x;
                    n.u[7] = RD.u.x; n.w[7] = RD.w.x; n.g[7] = RD.g.x;
                    if (VPRE) {
                        vd.du71 = dU.x; vd.dw71 = dW.x;
                        vd.U4 = dU.y - lane_shr1(dU.y); vd.W4 = dW.y - lane_shr1(dW.y);
                    } else {
                        n.u[2] = RU.u.y; n.w[2] = RU.w.y;
                        n.u[8] = RD.u.y; n.w[8] = RD.w.y;
                        n.u[0] = lane_shr1(RU.u.y); n.w[0] = lane_shr1(RU.w.y);
                        n.u[6] = lane_shr1(RD.u.y); n.w[6] = lane_shr1(RD.w.y);
                    }
                }
                if (needLR) {
                    n.u[5] = RC.u.y; n.w[5] = RC.w.y; n.g[5] = RC.g.y;
                    n.u[3] = lane_shr1(RC.u.y); n.w[3] = lane_shr1(RC.w.y); n.g[3] = lane_shr1(RC.g.y);
                }
                if (BORDER && glE) {   // ghost column -1 mirrors column 1
                    gl = true;
                    n.u[0] = n.u[2]; n.w[0] = n.w[2]; n.u[3] = n.u[5]; n.w[3] = n.w[5]; n.g[3] = n.g[5]; n.u[6] = n.u[8]; n.w[6] = n.w[8];
                }
                imv[0] = iuA; imv[1] = iuB; imv[2] = iuC; imv[3] = icA; imv[4] = icB; imv[5] = icC; imv[6] = idA; imv[7] = idB; imv[8] = idC;
            } else {          // odd columns: left neighbour = the lane's even column, right neighbour = even column of lane + 1
                if (needUD) {
                    n.u[1] = RU.u.y; n.w[1] = RU.w.y; n.g[1] = RU.g.y;
                    n.u[7] = RD.u.y; n.w[7] = RD.w.y; n.g[7] = RD.g.y;
                    if (VPRE) {
                        vd.du71 = dU.y; vd.dw71 = dW.y;
                        vd.U4 = lane_shl1(dU.x) - dU.x; vd.W4 = lane_shl1(dW.x) - dW.x;
                    } else {
                        n.u[0] = RU.u.x; n.w[0] = RU.w.x;
                        n.u[6] = RD.u.x; n.w[6] = RD.w.x;
                        n.u[2] = lane_shl1(RU.u.x); n.w[2] = lane_shl1(RU.w.x);
                        n.u[8] = lane_shl1(RD.u.x); n.w[8] = lane_shl1(RD.w.x);
                    }
                }
                if (needLR) {
                    n.u[3] = RC.u.x; n.w[3] = RC.w.x; n.g[3] = RC.g.x;
                    n.u[5] = lane_shl1(RC.u.x); n.w[5] = lane_shl1(RC.w.x); n.g[5] = lane_shl1(RC.g.x);
                }
                if (BORDER && grO) {   // ghost column n_j mirrors column n_j - 2
                    gr = true;
                    n.u[2] = n.u[0]; n.w[2] = n.w[0]; n.u[5] = n.u[3]; n.w[5] = n.w[3]; n.g[5] = n.g[3]; n.u[8] = n.u[6]; n.w[8] = n.w[6];
                }
                imv[0] = iuB; imv[1] = iuC; imv[2] = iuD; imv[3] = icB; imv[4] = icC; imv[5] = icD; imv[6] = idB; imv[7] = idC; imv[8] = idD;
            }
            const double c0 = par ? bs0.y : bs0.x, c1 = par ? bs1.y : bs1.x, c2 = par ? bs2.y : bs2.x;
            double u, w, gm;
            Diag0 dg;
            if (DCM == 2) {
                if (par == 0) { dg.axx = dq[0].x; dg.ayy = dq[1].x; dg.c = dq[2].x; dg.inv = dq[3].x; }
                else { dg.axx = dq[0].y; dg.ayy = dq[1].y; dg.c = dq[2].y; dg.inv = dq[3].y; }
            }
            auto update = [&](auto zero_tag) {
                constexpr int Z = decltype(zero_tag)::value;
                if (EDGE) {
                    const double sUL = (oU && gl) ? 2.0 : 1.0, sUR = (oU && gr) ? 2.0 : 1.0;
                    const double sDL = (oD && gl) ? 2.0 : 1.0, sDR = (oD && gr) ? 2.0 : 1.0;
                    gs0_point<true, Z, DCM>(imv, n, sUL, sUR, sDL, sDR, alpha, beta, inv_g, quirks, c0, c1, c2, u, w, gm, &dg);
                } else {
                    gs0_point<false, Z, DCM, 1>(imv, n, 1.0, 1.0, 1.0, 1.0, alpha, beta, inv_g, quirks, c0, c1, c2, u, w, gm, &dg, &vd);
                }
            };
            if (ZM == 1) update(std::integral_constant<int, 1>{});
            else if (ZM == 2) update(std::integral_constant<int, 2>{});
            else if (ZM == 3) update(std::integral_constant<int, 3>{});
            else update(std::integral_constant<int, 0>{});
            if (par == 0) { RC.u.x = u; RC.w.x = w; RC.g.x = gm; }
            else { RC.u.y = u; RC.w.y = w; RC.g.y = gm; }
            if (DCM == 1) {
                if (par == 0) { dq[0].x = dg.axx; dq[1].x = dg.ayy; dq[2].x = dg.c; dq[3].x = dg.inv; }
                else { dq[0].y = dg.axx; dq[1].y = dg.ayy; dq[2].y = dg.c; dq[3].y = dg.inv; }
            }
        }
        if constexpr (DCM == 1) {
            char* dw = drow(stg_dc);
#pragma unroll
            for (int v = 0; v < 4; ++v) *reinterpret_cast<double2*>(dw + v * 1024) = dq[v];
        }
    };

    // ---- trailing operator stage on window row jc (final, and so are its neighbours): v = A x_out + the dot products
    auto trail_row = [&](auto edge_tag, auto border_tag, auto jc_tag, int rr, auto slot_tag, double2 (&yout)[3]) -> bool {
        constexpr bool EDGE = decltype(edge_tag)::value;
        constexpr bool BORDER = decltype(border_tag)::value;
        constexpr int jc = decltype(jc_tag)::value < 1 ? 1 : decltype(jc_tag)::value;   // (never instantiated below 1 when TRAIL is set)
        constexpr int slot = decltype(slot_tag)::value;
        const int p = p0 + rr;
        const double2 t0 = tn[slot][0], t1 = tn[slot][1], t2 = tn[slot][2];
        if constexpr (!RR) {   // the dot partner of the NEXT step's row
            const bool nrow = EDGE ? (rr + 2 >= 0 && rr + 2 < TI && p + 2 >= 0 && p + 2 < ni) : true;
            if (EDGE) tn[slot][0] = tn[slot][1] = tn[slot][2] = double2{0.0, 0.0};
            // (no dot partner given: the loads read v itself - valid memory, values unused - so that the steady state has no branch)
            const double* dv = tr.dotvec ? tr.dotvec : tr.v;
            if (EDGE ? (tr.dotvec && nrow && st_ok) : true) {
                const double* drow = dv + off + (size_t)(p + 2) * nj + qg;
                tn[slot][0] = *reinterpret_cast<const double2*>(drow);
                tn[slot][1] = *reinterpret_cast<const double2*>(drow + npts);
                tn[slot][2] = *reinterpret_cast<const double2*>(drow + 2 * npts);
            }
        }
        const bool rowok = EDGE ? (rr >= (RR ? -1 : 0) && rr < TI && p >= 0 && p < ni) : true;   // (RR: the coarse row of fine row 0 takes row -1 in)
        if (!rowok) return false;
        const bool oU = EDGE && p - 1 < 0, oD = EDGE && p + 1 >= ni;
        S0RRow RU = X[jc - 1], RD = X[jc + 1];
        const S0RRow RC = X[jc];
        if (EDGE) {
            if (oU) RU = X[jc + 1];
            if (oD) RD = X[jc - 1];
        }
        const char* iu = irow(jc - 1);
        const char* ic = irow(jc);
        const char* id = irow(jc + 1);
        auto LD = [](const char* r, int o) { return *reinterpret_cast<const double*>(r + o); };
        const int ie = lane * 8, io = IHB + lane * 8;
        const double iuA = LD(iu, ie), iuB = LD(iu, io), iuC = LD(iu, ie + 8), iuD = LD(iu, io + 8);
        const double icA = LD(ic, ie), icB = LD(ic, io), icC = LD(ic, ie + 8), icD = LD(ic, io + 8);
        const double idA = LD(id, ie), idB = LD(id, io), idC = LD(id, ie + 8), idD = LD(id, io + 8);
        constexpr int DCT = G::DCT ? 2 : 0;
        double2 dq[3] = {{0, 0}, {0, 0}, {0, 0}};                           // axx, ayy, c of the row (stored by the first sweep's stage)
        if constexpr (G::DCT) {
            const char* dr = drow(LO - jc);
#pragma unroll
            for (int v = 0; v < 3; ++v) dq[v] = *reinterpret_cast<const double2*>(dr + v * 1024);
        }
        double2 y[3];
        constexpr int VPRE = EDGE ? 0 : 1;                                  // (see stage())
        double2 dU = {0.0, 0.0}, dW = {0.0, 0.0};
        if (VPRE) {
            dU.x = RD.u.x - RU.u.x; dU.y = RD.u.y - RU.u.y;
            dW.x = RD.w.x - RU.w.x; dW.y = RD.w.y - RU.w.y;
        }
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            Nbr n;
            double imv[9];
            bool gl = false, gr = false;
            VDiff0 vd = {0.0, 0.0, 0.0, 0.0};
            if (par == 0) {
                n.u[1] = RU.u.x; n.w[1] = RU.w.x; n.g[1] = RU.g.x;
                n.u[4] = RC.u.x; n.w[4] = RC.w.x; n.g[4] = RC.g.x;
                n.u[7] = RD.u.x; n.w[7] = RD.w.x; n.g[7] = RD.g.x;
                n.u[5] = RC.u.y; n.w[5] = RC.w.y; n.g[5] = RC.g.y;
                n.u[3] = lane_shr1(RC.u.y); n.w[3] = lane_shr1(RC.w.y); n.g[3] = lane_shr1(RC.g.y);
                if (VPRE) {
                    vd.du71 = dU.x; vd.dw71 = dW.x;
                    vd.U4 = dU.y - lane_shr1(dU.y); vd.W4 = dW.y - lane_shr1(dW.y);
                } else {
                    n.u[2] = RU.u.y; n.w[2] = RU.w.y;
                    n.u[8] = RD.u.y; n.w[8] = RD.w.y;
                    n.u[0] = lane_shr1(RU.u.y); n.w[0] = lane_shr1(RU.w.y);
                    n.u[6] = lane_shr1(RD.u.y); n.w[6] = lane_shr1(RD.w.y);
                }
                if (BORDER && glE) {
                    gl = true;
                    n.u[0] = n.u[2]; n.w[0] = n.w[2]; n.u[3] = n.u[5]; n.w[3] = n.w[5]; n.g[3] = n.g[5]; n.u[6] = n.u[8]; n.w[6] = n.w[8];
                }
                imv[0] = iuA; imv[1] = iuB; imv[2] = iuC; imv[3] = icA; imv[4] = icB; imv[5] = icC; imv[6] = idA; imv[7] = idB; imv[8] = idC;
            } else {
                n.u[1] = RU.u.y; n.w[1] = RU.w.y; n.g[1] = RU.g.y;
                n.u[4] = RC.u.y; n.w[4] = RC.w.y; n.g[4] = RC.g.y;
                n.u[7] = RD.u.y; n.w[7] = RD.w.y; n.g[7] = RD.g.y;
                n.u[3] = RC.u.x; n.w[3] = RC.w.x; n.g[3] = RC.g.x;
                n.u[5] = lane_shl1(RC.u.x); n.w[5] = lane_shl1(RC.w.x); n.g[5] = lane_shl1(RC.g.x);
                if (VPRE) {
                    vd.du71 = dU.y; vd.dw71 = dW.y;
                    vd.U4 = lane_shl1(dU.x) - dU.x; vd.W4 = lane_shl1(dW.x) - dW.x;
                } else {
                    n.u[0] = RU.u.x; n.w[0] = RU.w.x;
                    n.u[6] = RD.u.x; n.w[6] = RD.w.x;
                    n.u[2] = lane_shl1(RU.u.x); n.w[2] = lane_shl1(RU.w.x);
                    n.u[8] = lane_shl1(RD.u.x); n.w[8] = lane_shl1(RD.w.x);
                }
                if (BORDER && grO) {
                    gr = true;
                    n.u[2] = n.u[0]; n.w[2] = n.w[0]; n.u[5] = n.u[3]; n.w[5] = n.w[3]; n.g[5] = n.g[3]; n.u[8] = n.u[6]; n.w[8] = n.w[6];
                }
                imv[0] = iuB; imv[1] = iuC; imv[2] = iuD; imv[3] = icB; imv[4] = icC; imv[5] = icD; imv[6] = idB; imv[7] = idC; imv[8] = idD;
            }
            double y0, y1, y2;
            Diag0 dg;
            if (par == 0) { dg.axx = dq[0].x; dg.ayy = dq[1].x; dg.c = dq[2].x; }
            else { dg.axx = dq[0].y; dg.ayy = dq[1].y; dg.c = dq[2].y; }
            dg.inv = 0.0;
            if (EDGE) {
                const double sUL = (oU && gl) ? 2.0 : 1.0, sUR = (oU && gr) ? 2.0 : 1.0;
                const double sDL = (oD && gl) ? 2.0 : 1.0, sDR = (oD && gr) ? 2.0 : 1.0;
                apply0_point<true, DCT>(imv, n, sUL, sUR, sDL, sDR, alpha, beta, quirks, y0, y1, y2, &dg);
            } else {
                apply0_point<false, DCT, 1>(imv, n, 1.0, 1.0, 1.0, 1.0, alpha, beta, quirks, y0, y1, y2, &dg, &vd);
            }
            if (par == 0) { y[0].x = y0; y[1].x = y1; y[2].x = y2; }
            else { y[0].y = y0; y[1].y = y1; y[2].y = y2; }
        }
        if constexpr (RR) { yout[0] = y[0]; yout[1] = y[1]; yout[2] = y[2]; return true; }
        // (only the three stores sit under the lane predicate: a short predicated region gets no skip branch, so the steady-state
        // step stays one basic block; the sums of the lanes outside the owned columns are masked with a select)
        if (st_ok) {
            double* vrow = tr.v + off + (size_t)p * nj + qg;
            *reinterpret_cast<double2*>(vrow) = y[0];
            *reinterpret_cast<double2*>(vrow + npts) = y[1];
            *reinterpret_cast<double2*>(vrow + 2 * npts) = y[2];
        }
        const double vv = (y[0].x * y[0].x + y[1].x * y[1].x + y[2].x * y[2].x) + (y[0].y * y[0].y + y[1].y * y[1].y + y[2].y * y[2].y);
        const double vt = (y[0].x * t0.x + y[1].x * t1.x + y[2].x * t2.x) + (y[0].y * t0.y + y[1].y * t1.y + y[2].y * t2.y);
        ts0 += st_ok ? (tr.dotvec ? vt : vv) : 0.0;
        ts1 += st_ok ? vv : 0.0;
        return true;
    };

    auto step = [&](auto edge_tag, auto border_tag, const int e) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        // ---- (1) requests for rows e + 2, e + 3 (into the top of the window, first touched by the rotation below)
        request_rows(edge_tag, e);
        // ---- (2) the stages in order E_0, O_0, E_1, O_1, ...: stage st on row e - st = window row LO - st
        auto run_stage = [&](auto st_tag) {
            constexpr int ST = decltype(st_tag)::value;
            const int rr = e - ST, p = p0 + rr;
            const bool rowok = EDGE ? (rr >= st_lo(ST) && rr <= st_hi(ST) && p >= 0 && p < ni) : true;
            if constexpr (BF != 0 && ST < 2) {
                // form the row of b (the operations of k_update_s / k_update_p), store the owned part, hand the row on
                double2 bs[3];
#pragma unroll
                for (int f = 0; f < 3; ++f) {
                    if (BF == 3) {
                        bs[f] = B[ST][f];
                    } else if (BF == 1) {
                        bs[f].x = fma(-bcA, BV[ST][f].x, B[ST][f].x);
                        bs[f].y = fma(-bcA, BV[ST][f].y, B[ST][f].y);
                    } else {
                        bs[f].x = fma(bcA, fma(-bcB, BV[ST][f].x, BQ[ST][f].x), B[ST][f].x);
                        bs[f].y = fma(bcA, fma(-bcB, BV[ST][f].y, BQ[ST][f].y), B[ST][f].y);
                    }
                    BC[1][ST][f] = bs[f];
                }
                const bool own = st_ok && rr >= 0 && rr < TI && p < ni;    // (p >= 0: the bands of a forward pass start at row 0)
                if (BF != 3 && own) {
                    double* orow = bfo + (size_t)p * nj + qg;
                    *reinterpret_cast<double2*>(orow) = bs[0];
                    *reinterpret_cast<double2*>(orow + npts) = bs[1];
                    *reinterpret_cast<double2*>(orow + 2 * npts) = bs[2];
                }
                if (BF == 1) {
                    const double q = (bs[0].x * bs[0].x + bs[0].y * bs[0].y) + (bs[1].x * bs[1].x + bs[1].y * bs[1].y) + (bs[2].x * bs[2].x + bs[2].y * bs[2].y);
                    bss += own ? q : 0.0;
                }
                if (rowok) stage(edge_tag, border_tag, std::integral_constant<int, LO - ST>{}, rr, bs);
            } else if constexpr (BF != 0) {
                if (rowok) stage(edge_tag, border_tag, std::integral_constant<int, LO - ST>{}, rr, BC[0][ST - 2]);
            } else if constexpr (BL && ST < 2) {
                if (rowok) stage(edge_tag, border_tag, std::integral_constant<int, LO - ST>{}, rr, B[ST]);
                char* bw = bring + (ST * 2 + bph) * G::BRB + lane * 16;
#pragma unroll
                for (int f = 0; f < 3; ++f) *reinterpret_cast<double2*>(bw + f * 1024) = B[ST][f];
            } else if constexpr (BL) {
                double2 bl[3];
                const char* br = bring + ((ST - 2) * 2 + (bph ^ 1)) * G::BRB + lane * 16;
#pragma unroll
                for (int f = 0; f < 3; ++f) bl[f] = *reinterpret_cast<const double2*>(br + f * 1024);
                if (rowok) stage(edge_tag, border_tag, std::integral_constant<int, LO - ST>{}, rr, bl);
            } else {
                if (rowok) stage(edge_tag, border_tag, std::integral_constant<int, LO - ST>{}, rr, B[ST]);
            }
            request_b(edge_tag, st_tag, e + 2);
        };
        run_stage(std::integral_constant<int, 0>{});
        run_stage(std::integral_constant<int, 1>{});
        if constexpr (NST > 2) {
            run_stage(std::integral_constant<int, 2>{});
            run_stage(std::integral_constant<int, 3>{});
        }
        // ---- (3) trailing stage on the rows that have just become final: e - 2 NS + 1 and e - 2 NS
        if constexpr (TRAIL == 1) {
            double2 yd[3];
            trail_row(edge_tag, border_tag, std::integral_constant<int, LO - 2 * NS + 1>{}, e - 2 * NS + 1, std::integral_constant<int, 1>{}, yd);
            trail_row(edge_tag, border_tag, std::integral_constant<int, LO - 2 * NS>{}, e - 2 * NS, std::integral_constant<int, 0>{}, yd);
        }
        if constexpr (RR) {
            // residual rows e - 4 (even fine row 2 cp) and e - 3 (2 cp + 1); b of the two rows: what stage 2 used in the previous step
            // and what stage 3 used in this one.  Lanes without a column pair and rows outside the image contribute zero.
            double2 ya[3], yb[3];
            const bool okb = trail_row(edge_tag, border_tag, std::integral_constant<int, LO - 2 * NS + 1>{}, e - 2 * NS + 1, std::integral_constant<int, 1>{}, yb);
            const bool oka = trail_row(edge_tag, border_tag, std::integral_constant<int, LO - 2 * NS>{}, e - 2 * NS, std::integral_constant<int, 0>{}, ya);
            const int rrA = e - 2 * NS, pA = p0 + rrA, cp = pA >> 1;
            const int cq = (qs >> 1) + lane;                                   // the lane's coarse column (fine columns 2 cq, 2 cq + 1)
            const double wR = (EDGE && cq + 1 >= ncj) ? 1.0 : 0.5;             // the last coarse column takes its orphan right neighbour whole
            const double wD = (EDGE && cp + 1 >= nci) ? 1.0 : 0.5;             // ... the last coarse row likewise (pweight)
            double cv[3];
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                double2 ra = {0.0, 0.0}, rb = {0.0, 0.0};
                if (EDGE ? (oka && pair_ok) : true) { ra.x = BCo[f].x - ya[f].x; ra.y = BCo[f].y - ya[f].y; }
                if (EDGE ? (okb && pair_ok) : true) { rb.x = BC[0][1][f].x - yb[f].x; rb.y = BC[0][1][f].y - yb[f].y; }
                // columns 2 cq - 1 (the left lane's odd column), 2 cq, 2 cq + 1
                const double ha = fma(0.5, lane_shr1(ra.y), fma(wR, ra.y, ra.x));
                const double hb = fma(0.5, lane_shr1(rb.y), fma(wR, rb.y, rb.x));
                cv[f] = 0.25 * fma(0.5, hprev[f], fma(wD, hb, ha));
                hprev[f] = hb;
            }
            const bool own = st_ok && rrA >= 0 && rrA < TI && pA < ni && (EDGE ? (cq < ncj && cp < nci) : true);
            if (own) {
                ET* bc = reinterpret_cast<ET*>(tr.v) + (size_t)pair * 3 * ncpts + (size_t)cp * ncj + cq;
                bc[0] = (ET)cv[0]; bc[ncpts] = (ET)cv[1]; bc[2 * ncpts] = (ET)cv[2];
            }
#pragma unroll
            for (int f = 0; f < 3; ++f) BCo[f] = BC[0][0][f];                  // this step's row e - 2 is the next step's row e - 4
        }
        // ---- (4) write-out of rows e - 2 NS, e - 2 NS + 1
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int rrW = e - 2 * NS + r, pW = p0 + rrW;
            const bool rowok = EDGE ? (rrW >= 0 && rrW < TI && pW >= 0 && pW < ni) : true;
            if (rowok && st_ok) {
                const S0RRow& s = X[LO - 2 * NS + r];
                double* dst = xout + (size_t)pW * nj + qg;
                *reinterpret_cast<double2*>(dst) = s.u;
                *reinterpret_cast<double2*>(dst + npts) = s.w;
                *reinterpret_cast<double2*>(dst + 2 * npts) = s.g;
            }
        }
        // ---- (5) the image rows in flight take the ring slots of rows e - LO, e - LO + 1 (dead from here on)
        {
            const bool do_load = EDGE ? (e + 3 <= TI + 2 * (NS + EXT) - 1) : true;
            if (do_load) {
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    char* dst = irow(r);
                    *reinterpret_cast<double*>(dst + lane * 8) = li[r].x;
                    *reinterpret_cast<double*>(dst + IHB + lane * 8) = li[r].y;
                    if (EDGE ? lane == 0 : true) {   // (steady state: all lanes hold the same pair and write it to the same place)
                        *reinterpret_cast<double*>(dst + 64 * 8) = lix[r].x;
                        *reinterpret_cast<double*>(dst + IHB + 64 * 8) = lix[r].y;
                    }
                }
            }
            islot += 2;
            if (islot >= NRI) islot -= NRI;
            if (DC) { dslot += 2; if (dslot >= ND) dslot -= ND; }
            if (BL) bph ^= 1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // LDS operations of a wave execute in order; keeps the compiler from
            __builtin_amdgcn_wave_barrier();                         // moving next step's reads above these writes
        }
        // ---- (6) rotation of the window by two rows; the rows that enter get the interpolated coarse-grid correction (EC)
        if constexpr (EC) {
            // fine rows pL = p0 + e + 2 + r; coarse rows cp = pL >> 1 (and cp + 1 for an odd row).  CR[0] = coarse row of the LOWER of
            // the two entering rows' cp; after this step CR shifts by one coarse row (two fine rows = one coarse row)
            const int pL0 = p0 + e + 2;
            double cnew[3];
#pragma unroll
            for (int f = 0; f < 3; ++f) cnew[f] = (double)crv[f];
            // coarse rows available: CR[0] = row (pL0 >> 1), CR[1] = row (pL0 >> 1) + 1 when pL0 is even ... see below
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int pL = pL0 + r;
                const bool rowok = EDGE ? (e + 3 <= TI + 2 * (NS + EXT) - 1 && pL >= 0 && pL < ni) : true;
                if (EDGE ? (rowok && pair_ok) : true) {
                    const int cp = pL >> 1;
                    // (steady state: the row parity is that of PO + r - TI and e are even -, the coarse row above exists and
                    // every lane's odd column has its right coarse neighbour: no branch left)
                    const bool ipi = EDGE ? ((pL & 1) && (cp + 1 < nci)) : (((PO + r) & 1) != 0);
                    const bool ipj = EDGE ? ((qpair >> 1) + 1 < ncj) : true;   // the odd column has a right coarse neighbour
                    // coarse row cp: CR[cp - (pL0 >> 1)] (0 or 1); coarse row cp + 1: CR[1] or the row that has just arrived
                    const int i0 = cp - (pL0 >> 1);
                    S0RRow& d = X[LO + 2 + r];
#pragma unroll
                    for (int f = 0; f < 3; ++f) {
                        const double a0 = i0 ? CR[1][f] : CR[0][f];
                        const double a1 = i0 ? cnew[f] : CR[1][f];
                        const double a0r = lane_shl1(a0), a1r = lane_shl1(a1);
                        const double wi0 = ipi ? 0.5 : 1.0;
                        // same terms in the same order as k_prolong_add: (cp, cq), (cp, cq + 1), (cp + 1, cq), (cp + 1, cq + 1)
                        double ve = wi0 * a0;
                        double vo = (ipj ? wi0 * 0.5 : wi0) * a0;
                        if (ipj) vo += wi0 * 0.5 * a0r;
                        if (ipi) {
                            ve += 0.5 * a1;
                            vo += (ipj ? 0.25 : 0.5) * a1;
                            if (ipj) vo += 0.25 * a1r;
                        }
                        double2& t = f == 0 ? d.u : (f == 1 ? d.w : d.g);
                        t.x += ve; t.y += vo;
                    }
                }
            }
            // next step's entering rows start one coarse row higher
#pragma unroll
            for (int f = 0; f < 3; ++f) { CR[0][f] = CR[1][f]; CR[1][f] = cnew[f]; }
        }
#pragma unroll
        for (int j = 0; j + 2 < NRW; ++j) X[j] = X[j + 2];
        if constexpr (BF != 0) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int f = 0; f < 3; ++f) BC[0][a][f] = BC[1][a][f];
        }
    };

    // ---- prologue: b of the first step's stages; EC: the coarse rows the first entering rows need
    request_b(std::true_type{}, std::integral_constant<int, 0>{}, e0);
    request_b(std::true_type{}, std::integral_constant<int, 1>{}, e0);
    if constexpr (NST > 2) {
        request_b(std::true_type{}, std::integral_constant<int, 2>{}, e0);
        request_b(std::true_type{}, std::integral_constant<int, 3>{}, e0);
    }
    if constexpr (EC) {
        // the first step brings in fine rows pL0 = p0 + e0 + 2, + 1: coarse rows k0 = pL0 >> 1 and k0 + 1 (+ 2 arrives with the step)
        const int k0 = (p0 + e0 + 2) >> 1;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                const int k = k0 + d, c0 = cqs + lane;
                CR[d][f] = (k >= 0 && k < nci && c0 >= 0 && c0 < ncj) ? (double)ec[(size_t)f * ncpts + (size_t)k * ncj + c0] : 0.0;
            }
    }

    for (int s = s_first; s <= s_last; ++s) {
        const int e = 2 * s;
        // (strips that touch a side of the image run the edge version throughout: the steady-state version then has no
        // per-phase branch at all and is one basic block per step)
        if (e >= e_lo && e <= e_hi && !border_strip) step(std::false_type{}, std::false_type{}, e);
        else step(std::true_type{}, std::true_type{}, e);
    }
    if constexpr (BF == 1) {      // per-block partial sums of (s, s), laid out like the trailing product's
        const double a0 = wave_sum(bss);
        if (lane == 0) bsrc.partials[((size_t)pair * 3) * ((size_t)nx * ny) + (size_t)by * nx + bx] = a0;
    }
    if constexpr (TRAIL == 1) {   // per-block partial sums of the dot products
        const double a0 = wave_sum(ts0), a1 = wave_sum(ts1);
        if (lane == 0 && tr.partials) {
            const int nblk = nx * ny, blk = by * nx + bx;
            double* pp = tr.partials + ((size_t)pair * 3) * nblk + blk;
            pp[0] = a0;
            if (tr.dotvec && tr.want_vv) pp[nblk] = a1;
        }
    }
}

}  // namespace vof
