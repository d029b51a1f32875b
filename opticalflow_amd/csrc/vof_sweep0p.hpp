// vof_sweep0p.hpp - k_sweep0p: the register-resident level-0 smoother pass (vof_sweep0r.hpp) for FLOAT32 cycle vectors, in
// PACKED float32 arithmetic.
//
// k_sweep0r is bound by the FP64 vector pipes (profiles/r03_sq_sweep0.md: 78 FP64 instructions per point update at 4-5 cycles
// each, 58-61 % of all SIMD cycles busy with them at 1.7-1.8 GHz), not by HBM.  With float32 cycle vectors
// (vcycle_precision 1 / 2: x, b and the coarse-grid correction of the cycle are float32 in HBM; the Krylov vectors, the
// operator products, the residuals and the stopping rule stay float64) the smoother can do its arithmetic in float32 as well:
// it is part of the preconditioner, and float32 arithmetic there leaves the iteration counts of the well-conditioned regimes
// unchanged (CPU prototype experiment; the grad-div dominated regimes need float64, which is what
// the solver switches to after AUTO_F64_AFTER iterations in mode 2).
//
// One wave owns TWO strips (A = strip 2 j, B = strip 2 j + 1; 2 x 128 columns) and carries them as the two halves of packed
// registers: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 process both strips in one instruction at the cost of one FP64
// instruction, a DPP lane shift moves both, and the loads in flight are half the size of k_sweep0r's.  Same schedule as
// k_sweep0r (stage st on row e - st, register window rows e - 2 NS .. e + 3, image rows through a wave-private LDS ring -
// here of float pairs {A, B}), same ghost folding at the image sides and rows (per strip), no trailing operator stage (the
// Krylov product is float64 and runs as its own kernel).  Results agree with the float64-arithmetic sweeps to float32
// rounding (tests), not bit for bit.
#pragma once
#include "vof_sweep0r.hpp"

namespace vof {

typedef float f2 __attribute__((ext_vector_type(2)));   // {strip A, strip B}

__device__ __forceinline__ f2 f2_shr1(f2 v) {
    int lo = __float_as_int(v.x), hi = __float_as_int(v.y);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x138, 0xF, 0xF, true);   // wave_shr:1
    hi = __builtin_amdgcn_mov_dpp(hi, 0x138, 0xF, 0xF, true);
    return f2{__int_as_float(lo), __int_as_float(hi)};
}
__device__ __forceinline__ f2 f2_shl1(f2 v) {
    int lo = __float_as_int(v.x), hi = __float_as_int(v.y);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x130, 0xF, 0xF, true);   // wave_shl:1
    hi = __builtin_amdgcn_mov_dpp(hi, 0x130, 0xF, 0xF, true);
    return f2{__int_as_float(lo), __int_as_float(hi)};
}
__device__ __forceinline__ f2 f2_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 f2_bc(float v) { return f2{v, v}; }
__device__ __forceinline__ f2 f2_sel(bool ca, bool cb, f2 a, f2 b) { return f2{ca ? a.x : b.x, cb ? a.y : b.y}; }   // per strip: c ? a : b

struct NbrP { f2 u[9], w[9], g[9]; };

// gs0_point in packed float32, reference quirk 'dy' == 'dx' built in.  Same formulas as gs0_point (CORNERS: the factor 2 of
// a corner ghost, per strip); the 2x2 determinant is inverted with v_rcp_f32 + one Newton step.
template <bool CORNERS>
__device__ __forceinline__ void gs0_point_p(const f2* im, const NbrP& n, f2 sUL, f2 sUR, f2 sDL, f2 sDR, f2 alpha, f2 beta, f2 inv_g, f2 b0,
                                            f2 b1, f2 b2, f2& u, f2& w, f2& gm) {
    const f2 P = im[4];
    const f2 Dx = (im[7] - im[1]) * 0.5f;
    const f2 Dxx = f2_fma(f2_bc(-2.0f), P, im[7] + im[1]);
    const f2 Dyy = f2_fma(f2_bc(-2.0f), P, im[5] + im[3]);
    const f2 Dxy = (im[8] - im[6] - im[2] + im[0]) * 0.25f;
    const f2 PP = P * P, PDx = P * Dx, hP = P * 0.5f;
    const f2 A1 = PP + alpha, qPP = PP * 0.25f, hPDx = PDx * 0.5f;
    const f2 du71 = n.u[7] - n.u[1], du53 = n.u[5] - n.u[3], dw71 = n.w[7] - n.w[1], dw53 = n.w[5] - n.w[3];
    f2 W4, U4;
    if (CORNERS) {
        W4 = sUL * n.w[0] + sDR * n.w[8] - sUR * n.w[2] - sDL * n.w[6];
        U4 = sUL * n.u[0] + sDR * n.u[8] - sUR * n.u[2] - sDL * n.u[6];
    } else {
        W4 = n.w[0] + n.w[8] - n.w[2] - n.w[6];
        U4 = n.u[0] + n.u[8] - n.u[2] - n.u[6];
    }
    const f2 y0a = f2_fma(hPDx, dw53, f2_fma(alpha, n.u[3] + n.u[5], A1 * (n.u[1] + n.u[7])));
    const f2 y0b = f2_fma(hP, n.g[1] - n.g[7], f2_fma(qPP, W4, f2_fma(hPDx, dw71, PDx * du71)));
    const f2 y1a = f2_fma(hPDx, du71, f2_fma(alpha, n.w[1] + n.w[7], A1 * (n.w[3] + n.w[5])));
    const f2 y1b = f2_fma(hP, n.g[3] - n.g[5], f2_fma(qPP, U4, f2_fma(hPDx, du53, PDx * dw53)));
    const f2 y2 = f2_fma(hP, du71 + dw53, beta * ((n.g[1] + n.g[7]) + (n.g[3] + n.g[5])));
    const f2 r0 = b0 - (y0a + y0b), r1 = b1 - (y1a + y1b), r2 = b2 - y2;
    const f2 m4a = alpha * -4.0f;
    const f2 axx = f2_fma(P, f2_fma(f2_bc(-2.0f), P, Dxx), m4a), ayy = f2_fma(P, f2_fma(f2_bc(-2.0f), P, Dyy), m4a), c = P * Dxy;
    const f2 det = f2_fma(axx, ayy, -(c * c));
    f2 inv = f2{__builtin_amdgcn_rcpf(det.x), __builtin_amdgcn_rcpf(det.y)};
    inv = f2_fma(f2_fma(-det, inv, f2_bc(1.0f)), inv, inv);
    u = f2_fma(r0, ayy, -(c * r1)) * inv;
    w = f2_fma(axx, r1, -(c * r0)) * inv;
    gm = f2_fma(-Dx, w, f2_fma(-Dx, u, r2)) * inv_g;
}

struct S0PRow { f2 ux, uy, wx, wy, gx, gy; };   // one x row of the two strips: ?x = column 2 lane, ?y = column 2 lane + 1

// grid: nxp (strip pairs: strips 2 j, 2 j + 1; the last pair may lack its second strip) x ny x pairs blocks of one wave
template <int NS, bool EC, bool FROM_ZERO, int PO>
__global__ __launch_bounds__(64) void k_sweep0p(
    Fine0 pol, int ni, int nj, int TI, int nx, int nxp, int ny, int nz, const float* __restrict__ x_in,
    float* __restrict__ x_out, const float* __restrict__ b, const int* __restrict__ active,
    const float* __restrict__ ecoarse, int nci, int ncj) {
    typedef S0R<NS, 0> G;
    constexpr int W = S0_W, LO = G::LO, NRW = G::NRW, NRI = G::NRI, IRB = G::IRB, IHB = G::IPW * 8;
    constexpr int NST = 2 * NS, po = PO;
    extern __shared__ double sw_lds[];
    char* iring = reinterpret_cast<char*>(sw_lds);
    const unsigned nblocks = (unsigned)nxp * ny * nz;
    unsigned lb = blockIdx.x;
    if ((nblocks & 7u) == 0) lb = (lb & 7u) * (nblocks >> 3) + (lb >> 3);   // XCD-aware remap, see k_sweep
    const int bxp = lb % nxp, by = (lb / nxp) % ny;
    const int pair = lb / (nxp * ny);
    if (active && !active[pair]) return;
    const int lane = threadIdx.x;
    const int p0 = by * TI - po;
    const int qsA = (2 * bxp) * G::OUT - G::HALO, qsB = qsA + G::OUT;
    const bool hasB = 2 * bxp + 1 < nx;                                      // (wave-uniform) the pair's second strip exists
    const size_t npts = (size_t)ni * nj, off = (size_t)pair * 3 * npts;
    const float* xin = FROM_ZERO ? nullptr : x_in + off;
    float* xout = x_out + off;
    const float* bp = b + off;
    const size_t ncpts = (size_t)nci * ncj;
    const float* ec = EC ? ecoarse + (size_t)pair * 3 * ncpts : nullptr;
    double alpha_d = pol.alpha, beta_d = pol.beta;
    int fidx = pair;
    if (pol.pp) { alpha_d = pol.pp[pair].alpha; beta_d = pol.pp[pair].beta; fidx = pol.pp[pair].frame; }
    const double* img = pol.frames + (size_t)fidx * pol.frame_stride;
    const int Nj = pol.Nj;
    const f2 alpha = f2_bc((float)alpha_d), beta = f2_bc((float)beta_d), inv_g = f2_bc((float)(1.0 / (-1 - 4 * beta_d)));
    // per strip: lane <-> column pair (2 lane, 2 lane + 1); pair validity is all-or-nothing (qs and nj are even)
    const int qpA = qsA + 2 * lane, qpB = qsB + 2 * lane;
    const bool okA = qpA >= 0 && qpA + 1 < nj, okB = hasB && qpB >= 0 && qpB + 1 < nj;
    const size_t qgA = okA ? (size_t)qpA : 0, qgB = okB ? (size_t)qpB : 0;
    const bool iokA = qpA >= 0 && qpA + 1 <= nj + 1, iokB = hasB && qpB >= 0 && qpB + 1 <= nj + 1;   // image columns (full image)
    const size_t iqA = iokA ? (size_t)qpA : 0, iqB = iokB ? (size_t)qpB : 0;
    const int xqA = qsA + 128, xqB = qsB + 128;                              // image pair 64 of a strip (columns 128, 129)
    const bool xokA = xqA >= 0 && xqA + 1 <= nj + 1, xokB = hasB && xqB >= 0 && xqB + 1 <= nj + 1;
    const bool own = lane >= G::HALO / 2 && lane < (W - G::HALO) / 2;        // owned column pairs (the same lanes in both strips)
    const bool stA = okA && own, stB = okB && own;
    const bool glA = qpA == 0, glB = hasB && qpB == 0, grA = qpA + 1 == nj - 1, grB = hasB && qpB + 1 == nj - 1;   // ghost columns
    const bool interior = qsA >= 0 && hasB && qsB + W <= nj;                 // (wave-uniform) neither strip touches a side of the image
    const int cqsA = qsA >> 1, cqsB = qsB >> 1;

    S0PRow X[NRW];
#pragma unroll
    for (int j = 0; j < NRW; ++j) X[j].ux = X[j].uy = X[j].wx = X[j].wy = X[j].gx = X[j].gy = f2_bc(0.f);
    f2 Bx[NST][3], By[NST][3];                                              // b of the stage's row, even / odd column
    f2 lix_[2] = {f2_bc(0.f), f2_bc(0.f)}, liy_[2] = {f2_bc(0.f), f2_bc(0.f)};   // image rows e + 2, e + 3 in flight
    f2 lxx_[2] = {f2_bc(0.f), f2_bc(0.f)}, lxy_[2] = {f2_bc(0.f), f2_bc(0.f)};   // ... their pair 64
    f2 crv[3] = {f2_bc(0.f), f2_bc(0.f), f2_bc(0.f)};
    f2 CR[2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int f = 0; f < 3; ++f) CR[r][f] = f2_bc(0.f);

    auto st_lo = [](int st) { const int k = st >> 1, odd = st & 1; return -2 * (NS - 1 - k) + odd; };
    auto st_hi = [TI](int st) { const int k = st >> 1, odd = st & 1; return TI + 2 * (NS - 1 - k) - odd; };
    const int s_first = -NS - 1, s_last = (TI + 2 * NS - 2) / 2;
    const int e0 = 2 * s_first;
    int e_lo = 2 * NS, e_hi = TI + 2 * NS - 4;
    e_lo = max(e_lo, 2 * NS - p0);  e_hi = min(e_hi, ni - 4 - p0);
    e_hi = min(e_hi, TI + 2 * NS - 2);
#pragma unroll
    for (int st = 0; st < NST; ++st) {
        e_lo = max(e_lo, max(st_lo(st) + st, 1 - p0 + st));
        e_hi = min(e_hi, min(st_hi(st) + st - 2, ni - 4 - p0 + st));
    }
    if (EC) { e_lo = max(e_lo, -p0); e_hi = min(e_hi, 2 * nci - 8 - p0); }

    int islot = 0;
    auto irow = [&](int j) { int s = islot + j; if (s >= NRI) s -= NRI; return iring + s * IRB; };
    auto pack = [](float a, float b2) { return f2{a, b2}; };

    // a float2 of one strip at `base + q` (steady state: unconditional)
    auto ld2 = [](auto edge_tag, const float* base, size_t q, bool ok) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        if (EDGE && !ok) return make_float2(0.f, 0.f);
        return *reinterpret_cast<const float2*>(base + q);
    };

    auto request_rows = [&](auto edge_tag, int e) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        const bool do_load = EDGE ? (e + 3 <= TI + 2 * NS - 1) : true;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int pL = p0 + e + 2 + r;
            S0PRow& d = X[LO + 2 + r];
            if (FROM_ZERO || EDGE) d.ux = d.uy = d.wx = d.wy = d.gx = d.gy = f2_bc(0.f);
            if (!FROM_ZERO) {
                const bool rowok = EDGE ? (do_load && pL >= 0 && pL < ni) : true;
                if (rowok) {
                    const float* row = xin + (size_t)pL * nj;
                    const float2 ua = ld2(edge_tag, row, qgA, okA), ub = ld2(edge_tag, row, qgB, okB);
                    const float2 wa = ld2(edge_tag, row + npts, qgA, okA), wb = ld2(edge_tag, row + npts, qgB, okB);
                    const float2 ga = ld2(edge_tag, row + 2 * npts, qgA, okA), gb = ld2(edge_tag, row + 2 * npts, qgB, okB);
                    d.ux = pack(ua.x, ub.x); d.uy = pack(ua.y, ub.y);
                    d.wx = pack(wa.x, wb.x); d.wy = pack(wa.y, wb.y);
                    d.gx = pack(ga.x, gb.x); d.gy = pack(ga.y, gb.y);
                }
            }
            {
                const int pI = pL + 1;
                const bool rowok = EDGE ? (do_load && pI >= 0 && pI <= ni + 1) : true;
                if (EDGE) { lix_[r] = liy_[r] = lxx_[r] = lxy_[r] = f2_bc(0.f); }
                if (rowok) {
                    const double* frow = img + (size_t)pI * Nj;
                    double2 ia = {0, 0}, ib = {0, 0}, xa = {0, 0}, xb = {0, 0};
                    if (EDGE ? iokA : true) ia = *reinterpret_cast<const double2*>(frow + iqA);
                    if (EDGE ? iokB : true) ib = *reinterpret_cast<const double2*>(frow + iqB);
                    if (EDGE ? xokA : true) xa = *reinterpret_cast<const double2*>(frow + (EDGE ? (xokA ? xqA : 0) : xqA));
                    if (EDGE ? xokB : true) xb = *reinterpret_cast<const double2*>(frow + (EDGE ? (xokB ? xqB : 0) : xqB));
                    lix_[r] = pack((float)ia.x, (float)ib.x); liy_[r] = pack((float)ia.y, (float)ib.y);
                    lxx_[r] = pack((float)xa.x, (float)xb.x); lxy_[r] = pack((float)xa.y, (float)xb.y);
                }
            }
        }
        if constexpr (EC) {
            const int knew = ((p0 + e + 2) >> 1) + 2;
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                crv[f] = f2_bc(0.f);
                if (EDGE ? (knew >= 0 && knew < nci) : true) {
                    const float* er = ec + (size_t)f * ncpts + (size_t)knew * ncj;
                    const int ca = cqsA + lane, cb = cqsB + lane;
                    if (EDGE) crv[f] = f2{(ca >= 0 && ca < ncj) ? er[ca] : 0.f, (hasB && cb >= 0 && cb < ncj) ? er[cb] : 0.f};
                    else crv[f] = f2{er[ca], er[cb]};
                }
            }
        }
    };

    auto request_b = [&](auto edge_tag, auto st_tag, int e_next) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        constexpr int st = decltype(st_tag)::value;
        const int rr = e_next - st, p = p0 + rr;
        const bool rowok = EDGE ? (rr >= st_lo(st) && rr <= st_hi(st) && p >= 0 && p < ni) : true;
        if (EDGE) {
#pragma unroll
            for (int f = 0; f < 3; ++f) Bx[st][f] = By[st][f] = f2_bc(0.f);
        }
        if (rowok) {
            const float* row = bp + (size_t)p * nj;
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                const float2 va = ld2(edge_tag, row + (size_t)f * npts, qgA, okA), vb = ld2(edge_tag, row + (size_t)f * npts, qgB, okB);
                Bx[st][f] = pack(va.x, vb.x);
                By[st][f] = pack(va.y, vb.y);
            }
        }
    };

    auto stage = [&](auto edge_tag, auto jc_tag, int rr, auto st_tag) {
        constexpr bool EDGE = decltype(edge_tag)::value;   // also: the strips may touch a side of the image
        constexpr int jc = decltype(jc_tag)::value;
        constexpr int st = decltype(st_tag)::value;
        const int p = p0 + rr;
        const bool oU = EDGE && p - 1 < 0, oD = EDGE && p + 1 >= ni;
        S0PRow RU = X[jc - 1], RD = X[jc + 1];
        if (EDGE) {
            if (oU) RU = X[jc + 1];
            if (oD) RD = X[jc - 1];
        }
        const char* iu = irow(jc - 1);
        const char* ic = irow(jc);
        const char* id = irow(jc + 1);
        auto LD = [](const char* r, int o) { return *reinterpret_cast<const f2*>(r + o); };
        const int ie = lane * 8, io = IHB + lane * 8;
        const f2 iuA = LD(iu, ie), iuB = LD(iu, io), iuC = LD(iu, ie + 8), iuD = LD(iu, io + 8);
        const f2 icA = LD(ic, ie), icB = LD(ic, io), icC = LD(ic, ie + 8), icD = LD(ic, io + 8);
        const f2 idA = LD(id, ie), idB = LD(id, io), idC = LD(id, ie + 8), idD = LD(id, io + 8);
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            constexpr int par_of_ph[2] = {po, 1 - po};
            const int par = par_of_ph[ph];
            S0PRow& RC = X[jc];
            NbrP n;
            f2 imv[9];
            bool gla = false, glb = false, gra = false, grb = false;
            if (par == 0) {
                n.u[1] = RU.ux; n.w[1] = RU.wx; n.g[1] = RU.gx;
                n.u[7] = RD.ux; n.w[7] = RD.wx; n.g[7] = RD.gx;
                n.u[2] = RU.uy; n.w[2] = RU.wy;
                n.u[5] = RC.uy; n.w[5] = RC.wy; n.g[5] = RC.gy;
                n.u[8] = RD.uy; n.w[8] = RD.wy;
                n.u[0] = f2_shr1(RU.uy); n.w[0] = f2_shr1(RU.wy);
                n.u[3] = f2_shr1(RC.uy); n.w[3] = f2_shr1(RC.wy); n.g[3] = f2_shr1(RC.gy);
                n.u[6] = f2_shr1(RD.uy); n.w[6] = f2_shr1(RD.wy);
                if (EDGE) {   // ghost column -1 mirrors column 1 (per strip)
                    gla = glA; glb = glB;
                    n.u[0] = f2_sel(gla, glb, n.u[2], n.u[0]); n.w[0] = f2_sel(gla, glb, n.w[2], n.w[0]);
                    n.u[3] = f2_sel(gla, glb, n.u[5], n.u[3]); n.w[3] = f2_sel(gla, glb, n.w[5], n.w[3]); n.g[3] = f2_sel(gla, glb, n.g[5], n.g[3]);
                    n.u[6] = f2_sel(gla, glb, n.u[8], n.u[6]); n.w[6] = f2_sel(gla, glb, n.w[8], n.w[6]);
                }
                imv[0] = iuA; imv[1] = iuB; imv[2] = iuC; imv[3] = icA; imv[4] = icB; imv[5] = icC; imv[6] = idA; imv[7] = idB; imv[8] = idC;
            } else {
                n.u[1] = RU.uy; n.w[1] = RU.wy; n.g[1] = RU.gy;
                n.u[7] = RD.uy; n.w[7] = RD.wy; n.g[7] = RD.gy;
                n.u[0] = RU.ux; n.w[0] = RU.wx;
                n.u[3] = RC.ux; n.w[3] = RC.wx; n.g[3] = RC.gx;
                n.u[6] = RD.ux; n.w[6] = RD.wx;
                n.u[2] = f2_shl1(RU.ux); n.w[2] = f2_shl1(RU.wx);
                n.u[5] = f2_shl1(RC.ux); n.w[5] = f2_shl1(RC.wx); n.g[5] = f2_shl1(RC.gx);
                n.u[8] = f2_shl1(RD.ux); n.w[8] = f2_shl1(RD.wx);
                if (EDGE) {   // ghost column n_j mirrors column n_j - 2 (per strip)
                    gra = grA; grb = grB;
                    n.u[2] = f2_sel(gra, grb, n.u[0], n.u[2]); n.w[2] = f2_sel(gra, grb, n.w[0], n.w[2]);
                    n.u[5] = f2_sel(gra, grb, n.u[3], n.u[5]); n.w[5] = f2_sel(gra, grb, n.w[3], n.w[5]); n.g[5] = f2_sel(gra, grb, n.g[3], n.g[5]);
                    n.u[8] = f2_sel(gra, grb, n.u[6], n.u[8]); n.w[8] = f2_sel(gra, grb, n.w[6], n.w[8]);
                }
                imv[0] = iuB; imv[1] = iuC; imv[2] = iuD; imv[3] = icB; imv[4] = icC; imv[5] = icD; imv[6] = idB; imv[7] = idC; imv[8] = idD;
            }
            f2 u, w, gm;
            const f2 c0 = par ? By[st][0] : Bx[st][0], c1 = par ? By[st][1] : Bx[st][1], c2 = par ? By[st][2] : Bx[st][2];
            if (EDGE) {
                const f2 one = f2_bc(1.f), two = f2_bc(2.f);
                const f2 sUL = f2_sel(oU && gla, oU && glb, two, one), sUR = f2_sel(oU && gra, oU && grb, two, one);
                const f2 sDL = f2_sel(oD && gla, oD && glb, two, one), sDR = f2_sel(oD && gra, oD && grb, two, one);
                gs0_point_p<true>(imv, n, sUL, sUR, sDL, sDR, alpha, beta, inv_g, c0, c1, c2, u, w, gm);
            } else {
                gs0_point_p<false>(imv, n, f2_bc(1.f), f2_bc(1.f), f2_bc(1.f), f2_bc(1.f), alpha, beta, inv_g, c0, c1, c2, u, w, gm);
            }
            if (par == 0) { RC.ux = u; RC.wx = w; RC.gx = gm; }
            else { RC.uy = u; RC.wy = w; RC.gy = gm; }
        }
    };

    auto step = [&](auto edge_tag, const int e) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        request_rows(edge_tag, e);
        auto run_stage = [&](auto st_tag) {
            constexpr int ST = decltype(st_tag)::value;
            const int rr = e - ST, p = p0 + rr;
            const bool rowok = EDGE ? (rr >= st_lo(ST) && rr <= st_hi(ST) && p >= 0 && p < ni) : true;
            if (rowok) stage(edge_tag, std::integral_constant<int, LO - ST>{}, rr, st_tag);
            request_b(edge_tag, st_tag, e + 2);
        };
        run_stage(std::integral_constant<int, 0>{});
        run_stage(std::integral_constant<int, 1>{});
        if constexpr (NST > 2) {
            run_stage(std::integral_constant<int, 2>{});
            run_stage(std::integral_constant<int, 3>{});
        }
        // write-out of rows e - 2 NS, e - 2 NS + 1
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int rrW = e - 2 * NS + r, pW = p0 + rrW;
            const bool rowok = EDGE ? (rrW >= 0 && rrW < TI && pW >= 0 && pW < ni) : true;
            if (rowok) {
                const S0PRow& s = X[LO - 2 * NS + r];
                float* row = xout + (size_t)pW * nj;
                if (stA) {
                    *reinterpret_cast<float2*>(row + qgA) = make_float2(s.ux.x, s.uy.x);
                    *reinterpret_cast<float2*>(row + npts + qgA) = make_float2(s.wx.x, s.wy.x);
                    *reinterpret_cast<float2*>(row + 2 * npts + qgA) = make_float2(s.gx.x, s.gy.x);
                }
                if (stB) {
                    *reinterpret_cast<float2*>(row + qgB) = make_float2(s.ux.y, s.uy.y);
                    *reinterpret_cast<float2*>(row + npts + qgB) = make_float2(s.wx.y, s.wy.y);
                    *reinterpret_cast<float2*>(row + 2 * npts + qgB) = make_float2(s.gx.y, s.gy.y);
                }
            }
        }
        {   // the image rows in flight take the ring slots of rows e - LO, e - LO + 1
            const bool do_load = EDGE ? (e + 3 <= TI + 2 * NS - 1) : true;
            if (do_load) {
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    char* dst = irow(r);
                    *reinterpret_cast<f2*>(dst + lane * 8) = lix_[r];
                    *reinterpret_cast<f2*>(dst + IHB + lane * 8) = liy_[r];
                    *reinterpret_cast<f2*>(dst + 64 * 8) = lxx_[r];              // (every lane holds the same pair and writes it to the same place)
                    *reinterpret_cast<f2*>(dst + IHB + 64 * 8) = lxy_[r];
                }
            }
            islot += 2;
            if (islot >= NRI) islot -= NRI;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if constexpr (EC) {   // x + P e for the rows that enter (terms as in k_prolong_add; float32 here)
            const int pL0 = p0 + e + 2;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int pL = pL0 + r;
                const bool rowok = EDGE ? (e + 3 <= TI + 2 * NS - 1 && pL >= 0 && pL < ni) : true;
                if (rowok) {
                    const int cp = pL >> 1;
                    const bool ipi = EDGE ? ((pL & 1) && (cp + 1 < nci)) : (((PO + r) & 1) != 0);
                    // per strip: the odd column has a right coarse neighbour (steady state: always)
                    const bool ipjA = EDGE ? ((qpA >> 1) + 1 < ncj) : true, ipjB = EDGE ? ((qpB >> 1) + 1 < ncj) : true;
                    const int i0 = cp - (pL0 >> 1);
                    S0PRow& d = X[LO + 2 + r];
#pragma unroll
                    for (int f = 0; f < 3; ++f) {
                        const f2 a0 = i0 ? CR[1][f] : CR[0][f];
                        const f2 a1 = i0 ? crv[f] : CR[1][f];
                        const f2 a0r = f2_shl1(a0), a1r = f2_shl1(a1);
                        const float wi0 = ipi ? 0.5f : 1.0f;
                        const f2 wj0 = f2_sel(ipjA, ipjB, f2_bc(0.5f), f2_bc(1.0f)), wj1 = f2_sel(ipjA, ipjB, f2_bc(0.5f), f2_bc(0.0f));
                        f2 ve = a0 * wi0;
                        f2 vo = a0 * (wj0 * wi0) + a0r * (wj1 * wi0);
                        if (ipi) {
                            ve += a1 * 0.5f;
                            vo += a1 * (wj0 * 0.5f) + a1r * (wj1 * 0.5f);
                        }
                        if (f == 0) { d.ux += ve; d.uy += vo; }
                        else if (f == 1) { d.wx += ve; d.wy += vo; }
                        else { d.gx += ve; d.gy += vo; }
                    }
                    if (EDGE) {   // keep the columns outside the image at zero
                        if (!okA) { d.ux.x = d.uy.x = d.wx.x = d.wy.x = d.gx.x = d.gy.x = 0.f; }
                        if (!okB) { d.ux.y = d.uy.y = d.wx.y = d.wy.y = d.gx.y = d.gy.y = 0.f; }
                    }
                }
            }
#pragma unroll
            for (int f = 0; f < 3; ++f) { CR[0][f] = CR[1][f]; CR[1][f] = crv[f]; }
        }
#pragma unroll
        for (int j = 0; j + 2 < NRW; ++j) X[j] = X[j + 2];
    };

    request_b(std::true_type{}, std::integral_constant<int, 0>{}, e0);
    request_b(std::true_type{}, std::integral_constant<int, 1>{}, e0);
    if constexpr (NST > 2) {
        request_b(std::true_type{}, std::integral_constant<int, 2>{}, e0);
        request_b(std::true_type{}, std::integral_constant<int, 3>{}, e0);
    }
    if constexpr (EC) {
        const int k0 = (p0 + e0 + 2) >> 1;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                const int k = k0 + d, ca = cqsA + lane, cb = cqsB + lane;
                const bool rowok = k >= 0 && k < nci;
                const float* er = ec + (size_t)f * ncpts + (size_t)(rowok ? k : 0) * ncj;
                CR[d][f] = f2{(rowok && ca >= 0 && ca < ncj) ? er[ca] : 0.f, (rowok && hasB && cb >= 0 && cb < ncj) ? er[cb] : 0.f};
            }
    }
    for (int s = s_first; s <= s_last; ++s) {
        const int e = 2 * s;
        // (strip pairs that touch a side of the image run the edge version throughout)
        if (e >= e_lo && e <= e_hi && interior) step(std::false_type{}, e);
        else step(std::true_type{}, e);
    }
}

}  // namespace vof
