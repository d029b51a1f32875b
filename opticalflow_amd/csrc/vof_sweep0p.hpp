// vof_sweep0p.hpp - k_sweep0p: the register-resident level-0 smoother pass (vof_sweep0r.hpp) in PACKED FLOAT32 arithmetic.
//
// k_sweep0r is bound by the FP64 vector pipes (measured, scripts/gpu_valu_rate.hip + profiles/r03_sq_sweep0.md: 78 FP64
// instructions per point update at 4.1-5 cycles each, 58-61 % of all SIMD cycles busy with them, clocks at 1.7-1.8 GHz), not
// by HBM.  The smoother is part of the PRECONDITIONER: the Krylov iteration, its operator products, the residuals and the
// stopping rule stay float64, and the cycle vectors below level 0 are float32 already (vcycle_precision 3).  Float32
// arithmetic in the level-0 sweeps leaves every iteration count of the well-conditioned regimes unchanged (CPU prototype
// experiment, DESIGN.md section 3.1: N, 8-bit alpha 1e5, alpha = beta = 1, alpha 0.5, 8-bit alpha = beta = 1e6: the same
// counts to rtol 1e-6 and 1e-10); the grad-div dominated regime T needs float64 there, which is what the solver switches
// to after AUTO_F64_AFTER iterations anyway (together with the float64 cycle vectors).
//
// One wave owns TWO interior strips (A, B: 2 x 128 columns) and carries them as the two halves of packed registers:
// v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 process both strips in one instruction at the rate of one FP64 instruction,
// and a DPP lane shift moves both.  Same schedule as k_sweep0r (stage st on row e - st, window rows e - 2 NS .. e + 3, image
// rows through a wave-private LDS ring - here of float pairs {A, B}), no trailing operator stage (the Krylov product needs
// float64 and runs as its own kernel in this mode).  The strips at the left / right side of the image (ghost columns) are
// left to k_sweep0r: they are a fifth of the image at 1024^2.  x and b are float64 in HBM and converted on the way.
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

struct NbrP { f2 u[9], w[9], g[9]; };

// gs0_point in packed float32 (interior points: no corner factors), reference quirk 'dy' == 'dx' built in.  Same formulas
// as gs0_point; the 2x2 determinant is inverted with v_rcp_f32 + one Newton step.
__device__ __forceinline__ void gs0_point_p(const f2* im, const NbrP& n, f2 alpha, f2 beta, f2 inv_g, f2 b0, f2 b1, f2 b2, f2& u, f2& w, f2& gm) {
    const f2 P = im[4];
    const f2 Dx = (im[7] - im[1]) * 0.5f;
    const f2 Dxx = f2_fma(f2_bc(-2.0f), P, im[7] + im[1]);
    const f2 Dyy = f2_fma(f2_bc(-2.0f), P, im[5] + im[3]);
    const f2 Dxy = (im[8] - im[6] - im[2] + im[0]) * 0.25f;
    const f2 PP = P * P, PDx = P * Dx, hP = P * 0.5f;
    const f2 A1 = PP + alpha, qPP = PP * 0.25f, hPDx = PDx * 0.5f;
    const f2 du71 = n.u[7] - n.u[1], du53 = n.u[5] - n.u[3], dw71 = n.w[7] - n.w[1], dw53 = n.w[5] - n.w[3];
    const f2 W4 = n.w[0] + n.w[8] - n.w[2] - n.w[6];
    const f2 U4 = n.u[0] + n.u[8] - n.u[2] - n.u[6];
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

// grid: (number of strip pairs) x ny x pairs blocks of one wave; strip pair j = strips bx_first + 2 j, bx_first + 2 j + 1
template <int NS, bool EC, bool FROM_ZERO, typename ET, int PO>
__global__ __launch_bounds__(64) void k_sweep0p(
    Fine0 pol, int ni, int nj, int TI, int bx_first, int nxp, int ny, int nz, const double* __restrict__ x_in,
    double* __restrict__ x_out, const double* __restrict__ b, const int* __restrict__ active,
    const ET* __restrict__ ecoarse, int nci, int ncj) {
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
    const int qsA = (bx_first + 2 * bxp) * G::OUT - G::HALO, qsB = qsA + G::OUT;   // both strips lie inside the image: 0 <= qs, qs + 128 <= nj
    const size_t npts = (size_t)ni * nj, off = (size_t)pair * 3 * npts;
    const double* xin = FROM_ZERO ? nullptr : x_in + off;
    double* xout = x_out + off;
    const double* bp = b + off;
    const size_t ncpts = (size_t)nci * ncj;
    const ET* ec = EC ? ecoarse + (size_t)pair * 3 * ncpts : nullptr;
    double alpha_d = pol.alpha, beta_d = pol.beta;
    int fidx = pair;
    if (pol.pp) { alpha_d = pol.pp[pair].alpha; beta_d = pol.pp[pair].beta; fidx = pol.pp[pair].frame; }
    const double* img = pol.frames + (size_t)fidx * pol.frame_stride;
    const int Nj = pol.Nj;
    const f2 alpha = f2_bc((float)alpha_d), beta = f2_bc((float)beta_d), inv_g = f2_bc((float)(1.0 / (-1 - 4 * beta_d)));
    const size_t qgA = (size_t)(qsA + 2 * lane), qgB = (size_t)(qsB + 2 * lane);
    const size_t xqA = (size_t)(qsA + 128), xqB = (size_t)(qsB + 128);       // image pair 64 of a strip (columns 128, 129)
    const bool st_ok = lane >= G::HALO / 2 && lane < (W - G::HALO) / 2;      // owned column pairs (the same lanes in both strips)
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
    auto pack = [](double a, double b2) { return f2{(float)a, (float)b2}; };

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
                    const double* sa = xin + (size_t)pL * nj + qgA;
                    const double* sb = xin + (size_t)pL * nj + qgB;
                    const double2 ua = *reinterpret_cast<const double2*>(sa), ub = *reinterpret_cast<const double2*>(sb);
                    const double2 wa = *reinterpret_cast<const double2*>(sa + npts), wb = *reinterpret_cast<const double2*>(sb + npts);
                    const double2 ga = *reinterpret_cast<const double2*>(sa + 2 * npts), gb = *reinterpret_cast<const double2*>(sb + 2 * npts);
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
                    const double2 ia = *reinterpret_cast<const double2*>(frow + qgA), ib = *reinterpret_cast<const double2*>(frow + qgB);
                    const double2 xa = *reinterpret_cast<const double2*>(frow + xqA), xb = *reinterpret_cast<const double2*>(frow + xqB);
                    lix_[r] = pack(ia.x, ib.x); liy_[r] = pack(ia.y, ib.y);
                    lxx_[r] = pack(xa.x, xb.x); lxy_[r] = pack(xa.y, xb.y);
                }
            }
        }
        if constexpr (EC) {
            const int knew = ((p0 + e + 2) >> 1) + 2;
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                crv[f] = f2_bc(0.f);
                if (EDGE ? (knew >= 0 && knew < nci) : true) {
                    const ET* er = ec + (size_t)f * ncpts + (size_t)knew * ncj;
                    crv[f] = f2{(float)er[cqsA + lane], (float)er[cqsB + lane]};   // (interior strips: every lane's coarse column exists)
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
            const double* ba = bp + (size_t)p * nj + qgA;
            const double* bb = bp + (size_t)p * nj + qgB;
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                const double2 va = *reinterpret_cast<const double2*>(ba + (size_t)f * npts), vb = *reinterpret_cast<const double2*>(bb + (size_t)f * npts);
                Bx[st][f] = pack(va.x, vb.x);
                By[st][f] = pack(va.y, vb.y);
            }
        }
    };

    auto stage = [&](auto edge_tag, auto jc_tag, int rr, auto st_tag) {
        constexpr bool EDGE = decltype(edge_tag)::value;
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
            if (par == 0) {
                n.u[1] = RU.ux; n.w[1] = RU.wx; n.g[1] = RU.gx;
                n.u[7] = RD.ux; n.w[7] = RD.wx; n.g[7] = RD.gx;
                n.u[2] = RU.uy; n.w[2] = RU.wy;
                n.u[5] = RC.uy; n.w[5] = RC.wy; n.g[5] = RC.gy;
                n.u[8] = RD.uy; n.w[8] = RD.wy;
                n.u[0] = f2_shr1(RU.uy); n.w[0] = f2_shr1(RU.wy);
                n.u[3] = f2_shr1(RC.uy); n.w[3] = f2_shr1(RC.wy); n.g[3] = f2_shr1(RC.gy);
                n.u[6] = f2_shr1(RD.uy); n.w[6] = f2_shr1(RD.wy);
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
                imv[0] = iuB; imv[1] = iuC; imv[2] = iuD; imv[3] = icB; imv[4] = icC; imv[5] = icD; imv[6] = idB; imv[7] = idC; imv[8] = idD;
            }
            f2 u, w, gm;
            if (par == 0) gs0_point_p(imv, n, alpha, beta, inv_g, Bx[st][0], Bx[st][1], Bx[st][2], u, w, gm);
            else gs0_point_p(imv, n, alpha, beta, inv_g, By[st][0], By[st][1], By[st][2], u, w, gm);
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
        // write-out of rows e - 2 NS, e - 2 NS + 1 (float64 in HBM)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int rrW = e - 2 * NS + r, pW = p0 + rrW;
            const bool rowok = EDGE ? (rrW >= 0 && rrW < TI && pW >= 0 && pW < ni) : true;
            if (rowok && st_ok) {
                const S0PRow& s = X[LO - 2 * NS + r];
                double* da = xout + (size_t)pW * nj + qgA;
                double* db = xout + (size_t)pW * nj + qgB;
                *reinterpret_cast<double2*>(da) = double2{(double)s.ux.x, (double)s.uy.x};
                *reinterpret_cast<double2*>(db) = double2{(double)s.ux.y, (double)s.uy.y};
                *reinterpret_cast<double2*>(da + npts) = double2{(double)s.wx.x, (double)s.wy.x};
                *reinterpret_cast<double2*>(db + npts) = double2{(double)s.wx.y, (double)s.wy.y};
                *reinterpret_cast<double2*>(da + 2 * npts) = double2{(double)s.gx.x, (double)s.gy.x};
                *reinterpret_cast<double2*>(db + 2 * npts) = double2{(double)s.gx.y, (double)s.gy.y};
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
        if constexpr (EC) {   // x + P e for the rows that enter (as k_sweep0r; float32 here)
            const int pL0 = p0 + e + 2;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int pL = pL0 + r;
                const bool rowok = EDGE ? (e + 3 <= TI + 2 * NS - 1 && pL >= 0 && pL < ni) : true;
                if (rowok) {
                    const int cp = pL >> 1;
                    const bool ipi = EDGE ? ((pL & 1) && (cp + 1 < nci)) : (((PO + r) & 1) != 0);
                    // (interior strips: the odd column always has its right coarse neighbour)
                    const int i0 = cp - (pL0 >> 1);
                    S0PRow& d = X[LO + 2 + r];
#pragma unroll
                    for (int f = 0; f < 3; ++f) {
                        const f2 a0 = i0 ? CR[1][f] : CR[0][f];
                        const f2 a1 = i0 ? crv[f] : CR[1][f];
                        const f2 a0r = f2_shl1(a0), a1r = f2_shl1(a1);
                        const float wi0 = ipi ? 0.5f : 1.0f;
                        f2 ve = a0 * wi0;
                        f2 vo = a0 * (wi0 * 0.5f) + a0r * (wi0 * 0.5f);
                        if (ipi) {
                            ve += a1 * 0.5f;
                            vo += a1 * 0.25f + a1r * 0.25f;
                        }
                        if (f == 0) { d.ux += ve; d.uy += vo; }
                        else if (f == 1) { d.wx += ve; d.wy += vo; }
                        else { d.gx += ve; d.gy += vo; }
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
                const int k = k0 + d;
                const bool rowok = k >= 0 && k < nci;
                const ET* er = ec + (size_t)f * ncpts + (size_t)(rowok ? k : 0) * ncj;
                CR[d][f] = rowok ? f2{(float)er[cqsA + lane], (float)er[cqsB + lane]} : f2_bc(0.f);
            }
    }
    for (int s = s_first; s <= s_last; ++s) {
        const int e = 2 * s;
        if (e >= e_lo && e <= e_hi) step(std::false_type{}, e);
        else step(std::true_type{}, e);
    }
}

}  // namespace vof
