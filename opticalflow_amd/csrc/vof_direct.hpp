// vof_direct.hpp - device code of the direct preconditioner: block-tridiagonal LU of the level-0 operator by image rows.
//
// With the unknowns of one image row as a block (m = 3 n_j: field-major inside the row), the 9-point operator is block
// tridiagonal: A = tridiag(L_p, D_p, U_p), L_p = A(row p, row p-1), U_p = A(row p, row p+1) (the mirror rows of the
// reference, OF.py:964-1070, fold onto rows 1 and n-2, i.e. stay inside the pattern).  Block elimination
//     S_0 = D_0,   S_p = D_p - L_p S_{p-1}^{-1} U_{p-1},   T_p = S_p^{-1} (dense m x m, rocSOLVER getrf + getri),
// then z = A^{-1} r by  y_p = r_p - L_p T_{p-1} y_{p-1}  (forward)  and  x_p = T_p (y_p - U_p x_{p+1})  (backward).
// L, D, U are never stored as matrices: their 3x3 blocks come from a per-row table of the folded stencil blocks, so the
// Schur update costs 9 terms per entry instead of a dense product, and only the dense inverses T_p are kept
// (n_i m^2 doubles per pair: 0.23 GB at 150^2, 9.7 GB at 514^2).  Used as the preconditioner of the same Krylov iteration
// and stopping rule as the multigrid cycle (one or two iterations), for the regimes in which the cycle does not converge
//  and for use_direct_solver=True (the reference's SuperLU branch, OF.py:1146-1147).
#pragma once
#include "vof_device.hpp"

namespace vof {

// Stencil blocks of one image row: tab[((oi + 1) * nj + q) * 3 + (oj + 1)][9] = folded 3x3 block of offset (oi, oj) at
// point (p, q); zero where the target lies outside the grid.  Layout per pair and row: 3 * nj * 3 * 9 doubles.
constexpr int DIR_TAB = 81;   // doubles per point

__global__ __launch_bounds__(256) void k_dir_tables(const double* __restrict__ frames, size_t frame_stride, int Nj, double alpha,
                                                    double beta, int quirks, int ni, int nj, double* __restrict__ tabs,
                                                    const PairParam* __restrict__ pp) {
    // grid: (ceil(nj / 256), ni, pairs)
    const int q = blockIdx.x * blockDim.x + threadIdx.x, p = blockIdx.y, pair = blockIdx.z;
    if (q >= nj) return;
    int fidx = pair;
    if (pp) { alpha = pp[pair].alpha; beta = pp[pair].beta; fidx = pp[pair].frame; }
    const PixCoef k = pix_coef(frames + (size_t)fidx * frame_stride, Nj, p, q, quirks);
    double* out = tabs + ((size_t)pair * ni + p) * (size_t)nj * DIR_TAB;
    for (int oi = -1; oi <= 1; ++oi)
        for (int oj = -1; oj <= 1; ++oj) {
            double blk[9];
            const int tp = p + oi, tq = q + oj;
            if (tp < 0 || tp >= ni || tq < 0 || tq >= nj) {
                for (int t = 0; t < 9; ++t) blk[t] = 0.0;
            } else {
                folded_block(k, alpha, beta, p, q, ni, nj, oi, oj, blk);
            }
            double* o = out + (((size_t)(oi + 1) * nj + q) * 3 + (oj + 1)) * 9;
            for (int t = 0; t < 9; ++t) o[t] = blk[t];
        }
}

// entry (i, j) of the block (oi) of row p from its table: i = r * nj + q (row field r, column q), j = c * nj + q'
__device__ __forceinline__ double dir_entry(const double* __restrict__ tab, int nj, int oi, int i, int j) {
    const int r = i / nj, q = i - r * nj, c = j / nj, q2 = j - c * nj;
    const int oj = q2 - q;
    if (oj < -1 || oj > 1) return 0.0;
    return tab[(((size_t)(oi + 1) * nj + q) * 3 + (oj + 1)) * 9 + r * 3 + c];
}

// W = T_{p-1} U_{p-1}  (m x m, column-major): W[i, j] = sum_k T[i, k] U(k, j), U(k, j) != 0 only for the <= 9 rows k = (r, q_j + d)
// (ld: leading dimension of the dense blocks T, W, S - m, or m rounded up to the tile size of the blocked inverse)
__global__ __launch_bounds__(256) void k_dir_W(const double* __restrict__ T, size_t strideT, const double* __restrict__ tabs_prev,
                                               size_t stride_tab, int nj, double* __restrict__ W, size_t strideW, int ld) {
    const int m = 3 * nj;
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y, pair = blockIdx.z;
    if (i >= m) return;
    const double* Tp = T + (size_t)pair * strideT;
    const double* tab = tabs_prev + (size_t)pair * stride_tab;   // table of row p - 1: U_{p-1} = its oi = +1 blocks
    const int c = j / nj, q2 = j - c * nj;
    double acc = 0.0;
#pragma unroll
    for (int d = -1; d <= 1; ++d) {
        const int qk = q2 + d;          // column of the row-(p-1) point whose (+1, q2 - qk) block reaches (p, q2)
        if (qk < 0 || qk >= nj) continue;
        const double* blk = tab + (((size_t)2 * nj + qk) * 3 + (q2 - qk + 1)) * 9;
#pragma unroll
        for (int r = 0; r < 3; ++r) acc += Tp[(size_t)(r * nj + qk) * ld + i] * blk[r * 3 + c];
    }
    W[(size_t)pair * strideW + (size_t)j * ld + i] = acc;
}

// S = D_p - L_p W  (W = nullptr: S = D_p), written column-major into the slot of T_p
__global__ __launch_bounds__(256) void k_dir_schur(const double* __restrict__ tabs_row, size_t stride_tab, int nj,
                                                   const double* __restrict__ W, size_t strideW, double* __restrict__ S,
                                                   size_t strideS, int ld) {
    const int m = 3 * nj;
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y, pair = blockIdx.z;   // grid: ceil(ld / 256) x ld x pairs
    if (i >= ld) return;
    if (i >= m || j >= m) {   // padding up to the tile size of the blocked inverse: the identity (inverse: the identity)
        S[(size_t)pair * strideS + (size_t)j * ld + i] = (i == j) ? 1.0 : 0.0;
        return;
    }
    const double* tab = tabs_row + (size_t)pair * stride_tab;
    double v = dir_entry(tab, nj, 0, i, j);
    if (W) {
        const double* Wp = W + (size_t)pair * strideW + (size_t)j * ld;
        const int r = i / nj, q = i - r * nj;
#pragma unroll
        for (int d = -1; d <= 1; ++d) {
            const int qk = q + d;
            if (qk < 0 || qk >= nj) continue;
            const double* blk = tab + (((size_t)0 * nj + q) * 3 + (d + 1)) * 9;   // L_p: oi = -1 blocks of row p
#pragma unroll
            for (int c = 0; c < 3; ++c) v -= blk[r * 3 + c] * Wp[c * nj + qk];
        }
    }
    S[(size_t)pair * strideS + (size_t)j * ld + i] = v;
}

// In-place inverse of a dense m x m matrix (column-major) by Gauss-Jordan elimination with partial pivoting, one workgroup
// of 1024 threads per matrix of the batch.  Row exchanges are recorded and undone as column exchanges at the end
// (A^{-1} = (P A)^{-1} P).  Every step rewrites the whole matrix through one CU (~2 x 8 m^2 bytes), i.e. ~10 ms per 444 x 444
// matrix: fine for the down-sampled images the reference runs its parameter sweeps on (150^2: 148 Schur blocks per pair,
// pairs in parallel on different CUs); wide images go to rocSOLVER instead (vof.hip).  info[pair] = 1: a zero pivot.
__global__ __launch_bounds__(1024) void k_dir_invert(double* __restrict__ A, size_t strideA, int m, int* __restrict__ ipiv,
                                                     int* __restrict__ info) {
    extern __shared__ double dinv_sh[];
    double* prow = dinv_sh;        // row k before the step
    double* pcol = dinv_sh + m;    // column k before the step
    __shared__ double s_best[16];
    __shared__ int s_bi[16];
    __shared__ int s_piv;
    const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double* Ap = A + (size_t)pair * strideA;
    int* ip = ipiv + (size_t)pair * m;
    bool singular = false;
    for (int k = 0; k < m; ++k) {
        // pivot: largest |a_ik|, i >= k (lowest index on ties)
        double best = -1.0;
        int bi = k;
        for (int i = k + tid; i < m; i += 1024) {
            const double v = fabs(Ap[(size_t)k * m + i]);
            if (v > best) { best = v; bi = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double v2 = __shfl_down(best, o, 64);
            const int i2 = __shfl_down(bi, o, 64);
            if (v2 > best || (v2 == best && i2 < bi)) { best = v2; bi = i2; }
        }
        if (lane == 0) { s_best[wv] = best; s_bi[wv] = bi; }
        __syncthreads();
        if (tid == 0) {
            double b = s_best[0];
            int q = s_bi[0];
            for (int w = 1; w < 16; ++w)
                if (s_best[w] > b || (s_best[w] == b && s_bi[w] < q)) { b = s_best[w]; q = s_bi[w]; }
            s_piv = q;
            ip[k] = q;
        }
        __syncthreads();
        const int piv = s_piv;
        if (piv != k)
            for (int j = tid; j < m; j += 1024) {
                const double a = Ap[(size_t)j * m + k], b2 = Ap[(size_t)j * m + piv];
                Ap[(size_t)j * m + k] = b2;
                Ap[(size_t)j * m + piv] = a;
            }
        __syncthreads();
        for (int j = tid; j < m; j += 1024) { prow[j] = Ap[(size_t)j * m + k]; pcol[j] = Ap[(size_t)k * m + j]; }
        __syncthreads();
        const double pv = prow[k];
        if (pv == 0.0) singular = true;
        const double pinv = pv != 0.0 ? 1.0 / pv : 0.0;
        const size_t mm = (size_t)m * m;
        for (size_t idx = tid; idx < mm; idx += 1024) {
            const int j = (int)(idx / m), i = (int)(idx - (size_t)j * m);
            double v;
            if (i == k) v = (j == k) ? pinv : prow[j] * pinv;
            else if (j == k) v = -pcol[i] * pinv;
            else v = Ap[idx] - pcol[i] * (prow[j] * pinv);
            Ap[idx] = v;
        }
        __syncthreads();
    }
    for (int k = m - 1; k >= 0; --k) {   // undo the row exchanges: columns k <-> ipiv[k], last first
        const int piv = ip[k];
        if (piv != k)
            for (int i = tid; i < m; i += 1024) {
                const double a = Ap[(size_t)k * m + i], b2 = Ap[(size_t)piv * m + i];
                Ap[(size_t)k * m + i] = b2;
                Ap[(size_t)piv * m + i] = a;
            }
        __syncthreads();
    }
    if (tid == 0 && info && singular) info[pair] = 1;   // (accumulates over the image rows: cleared once per set-up)
}

// ------------------------------------------------------------------------------------------
// Blocked in-place inverse of dense ld x ld matrices (column-major, ld a multiple of DNB = 64) for the wide images
// (m = 3 n_j > 640: 258^2 ... 1026^2 and beyond), batched over the frame pairs: block Gauss-Jordan over 64 x 64 tiles,
//   for every tile index k:   D = A[k,k]^-1 (partial pivoting INSIDE the tile),  R[J] = D A[k,J],  C[I] = A[I,k],
//                             A[I,J] -= C[I] R[J] (I, J != k),  A[I,k] = -C[I] D,  A[k,J] = R[J],  A[k,k] = D,
// two launches per tile index (panel, update), every 64 x 64 x 64 product on the FP64 matrix cores
// (v_mfma_f64_16x16x4_f64).  No pivoting ACROSS tiles: the Schur blocks S_p inherit the diagonal dominance of the row
// blocks D_p (|4 alpha + 2 P^2| on the diagonal against 2 alpha beside it), and the result is only ever used as the
// preconditioner of the Krylov iteration whose independent residual decides `converged`.  2 m^3 flops per block as any
// inverse; round 2 used rocSOLVER getrf + getri here (a 0.9 GB library on the product path, loaded with dlopen).
// ------------------------------------------------------------------------------------------
constexpr int DNB = 64, DNB_LDB = DNB + 1;
typedef double dir_d4 __attribute__((ext_vector_type(4)));

// C tile (64 x 64) = A-operand (ldsA[kk][i], i contiguous) x B-operand (ldsB[j][kk], row stride DNB_LDB); wave w owns rows
// 16 w .. 16 w + 15.  The product is formed transposed (D' = B^T A^T) so that a lane's results are consecutive ROWS of one
// column - contiguous in the column-major matrices: acc[jt][reg] = C[16 w + (lane & 15)][16 jt + (lane >> 4) + 4 reg].
__device__ __forceinline__ void dir_tile_mma(const double* __restrict__ ldsA, const double* __restrict__ ldsB, int w, int lane, dir_d4 (&acc)[4]) {
    const int r16 = lane & 15, k4 = lane >> 4;
#pragma unroll 4
    for (int k0 = 0; k0 < DNB; k0 += 4) {
        const double bop = ldsA[(k0 + k4) * DNB + 16 * w + r16];          // (A^T)[k][i]
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            const double aop = ldsB[(16 * jt + r16) * DNB_LDB + k0 + k4];  // (B^T)[j][k]
            acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc[jt], 0, 0, 0);
        }
    }
}

// In-LDS inverse of the 64 x 64 tile a[i][j] (row stride DNB_LDB) by Gauss-Jordan with partial pivoting inside the tile,
// 256 threads; returns true if a pivot was exactly zero.
__device__ __forceinline__ bool dir_tile_invert(double* a, double* prow, double* pcol, int* ipiv, int tid) {
    __shared__ int s_piv;
    bool singular = false;
    for (int k = 0; k < DNB; ++k) {
        if (tid < 64) {   // wave 0: largest |a[i][k]|, i >= k (lowest index on ties)
            double best = tid >= k ? fabs(a[tid * DNB_LDB + k]) : -1.0;
            int bi = tid;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double v2 = __shfl_down(best, o, 64);
                const int i2 = __shfl_down(bi, o, 64);
                if (v2 > best || (v2 == best && i2 < bi)) { best = v2; bi = i2; }
            }
            if (tid == 0) { s_piv = bi; ipiv[k] = bi; }
        }
        __syncthreads();
        const int piv = s_piv;
        if (piv != k && tid < DNB) {   // row exchange k <-> piv
            const double t0 = a[k * DNB_LDB + tid], t1 = a[piv * DNB_LDB + tid];
            a[k * DNB_LDB + tid] = t1;
            a[piv * DNB_LDB + tid] = t0;
        }
        __syncthreads();
        if (tid < DNB) { prow[tid] = a[k * DNB_LDB + tid]; pcol[tid] = a[tid * DNB_LDB + k]; }
        __syncthreads();
        const double pv = prow[k];
        if (pv == 0.0) singular = true;
        const double pinv = pv != 0.0 ? 1.0 / pv : 0.0;
        for (int idx = tid; idx < DNB * DNB; idx += 256) {
            const int i = idx >> 6, j = idx & 63;
            double v;
            if (i == k) v = (j == k) ? pinv : prow[j] * pinv;
            else if (j == k) v = -pcol[i] * pinv;
            else v = a[i * DNB_LDB + j] - pcol[i] * (prow[j] * pinv);
            a[i * DNB_LDB + j] = v;
        }
        __syncthreads();
    }
    for (int k = DNB - 1; k >= 0; --k) {   // undo the row exchanges: columns k <-> ipiv[k], last first
        const int piv = ipiv[k];
        if (piv != k && tid < DNB) {
            const double t0 = a[tid * DNB_LDB + k], t1 = a[tid * DNB_LDB + piv];
            a[tid * DNB_LDB + k] = t1;
            a[tid * DNB_LDB + piv] = t0;
        }
        __syncthreads();
    }
    return singular;
}

// Panel launch of tile index k: grid (nt, 2, pairs).  y = 0, x = J: R[J] = D A[k,J] into Rbuf (tile J = k: D itself into
// Dbuf); y = 1, x = I: copy of the column-panel tile A[I,k] into Cbuf.  Every y = 0 block inverts the diagonal tile itself
// (6 us, in parallel) instead of waiting for a separate launch.  Tiles in the buffers are 64 x 64 column-major.
__global__ __launch_bounds__(256) void k_dir_bgj_panel(const double* __restrict__ A, size_t strideA, int ld, int kt, double* __restrict__ Rbuf,
                                                       double* __restrict__ Cbuf, double* __restrict__ Dbuf, int* __restrict__ info) {
    extern __shared__ double bgj_sh[];
    double* shA = bgj_sh;                       // [64][64]      A-operand [kk][i]
    double* shB = bgj_sh + DNB * DNB;           // [64][65]      B-operand [j][kk] / the tile being inverted [i][j]
    __shared__ double prow[DNB], pcol[DNB];
    __shared__ int ipiv[DNB];
    const int nt = ld / DNB, pair = blockIdx.z, t = blockIdx.x, tid = threadIdx.x;
    const double* Ap = A + (size_t)pair * strideA;
    const size_t tile = (size_t)DNB * DNB;
    if (blockIdx.y == 1) {   // column-panel copy
        double* dst = Cbuf + ((size_t)pair * nt + t) * tile;
        for (int idx = tid; idx < DNB * DNB; idx += 256) {
            const int kk = idx >> 6, i = idx & 63;
            dst[idx] = Ap[(size_t)(kt * DNB + kk) * ld + t * DNB + i];
        }
        return;
    }
    // the diagonal tile, as a[i][j]
    for (int idx = tid; idx < DNB * DNB; idx += 256) {
        const int j = idx >> 6, i = idx & 63;
        shB[i * DNB_LDB + j] = Ap[(size_t)(kt * DNB + j) * ld + kt * DNB + i];
    }
    __syncthreads();
    const bool singular = dir_tile_invert(shB, prow, pcol, ipiv, tid);
    if (t == kt) {
        double* dst = Dbuf + (size_t)pair * tile;
        for (int idx = tid; idx < DNB * DNB; idx += 256) {
            const int j = idx >> 6, i = idx & 63;
            dst[idx] = shB[i * DNB_LDB + j];
        }
        if (tid == 0 && singular && info) info[pair] = 1;
        return;
    }
    // A-operand: D as [kk][i]; then the B-operand A[k, t] as [j][kk] takes the place of the tile
    for (int idx = tid; idx < DNB * DNB; idx += 256) {
        const int kk = idx >> 6, i = idx & 63;
        shA[kk * DNB + i] = shB[i * DNB_LDB + kk];
    }
    __syncthreads();
    for (int idx = tid; idx < DNB * DNB; idx += 256) {
        const int j = idx >> 6, kk = idx & 63;
        shB[j * DNB_LDB + kk] = Ap[(size_t)(t * DNB + j) * ld + kt * DNB + kk];
    }
    __syncthreads();
    dir_d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const int w = tid >> 6, lane = tid & 63;
    dir_tile_mma(shA, shB, w, lane, acc);
    double* dst = Rbuf + ((size_t)pair * nt + t) * tile;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(size_t)(16 * jt + (lane >> 4) + 4 * r) * DNB + 16 * w + (lane & 15)] = acc[jt][r];
}

// Update launch of tile index k: grid (nt, nt, pairs), block (I, J).
__global__ __launch_bounds__(256) void k_dir_bgj_update(double* __restrict__ A, size_t strideA, int ld, int kt, const double* __restrict__ Rbuf,
                                                        const double* __restrict__ Cbuf, const double* __restrict__ Dbuf) {
    extern __shared__ double bgj_sh[];
    double* shA = bgj_sh;
    double* shB = bgj_sh + DNB * DNB;
    const int nt = ld / DNB, pair = blockIdx.z, I = blockIdx.x, J = blockIdx.y, tid = threadIdx.x;
    double* Ap = A + (size_t)pair * strideA;
    const size_t tile = (size_t)DNB * DNB;
    const double* Rt = Rbuf + ((size_t)pair * nt + J) * tile;
    const double* Ct = Cbuf + ((size_t)pair * nt + I) * tile;
    const double* Dt = Dbuf + (size_t)pair * tile;
    if (I == kt) {   // pivot row of tiles: A[k,J] = R[J], A[k,k] = D
        const double* src = (J == kt) ? Dt : Rt;
        for (int idx = tid; idx < DNB * DNB; idx += 256) {
            const int j = idx >> 6, i = idx & 63;
            Ap[(size_t)(J * DNB + j) * ld + kt * DNB + i] = src[idx];
        }
        return;
    }
    const double* Bsrc = (J == kt) ? Dt : Rt;     // 64 x 64 column-major: element (kk, j) at [j * 64 + kk]
    for (int idx = tid; idx < DNB * DNB; idx += 256) {
        const int hi = idx >> 6, lo = idx & 63;
        shA[idx] = Ct[idx];                       // [kk][i]
        shB[hi * DNB_LDB + lo] = Bsrc[idx];       // [j][kk]
    }
    __syncthreads();
    dir_d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const int w = tid >> 6, lane = tid & 63;
    dir_tile_mma(shA, shB, w, lane, acc);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double* dst = Ap + (size_t)(J * DNB + 16 * jt + (lane >> 4) + 4 * r) * ld + I * DNB + 16 * w + (lane & 15);
            *dst = (J == kt) ? -acc[jt][r] : *dst - acc[jt][r];
        }
}

// y = T x for one row block and every pair: 64 rows per block, the k range split over 4 thread groups
__global__ __launch_bounds__(256) void k_dir_gemv(const double* __restrict__ T, size_t strideT, int m, const double* __restrict__ x,
                                                  size_t stridex, double* __restrict__ y, size_t stridey, int ld) {
    __shared__ double part[4][64];
    const int pair = blockIdx.y;
    const int li = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + li;
    const double* Tp = T + (size_t)pair * strideT;
    const double* xp = x + (size_t)pair * stridex;
    double acc = 0.0;
    if (i < m)
        for (int k = g; k < m; k += 4) acc += Tp[(size_t)k * ld + i] * xp[k];
    part[g][li] = acc;
    __syncthreads();
    if (g == 0 && i < m) y[(size_t)pair * stridey + i] = part[0][li] + part[1][li] + part[2][li] + part[3][li];
}

// out = a - B t, B = the oi blocks of row p's table (oi = -1: L_p acting on row p - 1, oi = +1: U_p acting on row p + 1); t == nullptr: out = a
__global__ __launch_bounds__(256) void k_dir_rowupdate(const double* __restrict__ tabs_row, size_t stride_tab, int nj, int oi,
                                                       const double* __restrict__ a, size_t stridea, const double* __restrict__ t,
                                                       size_t stridet, double* __restrict__ out, size_t strideo) {
    const int m = 3 * nj;
    const int i = blockIdx.x * blockDim.x + threadIdx.x, pair = blockIdx.y;
    if (i >= m) return;
    double v = a[(size_t)pair * stridea + i];
    if (t) {
        const double* tab = tabs_row + (size_t)pair * stride_tab;
        const double* tp = t + (size_t)pair * stridet;
        const int r = i / nj, q = i - r * nj;
#pragma unroll
        for (int d = -1; d <= 1; ++d) {
            const int qk = q + d;
            if (qk < 0 || qk >= nj) continue;
            const double* blk = tab + (((size_t)(oi + 1) * nj + q) * 3 + (d + 1)) * 9;
#pragma unroll
            for (int c = 0; c < 3; ++c) v -= blk[r * 3 + c] * tp[c * nj + qk];
        }
    }
    out[(size_t)pair * strideo + i] = v;
}

// SoA level-0 vector [pair][3][ni][nj] <-> row-block layout [pair][ni][3][nj]
template <typename VT, bool TO_BLOCKS>
__global__ __launch_bounds__(256) void k_dir_permute(VT* __restrict__ soa, double* __restrict__ blocks, int ni, int nj) {
    const size_t npts = (size_t)ni * nj, len = 3 * npts;
    const int pair = blockIdx.y;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < len; t += (size_t)gridDim.x * blockDim.x) {
        const int f = (int)(t / npts);
        const size_t r = t - (size_t)f * npts;
        const int p = (int)(r / nj), q = (int)(r - (size_t)p * nj);
        const size_t b = ((size_t)p * 3 + f) * nj + q;
        if constexpr (TO_BLOCKS) blocks[(size_t)pair * len + b] = (double)soa[(size_t)pair * len + t];
        else soa[(size_t)pair * len + t] = (VT)blocks[(size_t)pair * len + b];
    }
}

}  // namespace vof
