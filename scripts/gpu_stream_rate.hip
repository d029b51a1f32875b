// What does a pure streaming kernel reach on this GPU?  (ceiling for the BiCGStab vector updates: 5 reads + 2 writes per element)
// build: hipcc -O3 --offload-arch=gfx950 -o gpu_stream_rate scripts/gpu_stream_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));
template <bool NT> __device__ __forceinline__ v2d ld(const v2d* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(v2d* p, v2d v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// x += a y + w z ; r = s - w t ; (r, r), (h, r): the access pattern of k_update_xr
template <bool NT, int U>
__global__ __launch_bounds__(256) void k_xr(v2d* x, const v2d* y, const v2d* z, v2d* r, const v2d* t, const v2d* h, size_t n2, double a, double w, double* out) {
    double rr = 0, rho = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n2; i0 += stride * U) {
        v2d X[U], Y[U], Z[U], R[U], T[U], H[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { size_t i = i0 + u * stride; if (i < n2) { X[u] = ld<NT>(x + i); Y[u] = ld<NT>(y + i); Z[u] = ld<NT>(z + i); R[u] = ld<NT>(r + i); T[u] = ld<NT>(t + i); H[u] = ld<NT>(h + i); } }
#pragma unroll
        for (int u = 0; u < U; ++u) { size_t i = i0 + u * stride; if (i < n2) {
            X[u] += a * Y[u] + w * Z[u]; R[u] -= w * T[u]; st<NT>(x + i, X[u]); st<NT>(r + i, R[u]);
            rr += R[u].x * R[u].x + R[u].y * R[u].y; rho += H[u].x * R[u].x + H[u].y * R[u].y; } }
    }
    if (rr == 12345.678 && rho == 1.0) out[0] = rr;   // keep the sums alive
}
template <bool NT, int U>
__global__ __launch_bounds__(256) void k_copy(v2d* x, const v2d* y, size_t n2) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n2; i0 += stride * U) {
        v2d Y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { size_t i = i0 + u * stride; if (i < n2) Y[u] = ld<NT>(y + i); }
#pragma unroll
        for (int u = 0; u < U; ++u) { size_t i = i0 + u * stride; if (i < n2) st<NT>(x + i, Y[u]); }
    }
}
template <bool NT, int U>
__global__ __launch_bounds__(256) void k_read(const v2d* y, size_t n2, double* out) {
    const size_t stride = (size_t)gridDim.x * 256;
    v2d acc = {0, 0};
    for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n2; i0 += stride * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) { size_t i = i0 + u * stride; if (i < n2) acc += ld<NT>(y + i); }
    }
    if (acc.x == 12345.678) out[0] = acc.y;
}
int main() {
    const size_t n = (size_t)170 * 3 * 1022 * 1022;   // one 170-pair batch of level-0 vectors (4.26 GB each)
    const size_t n2 = n / 2;
    double* buf[6]; double* out;
    for (auto& b : buf) { CHK(hipMalloc(&b, n * 8)); CHK(hipMemset(b, 0, n * 8)); }
    CHK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, double bytes, auto launch) {
        launch(); CHK(hipDeviceSynchronize());
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) { CHK(hipEventRecord(e0)); launch(); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; }
        printf("%-44s %8.3f ms  %7.0f GB/s\n", name, best, bytes / best / 1e6); fflush(stdout);
        return 0;
    };
    v2d** B = reinterpret_cast<v2d**>(buf);
    for (int blocks : {1024, 2048, 4096, 8192, 16384, 65536}) {
        char nm[128];
#define RUN(K, NT, U, BYTES, ...) snprintf(nm, sizeof nm, #K "<nt=%d, unroll %d> %d blocks", NT, U, blocks); if (timeit(nm, BYTES, [&] { K<NT, U><<<blocks, 256>>>(__VA_ARGS__); })) return 1;
        RUN(k_xr, true, 1, 7.0 * n * 8, B[0], B[1], B[2], B[3], B[4], B[5], n2, 0.5, 0.25, out)
        RUN(k_xr, true, 2, 7.0 * n * 8, B[0], B[1], B[2], B[3], B[4], B[5], n2, 0.5, 0.25, out)
        RUN(k_xr, false, 1, 7.0 * n * 8, B[0], B[1], B[2], B[3], B[4], B[5], n2, 0.5, 0.25, out)
        RUN(k_xr, false, 2, 7.0 * n * 8, B[0], B[1], B[2], B[3], B[4], B[5], n2, 0.5, 0.25, out)
        RUN(k_copy, true, 4, 2.0 * n * 8, B[0], B[1], n2)
        RUN(k_copy, false, 4, 2.0 * n * 8, B[0], B[1], n2)
        RUN(k_read, true, 4, 1.0 * n * 8, B[1], n2, out)
        RUN(k_read, false, 4, 1.0 * n * 8, B[1], n2, out)
    }
    return 0;
}
