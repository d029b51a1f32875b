// Issue cost of the vector instructions the level-0 smoother is made of (cycles per wave64 instruction and SIMD), measured
// with long dependent / independent chains: hipcc --offload-arch=gfx950 -O3 scripts/gpu_valu_rate.hip -o /tmp/vr && /tmp/vr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float pf2 __attribute__((ext_vector_type(2)));
template <int ILP, int DEP>
__global__ __launch_bounds__(64) void kpk(double* out, int iters, double seed) {   // v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32
    pf2 a[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) a[i] = pf2{(float)seed + threadIdx.x + i, (float)seed - i};
    const pf2 m = pf2{1.0000001f, 0.9999999f}, c = pf2{1e-9f, 2e-9f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (DEP == 0) a[i] = __builtin_elementwise_fma(a[i], m, c);
                else if (DEP == 1) a[i] = a[i] * m;
                else a[i] = a[i] + c;
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int KIND, int ILP>
__global__ __launch_bounds__(64) void k(double* out, int iters, double seed) {
    double a[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) a[i] = seed + threadIdx.x + i;
    const double m = 1.0000001, c = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (KIND == 0) a[i] = __builtin_fma(a[i], m, c);                      // v_fma_f64
                else if (KIND == 1) a[i] = a[i] + c;                                  // v_add_f64
                else if (KIND == 2) a[i] = a[i] * m;                                  // v_mul_f64
                else if (KIND == 3) {                                                 // 2 x v_mov_b32_dpp wave_shr:1
                    int lo = __double2loint(a[i]), hi = __double2hiint(a[i]);
                    lo = __builtin_amdgcn_mov_dpp(lo, 0x138, 0xF, 0xF, true);
                    hi = __builtin_amdgcn_mov_dpp(hi, 0x138, 0xF, 0xF, true);
                    a[i] = __hiloint2double(hi, lo);
                } else if (KIND == 4) {                                               // v_fma_f32
                    float f = (float)a[i]; f = __builtin_fmaf(f, 1.0000001f, 1e-9f); a[i] = f;
                } else if (KIND == 5) a[i] = __builtin_amdgcn_rcp(a[i]);               // v_rcp_f64
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int KIND, int ILP>
int run(const char* name, int waves_per_simd, int per_elem) {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount, blocks = cus * 4 * waves_per_simd, iters = 20000;
    double* out;
    CHECK(hipMalloc(&out, (size_t)blocks * 64 * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<KIND, ILP><<<blocks, 64>>>(out, 100, 1.0);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k<KIND, ILP><<<blocks, 64>>>(out, iters, 1.0);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    int clk_khz = 0;
    CHECK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
    const double instr_per_wave = (double)iters * 16 * ILP * per_elem;
    const double ns_per_instr_simd = ms * 1e6 / (instr_per_wave * waves_per_simd);
    printf("%-28s ILP %d, %d wave(s)/SIMD: %7.2f ms, %6.2f ns per wave instruction and SIMD = %5.2f cycles at the nominal %d MHz\n", name, ILP,
           waves_per_simd, ms, ns_per_instr_simd, ns_per_instr_simd * clk_khz * 1e-6, clk_khz / 1000);
    hipFree(out);
    return 0;
}

template <int ILP, int DEP>
int runpk(const char* name, int waves_per_simd) {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount, blocks = cus * 4 * waves_per_simd, iters = 20000;
    double* out;
    CHECK(hipMalloc(&out, (size_t)blocks * 64 * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    kpk<ILP, DEP><<<blocks, 64>>>(out, 100, 1.0);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    kpk<ILP, DEP><<<blocks, 64>>>(out, iters, 1.0);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ns = ms * 1e6 / ((double)iters * 16 * ILP * waves_per_simd);
    printf("%-28s ILP %d, %d wave(s)/SIMD: %7.2f ms, %6.2f ns per wave instruction and SIMD = %5.2f cycles at the nominal 2400 MHz\n", name, ILP, waves_per_simd, ms, ns, ns * 2.4);
    hipFree(out);
    return 0;
}

int main() {
    runpk<8, 0>("v_pk_fma_f32 independent", 1);
    runpk<8, 0>("v_pk_fma_f32 independent", 2);
    runpk<1, 0>("v_pk_fma_f32 dependent", 1);
    runpk<8, 1>("v_pk_mul_f32 independent", 1);
    runpk<8, 2>("v_pk_add_f32 independent", 1);
    run<0, 1>("v_fma_f64 dependent", 1, 1);
    run<0, 8>("v_fma_f64 independent", 1, 1);
    run<0, 8>("v_fma_f64 independent", 2, 1);
    run<1, 8>("v_add_f64 independent", 1, 1);
    run<1, 1>("v_add_f64 dependent", 1, 1);
    run<2, 8>("v_mul_f64 independent", 1, 1);
    run<2, 8>("v_mul_f64 independent", 2, 1);
    run<3, 8>("v_mov_b32_dpp wave_shr", 1, 2);
    run<3, 1>("v_mov_b32_dpp dependent", 1, 2);
    run<4, 8>("v_fma_f32 (+2 cvt)", 1, 3);
    run<5, 4>("v_rcp_f64", 1, 1);
    return 0;
}
