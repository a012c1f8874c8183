// Issue-rate microbenchmark: how many SIMD cycles does a wave64 v_exp_f32 / v_exp_f16 / v_rcp_f16 / v_pk_mul_f32 take?
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/trans_rate.hip -o /tmp/trans_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#define REP8(OP) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
        if constexpr (MODE == 0) {
#define OPX(v) asm volatile("v_exp_f32 %0, %0" : "+v"(v));
            REP8(OPX) REP8(OPX)
        } else if constexpr (MODE == 1) {
#define OPH(v) asm volatile("v_exp_f16 %0, %0" : "+v"(v));
            REP8(OPH) REP8(OPH)
        } else if constexpr (MODE == 2) {
#define OPR(v) asm volatile("v_rcp_f16 %0, %0" : "+v"(v));
            REP8(OPR) REP8(OPR)
        } else if constexpr (MODE == 3) {
#define OPM(v) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(v));
            REP8(OPM) REP8(OPM)
        } else if constexpr (MODE == 4) {
#define OPC(v) asm volatile("v_rcp_f32 %0, %0" : "+v"(v));
            REP8(OPC) REP8(OPC)
        } else if constexpr (MODE == 5) {
#define OPK(v) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(v));
            REP8(OPK) REP8(OPK)
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(t1 - t0);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}

template <int MODE> void run(const char* name, int waves_per_simd) {
    float* d; hipMalloc(&d, 1 << 24);
    const int iters = 2000;
    // one block of 256 threads = one wave per SIMD of a CU; more blocks per CU -> more waves per SIMD
    hipLaunchKernelGGL(k<MODE>, dim3(256 * waves_per_simd), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * waves_per_simd), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float cyc; hipMemcpy(&cyc, d, 4, hipMemcpyDeviceToHost);
    const double n = 16.0 * iters;
    printf("%-14s waves/SIMD %d: %.2f cycles per instruction per wave (s_memtime), %.2f SIMD-cycles per instruction at 2.4 GHz wall\n", name, waves_per_simd,
           cyc / n, ms * 1e-3 * 2.4e9 / (n * waves_per_simd));
    hipFree(d);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<3>("v_mul_f32", w); run<0>("v_exp_f32", w); run<4>("v_rcp_f32", w); run<1>("v_exp_f16", w); run<2>("v_rcp_f16", w); run<5>("v_cvt_f16_f32", w);
    }
    return 0;
}
