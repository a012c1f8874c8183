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
    f32x2 p0 = {a0, a1}, p1 = {a1, a2}, p2 = {a2, a3}, p3 = {a3, a4}, p4 = {a4, a5}, p5 = {a5, a6}, p6 = {a6, a7}, p7 = {a7, a0};
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
        } else if constexpr (MODE == 6) {
#define OPP(v) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(v));
            OPP(p0) OPP(p1) OPP(p2) OPP(p3) OPP(p4) OPP(p5) OPP(p6) OPP(p7) OPP(p0) OPP(p1) OPP(p2) OPP(p3) OPP(p4) OPP(p5) OPP(p6) OPP(p7)
        } else if constexpr (MODE == 7) {
#define OPQ(v) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(v));
            OPQ(p0) OPQ(p1) OPQ(p2) OPQ(p3) OPQ(p4) OPQ(p5) OPQ(p6) OPQ(p7) OPQ(p0) OPQ(p1) OPQ(p2) OPQ(p3) OPQ(p4) OPQ(p5) OPQ(p6) OPQ(p7)
        } else if constexpr (MODE == 8) {
#define OPA(v) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(v));
            OPA(p0) OPA(p1) OPA(p2) OPA(p3) OPA(p4) OPA(p5) OPA(p6) OPA(p7) OPA(p0) OPA(p1) OPA(p2) OPA(p3) OPA(p4) OPA(p5) OPA(p6) OPA(p7)
        } else if constexpr (MODE == 9) {
#define OPF(v) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v));
            REP8(OPF) REP8(OPF)
        } else if constexpr (MODE == 10) {
#define OPB(v) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(v));
            REP8(OPB) REP8(OPB)
        } else if constexpr (MODE == 11) {
#define OPN(v) asm volatile("v_and_b32 %0, %0, %0" : "+v"(v));
            REP8(OPN) REP8(OPN)
        } else if constexpr (MODE == 12) {                 // the SiLU pair as the kernels issue it: 7 instructions per 2 values
#define OPS(v, x, y) asm volatile("v_pk_mul_f32 %0, %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_pk_add_f32 %0, %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_pk_mul_f32 %0, %0, %0" : "+v"(v), "+v"(x), "+v"(y));
            OPS(p0, a0, a1) OPS(p1, a2, a3) OPS(p2, a4, a5) OPS(p3, a6, a7) OPS(p4, a0, a1) OPS(p5, a2, a3) OPS(p6, a4, a5) OPS(p7, a6, a7)
        } else if constexpr (MODE == 13) {                 // the same with scalar multiplies / adds: 10 instructions per 2 values
#define OPT(v, w) asm volatile("v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1" : "+v"(v), "+v"(w));
            OPT(a0, a1) OPT(a2, a3) OPT(a4, a5) OPT(a6, a7) OPT(a0, a1) OPT(a2, a3) OPT(a4, a5) OPT(a6, a7)
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(t1 - t0) + (p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7)[0] + (p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7)[1];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}

template <int MODE> void run(const char* name, int waves_per_simd, double per_iter = 16.0) {
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
    const double n = per_iter * iters;
    printf("%-26s waves/SIMD %d: %.2f cycles per instruction per wave (s_memtime), %.2f SIMD-cycles per instruction at 2.4 GHz wall\n", name, waves_per_simd,
           cyc / n, ms * 1e-3 * 2.4e9 / (n * waves_per_simd));
    hipFree(d);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<3>("v_mul_f32", w); run<0>("v_exp_f32", w); run<4>("v_rcp_f32", w); run<1>("v_exp_f16", w); run<2>("v_rcp_f16", w); run<5>("v_cvt_f16_f32", w);
        run<9>("v_fma_f32", w); run<6>("v_pk_mul_f32", w); run<7>("v_pk_fma_f32", w); run<8>("v_pk_add_f32", w); run<10>("v_cvt_pk_bf16", w); run<11>("v_and_b32", w);
        run<12>("silu2 packed (7 instr)", w, 8.0); run<13>("silu2 scalar (10 instr)", w, 8.0);
    }
    return 0;
}
