// Issue rates of the vector instructions the cull kernels are made of, measured on the device: each kernel runs a long
// chain of one instruction kind per wave, 8 waves per SIMD, and reports wave-instructions per SIMD-cycle-quad.
//   hipcc -O3 --offload-arch=gfx950 tools/bench_valu_rates.hip -o tools/bench_valu_rates && tools/bench_valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2v __attribute__((ext_vector_type(2)));
#define REP 64
template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, float s, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, ps = {s, s * 2};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
            if (KIND == 0) { a0 = a0 * s; a1 = a1 * s; a2 = a2 * s; a3 = a3 * s; a4 = a4 * s; a5 = a5 * s; a6 = a6 * s; a7 = a7 * s; }
            if (KIND == 1) { a0 = __builtin_fmaf(a0, s, a1); a1 = __builtin_fmaf(a1, s, a2); a2 = __builtin_fmaf(a2, s, a3); a3 = __builtin_fmaf(a3, s, a0);
                             a4 = __builtin_fmaf(a4, s, a5); a5 = __builtin_fmaf(a5, s, a6); a6 = __builtin_fmaf(a6, s, a7); a7 = __builtin_fmaf(a7, s, a4); }
            if (KIND == 2) { p0 = p0 * ps; p1 = p1 * ps; p2 = p2 * ps; p3 = p3 * ps; p0 = p0 * p1; p1 = p1 * p2; p2 = p2 * p3; p3 = p3 * p0; }
            if (KIND == 3) { p0 = __builtin_elementwise_fma(p0, ps, p1); p1 = __builtin_elementwise_fma(p1, ps, p2); p2 = __builtin_elementwise_fma(p2, ps, p3); p3 = __builtin_elementwise_fma(p3, ps, p0);
                             p0 = __builtin_elementwise_fma(p0, p1, p2); p1 = __builtin_elementwise_fma(p1, p2, p3); p2 = __builtin_elementwise_fma(p2, p3, p0); p3 = __builtin_elementwise_fma(p3, p0, p1); }
            if (KIND == 4) { a0 = __builtin_amdgcn_rsqf(a0); a1 = __builtin_amdgcn_rsqf(a1); a2 = __builtin_amdgcn_rsqf(a2); a3 = __builtin_amdgcn_rsqf(a3);
                             a4 = __builtin_amdgcn_rsqf(a4); a5 = __builtin_amdgcn_rsqf(a5); a6 = __builtin_amdgcn_rsqf(a6); a7 = __builtin_amdgcn_rsqf(a7); }
            if (KIND == 5) { a0 = a0 < s ? a1 : a0; a1 = a1 < s ? a2 : a1; a2 = a2 < s ? a3 : a2; a3 = a3 < s ? a0 : a3;   // v_cmp + v_cndmask
                             a4 = a4 < s ? a5 : a4; a5 = a5 < s ? a6 : a5; a6 = a6 < s ? a7 : a6; a7 = a7 < s ? a4 : a7; }
            if (KIND == 6) { p0 = p0 + ps; p1 = p1 + ps; p2 = p2 + ps; p3 = p3 + ps; p0 = p0 + p1; p1 = p1 + p2; p2 = p2 + p3; p3 = p3 + p0; }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int KIND>
void run(const char *name, int per_rep) {
    float *out; hipMalloc(&out, 4 * 256 * 2048);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200;
    k<KIND><<<2048, 256>>>(out, 1.0001f, 10);  // warm-up
    hipEventRecord(e0); k<KIND><<<2048, 256>>>(out, 1.0001f, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // 2048 workgroups x 4 waves; 1024 SIMDs -> 8 waves per SIMD
    const double wave_instr_per_simd = 8.0 * iters * REP * per_rep;
    printf("%-28s %8.3f ms  -> %.2f ns per wave-instruction per SIMD (4 cycles at 2.4 GHz = 1.67 ns)\n", name, ms, ms * 1e6 / wave_instr_per_simd);
    hipFree(out);
}
int main() {
    run<0>("v_mul_f32", 8); run<1>("v_fma_f32", 8); run<2>("v_pk_mul_f32", 8); run<3>("v_pk_fma_f32", 8); run<6>("v_pk_add_f32", 8);
    run<4>("v_rsq_f32", 8); run<5>("v_cmp + v_cndmask", 16);
    return 0;
}
