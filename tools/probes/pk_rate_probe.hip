// Issue rate of packed f32 VALU (v_pk_mul_f32 / v_pk_add_f32) against the scalar forms on this GPU, with the wave
// occupancy of level_kernel's SOR sweep (16 wavefronts per CU, one workgroup per CU).  Independent accumulator chains:
// throughput, not latency.  Prints ns per wave-instruction per SIMD-equivalent and the packed / scalar ratio.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP 64
template <int KIND> __global__ __launch_bounds__(1024) void probe(float* out, int iters, float seed)
{
    f2 a[8];
    for (int k = 0; k < 8; k++) a[k] = f2{seed + threadIdx.x + k, seed - k};
    f2 m = f2{1.0000001f, 0.9999999f};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (KIND == 0) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k].x) : "v"(m.x)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k].y) : "v"(m.y)); }
                if (KIND == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
                if (KIND == 2) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k].x) : "v"(m.x)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k].y) : "v"(m.y)); }
                if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
                if (KIND == 4) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(a[k]) : "v"(m));   // broadcast m.x
            }
        }
    }
    float s = 0;
    for (int k = 0; k < 8; k++) s += a[k].x + a[k].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND> double run(float* d, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<KIND><<<256, 1024>>>(d, 10, 1.f);
    hipEventRecord(e0);
    probe<KIND><<<256, 1024>>>(d, iters, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main()
{
    float* d; hipMalloc(&d, 256 * 1024 * 4);
    const int iters = 20000;
    const char* names[5] = {"2 x v_mul_f32", "v_pk_mul_f32", "2 x v_add_f32", "v_pk_add_f32", "v_pk_mul_f32 op_sel_hi:[1,0]"};
    double ms[5] = {run<0>(d, iters), run<1>(d, iters), run<2>(d, iters), run<3>(d, iters), run<4>(d, iters)};
    for (int k = 0; k < 5; k++)
        printf("%-30s %8.3f ms for %d x %d element-pair ops per lane  (%.2f cycles @2.4GHz per pair-op per wave on a SIMD with 4 waves)\n", names[k], ms[k], iters, REP,
               ms[k] * 1e-3 * 2.4e9 / ((double)iters * REP * 4));
    float h[4]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("check %g\n", h[0]);
    return 0;
}
