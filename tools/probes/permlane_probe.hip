// Prints what v_permlane16_swap_b32 and the masked DPP row_shr / row_ror adds do on this GPU (lane -> value).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int* out)
{
    const int lane = threadIdx.x;
    unsigned a = 100 + lane, b = 200 + lane;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[lane] = r[0];
    out[64 + lane] = r[1];
    float v = (float)lane;
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0x5 bank_mask:0x2" : "+v"(v));
    out[128 + lane] = (int)v;
    float t = (float)(1000 + lane), u = (float)lane;
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %0 row_ror:4 row_mask:0xa bank_mask:0x1" : "+v"(u) : "v"(t));
    out[192 + lane] = (int)u;
}
int main()
{
    int* d; hipMalloc(&d, 256 * 4);
    probe<<<1, 64>>>(d);
    int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"swap.vdst", "swap.src ", "shr4 m5b2", "ror4 mab1"};
    for (int k = 0; k < 4; k++) { printf("%s:", names[k]); for (int i = 0; i < 64; i++) printf(" %d", h[k * 64 + i]); printf("\n"); }
    return 0;
}
