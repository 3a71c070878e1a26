// Does gfx950's LDS serve MISALIGNED ds_read_b64 / ds_read_b32 / ds_read_u16 exactly (any byte address), and at what rate
// against the four ds_read_u8 they could replace in the DIS patch search's bilinear window (pis4_kernel: bytes c, c+1, c+4,
// c+5 of a row at an arbitrary byte offset)?  Prints the number of wrong lanes per form and the time per form.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(256) void probe(unsigned* bad, unsigned long long* sink, int iters, int mode)
{
    __shared__ unsigned char lds[4096 + 64];
    for (int i = threadIdx.x; i < 4096 + 64; i += 256) lds[i] = (unsigned char)((i * 37 + 11) & 0xff);
    __syncthreads();
    unsigned wrong = 0;
    unsigned long long acc = 0;
    for (int it = 0; it < iters; it++) {
        const unsigned a = (threadIdx.x * 13 + it * 7) & 4095;       // every alignment occurs
        const unsigned addr = (unsigned)(uintptr_t)(lds + a) & 0xffff;   // LDS byte address (generic -> low bits)
        unsigned long long v64 = 0; unsigned v32 = 0, v16 = 0, b0, b1, b4, b5;
        if (mode == 0) {
            asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v64) : "v"(addr) : "memory");
            b0 = v64 & 0xff; b1 = (v64 >> 8) & 0xff; b4 = (v64 >> 32) & 0xff; b5 = (v64 >> 40) & 0xff;
        } else if (mode == 1) {
            unsigned lo, hi;
            asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:4\n\ts_waitcnt lgkmcnt(0)" : "=v"(lo), "=v"(hi) : "v"(addr) : "memory");
            b0 = lo & 0xff; b1 = (lo >> 8) & 0xff; b4 = hi & 0xff; b5 = (hi >> 8) & 0xff; v32 = lo;
        } else if (mode == 2) {
            unsigned lo, hi;
            asm volatile("ds_read_u16 %0, %2\n\tds_read_u16 %1, %2 offset:4\n\ts_waitcnt lgkmcnt(0)" : "=v"(lo), "=v"(hi) : "v"(addr) : "memory");
            b0 = lo & 0xff; b1 = (lo >> 8) & 0xff; b4 = hi & 0xff; b5 = (hi >> 8) & 0xff; v16 = lo;
        } else {
            asm volatile("ds_read_u8 %0, %4\n\tds_read_u8 %1, %4 offset:1\n\tds_read_u8 %2, %4 offset:4\n\tds_read_u8 %3, %4 offset:5\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(b0), "=v"(b1), "=v"(b4), "=v"(b5) : "v"(addr) : "memory");
        }
        const unsigned e0 = ((a) * 37 + 11) & 0xff, e1 = ((a + 1) * 37 + 11) & 0xff, e4 = ((a + 4) * 37 + 11) & 0xff, e5 = ((a + 5) * 37 + 11) & 0xff;
        wrong += (b0 != e0) + (b1 != e1) + (b4 != e4) + (b5 != e5);
        acc += b0 + b1 + b4 + b5 + v32 + v16;
    }
    atomicAdd(bad + mode, wrong);
    if (acc == 0xdeadbeef) sink[0] = acc;
}
int main()
{
    unsigned* bad; unsigned long long* sink;
    hipMalloc(&bad, 16); hipMalloc(&sink, 8); hipMemset(bad, 0, 16);
    const char* names[4] = {"1 x ds_read_b64 (misaligned)", "2 x ds_read_b32 (misaligned)", "2 x ds_read_u16 (misaligned)", "4 x ds_read_u8"};
    for (int mode = 0; mode < 4; mode++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        probe<<<1024, 256>>>(bad, sink, 10, mode);
        hipMemset(bad + mode, 0, 4);
        hipEventRecord(e0);
        probe<<<1024, 256>>>(bad, sink, 20000, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned w; hipMemcpy(&w, bad + mode, 4, hipMemcpyDeviceToHost);
        printf("%-32s wrong bytes %u   %.3f ms\n", names[mode], w, ms);
    }
    return 0;
}
