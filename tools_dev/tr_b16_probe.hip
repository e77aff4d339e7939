// What does ds_read_b64_tr_b16 return?  LDS holds u16 values equal to their element index; every lane supplies an
// address and gets 4 values.  Two address patterns: (a) lane * 8 bytes (lane-linear 8-byte chunks), (b) a [16 k][32 col]
// bf16 image with 64-byte rows where lane l points at row (l >> 4) * 4 + ... -- printed so the mapping can be read off.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(unsigned* out, int pattern) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned addr;
    if (pattern == 0) addr = l * 8;                                     // element 4*l
    else if (pattern == 1) addr = ((l & 15) * 4 + (l >> 4) * 64) * 2;   // same thing written per group
    else addr = ((l >> 2 & 3) * 32 + (l & 3) * 4 + (l >> 4) * 128) * 2; // group g: rows 4g..4g+3 of a [16][32] image, lane t: row t/4, cols 4*(t%4)
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned short*)lds;
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr + base) : "memory");
    out[l * 4 + 0] = (unsigned)(v & 0xffff);
    out[l * 4 + 1] = (unsigned)((v >> 16) & 0xffff);
    out[l * 4 + 2] = (unsigned)((v >> 32) & 0xffff);
    out[l * 4 + 3] = (unsigned)((v >> 48) & 0xffff);
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4);
    unsigned h[256];
    for (int p = 0; p < 3; ++p) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, p);
        hipError_t e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        if (e != hipSuccess) printf("error %s\n", hipGetErrorString(e));
        printf("pattern %d\n", p);
        for (int l = 0; l < 64; ++l) printf("%2d:%4u %4u %4u %4u%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l % 4 == 3) ? "\n" : " | ");
    }
    return 0;
}
