// Which XCD does each workgroup of a 256-block grid run on, as HW_REG_XCC_ID reports it?  (stream3_kernel keys its
// per-XCD scratch on this register: DESIGN.md 4.9.)  hipcc --offload-arch=gfx950 -O2, run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
__global__ void k(int *out) {
    int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    if (threadIdx.x == 0) out[blockIdx.x] = xcc;
}
int main() {
    int *d, h[1024];
    hipMalloc(&d, sizeof h);
    for (int rep = 0; rep < 2; rep++) {
        k<<<256, 512>>>(d);
        hipMemcpy(h, d, 256 * sizeof(int), hipMemcpyDeviceToHost);
        int hist[16] = {};
        for (int i = 0; i < 256; i++) hist[h[i] & 15]++;
        printf("blocks per XCC_ID:");
        for (int i = 0; i < 16; i++) printf(" %d", hist[i]);
        printf("\nfirst 16 blocks:");
        for (int i = 0; i < 16; i++) printf(" %d", h[i]);
        printf("\n");
    }
    return 0;
}
