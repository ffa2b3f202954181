// where do the 10 waves of a 640-thread workgroup (160 KB LDS: one workgroup per CU) land?  prints simd_id per wave for a few workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(640) k(unsigned *out) {
    extern __shared__ unsigned char smem[];
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + threadIdx.x / 64] = hw;
    if (threadIdx.x == 0) smem[0] = 1;
    // stay resident a little so that placement is the steady-state one
    for (int i = 0; i < 2000; i++) __builtin_amdgcn_s_sleep(10);
}
int main() {
    unsigned *d; hipMalloc(&d, 512 * 16 * 4); hipMemset(d, 0, 512 * 16 * 4);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163264);
    hipLaunchKernelGGL(k, dim3(256), dim3(640), 163264, 0, d);
    hipDeviceSynchronize();
    std::vector<unsigned> h(256 * 16); hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int hist[5][11] = {{0}};
    for (int b = 0; b < 256; b++) {
        int cnt[4] = {0, 0, 0, 0};
        for (int w = 0; w < 10; w++) cnt[(h[b * 16 + w] >> 4) & 3]++;
        if (b < 6) { printf("wg %d: simd of waves 0..9:", b); for (int w = 0; w < 10; w++) printf(" %u", (h[b * 16 + w] >> 4) & 3); printf("   cu %u se %u raw %08x\n", (h[b*16] >> 8) & 15, (h[b*16] >> 13) & 7, h[b*16]); }
        int mx = 0; for (int i = 0; i < 4; i++) mx = cnt[i] > mx ? cnt[i] : mx;
        hist[0][mx]++;
    }
    printf("max waves of one workgroup on a SIMD -> number of workgroups:"); for (int i = 0; i <= 10; i++) if (hist[0][i]) printf("  %d:%d", i, hist[0][i]); printf("\n");
    return 0;
}
