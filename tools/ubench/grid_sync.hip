// micro-benchmark: what one grid-wide barrier (cooperative groups) costs on gfx950 with 4 blocks of 512 threads per CU resident, with and without traffic that
// has to become visible across XCDs.  Built and run on the GPU box: hipcc --offload-arch=gfx950 -O2 -o /tmp/grid_sync tools/ubench/grid_sync.hip && /tmp/grid_sync
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;
__global__ void __launch_bounds__(512) k(float *buf, int n_sync, int work, unsigned *bad) {
    cg::grid_group grid = cg::this_grid();
    const int nb = gridDim.x, tid = threadIdx.x;
    for (int i = 0; i < n_sync; i++) {
        if (work) {   // every block writes a 2-KB piece, then (behind the barrier) reads its neighbour's: the value must be this round's
            buf[(size_t)blockIdx.x * 512 + tid] = (float)(i + 1);
        }
        __threadfence();
        grid.sync();
        if (work) {
            const float v = buf[(size_t)((blockIdx.x + 1) % nb) * 512 + tid];
            if (v != (float)(i + 1)) atomicAdd(bad, 1u);
            __threadfence();
            grid.sync();
        }
    }
}
int main() {
    int dev = 0, ncu = 0, per = 0, coop = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, k, 512, 0);
    printf("CUs %d, cooperative launch %d, blocks of 512 per CU %d\n", ncu, coop, per);
    float *buf; unsigned *bad;
    hipMalloc(&buf, (size_t)ncu * 8 * 512 * 4); hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
    for (int bpc = 1; bpc <= per && bpc <= 4; bpc *= 2)
        for (int work = 0; work < 2; work++) {
            int n = 200; dim3 grid(ncu * bpc), block(512);
            void *args[] = {&buf, &n, &work, &bad};
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipError_t r = hipLaunchCooperativeKernel((void *)k, grid, block, args, 0, 0);
            if (r != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(r)); return 1; }
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchCooperativeKernel((void *)k, grid, block, args, 0, 0);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned hb; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
            printf("%4d blocks, work %d: %.2f us per barrier (%d barriers), stale reads %u\n", ncu * bpc, work, ms * 1e3 / (n * (work ? 2 : 1)), n * (work ? 2 : 1), hb);
        }
    return 0;
}
