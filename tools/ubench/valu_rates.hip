// micro-benchmark: what one wave64 VALU instruction of the SOR point update costs on gfx950, alone and with 2 / 3 waves per SIMD.
// Built and run on the GPU box:  hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rates tools/ubench/valu_rates.hip && /tmp/valu_rates
// Prints cycles (s_memtime ticks) per instruction per wave for independent (throughput) and dependent (latency) streams.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP16(X) X X X X X X X X X X X X X X X X

template <int KIND, bool DEP>
__global__ void k(unsigned long long *out, int iters, float seed) {
    v2f a0 = {seed, seed + 1}, a1 = {seed + 2, seed + 3}, a2 = {seed + 4, seed + 5}, a3 = {seed + 6, seed + 7};
    v2f a4 = a0 + 1.f, a5 = a1 + 1.f, a6 = a2 + 1.f, a7 = a3 + 1.f;
    v2f m = {1.0000001f, 0.9999999f};
    float s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3, s4 = seed + 4, s5 = seed + 5, s6 = seed + 6, s7 = seed + 7;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {          // v_pk_mul_f32
            if (DEP) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a0) : "v"(m));) }
            else { REP16(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));) }
        } else if (KIND == 1) {   // v_pk_add_f32
            if (DEP) { REP16(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a0) : "v"(m));) }
            else { REP16(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));) }
        } else if (KIND == 2) {   // v_mul_f32
            if (DEP) { REP16(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s0) : "v"(m.x));) }
            else { REP16(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7) : "v"(m.x));) }
        } else if (KIND == 3) {   // v_add_f32
            if (DEP) { REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(s0) : "v"(m.x));) }
            else { REP16(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7) : "v"(m.x));) }
        } else if (KIND == 4) {   // DPP mov wave_shr:1
            if (DEP) { REP16(asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1" : "+v"(s0));) }
            else { REP16(asm volatile("v_mov_b32_dpp %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7) : "v"(m.x));) }
        } else if (KIND == 6) {   // DPP mov row_shr:1 (inside the 16-lane rows)
            if (DEP) { REP16(asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1" : "+v"(s0));) }
            else { REP16(asm volatile("v_mov_b32_dpp %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7) : "v"(m.x));) }
        } else if (KIND == 7) {   // ds_bpermute_b32 (lane - 1) + v_cndmask for lane 0: the LDS crossbar instead of the DPP shift
            const int addr = ((int)(threadIdx.x & 63) - 1) * 4;
            if (DEP) { REP16(asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(s0) : "v"(addr));) }
            else { REP16(asm volatile("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7) : "v"(addr));) }
        } else if (KIND == 8) {   // four packed multiplies + one ds_bpermute per group: does the crossbar run beside the VALU?
            const int addr = ((int)(threadIdx.x & 63) - 1) * 4;
            REP16(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n ds_bpermute_b32 %4, %9, %4\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n ds_bpermute_b32 %5, %9, %5\n v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n s_waitcnt lgkmcnt(1)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7) : "v"(m), "v"(addr));)
        } else if (KIND == 9) {   // the same with the DPP shift
            REP16(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_mov_b32_dpp %4, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_mov_b32_dpp %5, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7) : "v"(m));)
        } else if (KIND == 5) {   // v_pk_mul_f32 with op_sel broadcast (as the kernel uses)
            if (DEP) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(a0) : "v"(m));) }
            else { REP16(asm volatile("v_pk_mul_f32 %0, %0, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %1, %1, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %2, %2, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %3, %3, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %4, %4, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %5, %5, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %6, %6, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %7, %7, %8 op_sel_hi:[1,0]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));) }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (a0.x + a1.x + a2.x + a3.x + a4.x + a5.x + a6.x + a7.x + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7 == 12345.678f) out[0] = 0;
}

template <int KIND, bool DEP>
static double run(int waves, unsigned long long *d) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<KIND, DEP>), dim3(1), dim3(64 * waves), 0, 0, d, iters, 1.0f);
    hipLaunchKernelGGL((k<KIND, DEP>), dim3(1), dim3(64 * waves), 0, 0, d, iters, 1.0f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(waves);
    hipMemcpy(h.data(), d, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mx = 0;
    for (auto v : h) mx = v > mx ? (double)v : mx;
    const double ninstr = (double)iters * 16 * (DEP ? 1 : 8);
    return mx / ninstr;
}

int main() {
    unsigned long long *d;
    hipMalloc(&d, 4096);
    const char *names[10] = {"v_pk_mul_f32", "v_pk_add_f32", "v_mul_f32", "v_add_f32", "v_mov_dpp wave_shr", "v_pk_mul_f32 op_sel", "v_mov_dpp row_shr", "ds_bpermute_b32",
                             "6 pk_mul + 2 bpermute /8", "6 pk_mul + 2 wave_shr /8"};
    printf("cycles (s_memtime/readcyclecounter ticks) per instruction PER WAVE; waves = wavefronts in ONE workgroup on one CU (4 SIMDs)\n");
    printf("%-22s %8s | %8s %8s %8s %8s\n", "instruction", "latency", "1 wave", "4 waves", "8 waves", "12 waves");
#define ROW(K) printf("%-22s %8.2f | %8.2f %8.2f %8.2f %8.2f\n", names[K], run<K, true>(1, d), run<K, false>(1, d), run<K, false>(4, d), run<K, false>(8, d), run<K, false>(12, d));
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8) ROW(9)
    return 0;
}
