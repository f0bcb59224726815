#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
__global__ void lat(long long *o, int iters, int *sink) {
    __shared__ int sm[256];
    sm[threadIdx.x] = (threadIdx.x + 1) & 63;
    __syncthreads();
    long long t[16];
    int p = threadIdx.x; float g = p; double f = p; int q = p; int s = iters; 
    t[0] = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) { REP64(p = sm[p];) }
    t[1] = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) { REP64(asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(g));) }
    t[2] = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) { REP64(asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(f));) }
    t[3] = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) { REP64(asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(q));) }
    t[4] = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) { REP64(asm volatile("v_readlane_b32 %0, %1, 3\n v_add_u32 %1, %0, %1" : "+s"(s), "+v"(q) : : "scc");) }
    t[5] = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) { REP64(asm volatile("s_add_u32 %0, %0, 1" : "+s"(s) : : "scc");) }
    t[6] = __builtin_readcyclecounter();
    // independent valu ops (4 chains)
    float g1 = g, g2 = g + 1, g3 = g + 2, g4 = g + 3;
    for (int i = 0; i < iters; i++) { REP64(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(g1), "+v"(g2), "+v"(g3), "+v"(g4));) }
    t[7] = __builtin_readcyclecounter();
    // v_cmp + cndmask chain (vcc dependency)
    for (int i = 0; i < iters; i++) { REP64(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(q) : "v"(p) : "vcc");) }
    t[8] = __builtin_readcyclecounter();
    // ballot-like: v_cmp -> s_ff1 -> v_readlane with sgpr lane select
    for (int i = 0; i < iters; i++) { REP8(asm volatile("v_cmp_eq_u32 vcc, %1, %2\n s_ff1_i32_b64 %0, vcc\n s_and_b32 %0, %0, 63\n s_nop 3\n v_readlane_b32 %0, %1, %0\n v_add_u32 %1, %0, %1" : "+s"(s), "+v"(q) : "v"(p) : "vcc", "scc");) }
    t[9] = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) { REP64(asm volatile("v_min_f64 %0, %0, %1" : "+v"(f) : "v"((double)g1));) }
    t[10] = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) { REP64(asm volatile("v_rsq_f64 %0, %0" : "+v"(f));) }
    t[11] = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) { REP64(asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(q) : "v"(p));) }
    t[12] = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { for (int k = 0; k < 12; k++) o[k] = t[k + 1] - t[k]; }
    sink[threadIdx.x] = p + (int)g + (int)f + q + s + (int)(g1 + g2 + g3 + g4);
}
int main() {
    long long *lo; int *sink; hipMalloc(&lo, 256); hipMalloc(&sink, 1024);
    const int it = 50;
    lat<<<1, 64>>>(lo, it, sink); lat<<<1, 64>>>(lo, it, sink); long long h[16]; hipMemcpy(h, lo, 96, hipMemcpyDeviceToHost);
    const char *names[] = {"ds_read dep", "v_fma_f32 dep", "v_fma_f64 dep", "v_mov_dpp dep", "readlane+v_add pair", "s_add dep", "4 indep v_fma_f32 (per group of 4)", "v_cmp+cndmask pair", "cmp/ff1/and/nop/readlane/add (6 instr)", "v_min_f64 dep", "v_rsq_f64 dep", "v_mul_hi_u32 dep"};
    const double per[] = {64, 64, 64, 64, 64, 64, 64, 64, 8, 64, 64, 64};
    for (int k = 0; k < 12; k++) printf("%-42s %.2f ticks\n", names[k], (double)h[k] / (it * per[k]));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a); lat<<<1, 64>>>(lo, 2000, sink); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
    hipMemcpy(h, lo, 96, hipMemcpyDeviceToHost); long long tot = 0; for (int k = 0; k < 12; k++) tot += h[k];
    printf("total ticks %lld in %.3f ms -> %.1f MHz tick rate\n", tot, ms, tot / ms / 1e3);
    return 0;
}
