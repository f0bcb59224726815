#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(int *o) {
    int v = threadIdx.x * 3 + 1;
    o[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);        // wave_shr:1
    o[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);   // wave_shl:1
    o[128 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x142, 0xf, 0xa, false);  // row_bcast15 rows 1,3
    o[192 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x143, 0xf, 0xc, false);  // row_bcast31 rows 2,3
}
__device__ __forceinline__ double fast_sqrt(double x) {
#pragma clang fp contract(off)
    double r = __builtin_amdgcn_rsq(x);
    double g = x * r, h = r * 0.5;
    double e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g); h = __builtin_fma(h, e, h);
    double d = __builtin_fma(-g, g, x); g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x); g = __builtin_fma(d, h, g);
    return x == 0.0 ? 0.0 : g;
}
__global__ void s(int n, int *bad, double *worst) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = (double)i;
    double a = sqrt(x), b = fast_sqrt(x);
    if (a != b) { atomicAdd(bad, 1); *worst = x; }
}
__global__ void lat(long long *o, int iters) {
    // dependent chain timing: ds_read, v_readlane, dpp, f64 ops
    __shared__ int sm[256];
    sm[threadIdx.x] = (threadIdx.x + 1) & 63;
    __syncthreads();
    int p = threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) p = sm[p];
    long long t1 = __builtin_readcyclecounter();
    double f = (double)p;
    for (int i = 0; i < iters; i++) f = __builtin_fma(f, 1.0000001, 0.5);
    long long t2 = __builtin_readcyclecounter();
    int q = p;
    for (int i = 0; i < iters; i++) q = __builtin_amdgcn_update_dpp(q, q + 1, 0x111, 0xf, 0xf, false);
    long long t3 = __builtin_readcyclecounter();
    int r = q;
    for (int i = 0; i < iters; i++) r = __builtin_amdgcn_readlane(r + threadIdx.x, (i * 7) & 63);
    long long t4 = __builtin_readcyclecounter();
    float g = (float)r;
    for (int i = 0; i < iters; i++) g = __builtin_fmaf(g, 1.0001f, 0.5f);
    long long t5 = __builtin_readcyclecounter();
    double s = f;
    for (int i = 0; i < iters; i++) s = sqrt(s + 2.0);
    long long t6 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { o[0] = t1 - t0; o[1] = t2 - t1; o[2] = t3 - t2; o[3] = t4 - t3; o[4] = t5 - t4; o[5] = t6 - t5; o[6] = (long long)(f + q + r + g + s); }
}
int main() {
    int *o; hipMalloc(&o, 256 * 4); k<<<1, 64>>>(o); std::vector<int> h(256); hipMemcpy(h.data(), o, 1024, hipMemcpyDeviceToHost);
    printf("wave_shr1: lane0 %d lane1 %d lane16 %d lane32 %d lane63 %d (expect -1, 1, 46, 94, 187)\n", h[0], h[1], h[16], h[32], h[63]);
    printf("wave_shl1: lane0 %d lane15 %d lane31 %d lane62 %d lane63 %d (expect 4, 49, 97, 190, -1)\n", h[64], h[64+15], h[64+31], h[64+62], h[64+63]);
    printf("row_bcast15 lane16 %d lane31 %d lane48 %d lane0 %d lane32 %d\n", h[128+16], h[128+31], h[128+48], h[128], h[128+32]);
    printf("row_bcast31 lane32 %d lane63 %d lane16 %d\n", h[192+32], h[192+63], h[192+16]);
    int *bad; double *worst; hipMalloc(&bad, 4); hipMalloc(&worst, 8); hipMemset(bad, 0, 4);
    int n = 1 << 22; s<<<(n + 255) / 256, 256>>>(n, bad, worst); int hb; double hw; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(&hw, worst, 8, hipMemcpyDeviceToHost);
    printf("fast_sqrt mismatches vs sqrt() over [0, 2^22): %d (last %.0f)\n", hb, hw);
    long long *lo; hipMalloc(&lo, 64); lat<<<1, 64>>>(lo, 1000); lat<<<1, 64>>>(lo, 1000); long long hl[8]; hipMemcpy(hl, lo, 56, hipMemcpyDeviceToHost);
    printf("cycles(100MHz ticks?) per dependent: ds_read %.2f  fma64 %.2f  dpp %.2f  readlane %.2f  fma32 %.2f  sqrt64 %.2f\n", hl[0]/1000., hl[1]/1000., hl[2]/1000., hl[3]/1000., hl[4]/1000., hl[5]/1000.);
    return 0;
}
