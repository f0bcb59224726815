#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(long long *o, int *sink) {
    __shared__ int sm[64];
    sm[threadIdx.x] = threadIdx.x;
    __syncthreads();
    long long acc[4] = {0, 0, 0, 0};
    long long tk = __builtin_readcyclecounter();
    int v = threadIdx.x;
    for (int i = 0; i < 1000; i++) {
        long long t = __builtin_readcyclecounter(); acc[0] += t - tk; tk = t;      // section 0: loop overhead only
        t = __builtin_readcyclecounter(); acc[1] += t - tk; tk = t;                // section 1: empty
        v = sm[v & 63] + 1;
        t = __builtin_readcyclecounter(); acc[2] += t - tk; tk = t;                // section 2: one dependent LDS read
        float g = v;
        for (int q = 0; q < 20; q++) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(g));
        v += (int)g;
        t = __builtin_readcyclecounter(); acc[3] += t - tk; tk = t;                // section 3: 20 dependent fma + 2 cvt
    }
    if (threadIdx.x == 0) for (int k = 0; k < 4; k++) o[k] = acc[k];
    sink[threadIdx.x] = v;
}
int main() {
    long long *lo; int *sink; hipMalloc(&lo, 64); hipMalloc(&sink, 1024);
    k<<<1, 64>>>(lo, sink); k<<<1, 64>>>(lo, sink); long long h[4]; hipMemcpy(h, lo, 32, hipMemcpyDeviceToHost);
    printf("per iteration: loop-only section %.1f, empty section %.1f, one LDS read %.1f, 20 dependent fma %.1f cycles\n", h[0] / 1000., h[1] / 1000., h[2] / 1000., h[3] / 1000.);
    return 0;
}
