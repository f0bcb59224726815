#include <hip/hip_runtime.h>
#include <cstdio>
// cost of s_barrier and of an LDS mailbox hand-over between the waves of one workgroup (one wave per SIMD)
template <int MODE>
__global__ void k(long long *o, int iters, int *sink) {
    __shared__ int mb[64];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < 64) mb[threadIdx.x] = 0;
    __syncthreads();
    int acc = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) { __builtin_amdgcn_s_barrier(); }
        if (MODE == 1) { if (lane == 0) mb[w] = i; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); acc += mb[(w + 1) & 3]; }
        if (MODE == 2) { if (lane == 0) mb[w] = i; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); acc += mb[(w + 1) & 3];
                         __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
        if (MODE == 3) { // unbalanced: wave 0 does 40 dependent VALU ops, the others wait at the barrier
            if (w == 0) { float g = acc; for (int q = 0; q < 40; q++) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(g)); acc += (int)g; }
            __builtin_amdgcn_s_barrier(); }
        if (MODE == 4) { // all four waves issue 40 dependent VALU each: do the SIMDs run concurrently?
            float g = acc; for (int q = 0; q < 40; q++) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(g)); acc += (int)g; }
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) o[0] = t1 - t0;
    sink[threadIdx.x] = acc;
}
int main() {
    long long *lo; int *sink; hipMalloc(&lo, 64); hipMalloc(&sink, 4096);
    const int it = 2000; long long h;
    const char *names[] = {"bare s_barrier", "mailbox write + barrier + read", "the same with a second barrier", "40 dependent v_fma on wave 0 + barrier", "40 dependent v_fma on every wave, no barrier"};
    for (int nw = 2; nw <= 4; nw++) {
        for (int m = 0; m < 5; m++) {
            for (int rep = 0; rep < 2; rep++) {
                if (m == 0) k<0><<<1, 64 * nw>>>(lo, it, sink); if (m == 1) k<1><<<1, 64 * nw>>>(lo, it, sink); if (m == 2) k<2><<<1, 64 * nw>>>(lo, it, sink);
                if (m == 3) k<3><<<1, 64 * nw>>>(lo, it, sink); if (m == 4) k<4><<<1, 64 * nw>>>(lo, it, sink);
                hipDeviceSynchronize();
            }
            hipMemcpy(&h, lo, 8, hipMemcpyDeviceToHost);
            printf("%d waves: %-48s %.1f cycles per iteration\n", nw, names[m], (double)h / it);
        }
    }
    return 0;
}
