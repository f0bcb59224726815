#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
// What a SMALL kernel in front of the tick costs on the eight XCDs of an MI355X (round 6, the neighbour-list kernels):
//   * a chain of empty / tiny kernels on one stream: time per launch (kernel boundary = L2 write-back / invalidate between XCDs);
//   * inside a kernel, 100 MHz wall clock: the first global load of data the PREVIOUS kernel wrote, a device-scope 64-bit atomic with and
//     without a returned value (distinct addresses), the same on ONE hot address from every workgroup, a second load of the same line.
__device__ __forceinline__ long long wall() { return (long long)__builtin_amdgcn_s_memrealtime(); }
__global__ void empty_k(int *p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ void write_k(float *d, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) d[i] = (float)i; }
__global__ void probe_k(const float *d, unsigned long long *cells, unsigned long long *hot, long long *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    long long t0 = wall();
    float v = d[(i * 97) % n];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t1 = wall();
    float v2 = d[((i * 97) % n) ^ 1];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t2 = wall();
    unsigned long long r = 0;
    if ((threadIdx.x & 31) == 0) r = atomicAdd(&cells[4 * (i >> 5)], 1ull);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t3 = wall();
    if ((threadIdx.x & 31) == 0) atomicMax(&cells[4 * (i >> 5) + 1], (unsigned long long)i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t4 = wall();
    if (threadIdx.x == 0) r += atomicAdd(hot, 1ull);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t5 = wall();
    if (threadIdx.x == 0) { long long *o = out + 8 * blockIdx.x; o[0] = t1 - t0; o[1] = t2 - t1; o[2] = t3 - t2; o[3] = t4 - t3; o[4] = t5 - t4; o[5] = (long long)(v + v2) + (long long)r; }
}
int main()
{
    const int nb = 128, nt = 256, n = nb * nt;
    float *d; unsigned long long *cells, *hot; long long *out;
    hipMalloc(&d, n * 4); hipMalloc(&cells, 32 * n / 32 * 4); hipMalloc(&hot, 64); hipMalloc(&out, 64 * nb);
    hipMemset(cells, 0, 32 * n / 32 * 4); hipMemset(hot, 0, 64);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto chain = [&](int grid, int block, int reps, const char *what) {
        for (int w = 0; w < 20; w++) empty_k<<<grid, block, 0, st>>>(nullptr);
        hipEventRecord(e0, st);
        for (int r = 0; r < reps; r++) empty_k<<<grid, block, 0, st>>>(nullptr);
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("chain of %d empty kernels, grid %4d x %4d lanes (%s): %.2f us per launch\n", reps, grid, block, what, 1e3 * ms / reps);
    };
    chain(1, 64, 200, "one wave");
    chain(128, 256, 200, "the build kernel's shape");
    chain(1024, 128, 200, "the query kernel's shape");
    chain(1024, 256, 200, "the throughput plan kernel's shape without its LDS");
    std::vector<long long> h(8 * nb);
    for (int rep = 0; rep < 3; rep++) {
        write_k<<<nb, nt, 0, st>>>(d, n);
        probe_k<<<nb, nt, 0, st>>>(d, cells, hot, out, n);
        hipStreamSynchronize(st);
        hipMemcpy(h.data(), out, 64 * nb, hipMemcpyDeviceToHost);
        const char *names[] = {"first load of the previous kernel's data", "second load, neighbouring word", "64-bit atomic add, value returned, own address",
                               "64-bit atomic max, no value, own address", "64-bit atomic add, value returned, ONE address for all workgroups"};
        for (int k = 0; k < 5; k++) {
            std::vector<long long> v;
            for (int b = 0; b < nb; b++) v.push_back(h[8 * b + k]);
            std::sort(v.begin(), v.end());
            printf("rep %d  %-68s median %.2f us  max %.2f us\n", rep, names[k], v[nb / 2] / 100.0, v[nb - 1] / 100.0);
        }
    }
    return 0;
}
