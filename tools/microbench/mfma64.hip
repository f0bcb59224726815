#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double *av, const double *bv, double *D, long long *cyc, int iters) {
    const int l = threadIdx.x;
    d4 c = {0, 0, 0, 0};
    const double a = av[l], b = bv[l];
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[l * 4 + r] = c[r];
    d4 e = c;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) { e = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, e, 0, 0, 0); asm volatile("" : "+v"(e)); }
    long long t1 = __builtin_readcyclecounter();
    if (l == 0) cyc[0] = t1 - t0;
    D[256 + l] = e[0] + e[1] + e[2] + e[3];
}
int main() {
    // one-hot probing: A lane la = 1 (others 0), B lane lb = 1 -> which (lane, reg) of D becomes 1?
    double *dA, *dB, *dD; long long *dc; hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 4096); hipMalloc(&dc, 8);
    double hA[64], hB[64], hD[512];
    int a_row[64], a_k[64], b_col[64], b_k[64];
    // find for every A lane and B lane the output position they feed when paired with "all ones" on the other side
    for (int la = 0; la < 64; la++) {
        for (int i = 0; i < 64; i++) { hA[i] = i == la ? 1.0 : 0.0; hB[i] = 1.0 + i; }   // B distinct: D[i][j] = B[k(la)][j] for i = row(la)
        hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
        k<<<1, 64>>>(dA, dB, dD, dc, 1); hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
        if (la < 4 || la == 16 || la == 17 || la == 63) {
            printf("A lane %2d -> nonzero D (lane,reg,value): ", la);
            int cnt = 0; for (int p = 0; p < 256 && cnt < 6; p++) if (hD[p] != 0) { printf("(%d,%d,%.0f) ", p / 4, p % 4, hD[p]); cnt++; }
            printf("\n");
        }
    }
    for (int i = 0; i < 64; i++) { hA[i] = 1.0; hB[i] = 1.0; }
    hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dD, dc, 1000); k<<<1, 64>>>(dA, dB, dD, dc, 1000); long long hc; hipMemcpy(&hc, dc, 8, hipMemcpyDeviceToHost);
    printf("1000 dependent f64 16x16x4 MFMAs: %.1f cycles each\n", hc / 1000.0);
    return 0;
}
