// Probe of v_mfma_f64_16x16x4_f64 on gfx950: operand / result layout and the latency of a dependent chain (tools/probes, not part of the product).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f64_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef double double4_ __attribute__((ext_vector_type(4)));

__global__ void k_layout(const double* A /*16x4*/, const double* B /*4x16*/, double* D /*16x16*/) {
    const int l = threadIdx.x;
    double a = A[(l & 15) * 4 + (l >> 4)];       // A[m = l%16][k = l/16]
    double b = B[(l >> 4) * 16 + (l & 15)];      // B[k = l/16][n = l%16]
    double4_ acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];     // row = (lane>>4) + 4*reg, col = lane&15
}
__global__ void k_chain(double* out, long long* cyc, int n) {
    const int l = threadIdx.x;
    double a = 1.0 + 1e-3 * l, b = 1.0 - 1e-3 * l;
    double4_ acc = {0, 0, 0, 0};
    long long t0 = clock64();
    for (int i = 0; i < n; i++) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        a = acc[0] * 1e-30 + a;                 // true dependence of the next A operand on the result (like P -> next stage)
    }
    long long t1 = clock64();
    out[l] = acc[0] + acc[1] + acc[2] + acc[3];
    if (l == 0) cyc[0] = t1 - t0;
}
__global__ void k_indep(double* out, long long* cyc, int n) {
    const int l = threadIdx.x;
    double a = 1.0 + 1e-3 * l, b = 1.0 - 1e-3 * l;
    double4_ acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    long long t0 = clock64();
    for (int i = 0; i < n; i++) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc3, 0, 0, 0);
    }
    long long t1 = clock64();
    out[l] = acc0[0] + acc1[1] + acc2[2] + acc3[3];
    if (l == 0) cyc[0] = t1 - t0;
}
__global__ void k_fma_chain(double* out, long long* cyc, int n) {
    double x = 1.0 + 1e-3 * threadIdx.x, y = 0.999;
    long long t0 = clock64();
    for (int i = 0; i < n; i++) x = fma(x, y, 1e-9);
    long long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double hA[64], hB[64], hD[256], ref[256];
    for (int i = 0; i < 64; i++) { hA[i] = sin(1.0 + i); hB[i] = cos(2.0 + 3 * i); }
    for (int m = 0; m < 16; m++) for (int n = 0; n < 16; n++) { double s = 0; for (int k = 0; k < 4; k++) s += hA[m * 4 + k] * hB[k * 16 + n]; ref[m * 16 + n] = s; }
    double *dA, *dB, *dD; long long* dc;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048); hipMalloc(&dc, 8);
    hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
    k_layout<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, 2048, hipMemcpyDeviceToHost);
    double e = 0; for (int i = 0; i < 256; i++) e = fmax(e, fabs(hD[i] - ref[i]));
    printf("layout check: max |D - A B| = %.3e  (A[m=l%%16][k=l/16], B[k=l/16][n=l%%16], D[row=(l>>4)+4r][col=l&15])\n", e);
    long long c; const int n = 4096;
    k_chain<<<1, 64>>>(dD, dc, n); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost); printf("dependent MFMA f64 16x16x4 (+1 dependent FMA): %.1f cycles per link\n", (double)c / n);
    k_indep<<<1, 64>>>(dD, dc, n); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost); printf("independent MFMA f64 16x16x4: %.1f cycles each (issue rate)\n", (double)c / (4.0 * n));
    k_fma_chain<<<1, 64>>>(dD, dc, n); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost); printf("dependent v_fma_f64 chain: %.1f cycles per FMA\n", (double)c / n);
    return 0;
}
