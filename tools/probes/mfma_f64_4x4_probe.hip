// Probe of v_mfma_f64_4x4x4_4b_f64 on gfx950: layout of the four 4x4x4 blocks and latency (tools/probes, not part of the product).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
__global__ void k_layout(const double* a, const double* b, double* d) {
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
__global__ void k_chain(double* out, long long* cyc, int n) {
    const int l = threadIdx.x;
    double a = 1.0 + 1e-3 * l, b = 1.0 - 1e-3 * l, acc = 0.0;
    long long t0 = clock64();
#pragma unroll 8
    for (int i = 0; i < n; i++) { acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 0); a = acc * 1e-30 + a; }
    long long t1 = clock64();
    out[l] = acc; if (l == 0) cyc[0] = t1 - t0;
}
__global__ void k_acc(double* out, long long* cyc, int n) {
    const int l = threadIdx.x;
    double a = 1.0 + 1e-3 * l, b = 1.0 - 1e-3 * l, acc = 0.0;
    long long t0 = clock64();
#pragma unroll 8
    for (int i = 0; i < n; i++) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 0);
    long long t1 = clock64();
    out[l] = acc; if (l == 0) cyc[0] = t1 - t0;
}
__global__ void k_indep(double* out, long long* cyc, int n) {
    const int l = threadIdx.x;
    double a = 1.0 + 1e-3 * l, b = 1.0 - 1e-3 * l, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    long long t0 = clock64();
#pragma unroll 2
    for (int i = 0; i < n; i++) {
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
    }
    long long t1 = clock64();
    out[l] = c0 + c1 + c2 + c3; if (l == 0) cyc[0] = t1 - t0;
}
int main() {
    double ha[64], hb[64], hd[64];
    double *da, *db, *dd; long long* dc; long long c;
    (void)hipMalloc(&da, 512); (void)hipMalloc(&db, 512); (void)hipMalloc(&dd, 512); (void)hipMalloc(&dc, 8);
    // find the layout with unit impulses: a = e_p, b = e_q -> which d lanes light up
    printf("impulse map (a lane p, b lane q -> d lanes):\n");
    for (int p = 0; p < 64; p += 1) {
        if (!(p < 20 || p == 32 || p == 48)) continue;
        for (int q = 0; q < 64; q++) {
            for (int i = 0; i < 64; i++) { ha[i] = i == p; hb[i] = i == q; }
            (void)hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
            k_layout<<<1, 64>>>(da, db, dd); (void)hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
            for (int i = 0; i < 64; i++) if (hd[i] != 0.0) printf("  a[%d] b[%d] -> d[%d]\n", p, q, i);
        }
    }
    const int n = 4096;
    k_chain<<<1, 64>>>(dd, dc, n); (void)hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost); printf("dependent (operand) chain: %.1f cycles per link\n", (double)c / n);
    k_acc<<<1, 64>>>(dd, dc, n); (void)hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost); printf("accumulator chain: %.1f cycles each\n", (double)c / n);
    k_indep<<<1, 64>>>(dd, dc, n); (void)hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost); printf("independent: %.1f cycles each\n", (double)c / (4.0 * n));
    return 0;
}
