// Probe (tools/probes, not part of the product): what does one stage of k_solve_lat's roll-out cost, and why?  The stage is six dependent v_fmac_f64_dpp + one LDS store;
// variants isolate the operand delivery: (0) operands in registers (no memory), (1) LDS store removed, (2) operands re-loaded from L2 every stage with the wait right
// behind the load, (3) operands loaded a group of stages ahead, (4) as 3 with only 20 of 64 lanes loading.  Run with 1 wavefront and with 1024 (one per SIMD).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lat_rollout_probe.hip -o /tmp/lat_probe && /tmp/lat_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define DPP(acc, bsrc, x, lane) "v_fmac_f64_dpp %" #acc ", %" #bsrc ", %" #x " row_newbcast:" #lane " row_mask:0xf bank_mask:0xf\n\t"

__device__ __forceinline__ double ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }

template <int MODE> __global__ __launch_bounds__(64, 1) void k(const double* __restrict__ rows, double* out, long long* cyc, int N, int reps) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x, c = lane & 15;
    const double* rp = rows + ((size_t)blockIdx.x * 4 + (lane >> 4)) * N * 56 + 8 * (c < 4 ? c : 0);
    double R[7];
    for (int i = 0; i < 7; i++) R[i] = ld(rp + i);
    double xr = 1e-3 * lane;
    const bool loads = MODE < 4 || c < 4 || c == 5;
    long long t0 = clock64();
    for (int r = 0; r < reps; r++) {
        if (MODE <= 1) {
#pragma unroll 1
            for (int k = 0; k < N; k++) {
                double acc = R[6], xn;
                asm volatile("s_nop 4\n\t" DPP(1, 2, 3, 0) DPP(1, 2, 4, 1) DPP(1, 2, 5, 2) DPP(1, 2, 6, 3) DPP(1, 2, 7, 4) "s_nop 1\n\tv_mov_b64 %0, %1\n\t" DPP(0, 1, 8, 5)
                             : "=&v"(xn), "+v"(acc) : "v"(xr), "v"(R[0]), "v"(R[1]), "v"(R[2]), "v"(R[3]), "v"(R[4]), "v"(R[5]));
                xr = xn * 1e-3;
                if (MODE == 0) lds[64 * (k & 7) + lane] = xn;
            }
        } else if (MODE == 2) {
#pragma unroll 1
            for (int k = 0; k < N; k++) {
                double Q[7];
                for (int i = 0; i < 7; i++) Q[i] = ld(rp + (size_t)56 * k + i);
                double acc = Q[6], xn;
                asm volatile("s_nop 4\n\t" DPP(1, 2, 3, 0) DPP(1, 2, 4, 1) DPP(1, 2, 5, 2) DPP(1, 2, 6, 3) DPP(1, 2, 7, 4) "s_nop 1\n\tv_mov_b64 %0, %1\n\t" DPP(0, 1, 8, 5)
                             : "=&v"(xn), "+v"(acc) : "v"(xr), "v"(Q[0]), "v"(Q[1]), "v"(Q[2]), "v"(Q[3]), "v"(Q[4]), "v"(Q[5]));
                xr = xn * 1e-3;
                lds[64 * (k & 7) + lane] = xn;
            }
        } else {
            constexpr int D = MODE == 5 ? 10 : 5;
            double A_[D][7], B_[D][7];
            for (int u = 0; u < D; u++) for (int i = 0; i < 7; i++) { A_[u][i] = 0.0; B_[u][i] = 0.0; }
            auto request = [&](int k0, double (*o)[7]) {
                if (loads) {
#pragma unroll
                    for (int u = 0; u < D; u++) { const int kk = k0 + u < N ? k0 + u : N - 1;
#pragma unroll
                        for (int i = 0; i < 7; i++) o[u][i] = ld(rp + (size_t)56 * kk + i); }
                } };
            auto touch = [&](double (*o)[7]) {
#pragma unroll
                for (int u = 0; u < D; u++) asm volatile("" : "+v"(o[u][0]), "+v"(o[u][1]), "+v"(o[u][2]), "+v"(o[u][3]), "+v"(o[u][4]), "+v"(o[u][5]), "+v"(o[u][6])); };
            auto compute = [&](int k0, double (*o)[7]) {
#pragma unroll
                for (int u = 0; u < D; u++) {
                    double acc = o[u][6], xn;
                    asm volatile("s_nop 4\n\t" DPP(1, 2, 3, 0) DPP(1, 2, 4, 1) DPP(1, 2, 5, 2) DPP(1, 2, 6, 3) DPP(1, 2, 7, 4) "s_nop 1\n\tv_mov_b64 %0, %1\n\t" DPP(0, 1, 8, 5)
                                 : "=&v"(xn), "+v"(acc) : "v"(xr), "v"(o[u][0]), "v"(o[u][1]), "v"(o[u][2]), "v"(o[u][3]), "v"(o[u][4]), "v"(o[u][5]));
                    xr = xn * 1e-3;
                    lds[64 * ((k0 + u) & 7) + lane] = xn;
                } };
            request(0, A_);
#pragma unroll 1
            for (int k0 = 0; k0 < N; k0 += 2 * D) { touch(A_); request(k0 + D, B_); compute(k0, A_); touch(B_); request(k0 + 2 * D, A_); compute(k0 + D, B_); }
            touch(A_);
        }
    }
    long long t1 = clock64();
    out[(size_t)blockIdx.x * 64 + lane] = xr + lds[lane];
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    const int N = 50, reps = 40, W = 1024;
    double *rows, *out; long long* cyc;
    hipMalloc(&rows, (size_t)W * 4 * N * 56 * 8); hipMalloc(&out, (size_t)W * 64 * 8); hipMalloc(&cyc, W * 8);
    std::vector<double> h((size_t)W * 4 * N * 56);
    for (size_t i = 0; i < h.size(); i++) h[i] = 1e-3 * ((i * 2654435761u) % 1000) - 0.5;
    hipMemcpy(rows, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    const char* names[6] = {"operands in registers, LDS store", "operands in registers, no store", "7 loads per stage, used at once", "groups of 5 stages requested ahead (all lanes)",
                            "groups of 5 ahead, 20 of 64 lanes load", "groups of 10 ahead, 20 of 64 lanes load"};
    for (int waves : {1, 256, 512, 1024}) {
        for (int m = 0; m < 6; m++) {
            for (int rep = 0; rep < 2; rep++) {
                switch (m) {
                    case 0: k<0><<<waves, 64, 4096>>>(rows, out, cyc, N, reps); break; case 1: k<1><<<waves, 64, 4096>>>(rows, out, cyc, N, reps); break;
                    case 2: k<2><<<waves, 64, 4096>>>(rows, out, cyc, N, reps); break; case 3: k<3><<<waves, 64, 4096>>>(rows, out, cyc, N, reps); break;
                    case 4: k<4><<<waves, 64, 4096>>>(rows, out, cyc, N, reps); break; case 5: k<5><<<waves, 64, 4096>>>(rows, out, cyc, N, reps); break;
                }
                hipDeviceSynchronize();
            }
            std::vector<long long> c(waves); hipMemcpy(c.data(), cyc, waves * 8, hipMemcpyDeviceToHost);
            double s = 0, mx = 0; for (auto v : c) { s += v; if (v > mx) mx = v; }
            printf("%5d wavefronts  %-48s %8.1f ticks per stage (mean), %8.1f (slowest wavefront)\n", waves, names[m], s / waves / reps / N, mx / reps / N);
        }
    }
    return 0;
}
