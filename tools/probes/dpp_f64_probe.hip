// Probe of the fp64 row-broadcast DPP forms on gfx950 (tools/probes, not part of the product): correctness of v_fmac_f64_dpp row_newbcast:k through inline
// assembly, and issue rate / dependent latency of   (a) plain v_fmac_f64   (b) v_mov_b64_dpp + v_fmac_f64 (what the compiler emits for
// __builtin_amdgcn_update_dpp on a 64-bit value)   (c) the fused v_fmac_f64_dpp.   The lateral solve kernel (k_solve_lat) builds its 5 x 5 stage products on them.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/dpp_f64_probe.hip -o /tmp/dpp_probe && /tmp/dpp_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>

template <int K> __device__ __forceinline__ double bc(double v) {
    long long r = __builtin_amdgcn_update_dpp((long long)0, __double_as_longlong(v), 0x150 + K, 0xF, 0xF, false);
    return __longlong_as_double(r);
}
#define FMAC_DPP(acc, bsrc, x, K) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bsrc), "v"(x))

__global__ void k_check(const double* in, double* out) {
    const int l = threadIdx.x;
    double p = in[l], x = in[64 + l];
    double a0 = 0.0, a1 = 0.0;
    asm volatile("s_nop 4");
    FMAC_DPP(a0, p, x, 3);           // a0 += p[lane 3 of my row] * x
    FMAC_DPP(a0, p, x, 11);
    a1 = bc<3>(p) * x + bc<11>(p) * x;
    out[l] = a0; out[64 + l] = a1;
}
template <int MODE> __global__ void k_rate(double* out, long long* cyc, int n) {
    const int l = threadIdx.x;
    double p0 = 1.0 + 1e-3 * l, p1 = 0.5 + 1e-3 * l, x = 1e-3;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    long long t0 = clock64();
    for (int i = 0; i < n; i++) {
        if (MODE == 0) {           // plain, 8 independent accumulators
            a0 = fma(p0, x, a0); a1 = fma(p1, x, a1); a2 = fma(p0, x, a2); a3 = fma(p1, x, a3); a4 = fma(p0, x, a4); a5 = fma(p1, x, a5); a6 = fma(p0, x, a6); a7 = fma(p1, x, a7);
        } else if (MODE == 1) {    // mov_dpp + fma, 8 independent
            a0 = fma(bc<0>(p0), x, a0); a1 = fma(bc<1>(p1), x, a1); a2 = fma(bc<2>(p0), x, a2); a3 = fma(bc<3>(p1), x, a3);
            a4 = fma(bc<4>(p0), x, a4); a5 = fma(bc<5>(p1), x, a5); a6 = fma(bc<6>(p0), x, a6); a7 = fma(bc<7>(p1), x, a7);
        } else if (MODE == 2) {    // fused, 8 independent
            FMAC_DPP(a0, p0, x, 0); FMAC_DPP(a1, p1, x, 1); FMAC_DPP(a2, p0, x, 2); FMAC_DPP(a3, p1, x, 3);
            FMAC_DPP(a4, p0, x, 4); FMAC_DPP(a5, p1, x, 5); FMAC_DPP(a6, p0, x, 6); FMAC_DPP(a7, p1, x, 7);
        } else if (MODE == 3) {    // plain, ONE dependent chain of 8
            a0 = fma(p0, x, a0); a0 = fma(p1, x, a0); a0 = fma(p0, x, a0); a0 = fma(p1, x, a0); a0 = fma(p0, x, a0); a0 = fma(p1, x, a0); a0 = fma(p0, x, a0); a0 = fma(p1, x, a0);
        } else if (MODE == 4) {    // fused, ONE dependent accumulate chain of 8
            FMAC_DPP(a0, p0, x, 0); FMAC_DPP(a0, p1, x, 1); FMAC_DPP(a0, p0, x, 2); FMAC_DPP(a0, p1, x, 3);
            FMAC_DPP(a0, p0, x, 4); FMAC_DPP(a0, p1, x, 5); FMAC_DPP(a0, p0, x, 6); FMAC_DPP(a0, p1, x, 7);
        } else if (MODE == 5) {    // fused, the broadcast SOURCE depends on the previous result (the Riccati chain: P_k -> M -> P_{k-1}); s_nop for the DPP read hazard
            FMAC_DPP(a0, p0, x, 0); asm volatile("s_nop 1"); FMAC_DPP(p0, a0, x, 1); asm volatile("s_nop 1"); FMAC_DPP(a0, p0, x, 2); asm volatile("s_nop 1"); FMAC_DPP(p0, a0, x, 3); asm volatile("s_nop 1");
            FMAC_DPP(a0, p0, x, 4); asm volatile("s_nop 1"); FMAC_DPP(p0, a0, x, 5); asm volatile("s_nop 1"); FMAC_DPP(a0, p0, x, 6); asm volatile("s_nop 1"); FMAC_DPP(p0, a0, x, 7); asm volatile("s_nop 1");
        } else if (MODE == 6) {    // mov_dpp + fma, source depends on the previous result
            a0 = fma(bc<0>(p0), x, a0); p0 = fma(bc<1>(a0), x, p0); a0 = fma(bc<2>(p0), x, a0); p0 = fma(bc<3>(a0), x, p0);
            a0 = fma(bc<4>(p0), x, a0); p0 = fma(bc<5>(a0), x, p0); a0 = fma(bc<6>(p0), x, a0); p0 = fma(bc<7>(a0), x, p0);
        } else if (MODE == 7) {    // v_rcp_f64 + 2 Newton steps, dependent (the S^-1 of a stage)
            double r = __builtin_amdgcn_rcp(p0 + a0); r = r * (2.0 - (p0 + a0) * r); r = r * (2.0 - (p0 + a0) * r); a0 = a0 * 0.5 + r * 1e-9;
        }
    }
    long long t1 = clock64();
    out[l] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0;
    if (l == 0) cyc[0] = t1 - t0;
}
int main() {
    double h[128], o[128]; double *din, *dout; long long* dc;
    for (int i = 0; i < 128; i++) h[i] = sin(1.0 + 0.37 * i);
    hipMalloc(&din, 1024); hipMalloc(&dout, 1024); hipMalloc(&dc, 8);
    hipMemcpy(din, h, 1024, hipMemcpyHostToDevice);
    k_check<<<1, 64>>>(din, dout);
    hipMemcpy(o, dout, 1024, hipMemcpyDeviceToHost);
    double worst = 0, wref = 0;
    for (int l = 0; l < 64; l++) {
        const int row = l & ~15; const double ref = h[row + 3] * h[64 + l] + h[row + 11] * h[64 + l];
        worst = fmax(worst, fabs(o[l] - ref)); wref = fmax(wref, fabs(o[64 + l] - ref));
    }
    printf("v_fmac_f64_dpp row_newbcast check: max err fused %.3e, mov_dpp+fma %.3e\n", worst, wref);
    const char* names[8] = {"plain fma x8 independent", "mov_b64_dpp+fma x8 independent", "fused fmac_dpp x8 independent", "plain fma chain of 8", "fused fmac_dpp accumulate chain of 8",
                            "fused, source-dependent chain (+s_nop 1)", "mov_dpp+fma source-dependent chain", "rcp + 2 Newton, dependent"};
    const int n = 20000;
    for (int m = 0; m < 8; m++) {
        for (int rep = 0; rep < 2; rep++) {
            switch (m) {
                case 0: k_rate<0><<<1, 64>>>(dout, dc, n); break; case 1: k_rate<1><<<1, 64>>>(dout, dc, n); break; case 2: k_rate<2><<<1, 64>>>(dout, dc, n); break;
                case 3: k_rate<3><<<1, 64>>>(dout, dc, n); break; case 4: k_rate<4><<<1, 64>>>(dout, dc, n); break; case 5: k_rate<5><<<1, 64>>>(dout, dc, n); break;
                case 6: k_rate<6><<<1, 64>>>(dout, dc, n); break; case 7: k_rate<7><<<1, 64>>>(dout, dc, n); break;
            }
            hipDeviceSynchronize();
        }
        long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        printf("%-44s %.2f clock64 ticks per %s\n", names[m], (double)c / n / (m == 7 ? 1 : 8), m == 7 ? "reciprocal" : "fma");
    }
    return 0;
}
