// Probe (round 6): does SQ_INSTS_VALU count v_accvgpr_read / v_accvgpr_write?  Two kernels with the same loop of four dependent v_fma_f64 per trip; the second one also
// moves 16 values into accumulation registers and back per trip.  Run under `rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace` and compare the two counts:
//   hipcc --offload-arch=gfx950 -O3 tools/probes/accvgpr_count_probe.hip -o /tmp/acc_probe && rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace -d gpurun_out/r6/accprobe -- /tmp/acc_probe
// Finding (MI355X, ROCm 7.2): see EXPERIMENTS.md 12.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(64) void k_fma_only(double* out, int trips) {
    double a = 1.0 + threadIdx.x * 1e-9, b = 0.999999, c = 1e-7;
    for (int i = 0; i < trips; i++) {
        asm volatile("v_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    }
    out[blockIdx.x * 64 + threadIdx.x] = a;
}
__global__ __launch_bounds__(64) void k_fma_and_acc_moves(double* out, int trips) {
    double a = 1.0 + threadIdx.x * 1e-9, b = 0.999999, c = 1e-7;
    int x = threadIdx.x;
    for (int i = 0; i < trips; i++) {
        asm volatile("v_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
        asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %0\n\tv_accvgpr_write_b32 a2, %0\n\tv_accvgpr_write_b32 a3, %0\n\t"
                     "v_accvgpr_write_b32 a4, %0\n\tv_accvgpr_write_b32 a5, %0\n\tv_accvgpr_write_b32 a6, %0\n\tv_accvgpr_write_b32 a7, %0\n\t"
                     "s_nop 4\n\t"
                     "v_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %0, a1\n\tv_accvgpr_read_b32 %0, a2\n\tv_accvgpr_read_b32 %0, a3\n\t"
                     "v_accvgpr_read_b32 %0, a4\n\tv_accvgpr_read_b32 %0, a5\n\tv_accvgpr_read_b32 %0, a6\n\tv_accvgpr_read_b32 %0, a7"
                     : "+v"(x) : : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");
    }
    out[blockIdx.x * 64 + threadIdx.x] = a + x;
}
int main() {
    double* d; hipMalloc(&d, 1024 * 64 * 8);
    const int trips = 10000;
    for (int r = 0; r < 3; r++) {
        hipLaunchKernelGGL(k_fma_only, dim3(1024), dim3(64), 0, 0, d, trips);
        hipLaunchKernelGGL(k_fma_and_acc_moves, dim3(1024), dim3(64), 0, 0, d, trips);
    }
    hipDeviceSynchronize();
    printf("1024 wavefronts x %d trips: k_fma_only issues 4 VALU per trip (+ loop overhead), k_fma_and_acc_moves 4 + 16 accvgpr moves\n", trips);
    return 0;
}
