#include <hip/hip_runtime.h>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* o) {
    unsigned a = threadIdx.x, b = threadIdx.x + 100;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    auto q = __builtin_amdgcn_permlane16_swap(r[0], r[0], false, false);
    o[threadIdx.x] = q[0]; o[64 + threadIdx.x] = q[1]; o[128 + threadIdx.x] = r[1];
}
int main() {
    unsigned* d; unsigned h[192];
    hipMalloc(&d, sizeof(h)); k<<<1, 64>>>(d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int j = 0; j < 3; j++) { for (int i = 0; i < 64; i += 8) printf("%u ", h[64 * j + i]); printf("\n"); }
    return 0;
}
