// tools/micro/xcc_id.hip — which XCD a workgroup runs on (HW_REG_XCC_ID, gfx942/gfx950) against its blockIdx: the dispatcher deals
// workgroups round-robin over the 8 XCDs.  hipcc --offload-arch=gfx950 -O2 -o xcc_id xcc_id.hip && ./xcc_id
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, size 4)
    const int id = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    if (threadIdx.x == 0) out[blockIdx.x] = id;
}
int main() {
    const int n = 2048;
    int* d; hipMalloc(&d, n * sizeof(int));
    hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, 0, d);
    int h[n]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int match = 0, hist[16] = {0};
    for (int b = 0; b < n; ++b) { match += (h[b] == b % 8); hist[h[b] & 15]++; }
    printf("blocks %d: xcc_id == blockIdx %% 8 for %d of them; first 24:", n, match);
    for (int b = 0; b < 24; ++b) printf(" %d", h[b]);
    printf("\nper XCD:"); for (int x = 0; x < 8; ++x) printf(" %d", hist[x]); printf("\n");
    return 0;
}
