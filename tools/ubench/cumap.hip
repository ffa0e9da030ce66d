// Probe: which CU does workgroup b of a 512-workgroup launch with the footprint of ntt16_fwd_kernel (1024 threads, 68 KiB LDS:
// two workgroups per CU) land on?  Prints blockIdx -> (XCC, SE, SH/SA, CU) and the pairs that share a CU.
// hipcc --offload-arch=gfx950 -O3 -o cumap cumap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void __launch_bounds__(1024) probe(unsigned* out, int spin) {
    extern __shared__ unsigned lds[];
    if (threadIdx.x == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID
        unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));     // HW_REG_XCC_ID, bits 3:0
        out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
    }
    lds[threadIdx.x] = threadIdx.x;
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);      // keep every workgroup resident until all are placed
    __syncthreads();
    if (lds[(threadIdx.x + 1) & 1023] == 12345678u) out[0] = 0;
}
int main() {
    const int W = 512;
    unsigned* d; hipMalloc(&d, 2 * W * sizeof(unsigned));
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 69632);
    hipLaunchKernelGGL(probe, dim3(W), dim3(1024), 69632, 0, d, 2000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(2 * W); hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;
    for (int b = 0; b < W; ++b) {
        const unsigned hw = h[2 * b], x = h[2 * b + 1] & 15;
        const unsigned cuid = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        if (b < 48) printf("b %3d  xcc %u se %u sh %u cu %2u  (hw_id %08x)\n", b, x, se, sh, cuid, hw);
        cu[(x << 16) | (se << 8) | (sh << 4) | cuid].push_back(b);
    }
    printf("distinct CUs %zu\n", cu.size());
    std::map<int, int> delta;
    for (auto& e : cu) { if (e.second.size() == 2) delta[e.second[1] - e.second[0]]++; else delta[-(int)e.second.size()]++; }
    for (auto& e : delta) printf("pairs with blockIdx difference %d: %d\n", e.first, e.second);
    int shown = 0;
    for (auto& e : cu) if (shown++ < 12) { printf("cu %06x:", e.first); for (int b : e.second) printf(" %d", b); printf("\n"); }
    return 0;
}
