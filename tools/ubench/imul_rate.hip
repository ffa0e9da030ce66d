// Microbenchmark: issue rate of the integer-multiply instructions a 64-bit
// Montgomery product is built from on gfx950, against full-rate VALU ops.
// Each kernel runs a dependent-free unrolled stream per lane; reports
// cycles per wave-instruction per SIMD (s_memtime based) and Gops/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 8;   // independent chains per lane

template <int OP>
__global__ void __launch_bounds__(256) k(uint64_t* out, uint32_t seed) {
    uint32_t a[UNROLL], b[UNROLL];
    uint64_t c[UNROLL];
    double d[UNROLL];
    for (int i = 0; i < UNROLL; ++i) {
        a[i] = seed * (threadIdx.x + 1) + i * 77u;
        b[i] = seed ^ (0x9e3779b9u * (i + 1));
        c[i] = (uint64_t)a[i] << 20 | b[i];
        d[i] = (double)a[i];
    }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if (OP == 0) {            // v_mad_u64_u32
                c[i] = (uint64_t)(uint32_t)c[i] * b[i] + c[i];
            } else if (OP == 1) {     // v_mul_lo_u32
                a[i] = a[i] * b[i];
            } else if (OP == 2) {     // v_mul_hi_u32
                a[i] = __umulhi(a[i], b[i]);
            } else if (OP == 3) {     // v_add_u32 (full rate reference)
                a[i] = a[i] + b[i];  b[i] ^= a[i];
            } else if (OP == 4) {     // 64x64 -> lo64 (compiler's choice)
                c[i] = c[i] * (c[i] | 1);
            } else if (OP == 5) {     // v_fma_f64
                d[i] = __builtin_fma(d[i], 1.0000001, 0.5);
            } else if (OP == 6) {     // v_mul_u32_u24
                a[i] = (a[i] & 0xffffffu) * (b[i] & 0xffffffu);
            } else if (OP == 7) {     // 64-bit add (v_add_co + v_addc)
                c[i] = c[i] + (((uint64_t)b[i] << 32) | a[i]);
            } else if (OP == 8) {     // mulhi64 (compiler's choice)
                c[i] = __umul64hi(c[i], c[i] | 0x8000000000000001ull);
            }
        }
    }
    uint64_t r = 0;
    for (int i = 0; i < UNROLL; ++i) r += a[i] + b[i] + c[i] + (uint64_t)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
int run(const char* name, int insts_per_op) {
    int blocks = 256 * 8, threads = 256;
    uint64_t* out;
    CHECK(hipMalloc(&out, (size_t)blocks * threads * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<OP><<<blocks, threads>>>(out, 12345u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k<OP><<<blocks, threads>>>(out, 12345u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double ops = (double)blocks * threads * ITERS * UNROLL;        // lane-ops
    double waveinst = ops / 64.0;
    // 256 CUs * 4 SIMDs; clock unknown -> report ns per wave-inst per SIMD
    double ns_per = ms * 1e6 / (waveinst / (256.0 * 4.0));
    printf("%-28s %8.3f ms  %8.2f Glane-op/s  %6.2f ns/wave-op/SIMD (~%4.1f cyc @2.4GHz, %d inst/op)\n",
           name, ms, ops / ms * 1e-6, ns_per, ns_per * 2.4, insts_per_op);
    CHECK(hipFree(out));
    return 0;
}

int main() {
    run<3>("v_add_u32+v_xor (2 inst)", 2);
    run<0>("v_mad_u64_u32", 1);
    run<1>("v_mul_lo_u32", 1);
    run<2>("v_mul_hi_u32", 1);
    run<6>("v_mul_u32_u24", 1);
    run<7>("add64 (2 inst)", 2);
    run<4>("mul64 lo", 3);
    run<8>("mulhi64", 4);
    run<5>("v_fma_f64", 1);
    return 0;
}
