// Microbenchmark #2: per-instruction VALU issue cost on gfx950 with inline asm (exact opcodes),
// 8 waves/SIMD, independent destination registers.  Prints cycles per wave-instruction per SIMD
// using the measured shader clock (s_memtime) instead of an assumed frequency.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
constexpr int ITERS = 2048;

#define REP8(S) S S S S S S S S
template <int OP>
__global__ void __launch_bounds__(256) k(uint64_t* out, unsigned long long* clk) {
    uint32_t a = threadIdx.x * 7 + 1, b = threadIdx.x * 13 + 5;
    uint64_t c0 = a, c1 = b, c2 = a + b, c3 = a * 3, d0 = 1, d1 = 2, d2 = 3, d3 = 4;
    uint32_t e0 = a, e1 = b, e2 = a ^ b, e3 = a + 7;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
        if (OP == 0) { REP8(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b) : "vcc");) }
        if (OP == 1) { REP8(asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(d0));) }
        if (OP == 2) { REP8(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %4\n v_mov_b32 %3, %5" : "=v"(e0), "=v"(e1), "=v"(e2), "=v"(e3) : "v"(a), "v"(b));) }
        if (OP == 3) { REP8(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %5, vcc\n v_add_co_u32 %2, vcc, %2, %4\n v_addc_co_u32 %3, vcc, %3, %5, vcc" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(a), "v"(b) : "vcc");) }
        if (OP == 4) { REP8(asm volatile("v_cmp_ge_u64 vcc, %0, %4\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_ge_u64 vcc, %1, %4\n v_cndmask_b32 %3, %3, %2, vcc" : "+v"(c0), "+v"(c1), "+v"(e2), "+v"(e3) : "v"(d0) : "vcc");) }
        if (OP == 5) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %5\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %5" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(a), "v"(b));) }
        if (OP == 6) { REP8(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %5" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(a), "v"(b));) }
        if (OP == 7) { REP8(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %5\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %5" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(a), "v"(b));) }
        if (OP == 8) { REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(a), "v"(b));) }
        if (OP == 9) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(d0));) }
        if (OP == 10) { REP8(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_add_u32 %6, %6, %4\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_add_u32 %7, %7, %5\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_add_u32 %6, %6, %5\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n v_add_u32 %7, %7, %4" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b), "v"(e0), "v"(e1) : "vcc");) }
        if (OP == 11) { REP8(asm volatile("v_sub_co_u32 %0, vcc, %0, %4\n v_subb_co_u32 %1, vcc, %1, %5, vcc\n v_cndmask_b32 %2, %2, %0, vcc\n v_cndmask_b32 %3, %3, %1, vcc" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(a), "v"(b) : "vcc");) }
        if (OP == 13) { REP8(asm volatile("v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %1, vcc, %4, %5, %1\n v_mad_i64_i32 %2, vcc, %4, %5, %2\n v_mad_i64_i32 %3, vcc, %4, %5, %3" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b) : "vcc");) }
        if (OP == 14) { REP8(asm volatile("v_ashrrev_i64 %0, 32, %0\n v_ashrrev_i64 %1, 32, %1\n v_ashrrev_i64 %2, 32, %2\n v_ashrrev_i64 %3, 32, %3" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));) }
        if (OP == 15) { REP8(asm volatile("v_ashrrev_i32 %0, 31, %0\n v_ashrrev_i32 %1, 31, %1\n v_ashrrev_i32 %2, 31, %2\n v_ashrrev_i32 %3, 31, %3" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));) }
        if (OP == 16) { REP8(asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %5\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %5" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(a), "v"(b));) }
        if (OP == 17) { REP8(asm volatile("v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %5, %4\n v_add3_u32 %2, %2, %4, %5\n v_add3_u32 %3, %3, %5, %4" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(a), "v"(b));) }
        if (OP == 12) { REP8(asm volatile("v_lshrrev_b64 %0, 32, %0\n v_lshrrev_b64 %1, 32, %1\n v_lshrrev_b64 %2, 32, %2\n v_lshrrev_b64 %3, 32, %3" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));) }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + e0 + e1 + e2 + e3 + d1 + d2 + d3;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
template <int OP> int run(const char* name) {
    int blocks = 256 * 8, threads = 256;   // 8 blocks/CU * 4 waves = 8 waves/SIMD
    uint64_t* out; unsigned long long* clk;
    CHECK(hipMalloc(&out, (size_t)blocks * threads * 8)); CHECK(hipMalloc(&clk, blocks * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<OP><<<blocks, threads>>>(out, clk); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); k<OP><<<blocks, threads>>>(out, clk); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2048]; CHECK(hipMemcpy(h, clk, blocks * 8, hipMemcpyDeviceToHost));
    double avg = 0; for (int i = 0; i < blocks; ++i) avg += h[i]; avg /= blocks;
    double insts_per_wave = (double)ITERS * 32;           // 8 reps * 4 insts (OP 10: 8 reps * 8)
    if (OP == 10) insts_per_wave *= 2;
    double waves_per_simd = 8.0;
    // wall-clock based: all waves of a SIMD share it; total inst per SIMD = waves_per_simd * insts_per_wave
    double ns_per_inst = ms * 1e6 / (waves_per_simd * insts_per_wave);
    printf("%-34s %7.3f ms  %6.3f ns/inst/SIMD   in-kernel s_memtime ticks/inst (8 waves sharing): %6.2f\n", name, ms, ns_per_inst, avg / insts_per_wave / 8.0 * 1.0);
    return 0;
}
int main() {
    run<6>("v_add_u32"); run<8>("v_fma_f32"); run<9>("v_pk_fma_f32"); run<2>("v_mov_b32"); run<3>("v_add_co/v_addc_co pair");
    run<11>("v_sub_co,v_subb,2x cndmask"); run<4>("v_cmp_ge_u64 + v_cndmask"); run<1>("v_lshl_add_u64"); run<12>("v_lshrrev_b64");
    run<13>("v_mad_i64_i32"); run<14>("v_ashrrev_i64 (by 32)"); run<15>("v_ashrrev_i32"); run<16>("v_xor_b32"); run<17>("v_add3_u32");
    run<0>("v_mad_u64_u32"); run<5>("v_mul_lo_u32"); run<7>("v_mul_hi_u32"); run<10>("v_mad_u64_u32 + v_add_u32 interleaved");
    return 0;
}
