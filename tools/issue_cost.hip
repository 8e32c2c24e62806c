// Issue cost of the instructions k_propagate is made of, measured on the card (gfx950): cycles a SIMD spends per
// wave-instruction, (a) one wave per SIMD (what a lone wave's stream pays) and (b) eight waves per SIMD (what the kernel's
// occupancy pays: throughput).  Every kernel runs ITER iterations of 16 copies of ONE instruction over four independent
// register sets (so that neither the dependency latency nor the loop overhead is what is measured; the loop adds one
// s_sub + s_cmp + s_cbranch per 16), stamped with s_memtime by every wave; cost = span of the waves of one SIMD / (16 ITER waves of that SIMD).
//   hipcc --offload-arch=gfx950 -O3 tools/issue_cost.hip -o tools/issue_cost && tools/issue_cost > profiles/r03/issue_costs.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <map>

constexpr int ITER = 512;

// what every wave leaves behind: its two stamps and where it ran (XCD, shader engine, CU, SIMD), so that the host can take
// the SPAN of all waves that shared one SIMD (the oldest wave of a SIMD wins the issue arbitration and finishes first:
// a wave's own t1 - t0 is not the time the SIMD needed for all of them)
__device__ __forceinline__ void mp_record(unsigned long long* out, unsigned long long t0, unsigned long long t1) {
    unsigned int hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
    const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    out[3 * w] = t0;
    out[3 * w + 1] = t1;
    out[3 * w + 2] = ((unsigned long long)(xcc & 0xFu) << 32) | (hw & 0xFFFFFFF0u);   // everything but the wave slot
}

#define R4(a, b, c, d) a "\n\t" b "\n\t" c "\n\t" d "\n\t"
// one instruction template applied to the four register sets (A0..A3: 64-bit accumulators; X, Y: 64-bit sources)
#define BODY16(I0, I1, I2, I3) R4(I0, I1, I2, I3) R4(I0, I1, I2, I3) R4(I0, I1, I2, I3) R4(I0, I1, I2, I3)

#define DEF_KERNEL(NAME, I0, I1, I2, I3)                                                                                          \
    __global__ void NAME(unsigned long long* out, double seed) {                                                                  \
        double a0 = seed + threadIdx.x, a1 = a0 * 1.5, a2 = a0 * 2.5, a3 = a0 * 3.5;                                               \
        double x = 1.000000001 + 1e-9 * threadIdx.x, y = 0.999999999;                                                             \
        unsigned long long t0, t1;                                                                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                                                \
        for (int it = 0; it < ITER; ++it) {                                                                                        \
            asm volatile(BODY16(I0, I1, I2, I3) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y) : "vcc", "s6", "s7", "memory"); \
        }                                                                                                                          \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                                                \
        if ((threadIdx.x & 63) == 0) mp_record(out, t0, t1);                  \
        if (a0 + a1 + a2 + a3 == 12345.6789) out[0] = 0;                                                                           \
    }

// operands: %0..%3 accumulators (64-bit VGPR pairs), %4 = x, %5 = y (64-bit VGPR pairs); %L / %H of a pair are written as
// v[..] sub-registers through the modifiers below
#define LO(n) "%" #n
// 32-bit views: clang's AMDGPU inline asm has no sub-register modifier, so 32-bit instructions get 64-bit operand pairs via
// the "v" constraint on 32-bit temporaries instead: see the K32 kernels below.
DEF_KERNEL(k_fma_f64, "v_fma_f64 %0, %4, %5, %0", "v_fma_f64 %1, %4, %5, %1", "v_fma_f64 %2, %4, %5, %2", "v_fma_f64 %3, %4, %5, %3")
DEF_KERNEL(k_mul_f64, "v_mul_f64 %0, %0, %4", "v_mul_f64 %1, %1, %4", "v_mul_f64 %2, %2, %4", "v_mul_f64 %3, %3, %4")
DEF_KERNEL(k_add_f64, "v_add_f64 %0, %0, %4", "v_add_f64 %1, %1, %4", "v_add_f64 %2, %2, %4", "v_add_f64 %3, %3, %4")
DEF_KERNEL(k_max_f64, "v_max_f64 %0, %0, %4", "v_max_f64 %1, %1, %4", "v_max_f64 %2, %2, %4", "v_max_f64 %3, %3, %4")
DEF_KERNEL(k_rcp_f64, "v_rcp_f64 %0, %0", "v_rcp_f64 %1, %1", "v_rcp_f64 %2, %2", "v_rcp_f64 %3, %3")
DEF_KERNEL(k_rsq_f64, "v_rsq_f64 %0, %0", "v_rsq_f64 %1, %1", "v_rsq_f64 %2, %2", "v_rsq_f64 %3, %3")
DEF_KERNEL(k_sqrt_f64, "v_sqrt_f64 %0, %0", "v_sqrt_f64 %1, %1", "v_sqrt_f64 %2, %2", "v_sqrt_f64 %3, %3")
DEF_KERNEL(k_div_scale_f64, "v_div_scale_f64 %0, vcc, %0, %4, %0", "v_div_scale_f64 %1, vcc, %1, %4, %1", "v_div_scale_f64 %2, vcc, %2, %4, %2",
           "v_div_scale_f64 %3, vcc, %3, %4, %3")
DEF_KERNEL(k_div_fmas_f64, "v_div_fmas_f64 %0, %0, %4, %5", "v_div_fmas_f64 %1, %1, %4, %5", "v_div_fmas_f64 %2, %2, %4, %5", "v_div_fmas_f64 %3, %3, %4, %5")
DEF_KERNEL(k_div_fixup_f64, "v_div_fixup_f64 %0, %0, %4, %5", "v_div_fixup_f64 %1, %1, %4, %5", "v_div_fixup_f64 %2, %2, %4, %5", "v_div_fixup_f64 %3, %3, %4, %5")
DEF_KERNEL(k_ldexp_f64, "v_ldexp_f64 %0, %0, 1", "v_ldexp_f64 %1, %1, 1", "v_ldexp_f64 %2, %2, 1", "v_ldexp_f64 %3, %3, 1")
DEF_KERNEL(k_rndne_f64, "v_rndne_f64 %0, %0", "v_rndne_f64 %1, %1", "v_rndne_f64 %2, %2", "v_rndne_f64 %3, %3")
DEF_KERNEL(k_ceil_f64, "v_ceil_f64 %0, %0", "v_ceil_f64 %1, %1", "v_ceil_f64 %2, %2", "v_ceil_f64 %3, %3")
DEF_KERNEL(k_frexp_mant_f64, "v_frexp_mant_f64 %0, %0", "v_frexp_mant_f64 %1, %1", "v_frexp_mant_f64 %2, %2", "v_frexp_mant_f64 %3, %3")
DEF_KERNEL(k_cmp_f64, "v_cmp_lt_f64 vcc, %0, %4", "v_cmp_lt_f64 vcc, %1, %4", "v_cmp_lt_f64 vcc, %2, %4", "v_cmp_lt_f64 vcc, %3, %4")
DEF_KERNEL(k_mov_b64, "v_mov_b64 %0, %4", "v_mov_b64 %1, %4", "v_mov_b64 %2, %4", "v_mov_b64 %3, %4")
DEF_KERNEL(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %4", "v_lshl_add_u64 %1, %1, 0, %4", "v_lshl_add_u64 %2, %2, 0, %4", "v_lshl_add_u64 %3, %3, 0, %4")
DEF_KERNEL(k_lshrrev_b64, "v_lshrrev_b64 %0, 1, %0", "v_lshrrev_b64 %1, 1, %1", "v_lshrrev_b64 %2, 1, %2", "v_lshrrev_b64 %3, 1, %3")
DEF_KERNEL(k_cmp_u64, "v_cmp_lt_u64 vcc, %0, %4", "v_cmp_lt_u64 vcc, %1, %4", "v_cmp_lt_u64 vcc, %2, %4", "v_cmp_lt_u64 vcc, %3, %4")
DEF_KERNEL(k_pk_fma_f32, "v_pk_fma_f32 %0, %4, %5, %0", "v_pk_fma_f32 %1, %4, %5, %1", "v_pk_fma_f32 %2, %4, %5, %2", "v_pk_fma_f32 %3, %4, %5, %3")
DEF_KERNEL(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %4", "v_pk_mul_f32 %1, %1, %4", "v_pk_mul_f32 %2, %2, %4", "v_pk_mul_f32 %3, %3, %4")

// 32-bit forms
#define DEF_KERNEL32(NAME, I0, I1, I2, I3)                                                                                        \
    __global__ void NAME(unsigned long long* out, double seed) {                                                                  \
        uint32_t a0 = (uint32_t)seed + threadIdx.x, a1 = a0 * 3u + 1u, a2 = a0 * 5u + 2u, a3 = a0 * 7u + 3u;                       \
        uint32_t x = 0xD2511F53u + threadIdx.x, y = 0xCD9E8D57u;                                                                  \
        unsigned long long b0 = a0, b1 = a1, b2 = a2, b3 = a3;                                                                     \
        unsigned long long t0, t1;                                                                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                                                \
        for (int it = 0; it < ITER; ++it) {                                                                                        \
            asm volatile(BODY16(I0, I1, I2, I3)                                                                                    \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3)                          \
                         : "v"(x), "v"(y)                                                                                          \
                         : "vcc", "s6", "s7", "memory");                                                                           \
        }                                                                                                                          \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                                                \
        if ((threadIdx.x & 63) == 0) mp_record(out, t0, t1);                  \
        if (a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3 == 12345u) out[0] = 0;                                                           \
    }
// %0..%3 u32 accumulators, %4..%7 u64 accumulators, %8 = x, %9 = y (u32)
DEF_KERNEL32(k_mad_u64_u32, "v_mad_u64_u32 %4, s[6:7], %8, %0, %4", "v_mad_u64_u32 %5, s[6:7], %8, %1, %5", "v_mad_u64_u32 %6, s[6:7], %8, %2, %6",
             "v_mad_u64_u32 %7, s[6:7], %8, %3, %7")
DEF_KERNEL32(k_mad_u64_u32_zero, "v_mad_u64_u32 %4, s[6:7], %8, %0, 0", "v_mad_u64_u32 %5, s[6:7], %8, %1, 0", "v_mad_u64_u32 %6, s[6:7], %8, %2, 0",
             "v_mad_u64_u32 %7, s[6:7], %8, %3, 0")
DEF_KERNEL32(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %8", "v_mul_hi_u32 %1, %1, %8", "v_mul_hi_u32 %2, %2, %8", "v_mul_hi_u32 %3, %3, %8")
DEF_KERNEL32(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %8", "v_mul_lo_u32 %1, %1, %8", "v_mul_lo_u32 %2, %2, %8", "v_mul_lo_u32 %3, %3, %8")
DEF_KERNEL32(k_mul_u32_u24, "v_mul_u32_u24 %0, %0, %8", "v_mul_u32_u24 %1, %1, %8", "v_mul_u32_u24 %2, %2, %8", "v_mul_u32_u24 %3, %3, %8")
DEF_KERNEL32(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %8, %9", "v_mad_u32_u24 %1, %1, %8, %9", "v_mad_u32_u24 %2, %2, %8, %9", "v_mad_u32_u24 %3, %3, %8, %9")
DEF_KERNEL32(k_mad_u32_u16, "v_mad_u32_u16 %0, %0, %8, %9", "v_mad_u32_u16 %1, %1, %8, %9", "v_mad_u32_u16 %2, %2, %8, %9", "v_mad_u32_u16 %3, %3, %8, %9")
DEF_KERNEL32(k_xor_b32, "v_xor_b32 %0, %0, %8", "v_xor_b32 %1, %1, %8", "v_xor_b32 %2, %2, %8", "v_xor_b32 %3, %3, %8")
DEF_KERNEL32(k_add_u32, "v_add_u32 %0, %0, %8", "v_add_u32 %1, %1, %8", "v_add_u32 %2, %2, %8", "v_add_u32 %3, %3, %8")
DEF_KERNEL32(k_add3_u32, "v_add3_u32 %0, %0, %8, %9", "v_add3_u32 %1, %1, %8, %9", "v_add3_u32 %2, %2, %8, %9", "v_add3_u32 %3, %3, %8, %9")
DEF_KERNEL32(k_mov_b32, "v_mov_b32 %0, %8", "v_mov_b32 %1, %8", "v_mov_b32 %2, %8", "v_mov_b32 %3, %8")
DEF_KERNEL32(k_mov_dpp, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf",
             "v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf")
DEF_KERNEL32(k_cndmask_b32, "v_cndmask_b32 %0, %0, %8, vcc", "v_cndmask_b32 %1, %1, %8, vcc", "v_cndmask_b32 %2, %2, %8, vcc", "v_cndmask_b32 %3, %3, %8, vcc")
DEF_KERNEL32(k_cndmask_e64, "v_cndmask_b32 %0, %0, %8, s[6:7]", "v_cndmask_b32 %1, %1, %8, s[6:7]", "v_cndmask_b32 %2, %2, %8, s[6:7]", "v_cndmask_b32 %3, %3, %8, s[6:7]")
DEF_KERNEL32(k_cmp_cndmask, "v_cmp_lt_u32 vcc, %0, %8", "v_cndmask_b32 %1, %1, %8, vcc", "v_cmp_lt_u32 vcc, %2, %8", "v_cndmask_b32 %3, %3, %8, vcc")
DEF_KERNEL32(k_cmp_u32, "v_cmp_lt_u32 vcc, %0, %8", "v_cmp_lt_u32 vcc, %1, %8", "v_cmp_lt_u32 vcc, %2, %8", "v_cmp_lt_u32 vcc, %3, %8")
DEF_KERNEL32(k_readlane, "v_readlane_b32 s6, %0, 3", "v_readlane_b32 s7, %1, 5", "v_readlane_b32 s6, %2, 7", "v_readlane_b32 s7, %3, 9")
DEF_KERNEL32(k_readfirstlane, "v_readfirstlane_b32 s6, %0", "v_readfirstlane_b32 s7, %1", "v_readfirstlane_b32 s6, %2", "v_readfirstlane_b32 s7, %3")
DEF_KERNEL32(k_cvt_f64_u32, "v_cvt_f64_u32 %4, %0", "v_cvt_f64_u32 %5, %1", "v_cvt_f64_u32 %6, %2", "v_cvt_f64_u32 %7, %3")
DEF_KERNEL32(k_cvt_u32_f64, "v_cvt_u32_f64 %0, %4", "v_cvt_u32_f64 %1, %5", "v_cvt_u32_f64 %2, %6", "v_cvt_u32_f64 %3, %7")
DEF_KERNEL32(k_cvt_f32_f64, "v_cvt_f32_f64 %0, %4", "v_cvt_f32_f64 %1, %5", "v_cvt_f32_f64 %2, %6", "v_cvt_f32_f64 %3, %7")
DEF_KERNEL32(k_cvt_f64_f32, "v_cvt_f64_f32 %4, %0", "v_cvt_f64_f32 %5, %1", "v_cvt_f64_f32 %6, %2", "v_cvt_f64_f32 %7, %3")
DEF_KERNEL32(k_fma_f32, "v_fma_f32 %0, %0, %8, %9", "v_fma_f32 %1, %1, %8, %9", "v_fma_f32 %2, %2, %8, %9", "v_fma_f32 %3, %3, %8, %9")
DEF_KERNEL32(k_log_f32, "v_log_f32 %0, %0", "v_log_f32 %1, %1", "v_log_f32 %2, %2", "v_log_f32 %3, %3")
DEF_KERNEL32(k_rcp_f32, "v_rcp_f32 %0, %0", "v_rcp_f32 %1, %1", "v_rcp_f32 %2, %2", "v_rcp_f32 %3, %3")
DEF_KERNEL32(k_rsq_f32, "v_rsq_f32 %0, %0", "v_rsq_f32 %1, %1", "v_rsq_f32 %2, %2", "v_rsq_f32 %3, %3")
DEF_KERNEL32(k_bfe_u32, "v_bfe_u32 %0, %0, 3, 7", "v_bfe_u32 %1, %1, 3, 7", "v_bfe_u32 %2, %2, 3, 7", "v_bfe_u32 %3, %3, 3, 7")
DEF_KERNEL32(k_lshl_or_b32, "v_lshl_or_b32 %0, %0, 3, %8", "v_lshl_or_b32 %1, %1, 3, %8", "v_lshl_or_b32 %2, %2, 3, %8", "v_lshl_or_b32 %3, %3, 3, %8")
DEF_KERNEL32(k_alignbit_b32, "v_alignbit_b32 %0, %0, %8, 12", "v_alignbit_b32 %1, %1, %8, 12", "v_alignbit_b32 %2, %2, %8, 12", "v_alignbit_b32 %3, %3, %8, 12")
DEF_KERNEL32(k_s_mul_i32, "s_mul_i32 s6, s6, s7", "s_mul_hi_u32 s7, s6, s7", "s_mul_i32 s6, s6, s7", "s_mul_hi_u32 s7, s6, s7")
DEF_KERNEL32(k_s_nop, "s_nop 0", "s_nop 0", "s_nop 0", "s_nop 0")

// a DEPENDENT Philox-like chain: what one wave alone pays for the serial rounds (latency, not issue)
__global__ void k_philox_chain(unsigned long long* out, double seed) {
    uint32_t c0 = (uint32_t)seed + threadIdx.x, c1 = 1u, c2 = 2u, c3 = 3u, k0 = 11u, k1 = 13u;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {   // 8 rounds = 16 multiplies per iteration, like the 16-instruction bodies above
            const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
            const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
            c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if ((threadIdx.x & 63) == 0) mp_record(out, t0, t1);
    if (c0 + c1 + c2 + c3 == 12345u) out[0] = 0;
}

typedef void (*kern_t)(unsigned long long*, double);
struct entry { const char* name; kern_t k; };

struct result { double cost; int waves; };
static result run(kern_t k, int block, int grid, unsigned long long* d, std::vector<unsigned long long>& h) {
    const int waves = grid * (block / 64);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, d, 1.0);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, d, 1.0);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), d, sizeof(unsigned long long) * 3 * waves, hipMemcpyDeviceToHost);
    // group the waves by the SIMD they ran on; cost = span of the group / (instructions per wave x waves of the group)
    std::map<unsigned long long, std::vector<int>> groups;
    for (int w = 0; w < waves; ++w) groups[h[3 * w + 2]].push_back(w);
    std::vector<double> costs;
    std::vector<int> counts;
    for (auto& g : groups) {
        unsigned long long lo = ~0ull, hi = 0;
        for (int w : g.second) { lo = std::min(lo, h[3 * w]); hi = std::max(hi, h[3 * w + 1]); }
        costs.push_back((double)(hi - lo) / (16.0 * ITER * g.second.size()));
        counts.push_back((int)g.second.size());
    }
    std::sort(costs.begin(), costs.end());
    std::sort(counts.begin(), counts.end());
    return {costs[costs.size() / 2], counts[counts.size() / 2]};
}

int main() {
    unsigned long long* d;
    const int max_waves = 512 * 16 * 3;
    (void)hipMalloc(&d, sizeof(unsigned long long) * max_waves);
    std::vector<unsigned long long> h(max_waves);
#define E(n) {#n, n}
    const entry es[] = {E(k_fma_f64), E(k_mul_f64), E(k_add_f64), E(k_max_f64), E(k_rcp_f64), E(k_rsq_f64), E(k_sqrt_f64), E(k_div_scale_f64),
                        E(k_div_fmas_f64), E(k_div_fixup_f64), E(k_ldexp_f64), E(k_rndne_f64), E(k_ceil_f64), E(k_frexp_mant_f64), E(k_cmp_f64),
                        E(k_mov_b64), E(k_lshl_add_u64), E(k_lshrrev_b64), E(k_cmp_u64), E(k_pk_fma_f32), E(k_pk_mul_f32),
                        E(k_mad_u64_u32), E(k_mad_u64_u32_zero), E(k_mul_hi_u32), E(k_mul_lo_u32), E(k_mul_u32_u24), E(k_mad_u32_u24), E(k_mad_u32_u16),
                        E(k_xor_b32), E(k_add_u32), E(k_add3_u32), E(k_mov_b32), E(k_mov_dpp), E(k_cndmask_b32), E(k_cndmask_e64), E(k_cmp_cndmask), E(k_cmp_u32),
                        E(k_readlane), E(k_readfirstlane), E(k_cvt_f64_u32), E(k_cvt_u32_f64), E(k_cvt_f32_f64), E(k_cvt_f64_f32), E(k_fma_f32),
                        E(k_log_f32), E(k_rcp_f32), E(k_rsq_f32), E(k_bfe_u32), E(k_lshl_or_b32), E(k_alignbit_b32), E(k_s_mul_i32), E(k_s_nop),
                        E(k_philox_chain)};
    std::printf("# gfx950 issue cost per wave-instruction, shader cycles per SIMD (median wave; %d x 16 instructions per wave)\n", ITER);
    std::printf("# span of the waves that shared one SIMD / (their instructions); in brackets: waves per SIMD (median SIMD)\n");
    std::printf("# %-22s %12s %12s %12s\n", "instruction", "256x256", "1024x256", "1024x512");
    for (const entry& e : es) {
        // 256 threads x 256 workgroups: one wave per SIMD; 1024 x 256: four; 1024 x 512: eight (two workgroups per CU)
        const result c1 = run(e.k, 256, 256, d, h), c4 = run(e.k, 1024, 256, d, h), c8 = run(e.k, 1024, 512, d, h);
        std::printf("%-24s %8.2f (%d) %8.2f (%d) %8.2f (%d)\n", e.name + 2, c1.cost, c1.waves, c4.cost, c4.waves, c8.cost, c8.waves);
    }
    (void)hipFree(d);
    return 0;
}
