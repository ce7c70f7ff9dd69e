// valu_rate.hip — microbenchmark: issue rate of the VALU ops k_prefilter is made of, on gfx950.
// Establishes the *measured* ceiling the prefilter's VALU roofline is priced against.
//   hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip && ./valu_rate
// For each op mix: cycles per wave-instruction per SIMD at 1, 2, 4, 8 waves per SIMD (all 256 CUs busy).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define ITER 2048


#define K4(KN, I0, I1, I2, I3) else if (KIND == KN) { REP8(asm volatile(I0 "\n" I1 "\n" I2 "\n" I3 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "s"(s) : "vcc");) }
template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned seed, unsigned long long *cyc) {
    unsigned a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x9e3779b9u, a2 = a0 * 3u, a3 = a1 * 5u;
    unsigned b0 = a0 + 1, b1 = a1 + 2, b2 = a2 + 3, b3 = a3 + 4;
    float f0 = a0, f1 = a1, f2 = a2, f3 = a3;
    unsigned s = seed | 1u;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITER; i++) {
        if (KIND == 0) {  // v_xor_b32 (VOP2, SGPR operand)
            REP8(asm volatile("v_xor_b32 %0, %4, %0\n v_xor_b32 %1, %4, %1\n v_xor_b32 %2, %4, %2\n v_xor_b32 %3, %4, %3"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(s));)
        } else if (KIND == 1) {  // v_bcnt_u32_b32
            REP8(asm volatile("v_bcnt_u32_b32 %0, %4, %0\n v_bcnt_u32_b32 %1, %5, %1\n v_bcnt_u32_b32 %2, %6, %2\n v_bcnt_u32_b32 %3, %7, %3"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
        } else if (KIND == 2) {  // v_min3_u32
            REP8(asm volatile("v_min3_u32 %0, %0, %4, %5\n v_min3_u32 %1, %1, %5, %6\n v_min3_u32 %2, %2, %6, %7\n v_min3_u32 %3, %3, %7, %4"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
        } else if (KIND == 3) {  // v_fma_f32 (reference: documented 2 cycles/wave-instr at >=2 waves/SIMD)
            REP8(asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %2\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %0"
                              : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));)
        } else if (KIND == 4) {  // the prefilter mix for 2 pairs: 2 xor, 2 bcnt, 1 min3 (x 6.4 = 32 instr)
            REP8(asm volatile("v_xor_b32 %1, %4, %5\n v_xor_b32 %2, %4, %6\n v_bcnt_u32_b32 %1, %1, 0\n v_bcnt_u32_b32 %2, %2, 0\n v_min3_u32 %0, %0, %1, %2"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(s), "v"(b0), "v"(b1));)
        } else if (KIND == 5) {  // v_min_u32 (VOP2)
            REP8(asm volatile("v_min_u32 %0, %4, %0\n v_min_u32 %1, %5, %1\n v_min_u32 %2, %6, %2\n v_min_u32 %3, %7, %3"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
        } else if (KIND == 6) {  // v_and_b32 + v_add_u32 (cheap int VOP2 pair)
            REP8(asm volatile("v_and_b32 %0, %4, %0\n v_add_u32 %1, %5, %1\n v_and_b32 %2, %6, %2\n v_add_u32 %3, %7, %3"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
        } else if (KIND == 7) {  // v_pk_fma_f32-free check: v_cmp_le_u32 writing vcc + v_addc (mask shift-in)
            REP8(asm volatile("v_cmp_le_u32 vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %1, vcc\n v_cmp_le_u32 vcc, %2, %5\n v_addc_co_u32 %3, vcc, %3, %3, vcc"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "vcc");)
        }
        K4(10, "v_xor_b32 %0, %4, %0", "v_xor_b32 %1, %5, %1", "v_xor_b32 %2, %6, %2", "v_xor_b32 %3, %7, %3")
        K4(11, "v_or_b32 %0, %4, %0", "v_or_b32 %1, %5, %1", "v_or_b32 %2, %6, %2", "v_or_b32 %3, %7, %3")
        K4(12, "v_and_b32 %0, %8, %0", "v_and_b32 %1, %8, %1", "v_and_b32 %2, %8, %2", "v_and_b32 %3, %8, %3")
        K4(13, "v_add_u32 %0, %8, %0", "v_add_u32 %1, %8, %1", "v_add_u32 %2, %8, %2", "v_add_u32 %3, %8, %3")
        K4(14, "v_mov_b32 %0, %4", "v_mov_b32 %1, %5", "v_mov_b32 %2, %6", "v_mov_b32 %3, %7")
        K4(15, "v_sub_u32 %0, %4, %0", "v_sub_u32 %1, %5, %1", "v_sub_u32 %2, %6, %2", "v_sub_u32 %3, %7, %3")
        K4(16, "v_lshlrev_b32 %0, 1, %0", "v_lshlrev_b32 %1, 1, %1", "v_lshlrev_b32 %2, 1, %2", "v_lshlrev_b32 %3, 1, %3")
        K4(17, "v_xad_u32 %0, %0, %4, %5", "v_xad_u32 %1, %1, %5, %6", "v_xad_u32 %2, %2, %6, %7", "v_xad_u32 %3, %3, %7, %4")
        K4(18, "v_sad_u8 %0, %0, %4, %5", "v_sad_u8 %1, %1, %5, %6", "v_sad_u8 %2, %2, %6, %7", "v_sad_u8 %3, %3, %7, %4")
        K4(19, "v_add3_u32 %0, %0, %4, %5", "v_add3_u32 %1, %1, %5, %6", "v_add3_u32 %2, %2, %6, %7", "v_add3_u32 %3, %3, %7, %4")
        K4(20, "v_and_or_b32 %0, %0, %4, %5", "v_and_or_b32 %1, %1, %5, %6", "v_and_or_b32 %2, %2, %6, %7", "v_and_or_b32 %3, %3, %7, %4")
        K4(21, "v_max_u32 %0, %4, %0", "v_max_u32 %1, %5, %1", "v_max_u32 %2, %6, %2", "v_max_u32 %3, %7, %3")
        K4(22, "v_mul_u32_u24 %0, %4, %0", "v_mul_u32_u24 %1, %5, %1", "v_mul_u32_u24 %2, %6, %2", "v_mul_u32_u24 %3, %7, %3")
        K4(23, "v_cndmask_b32 %0, %4, %0, vcc", "v_cndmask_b32 %1, %5, %1, vcc", "v_cndmask_b32 %2, %6, %2, vcc", "v_cndmask_b32 %3, %7, %3, vcc")
        K4(24, "v_pk_add_u16 %0, %0, %4", "v_pk_add_u16 %1, %1, %5", "v_pk_add_u16 %2, %2, %6", "v_pk_add_u16 %3, %3, %7")
        K4(25, "v_pk_min_u16 %0, %0, %4", "v_pk_min_u16 %1, %1, %5", "v_pk_min_u16 %2, %2, %6", "v_pk_min_u16 %3, %3, %7")
        K4(26, "v_bcnt_u32_b32 %0, %4, 0", "v_bcnt_u32_b32 %1, %5, 0", "v_bcnt_u32_b32 %2, %6, 0", "v_bcnt_u32_b32 %3, %7, 0")
        K4(27, "v_xor_b32 %0, %8, %4", "v_xor_b32 %1, %8, %5", "v_xor_b32 %2, %8, %6", "v_xor_b32 %3, %8, %7")
        K4(28, "v_bfe_u32 %0, %0, 1, 31", "v_bfe_u32 %1, %1, 1, 31", "v_bfe_u32 %2, %2, 1, 31", "v_bfe_u32 %3, %3, 1, 31")
        K4(29, "v_alignbit_b32 %0, %0, %4, 3", "v_alignbit_b32 %1, %1, %5, 3", "v_alignbit_b32 %2, %2, %6, 3", "v_alignbit_b32 %3, %3, %7, 3")
        K4(30, "v_mov_b32_dpp %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %1, %5 row_ror:8 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %2, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %3, %7 row_mirror row_mask:0xf bank_mask:0xf")
        K4(31, "v_permlane32_swap_b32 %0, %1", "v_permlane16_swap_b32 %2, %3", "v_permlane32_swap_b32 %1, %2", "v_permlane16_swap_b32 %3, %0")
        K4(32, "ds_bpermute_b32 %0, %4, %0", "ds_bpermute_b32 %1, %5, %1", "ds_bpermute_b32 %2, %6, %2", "ds_bpermute_b32 %3, %7, %3\n s_waitcnt lgkmcnt(0)")
        K4(33, "ds_swizzle_b32 %0, %0 offset:swizzle(BITMASK_PERM, \"0000p\")", "ds_swizzle_b32 %1, %1 offset:swizzle(BITMASK_PERM, \"000p0\")", "ds_swizzle_b32 %2, %2 offset:swizzle(BITMASK_PERM, \"00p00\")", "ds_swizzle_b32 %3, %3 offset:swizzle(BITMASK_PERM, \"p0000\")\n s_waitcnt lgkmcnt(0)")
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ (unsigned)(f0 + f1 + f2 + f3);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char *name, int instr_per_iter) {
    unsigned *out;
    unsigned long long *cyc;
    hipMalloc(&out, 256 * 8 * 256 * 4 * 4);
    hipMalloc(&cyc, 256 * 8 * 8);
    printf("%-28s", name);
    for (int wps : {1, 2, 4, 8}) {
        int blocks = 256 * wps;  // 256-thread blocks: 4 waves = one per SIMD
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        k<KIND><<<blocks, 256>>>(out, 1, cyc);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<KIND><<<blocks, 256>>>(out, 2, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0;
        for (auto v : h) avg += (double)v;
        avg /= blocks;
        // s_memtime ticks at 100 MHz on gfx9? report both: wall-based cycles at 2.4 GHz and raw ticks
        double instr_per_simd = (double)ITER * instr_per_iter * wps;
        double cyc_wall = ms * 1e-3 * 2.4e9 / instr_per_simd;
        printf("  w=%d: %6.2f cyc/instr (wall %.1f us, ticks/instr %.3f)", wps, cyc_wall, ms * 1e3, avg / ((double)ITER * instr_per_iter));
    }
    printf("\n");
}

int main() {
    run<3>("v_fma_f32", 32);
    run<0>("v_xor_b32 (sgpr src)", 32);
    run<1>("v_bcnt_u32_b32", 32);
    run<2>("v_min3_u32", 32);
    run<5>("v_min_u32", 32);
    run<6>("v_and_b32/v_add_u32", 32);
    run<4>("prefilter mix 2xor2bcnt1min3", 40);
    run<7>("v_cmp+v_addc", 32);
    run<10>("v_xor_b32 vgpr", 32);
    run<27>("v_xor_b32 sgpr,vgpr->new", 32);
    run<11>("v_or_b32", 32);
    run<12>("v_and_b32 sgpr", 32);
    run<13>("v_add_u32 sgpr", 32);
    run<14>("v_mov_b32", 32);
    run<15>("v_sub_u32", 32);
    run<16>("v_lshlrev_b32", 32);
    run<17>("v_xad_u32", 32);
    run<18>("v_sad_u8", 32);
    run<19>("v_add3_u32", 32);
    run<20>("v_and_or_b32", 32);
    run<21>("v_max_u32", 32);
    run<22>("v_mul_u32_u24", 32);
    run<23>("v_cndmask_b32", 32);
    run<24>("v_pk_add_u16", 32);
    run<25>("v_pk_min_u16", 32);
    run<26>("v_bcnt_u32_b32 x,0", 32);
    run<28>("v_bfe_u32", 32);
    run<29>("v_alignbit_b32", 32);
    run<30>("v_mov_b32_dpp", 32);
    run<31>("v_permlane32/16_swap", 32);
    run<32>("ds_bpermute_b32", 32);
    run<33>("ds_swizzle_b32", 32);
    return 0;
}
