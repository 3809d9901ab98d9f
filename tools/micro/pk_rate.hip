// Micro-benchmark: issue rate of VALU instruction kinds on gfx950, cycles per wave64 instruction per SIMD, against waves per SIMD.
// Eight independent chains per wave (no dependent-issue stalls) unless the name says "dep".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 64
template <int OP> __global__ void __launch_bounds__(256) k(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t a[8]; for (int i = 0; i < 8; i++) a[i] = seed * (threadIdx.x + i + 1);
    const uint32_t c = seed | 0x00030005u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 1) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 2) asm volatile("v_pk_sub_i16 %0, %0, %1 clamp" : "+v"(a[i]) : "v"(c));
                if (OP == 3) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c) : "vcc");
                if (OP == 5) asm volatile("v_cmp_ge_i32 vcc, %0, %1" : : "v"(a[i]), "v"(c) : "vcc");
                if (OP == 6) asm volatile("v_cmp_ge_i32 s[20:21], %0, %1" : : "v"(a[i]), "v"(c) : "s20", "s21");
                if (OP == 7) asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(c));
                if (OP == 8) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 9) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 10) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 11) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 12) asm volatile("v_cmp_ge_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c) : "vcc");
                if (OP == 13) asm volatile("v_cmp_ge_i32 s[20:21], %0, %1\n v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(c) : "s20", "s21");
                if (OP == 14) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[0]) : "v"(c));                       // dep: one chain
                if (OP == 15) asm volatile("v_pk_sub_i16 %0, %0, %1 clamp" : "+v"(a[0]) : "v"(c));              // dep
                if (OP == 16) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[i]));
                if (OP == 17) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 18) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 19) asm volatile("v_sub_u32 %0, %0, %1\n v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(c));      // alternating 32-bit / packed
                if (OP == 20) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(a[i]));                            // VOP2 with a 32-bit literal (8-byte encoding)
                if (OP == 21) asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(c));                    // the same operation in the VOP3 encoding
                if (OP == 22) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "s"(c));                        // SGPR operand
                if (OP == 23) asm volatile("v_addc_co_u32 %0, vcc, %0, %0, s[20:21]" : "+v"(a[i]) : : "vcc");        // shift a lane-mask bit in: carry-in from an SGPR pair
                if (OP == 24) asm volatile("v_cmp_ge_i32 s[20:21], %0, %1\n v_max_i32 %0, %0, %1\n s_nop 1\n v_addc_co_u32 %2, vcc, %2, %2, s[20:21]" : "+v"(a[i]), "+v"(a[(i + 1) & 7]) : "v"(c) : "vcc", "s20", "s21");
                if (OP == 25) asm volatile("v_cmp_ge_i32 s[20:21], %0, %1\n v_max_i32 %0, %0, %1\n v_cndmask_b32 %2, 0, 4, s[20:21]\n v_or_b32 %0, %0, %2" : "+v"(a[i]), "+v"(a[(i + 1) & 7]) : "v"(c) : "s20", "s21");
                if (OP == 26) asm volatile("v_addc_co_u32 %0, s[22:23], %0, %0, s[20:21]" : "+v"(a[i]) : : "s22", "s23");
            }
        }
    }
    uint32_t s = 0; for (int i = 0; i < 8; i++) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char *name, uint32_t *d)
{
    const int iters = 100;
    printf("%-44s", name);
    for (int w = 1; w <= 4; w++) {       // w blocks of 256 threads per CU = w waves per SIMD
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(256 * w), dim3(256), 0, 0, d, 4, 12345u); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(256 * w), dim3(256), 0, 0, d, iters, 12345u); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        const int perAsm = (OP == 12 || OP == 13 || OP == 19) ? 2 : (OP == 24 ? 3 : (OP == 25 ? 4 : 1));
        const double instr = (double)iters * REP * 8 * perAsm * w;       // per SIMD
        printf("  %dw: %5.2f", w, ms * 1e-3 * 2.4e9 / instr);
    }
    printf("\n");
}
int main()
{
    uint32_t *d; (void)hipMalloc(&d, 1 << 24);
    printf("cycles (at 2.4 GHz) per wave64 instruction per SIMD, with 1..4 waves resident per SIMD\n");
    run<0>("v_sub_u32 (VOP2)", d); run<1>("v_max_i32 (VOP2)", d); run<16>("v_lshrrev_b32 (VOP2)", d); run<17>("v_and_b32 (VOP2)", d); run<18>("v_mov_b32 (VOP1)", d);
    run<20>("v_add_u32 with literal (VOP2, 8 bytes)", d); run<21>("v_add_u32_e64 (VOP3)", d); run<22>("v_sub_u32 with SGPR operand", d);
    run<4>("v_cndmask_b32 vcc (VOP2)", d); run<5>("v_cmp_ge_i32 vcc (VOPC)", d); run<6>("v_cmp_ge_i32 sgpr pair (VOP3)", d); run<7>("v_cndmask_b32 sgpr pair (VOP3)", d);
    run<12>("v_cmp vcc + dependent v_cndmask", d); run<13>("v_cmp sgpr pair + dependent v_cndmask", d);
    run<8>("v_add3_u32 (VOP3)", d); run<9>("v_lshl_or_b32 (VOP3)", d); run<10>("v_bfi_b32 (VOP3)", d); run<11>("v_and_or_b32 (VOP3)", d);
    run<2>("v_pk_sub_i16 clamp (VOP3P)", d); run<3>("v_pk_max_i16 (VOP3P)", d); run<19>("v_sub_u32 / v_pk_max_i16 alternating", d);
    run<23>("v_addc_co_u32 vcc out, sgpr-pair carry in", d); run<26>("v_addc_co_u32 sgpr out, sgpr-pair carry in", d); run<24>("v_cmp sgpr + v_max + s_nop 1 + v_addc (per VALU)", d); run<25>("v_cmp sgpr + v_max + v_cndmask + v_or (per VALU)", d);
    run<14>("v_sub_u32, one dependent chain", d); run<15>("v_pk_sub_i16 clamp, one dependent chain", d);
    return 0;
}
