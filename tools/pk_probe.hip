// Issue-rate probe (gfx950): cycles per instruction of v_add_f32 / v_mul_f32 / v_pk_add_f32 / v_pk_mul_f32 / v_fma_f32 /
// v_add_f64 / v_cndmask streams, one to four waves per SIMD.  hipcc --offload-arch=gfx950 -O3 tools/pk_probe.hip -o pk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
template <int KIND> __global__ void probe(unsigned long long *out, int iters) {
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
    float2 p0 = {1.f, 2.f}, p1 = {3.f, 4.f}, p2 = {5.f, 6.f}, p3 = {7.f, 8.f};
    double d0 = 1.0, d1 = 2.0, d2 = 3.0, d3 = 4.0;
    const float k = 1.0000001f;
    const float2 kk = {k, k};
    const double kd = 1.0000001;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
            REP16(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k));)
        } else if (KIND == 1) {
            REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(kk));)
        } else if (KIND == 2) {
            REP16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(kk));)
        } else if (KIND == 3) {
            REP16(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k));)
        } else if (KIND == 4) {
            REP16(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(kd));)
        } else if (KIND == 5) {
            REP16(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k) : "vcc");)
        } else if (KIND == 6) {
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(kk));)
        } else if (KIND == 7) {
            REP16(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k));)
        } else if (KIND == 8) { // pk_add with op_sel / neg modifiers (a complex butterfly's form)
            REP16(asm volatile("v_pk_add_f32 %0, %0, %4 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %1, %1, %4 op_sel:[0,1] op_sel_hi:[1,0]\n v_pk_add_f32 %2, %2, %4 neg_lo:[0,1]\n v_pk_add_f32 %3, %3, %4"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(kk));)
        } else if (KIND == 10) { // mask in an SGPR pair other than vcc
            REP16(asm volatile("v_cndmask_b32 %0, %0, %4, s[20:21]\n v_cndmask_b32 %1, %1, %4, s[20:21]\n v_cndmask_b32 %2, %2, %4, s[20:21]\n v_cndmask_b32 %3, %3, %4, s[20:21]"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k) : "s20", "s21");)
        } else if (KIND == 11) { // an SGPR as a plain operand
            REP16(asm volatile("v_add_f32 %0, s20, %0\n v_add_f32 %1, s20, %1\n v_add_f32 %2, s20, %2\n v_add_f32 %3, s20, %3"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "s20");)
        } else if (KIND == 12) { // a literal constant
            REP16(asm volatile("v_add_f32 %0, 0x3f800001, %0\n v_add_f32 %1, 0x3f800001, %1\n v_add_f32 %2, 0x3f800001, %2\n v_add_f32 %3, 0x3f800001, %3"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (KIND == 13) { // compares writing vcc
            REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cmp_lt_f32 vcc, %1, %4\n v_cmp_lt_f32 vcc, %2, %4\n v_cmp_lt_f32 vcc, %3, %4"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k) : "vcc");)
        } else if (KIND == 14) { // compare + select pairs
            REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %4, vcc"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k) : "vcc");)
        } else if (KIND == 15) {
            REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (KIND == 16) {
            REP16(asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k));)
        } else if (KIND == 17) { // dependent chain
            REP16(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %0, %0, %4"
                               : "+v"(a0) : "v"(a1), "v"(a2), "v"(a3), "v"(k));)
        } else if (KIND == 18) {
            REP16(asm volatile("v_bfi_b32 %0, %4, %0, %1\n v_bfi_b32 %1, %4, %1, %2\n v_bfi_b32 %2, %4, %2, %3\n v_bfi_b32 %3, %4, %3, %0"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k));)
        } else if (KIND == 19) { // selects on a mask that was written by a compare before the loop
            if (i == 0) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a0), "v"(k) : "vcc");
            REP16(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k));)
        } else if (KIND == 20) {
            REP16(asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(kd));)
        } else if (KIND == 21) {
            REP16(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (KIND == 22) {
            REP16(asm volatile("v_mov_b32_dpp %0, %0 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 wave_ror:1 row_mask:0xf bank_mask:0xf"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (KIND == 23) { // one compare, three selects on it
            REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k) : "vcc");)
        } else if (KIND == 24) { // one compare, then an unrelated instruction, then selects
            REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_add_f32 %1, %1, %4\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k) : "vcc");)
        } else if (KIND == 25) { // VOP3 form with vcc named explicitly
            REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %4, vcc\n v_cndmask_b32_e64 %1, %1, %4, vcc\n v_cndmask_b32_e64 %2, %2, %4, vcc\n v_cndmask_b32_e64 %3, %3, %4, vcc"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k));)
        } else if (KIND == 26) { // compare into an SGPR pair, selects on it
            REP16(asm volatile("v_cmp_lt_f32 s[20:21], %0, %4\n v_cndmask_b32 %1, %1, %4, s[20:21]\n v_cndmask_b32 %2, %2, %4, s[20:21]\n v_cndmask_b32 %3, %3, %4, s[20:21]"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k) : "s20", "s21");)
        } else if (KIND == 27) { // compare into an SGPR pair, one select
            REP16(asm volatile("v_cmp_lt_f32 s[20:21], %0, %4\n v_cndmask_b32 %1, %1, %4, s[20:21]\n v_cmp_lt_f32 s[22:23], %2, %4\n v_cndmask_b32 %3, %3, %4, s[22:23]"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k) : "s20", "s21", "s22", "s23");)
        } else if (KIND == 28) { // compare e32 (vcc), two adds, one select
            REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_cndmask_b32 %3, %3, %4, vcc"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k) : "vcc");)
        } else if (KIND == 29) { // VOP2 with inline constant
            REP16(asm volatile("v_add_f32 %0, 1.0, %0\n v_add_f32 %1, 2.0, %1\n v_add_f32 %2, 0.5, %2\n v_add_f32 %3, 4.0, %3"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (KIND == 30) { // VOP3-encoded add (abs modifier)
            REP16(asm volatile("v_add_f32 %0, |%0|, %4\n v_add_f32 %1, |%1|, %4\n v_add_f32 %2, |%2|, %4\n v_add_f32 %3, |%3|, %4"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k));)
        } else if (KIND == 31) { // fmac (VOP2, accumulates into dst)
            REP16(asm volatile("v_fmac_f32 %0, %4, %4\n v_fmac_f32 %1, %4, %4\n v_fmac_f32 %2, %4, %4\n v_fmac_f32 %3, %4, %4"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k));)
        } else if (KIND == 32) { // fma with two distinct sources only
            REP16(asm volatile("v_fma_f32 %0, %4, %4, %0\n v_fma_f32 %1, %4, %4, %1\n v_fma_f32 %2, %4, %4, %2\n v_fma_f32 %3, %4, %4, %3"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k));)
        } else if (KIND == 33) { // fmaak (VOP2 with literal addend)
            REP16(asm volatile("v_fmaak_f32 %0, %0, %4, 0x3f800001\n v_fmaak_f32 %1, %1, %4, 0x3f800001\n v_fmaak_f32 %2, %2, %4, 0x3f800001\n v_fmaak_f32 %3, %3, %4, 0x3f800001"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k));)
        } else if (KIND == 34) { // ds_read_b32 stream
            REP16(asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:256\n ds_read_b32 %2, %4 offset:512\n ds_read_b32 %3, %4 offset:768\n s_waitcnt lgkmcnt(0)"
                               : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"((int)threadIdx.x * 4 & 255));)
        } else if (KIND == 9) {
            REP16(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k));)
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + (float)(d0 + d1 + d2 + d3);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (s == 12345.678f) out[1] = 1;
}

template <int KIND> static void run(const char *name, unsigned long long *d) {
    for (int waves : {1, 4}) {
        const int iters = 2000;
        // one block of waves*4 waves on one CU -> `waves` per SIMD
        for (int blocks : {1}) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256 * waves), 0, 0, d, iters);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256 * waves), 0, 0, d, iters);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            const double n = (double)iters * 64; // instructions per wave
            printf("%-22s waves/SIMD %d blocks %4d: %6.2f s_memtime ticks/instr/wave (%.2f per SIMD-instr), kernel %.3f ms\n", name, waves, blocks,
                   h[0] / n, h[0] / n / waves, ms);
        }
    }
}

int main() {
    unsigned long long *d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    run<23>("cmp + 3 cndmask vcc", d); run<24>("cmp, add, 2 cndmask vcc", d); run<28>("cmp, 2 add, cndmask vcc", d);
    run<25>("v_cndmask_e64 vcc", d); run<26>("cmp s[], 3 cndmask s[]", d); run<27>("cmp s[] + cndmask s[]", d);
    run<29>("v_add_f32 inline const", d); run<30>("v_add_f32 |abs| (VOP3)", d); run<31>("v_fmac_f32", d);
    run<32>("v_fma_f32 2 srcs", d); run<33>("v_fmaak_f32", d); run<34>("ds_read_b32 x4 + wait", d);
    return 0;
    run<0>("v_add_f32", d); run<7>("v_mul_f32", d); run<3>("v_fma_f32", d); run<1>("v_pk_add_f32", d); run<2>("v_pk_mul_f32", d);
    run<6>("v_pk_fma_f32", d); run<8>("v_pk_add_f32 modifiers", d); run<4>("v_add_f64", d); run<20>("v_fma_f64", d);
    run<5>("v_cndmask_b32 vcc", d); run<19>("v_cndmask vcc (set)", d); run<10>("v_cndmask_b32 s[20:21]", d);
    run<9>("v_mov_b32", d); run<11>("v_add_f32 sgpr", d); run<12>("v_add_f32 literal", d); run<13>("v_cmp_lt_f32 vcc", d);
    run<14>("cmp+cndmask pairs", d); run<15>("v_rcp_f32", d); run<21>("v_sqrt_f32", d); run<16>("v_and_b32", d);
    run<17>("v_add_f32 dependent", d); run<18>("v_bfi_b32", d); run<22>("v_mov_b32_dpp wave_ror", d);
    return 0;
}
