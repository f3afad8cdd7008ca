// Cost of dependent instruction chains for ONE wavefront alone on its SIMD, and for 2 / 3 wavefronts sharing it (gfx950).
// Build: hipcc --offload-arch=gfx950 -O2 -o issue_bench issue_bench.hip ; run: ./issue_bench
// Each test runs REP iterations of a block of 64 identical instructions and reports s_memtime ticks per instruction
// (s_memtime ticks at the shader clock on this part) and wall-clock ns per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP 2000

#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
#define R64(x) R16(x) R16(x) R16(x) R16(x)

struct Res { unsigned long long ticks, wall; };

template <int KIND> __global__ void k(Res *out, double seed, int iseed)
{
    __shared__ double lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (double)((i * 17) & 4095);
    __syncthreads();
    { int *li = (int *)lds; for (int i = threadIdx.x; i < 1024; i += blockDim.x) li[i] = ((i * 37) & 1023) * 4; }
    __syncthreads();
    double a = seed + threadIdx.x, b = seed * 0.5, c = 1.000001;
    int ia = iseed + threadIdx.x, ib = iseed;
    unsigned long long m = 0x5555555555555555ull ^ iseed, m2 = 0x3333333333333333ull;
    const unsigned long long w0 = wall_clock64();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REP; ++r) {
        if (KIND == 0) {            // dependent v_add_f64
            R64(asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(c));)
        } else if (KIND == 1) {     // independent v_add_f64 (4 chains)
            double a1 = a + 1, a2 = a + 2, a3 = a + 3;
            R16(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(a), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));)
            a += a1 + a2 + a3;
        } else if (KIND == 2) {     // dependent v_mul_f64
            R64(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(c));)
        } else if (KIND == 3) {     // dependent v_cndmask_b32 (vcc fixed)
            asm volatile("v_cmp_gt_i32 vcc, %0, %1" :: "v"(ia), "v"(ib) : "vcc");
            R64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ia) : "v"(ib) : "vcc");)
        } else if (KIND == 4) {     // dependent v_add_u32
            R64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia) : "v"(ib));)
        } else if (KIND == 5) {     // dependent s_and_b64 / s_or_b64
            R16(asm volatile("s_and_b64 %0, %0, %1\n s_or_b64 %0, %0, %1\n s_andn2_b64 %0, %0, %1\n s_xor_b64 %0, %0, %1" : "+s"(m) : "s"(m2) : "scc");)
        } else if (KIND == 6) {     // independent SALU (4 chains)
            unsigned long long m3 = m + 1, m4 = m + 2, m5 = m + 3;
            R16(asm volatile("s_and_b64 %0, %0, %4\n s_or_b64 %1, %1, %4\n s_andn2_b64 %2, %2, %4\n s_xor_b64 %3, %3, %4" : "+s"(m), "+s"(m3), "+s"(m4), "+s"(m5) : "s"(m2) : "scc");)
            m ^= m3 ^ m4 ^ m5;
        } else if (KIND == 7) {     // v_cmp -> s_and_saveexec -> restore (the divergent-if skeleton, no branch)
            R16(asm volatile("v_cmp_gt_i32 vcc, %0, %1\n s_and_saveexec_b64 s[40:41], vcc\n v_add_u32 %0, %0, %1\n s_or_b64 exec, exec, s[40:41]" : "+v"(ia) : "v"(ib) : "vcc", "s40", "s41", "scc");)
        } else if (KIND == 8) {     // the same with a never-taken s_cbranch_execz
            R16(asm volatile("v_cmp_ge_i32 vcc, %0, %0\n s_and_saveexec_b64 s[40:41], vcc\n s_cbranch_execz 1f\n v_add_u32 %0, %0, %1\n 1: s_or_b64 exec, exec, s[40:41]" : "+v"(ia) : "v"(ib) : "vcc", "s40", "s41", "scc");)
        } else if (KIND == 9) {     // always-taken s_cbranch_execz (exec = 0 inside)
            R16(asm volatile("v_cmp_lt_i32 vcc, %0, %0\n s_and_saveexec_b64 s[40:41], vcc\n s_cbranch_execz 1f\n v_add_u32 %0, %0, %1\n 1: s_or_b64 exec, exec, s[40:41]" : "+v"(ia) : "v"(ib) : "vcc", "s40", "s41", "scc");)
        } else if (KIND == 10) {    // dependent LDS read chain (pointer chase, b64)
            int idx = ia & 4095;
            for (int q = 0; q < 64; ++q) idx = (int)lds[idx] & 4095;
            ia += idx;
        } else if (KIND == 11) {    // v_cmp_f64 -> vcc -> v_cndmask (VALU -> VCC -> VALU)
            R16(asm volatile("v_cmp_gt_f64 vcc, %0, %2\n v_cndmask_b32 %1, %1, %3, vcc\n v_cmp_lt_f64 vcc, %0, %2\n v_cndmask_b32 %1, %1, %3, vcc" : "+v"(a), "+v"(ia) : "v"(c), "v"(ib) : "vcc");)
        } else if (KIND == 12) {    // v_cmp writes SGPR pair, SALU reads it (VALU -> SALU)
            R16(asm volatile("v_cmp_gt_i32 s[40:41], %1, %2\n s_and_b64 %0, %0, s[40:41]\n v_cmp_lt_i32 s[40:41], %1, %2\n s_or_b64 %0, %0, s[40:41]" : "+s"(m) : "v"(ia), "v"(ib) : "s40", "s41", "scc");)
        } else if (KIND == 13) {    // SALU writes SGPR, VALU reads it as mask (SALU -> VALU)
            R16(asm volatile("s_xor_b64 %0, %0, %2\n v_cndmask_b32 %1, %1, %3, %0\n s_xor_b64 %0, %0, %2\n v_cndmask_b32 %1, %1, %3, %0" : "+s"(m), "+v"(ia) : "s"(m2), "v"(ib) : "scc");)
        } else if (KIND == 14) {    // alternating independent VALU / SALU
            R16(asm volatile("v_add_u32 %0, %0, %2\n s_and_b64 %1, %1, %3\n v_add_u32 %0, %0, %2\n s_or_b64 %1, %1, %3" : "+v"(ia), "+s"(m) : "v"(ib), "s"(m2) : "scc");)
        } else if (KIND == 15) {    // v_mul_lo_u32 dependent
            R64(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(ia) : "v"(ib));)
        } else if (KIND == 16) {    // v_mov_b64 dependent
            R64(asm volatile("v_mov_b64 %0, %0" : "+v"(a));)
        } else if (KIND == 17) {    // dependent v_cndmask_b32, VOP3 with an SGPR-pair mask
            R64(asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(ia) : "v"(ib), "s"(m2));)
        } else if (KIND == 18) {    // 4 independent v_cndmask_b32 (vcc)
            int i1 = ia + 1, i2 = ia + 2, i3 = ia + 3;
            asm volatile("v_cmp_gt_i32 vcc, %0, %1" :: "v"(ia), "v"(ib) : "vcc");
            R16(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(ia), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(ib) : "vcc");)
            ia += i1 + i2 + i3;
        } else if (KIND == 19) {    // dependent v_bfi_b32 (bitwise select by a VGPR mask)
            R64(asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(ia) : "v"(ib), "v"(iseed));)
        } else if (KIND == 20) {    // dependent v_and_or_b32
            R64(asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(ia) : "v"(ib), "v"(iseed));)
        } else if (KIND == 21) {    // if-skeleton with v_cmpx: s_mov save, v_cmpx, body, s_mov restore (4 instr)
            R16(asm volatile("s_mov_b64 s[40:41], exec\n v_cmpx_ge_i32 %0, %0\n v_add_u32 %0, %0, %1\n s_mov_b64 exec, s[40:41]" : "+v"(ia) : "v"(ib) : "vcc", "s40", "s41", "scc");)
        } else if (KIND == 22) {    // s_cbranch_scc0 never taken after s_cmp (uniform branch, 3 instr)
            R16(asm volatile("s_cmp_eq_u32 %1, %1\n s_cbranch_scc0 1f\n v_add_u32 %0, %0, %2\n 1:" : "+v"(ia) : "s"(iseed), "v"(ib) : "scc");)
        } else if (KIND == 23) {    // s_cbranch_vccz never taken after v_cmp (3 instr)
            R16(asm volatile("v_cmp_ge_i32 vcc, %0, %0\n s_cbranch_vccz 1f\n v_add_u32 %0, %0, %1\n 1:" : "+v"(ia) : "v"(ib) : "vcc");)
        } else if (KIND == 24) {    // dependent ds_read_b32 chain
            int idx = (ia & 1023) * 4;
            R64(asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0xffc, %0" : "+v"(idx) :: "memory");)
            ia += idx;
        } else if (KIND == 25) {    // v_cmp_f64 -> s_and_saveexec -> v_add_f64 -> s_or exec: double compare skeleton
            R16(asm volatile("v_cmp_gt_f64 vcc, %0, %1\n s_and_saveexec_b64 s[40:41], vcc\n v_add_f64 %0, %0, %1\n s_or_b64 exec, exec, s[40:41]" : "+v"(a) : "v"(c) : "vcc", "s40", "s41", "scc");)
        } else if (KIND == 27) {    // what the compiler emits for a select of a double: v_cmp vcc, then two VOP2 v_cndmask on vcc (3 instr)
            int i1 = ia + 1;
            R16(asm volatile("v_cmp_gt_i32 vcc, %2, %3\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc" : "+v"(ia), "+v"(i1) : "v"(ib), "v"(iseed) : "vcc");)
            ia += i1;
        } else if (KIND == 28) {    // the same through an SGPR pair and the VOP3 form (3 instr)
            int i1 = ia + 1;
            R16(asm volatile("v_cmp_gt_i32 s[40:41], %2, %3\n v_cndmask_b32_e64 %0, %0, %2, s[40:41]\n v_cndmask_b32_e64 %1, %1, %2, s[40:41]" : "+v"(ia), "+v"(i1) : "v"(ib), "v"(iseed) : "s40", "s41");)
            ia += i1;
        } else if (KIND == 29) {    // v_cmp vcc, then FOUR VOP2 v_cndmask on vcc (5 instr)
            int i1 = ia + 1, i2 = ia + 2, i3 = ia + 3;
            R16(asm volatile("v_cmp_gt_i32 vcc, %4, %5\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(ia), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(ib), "v"(iseed) : "vcc");)
            ia += i1 + i2 + i3;
        } else if (KIND == 30) {    // ... FOUR VOP3 v_cndmask on an SGPR pair (5 instr)
            int i1 = ia + 1, i2 = ia + 2, i3 = ia + 3;
            R16(asm volatile("v_cmp_gt_i32 s[40:41], %4, %5\n v_cndmask_b32_e64 %0, %0, %4, s[40:41]\n v_cndmask_b32_e64 %1, %1, %4, s[40:41]\n v_cndmask_b32_e64 %2, %2, %4, s[40:41]\n v_cndmask_b32_e64 %3, %3, %4, s[40:41]" : "+v"(ia), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(ib), "v"(iseed) : "s40", "s41");)
            ia += i1 + i2 + i3;
        } else if (KIND == 31) {    // vcc written by the SALU, then VOP2 v_cndmask x2 (3 instr)
            int i1 = ia + 1;
            R16(asm volatile("s_mov_b64 vcc, %2\n v_cndmask_b32 %0, %0, %3, vcc\n v_cndmask_b32 %1, %1, %3, vcc" : "+v"(ia), "+v"(i1) : "s"(m2), "v"(ib) : "vcc");)
            ia += i1;
        } else if (KIND == 32) {    // the select chain of v_pick<4> as compiled: 3 compares, 6 selects (lo / hi), 9 instr
            int lo = ia, hi = ia + 1;
            R16(asm volatile("v_cmp_eq_u32 vcc, 1, %2\n v_cmp_eq_u32 s[40:41], 2, %2\n v_cmp_eq_u32 s[42:43], 3, %2\n"
                             "v_cndmask_b32 %0, %0, %3, vcc\n v_cndmask_b32 %1, %1, %3, vcc\n"
                             "v_cndmask_b32_e64 %0, %0, %4, s[40:41]\n v_cndmask_b32_e64 %1, %1, %4, s[40:41]\n"
                             "v_cndmask_b32_e64 %0, %0, %3, s[42:43]\n v_cndmask_b32_e64 %1, %1, %3, s[42:43]"
                             : "+v"(lo), "+v"(hi) : "v"(ib), "v"(iseed), "v"(ia) : "vcc", "s40", "s41", "s42", "s43");)
            ia += lo + hi;
        } else if (KIND == 33) {    // v_cmp (VOP2 encoding) writes vcc, two VOP3-encoded v_cndmask read vcc (3 instr)
            int i1 = ia + 1;
            R16(asm volatile("v_cmp_gt_i32 vcc, %2, %3\n v_cndmask_b32_e64 %0, %0, %2, vcc\n v_cndmask_b32_e64 %1, %1, %2, vcc" : "+v"(ia), "+v"(i1) : "v"(ib), "v"(iseed) : "vcc");)
            ia += i1;
        } else if (KIND == 34) {    // v_cmp (VOP3 encoding) writes vcc, two VOP2 v_cndmask read vcc (3 instr)
            int i1 = ia + 1;
            R16(asm volatile("v_cmp_gt_i32_e64 vcc, %2, %3\n v_cndmask_b32_e32 %0, %0, %2, vcc\n v_cndmask_b32_e32 %1, %1, %2, vcc" : "+v"(ia), "+v"(i1) : "v"(ib), "v"(iseed) : "vcc");)
            ia += i1;
        } else if (KIND == 35) {    // one VOP2 v_cndmask per v_cmp (2 instr)
            R16(asm volatile("v_cmp_gt_i32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_i32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ia) : "v"(ib), "v"(iseed) : "vcc");)
        } else if (KIND == 26) {    // predicated alternative: v_cmp_f64 -> v_add_f64 tmp -> v_cndmask x2 (4 instr, no exec change)
            double tmp;
            R16(asm volatile("v_cmp_gt_f64 vcc, %0, %2\n v_add_f64 %1, %0, %2\n v_cndmask_b32 %L0, %L0, %L1, vcc\n v_cndmask_b32 %H0, %H0, %H1, vcc" : "+v"(a), "=&v"(tmp) : "v"(c) : "vcc");)
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    if (threadIdx.x % 64 == 0) {
        out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = { t1 - t0, w1 - w0 };
    }
    if (a == 123.456 && ia == 77 && m == 99) out[0].ticks = 0;     // keep the chains alive
}

template <int KIND> void run(const char *name, Res *d)
{
    const int per[3] = { 256, 512, 768 };       // 1, 2, 3 wavefronts per SIMD (one workgroup on one CU)
    printf("%-48s", name);
    for (int c = 0; c < 3; ++c) {
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(per[c]), 0, 0, d, 1.5, 3);
        hipDeviceSynchronize();
        std::vector<Res> h(per[c] / 64);
        hipMemcpy(h.data(), d, h.size() * sizeof(Res), hipMemcpyDeviceToHost);
        double t = 0, w = 0;
        for (auto &r : h) { t += r.ticks; w += r.wall; }
        t /= h.size(); w /= h.size();
        printf("  %dw: %6.2f ticks %6.2f ns", c + 1, t / (REP * 64.0), w * 10.0 / (REP * 64.0));
    }
    printf("\n");
}

int main()
{
    Res *d;
    hipMalloc(&d, 64 * sizeof(Res));
    // warm the clocks
    for (int i = 0; i < 20; ++i) { hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, d, 1.5, 3); }
    hipDeviceSynchronize();
    printf("per instruction, one workgroup on one CU with 1 / 2 / 3 wavefronts per SIMD:\n");
    run<0>("dependent v_add_f64", d);
    run<1>("4 independent v_add_f64 chains", d);
    run<2>("dependent v_mul_f64", d);
    run<3>("dependent v_cndmask_b32", d);
    run<4>("dependent v_add_u32", d);
    run<15>("dependent v_mul_lo_u32", d);
    run<16>("dependent v_mov_b64", d);
    run<5>("dependent s_and/or/andn2/xor_b64", d);
    run<6>("4 independent SALU chains", d);
    run<14>("alternating independent VALU / SALU", d);
    run<7>("v_cmp, s_and_saveexec, v_add, s_or exec (4 instr)", d);
    run<8>("  + s_cbranch_execz never taken (5 instr)", d);
    run<9>("  + s_cbranch_execz always taken (4 executed)", d);
    run<11>("v_cmp_f64 -> vcc -> v_cndmask", d);
    run<12>("v_cmp -> sgpr -> s_and (VALU->SALU)", d);
    run<13>("s_xor -> sgpr -> v_cndmask (SALU->VALU)", d);
    run<17>("dependent v_cndmask_b32_e64 (sgpr mask)", d);
    run<18>("4 independent v_cndmask_b32 (vcc)", d);
    run<27>("v_cmp vcc + 2 VOP2 cndmask(vcc) (3 instr)", d);
    run<28>("v_cmp sgpr + 2 VOP3 cndmask(sgpr) (3 instr)", d);
    run<33>("v_cmp vcc + 2 VOP3-encoded cndmask(vcc) (3 instr)", d);
    run<34>("v_cmp_e64 vcc + 2 VOP2 cndmask(vcc) (3 instr)", d);
    run<35>("v_cmp vcc + 1 VOP2 cndmask(vcc) (2 instr)", d);
    run<29>("v_cmp vcc + 4 VOP2 cndmask(vcc) (5 instr)", d);
    run<30>("v_cmp sgpr + 4 VOP3 cndmask(sgpr) (5 instr)", d);
    run<31>("s_mov vcc + 2 VOP2 cndmask(vcc) (3 instr)", d);
    run<32>("v_pick<4> of a double as compiled (9 instr)", d);
    run<19>("dependent v_bfi_b32", d);
    run<20>("dependent v_and_or_b32", d);
    run<21>("s_mov save, v_cmpx, v_add, s_mov exec (4 instr)", d);
    run<22>("s_cmp, s_cbranch_scc0 not taken, v_add (3 instr)", d);
    run<23>("v_cmp, s_cbranch_vccz not taken, v_add (3 instr)", d);
    run<25>("v_cmp_f64, saveexec, v_add_f64, s_or exec (4)", d);
    run<24>("ds_read_b32 + waitcnt + v_and chain (3 instr)", d);
    return 0;
}
