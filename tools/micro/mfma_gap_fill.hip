// How much vector work hides in the gap behind a v_mfma_f32_32x32x16_f16?  The h2 pattern (3 MFMA + 2 ds_read_b128 per k-step, one
// wave per SIMD) with F plain VALU (v_fma_f32), packed (v_pk_fma_f32 / v_pk_mul_f32: two values per instruction) and T transcendental
// (v_exp_f32) instructions in EVERY MFMA gap; prints cycles per MFMA.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_gap_fill.hip -o tools/micro/mfma_gap_fill
#include <hip/hip_runtime.h>
#include <stdio.h>

#define FMA1 "v_fma_f32 v60, v60, v61, v62\n"
#define FMA2 FMA1 "v_fma_f32 v63, v63, v61, v62\n"
#define FMA3 FMA2 "v_fma_f32 v64, v64, v61, v62\n"
#define FMA4 FMA3 "v_fma_f32 v65, v65, v61, v62\n"
#define FMA5 FMA4 "v_fma_f32 v66, v66, v61, v62\n"
#define FMA6 FMA5 "v_fma_f32 v67, v67, v61, v62\n"
#define EXP1 "v_exp_f32 v68, v68\n"
#define EXP2 EXP1 "v_exp_f32 v69, v69\n"
#define EXP3 EXP2 "v_exp_f32 v70, v70\n"
#define PK1 "v_pk_fma_f32 v[72:73], v[72:73], v[74:75], v[76:77]\n"
#define PK2 PK1 "v_pk_fma_f32 v[78:79], v[78:79], v[74:75], v[76:77]\n"
#define PK3 PK2 "v_pk_fma_f32 v[80:81], v[80:81], v[74:75], v[76:77]\n"
#define PKM1 "v_pk_mul_f32 v[72:73], v[72:73], v[74:75]\n"
#define PKM2 PKM1 "v_pk_mul_f32 v[78:79], v[78:79], v[74:75]\n"
#define PKM3 PKM2 "v_pk_mul_f32 v[80:81], v[80:81], v[74:75]\n"
#define NONE ""

#define KSTEP(FILL, A0, A1, N0, N1, OFF)                                                  \
    "s_waitcnt lgkmcnt(0)\n"                                                               \
    "v_mfma_f32_32x32x16_f16 v[0:15], " A0 ", a[0:3], v[0:15]\n"                           \
    "ds_read_b128 " N0 ", %0 offset:" #OFF "\n"                                            \
    "ds_read_b128 " N1 ", %0 offset:" #OFF "+1024\n" FILL                                  \
    "v_mfma_f32_32x32x16_f16 v[16:31], " A0 ", a[4:7], v[16:31]\n" FILL                    \
    "v_mfma_f32_32x32x16_f16 v[16:31], " A1 ", a[0:3], v[16:31]\n" FILL

#define CLOB "a0","a1","a2","a3","a4","a5","a6","a7","v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81"

#define KERNEL(NAME, FILL)                                                                                     \
    __global__ __launch_bounds__(256, 1) void NAME(unsigned long long* out, int iters) {                       \
        __shared__ __attribute__((aligned(16))) char lds[65536];                                               \
        const unsigned la = (threadIdx.x & 63) * 16;                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)");                                                                  \
        unsigned long long t0 = __builtin_readcyclecounter();                                                  \
        for (int i = 0; i < iters; ++i)                                                                        \
            asm volatile(KSTEP(FILL, "v[32:35]", "v[44:47]", "v[48:51]", "v[52:55]", 0)                        \
                         KSTEP(FILL, "v[48:51]", "v[52:55]", "v[32:35]", "v[44:47]", 2048) ::"v"(la) : CLOB);  \
        asm volatile("s_nop 7\n s_nop 7");                                                                     \
        unsigned long long t1 = __builtin_readcyclecounter();                                                  \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                             \
        if (threadIdx.x == 12345) lds[threadIdx.x] = 1;                                                        \
    }

KERNEL(k_f0, NONE) KERNEL(k_f1, FMA1) KERNEL(k_f2, FMA2) KERNEL(k_f3, FMA3) KERNEL(k_f4, FMA4) KERNEL(k_f5, FMA5) KERNEL(k_f6, FMA6)
KERNEL(k_e1, EXP1) KERNEL(k_e2, EXP2) KERNEL(k_e3, EXP3)
KERNEL(k_p1, PK1) KERNEL(k_p2, PK2) KERNEL(k_p3, PK3) KERNEL(k_pm1, PKM1) KERNEL(k_pm2, PKM2) KERNEL(k_pm3, PKM3) KERNEL(k_e1p1, EXP1 PK1) KERNEL(k_e1p2, EXP1 PK2)
KERNEL(k_e1f1, EXP1 FMA1) KERNEL(k_e1f2, EXP1 FMA2) KERNEL(k_e1f3, EXP1 FMA3) KERNEL(k_e2f1, EXP2 FMA1) KERNEL(k_e2f2, EXP2 FMA2)

int main() {
    unsigned long long* d;
    (void)hipMalloc(&d, 64);
    const int iters = 20000;
    struct { const char* name; void (*fn)(unsigned long long*, int); } ks[] = {
        {"no filler", k_f0}, {"1 fma", k_f1}, {"2 fma", k_f2}, {"3 fma", k_f3}, {"4 fma", k_f4}, {"5 fma", k_f5}, {"6 fma", k_f6},
        {"1 pk_fma (2 values)", k_p1}, {"2 pk_fma (4 values)", k_p2}, {"3 pk_fma (6 values)", k_p3}, {"1 pk_mul", k_pm1}, {"2 pk_mul", k_pm2}, {"3 pk_mul", k_pm3}, {"1 exp + 1 pk_fma", k_e1p1}, {"1 exp + 2 pk_fma", k_e1p2},
        {"1 exp", k_e1}, {"2 exp", k_e2}, {"3 exp", k_e3}, {"1 exp + 1 fma", k_e1f1}, {"1 exp + 2 fma", k_e1f2}, {"1 exp + 3 fma", k_e1f3},
        {"2 exp + 1 fma", k_e2f1}, {"2 exp + 2 fma", k_e2f2}};
    for (auto& k : ks) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k.fn, dim3(256), dim3(256), 0, 0, d, iters);
            (void)hipDeviceSynchronize();
        }
        unsigned long long c = 0;
        (void)hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost);
        printf("per MFMA gap: %-16s %.2f cycles per MFMA\n", k.name, (double)c / iters / 6);
    }
    return 0;
}
