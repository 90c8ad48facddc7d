// Micro-benchmark: cycles per v_mfma_f32_32x32x16_f16 with the B operand in VGPRs or in AGPRs, C/D in VGPRs or AGPRs, and the h2
// core's accumulator pattern (acc_hi, acc_lo, acc_lo per k-step), optionally with two ds_read_b128 per three MFMAs.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_operand_rate.hip -o gpurun_out/mfma_rate && gpurun_out/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(unsigned long long* out, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    unsigned long long t0, t1;
    const unsigned la = (threadIdx.x & 63) * 16;
    asm volatile("s_waitcnt lgkmcnt(0)");
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {   // B in VGPR, C/D VGPR
            asm volatile(
                "v_mfma_f32_32x32x16_f16 v[0:15], v[32:35], v[36:39], v[0:15]\n"
                "v_mfma_f32_32x32x16_f16 v[16:31], v[32:35], v[40:43], v[16:31]\n"
                "v_mfma_f32_32x32x16_f16 v[16:31], v[44:47], v[36:39], v[16:31]\n" ::: "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31");
        } else if (MODE == 1) {   // B in AGPR, C/D VGPR (the h2 core with -amdgpu-mfma-vgpr-form=1)
            asm volatile(
                "v_mfma_f32_32x32x16_f16 v[0:15], v[32:35], a[0:3], v[0:15]\n"
                "v_mfma_f32_32x32x16_f16 v[16:31], v[32:35], a[4:7], v[16:31]\n"
                "v_mfma_f32_32x32x16_f16 v[16:31], v[44:47], a[0:3], v[16:31]\n" ::: "a0","a1","a2","a3","a4","a5","a6","a7","v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31");
        } else if (MODE == 2) {   // B in VGPR, C/D AGPR
            asm volatile(
                "v_mfma_f32_32x32x16_f16 a[0:15], v[32:35], v[36:39], a[0:15]\n"
                "v_mfma_f32_32x32x16_f16 a[16:31], v[32:35], v[40:43], a[16:31]\n"
                "v_mfma_f32_32x32x16_f16 a[16:31], v[44:47], v[36:39], a[16:31]\n" ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31");
        } else if (MODE == 3) {   // as 1 + two ds_read_b128 per k-step into the A registers of the NEXT k-step (double set)
            asm volatile(
                "s_waitcnt lgkmcnt(0)\n"
                "v_mfma_f32_32x32x16_f16 v[0:15], v[32:35], a[0:3], v[0:15]\n"
                "ds_read_b128 v[48:51], %0\n"
                "ds_read_b128 v[52:55], %0 offset:1024\n"
                "v_mfma_f32_32x32x16_f16 v[16:31], v[32:35], a[4:7], v[16:31]\n"
                "v_mfma_f32_32x32x16_f16 v[16:31], v[44:47], a[0:3], v[16:31]\n"
                "s_waitcnt lgkmcnt(0)\n"
                "v_mfma_f32_32x32x16_f16 v[0:15], v[48:51], a[0:3], v[0:15]\n"
                "ds_read_b128 v[32:35], %0 offset:2048\n"
                "ds_read_b128 v[44:47], %0 offset:3072\n"
                "v_mfma_f32_32x32x16_f16 v[16:31], v[48:51], a[4:7], v[16:31]\n"
                "v_mfma_f32_32x32x16_f16 v[16:31], v[52:55], a[0:3], v[16:31]\n" :: "v"(la) : "a0","a1","a2","a3","a4","a5","a6","a7","v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55");
        } else if (MODE == 4) {   // as 3 + six independent plain VALU per MFMA gap (the staged epilogue's load)
            asm volatile(
                "s_waitcnt lgkmcnt(0)\n"
                "v_mfma_f32_32x32x16_f16 v[0:15], v[32:35], a[0:3], v[0:15]\n"
                "ds_read_b128 v[48:51], %0\n"
                "ds_read_b128 v[52:55], %0 offset:1024\n"
                "v_fma_f32 v60, v60, v61, v62\n v_fma_f32 v63, v63, v61, v62\n v_fma_f32 v64, v64, v61, v62\n v_fma_f32 v65, v65, v61, v62\n v_fma_f32 v66, v66, v61, v62\n"
                "v_mfma_f32_32x32x16_f16 v[16:31], v[32:35], a[4:7], v[16:31]\n"
                "v_fma_f32 v60, v60, v61, v62\n v_fma_f32 v63, v63, v61, v62\n v_fma_f32 v64, v64, v61, v62\n v_fma_f32 v65, v65, v61, v62\n v_fma_f32 v66, v66, v61, v62\n v_fma_f32 v67, v67, v61, v62\n"
                "v_mfma_f32_32x32x16_f16 v[16:31], v[44:47], a[0:3], v[16:31]\n"
                "v_fma_f32 v60, v60, v61, v62\n v_fma_f32 v63, v63, v61, v62\n v_fma_f32 v64, v64, v61, v62\n v_fma_f32 v65, v65, v61, v62\n v_fma_f32 v66, v66, v61, v62\n"
                "s_waitcnt lgkmcnt(0)\n"
                "v_mfma_f32_32x32x16_f16 v[0:15], v[48:51], a[0:3], v[0:15]\n"
                "ds_read_b128 v[32:35], %0 offset:2048\n"
                "ds_read_b128 v[44:47], %0 offset:3072\n"
                "v_exp_f32 v60, v60\n v_exp_f32 v63, v63\n v_exp_f32 v64, v64\n"
                "v_mfma_f32_32x32x16_f16 v[16:31], v[48:51], a[4:7], v[16:31]\n"
                "v_exp_f32 v60, v60\n v_exp_f32 v63, v63\n v_exp_f32 v64, v64\n"
                "v_mfma_f32_32x32x16_f16 v[16:31], v[52:55], a[0:3], v[16:31]\n"
                "v_exp_f32 v60, v60\n v_exp_f32 v63, v63\n" :: "v"(la) : "a0","a1","a2","a3","a4","a5","a6","a7","v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v60","v61","v62","v63","v64","v65","v66","v67");
        }
    }
    asm volatile("s_nop 7\n s_nop 7");
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (threadIdx.x == 12345) lds[threadIdx.x] = 1;
}

int main() {
    unsigned long long* d; hipMalloc(&d, 64);
    const int iters = 20000;
    const char* names[5] = {"B vgpr, C/D vgpr", "B agpr, C/D vgpr (h2 core)", "B vgpr, C/D agpr", "h2 pattern + 2 ds_read_b128 per k-step", "  + staged-epilogue-like VALU in the gaps"};
    for (int grid : {1, 256}) {
        for (int m = 0; m < 5; ++m) {
            for (int rep = 0; rep < 2; ++rep) {
                switch (m) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, d, iters); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, d, iters); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, d, iters); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, d, iters); break;
                    default: hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, d, iters); break;
                }
                hipDeviceSynchronize();
            }
            unsigned long long c = 0; hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost);
            const int per_iter = (m >= 3) ? 6 : 3;
            printf("grid %3d  %-48s %.2f cycles per MFMA\n", grid, names[m], (double)c / iters / per_iter);
        }
    }
    return 0;
}
