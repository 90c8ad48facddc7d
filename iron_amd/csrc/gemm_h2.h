// Hand-written GEMM of the backward passes (libiron_train.so): fp32 in, fp32 out, fp32-accurate products on the f16 matrix pipe.
//
// The layer products of the closed-form backward (Z = X W^T, dX = dZ W, dW = dZ^T X; K up to the number of points) ran on
// rocBLAS SGEMM in round 1 (v_mfma_f32_32x32x2_f32: 1/16 of the f16 rate).  Here every fp32 operand element is split into two
// fp16 pieces as in the inference core (mlp_h2.h), x = xh + xl * 2^-11, and a product is three v_mfma_f32_32x32x16_f16
// (xh*yh into one fp32 accumulator, xh*yl + xl*yh into a second; the dropped xl*yl is 2^-22 relative): 5.3x the matrix rate
// of the fp32 MFMA at ~fp32 accuracy.  fp16 has a short exponent, so an operand may carry a power-of-two scale taken from
// its absolute maximum (gradients span 1e-8 .. 1e+3; k_absmax below), undone in the epilogue.
//
// One workgroup = 4 waves = a 128 x 128 tile of C (each wave 64 x 64 = 2 x 2 MFMA tiles, 4 x 2 x 16 accumulator registers);
// K advances 32 at a time through a double-buffered LDS stage that holds both operands already split and already in MFMA
// FRAGMENT ORDER (fragment = 32 rows x 16 k x fp16 = 1 KiB, lane l's 16 bytes at l * 16: conflict-free ds_read_b128, no
// transposes on the read side).  The global -> register -> split -> LDS path of tile k+1 is issued before the MFMAs of tile k.
// Operands are addressed as "row r of the operand, k" with either k contiguous in memory (X, W in Z = X W^T) or k strided
// (both operands of dW = dZ^T X, where k is the point index): two loader shapes, same LDS image.
// blockIdx.z splits K (dW: K = 65 536+ points, one or two output tiles): partial tiles go to a [splits, M*N] buffer that
// k_reduce_partials sums in a fixed order (deterministic; no float atomics).
#pragma once
#include <hip/hip_runtime.h>

namespace iron_train {

typedef _Float16 g_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 g_half2 __attribute__((ext_vector_type(2)));
typedef float g_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int g_u32x4 __attribute__((ext_vector_type(4)));

constexpr int kGemmTM = 128, kGemmTN = 128, kGemmTK = 32;
constexpr int kGemmOperandBytes = 4 * 2 * 2 * 1024;         // 4 row tiles x 2 k-steps x {hi, lo} x 1 KiB
constexpr int kGemmStageBytes = 2 * kGemmOperandBytes;      // A and B
constexpr float kGemmLoScale = 2048.0f, kGemmLoInv = 1.0f / 2048.0f;

struct GemmArgs {
    const float* A;   // operand indexed (m, k)
    const float* B;   // operand indexed (n, k)
    float* C;         // [M, N] row-major, or the partial buffer [splits, M * N] when splits > 1
    int lda, ldb, ldc;
    int M, N, K;
    int k_per_split;  // multiple of 32
    const float* a_absmax;  // device scalar (|A|_max) or null: A is used as it is
    float beta;       // C = beta * C + A B (splits == 1 only)
};

// |x|_max of a buffer into *out (which the caller zeroes): float bits of non-negative values order like integers
__global__ void k_absmax(const float* __restrict__ x, int64_t count, float* __restrict__ out) {
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(m));
}

// power of two that brings `amax` to [2^13, 2^14): far from fp16's 65504 for 32-deep partial sums' operands, and 2^-28 of it
// is still a normal fp16 number in the low piece
__device__ __forceinline__ float gemm_scale_for(float amax) {
    if (!(amax > 0.0f) || !isfinite(amax)) return 1.0f;
    int e;
    (void)frexpf(amax, &e);  // amax = f * 2^e, f in [0.5, 1)
    return ldexpf(1.0f, 14 - e);
}

// Range of the split: an operand element beyond fp16's largest number (65 504, after the gradient operand's power-of-two scale)
// or a non-finite one has no fp16 high piece -- the product would come out inf / NaN where the reference's fp32 arithmetic has
// an ordinary number.  Every kernel that splits an operand raises this sticky flag (one comparison per element, off the HBM-bound
// path); iron_train_numeric_status reads it.
__device__ int g_gemm_range_flag = 0;

// 8 fp32 -> 8 fp16 high pieces + 8 fp16 low pieces (both round to nearest; x - f32(hi) is exact); bad |= any element out of range
__device__ __forceinline__ void gemm_split8(const float* v, g_u32x4& hi, g_u32x4& lo, unsigned& bad) {
#pragma unroll
    for (int i = 0; i < 8; ++i) bad |= !(fabsf(v[i]) <= 65504.0f) ? 1u : 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        g_half2 h, l;
        h[0] = (_Float16)v[2 * i];
        h[1] = (_Float16)v[2 * i + 1];
        l[0] = (_Float16)((v[2 * i] - (float)h[0]) * kGemmLoScale);
        l[1] = (_Float16)((v[2 * i + 1] - (float)h[1]) * kGemmLoScale);
        hi[i] = __builtin_bit_cast(unsigned, h);
        lo[i] = __builtin_bit_cast(unsigned, l);
    }
}

// One operand's 128 x 32 slice of a K tile = 512 segments of 8 consecutive k for one row; 128 threads load it, 4 segments each
// (threads 0..127 take operand A, threads 128..255 operand B):
//   KSTRIDED = false: memory is [row][k].  thread w -> row w, its four k-chunks: 32 contiguous floats (eight dwordx4 when aligned)
//   KSTRIDED = true : memory is [k][row].  thread w -> rows 4 (w % 32) .. + 3, k-chunk w / 32: eight dwordx4, one per k, each
//                     holding the 4 rows -- a wave's lanes cover 512 contiguous bytes of every k row -- transposed in registers
//                     into the four 8-k segments
template <bool KSTRIDED>
struct GemmLoader {
    const float* base;
    int ld, rows, row0;
    bool vec_ok;  // ld % 4 == 0 and the base 16-byte aligned

    // v[q][i]: segment q of this thread, its 8 k values, times `scale`; zero outside the operand or beyond k_end
    __device__ __forceinline__ void load(int w, int kbase, int k_end, float scale, float (&v)[4][8]) const {
        if (KSTRIDED) {
            const int r4 = row0 + 4 * (w & 31), k0 = kbase + 8 * (w >> 5);
            if (vec_ok && r4 + 4 <= rows && k0 + 8 <= k_end) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float4 t = *reinterpret_cast<const float4*>(base + (size_t)(k0 + i) * ld + r4);
                    v[0][i] = t.x * scale; v[1][i] = t.y * scale; v[2][i] = t.z * scale; v[3][i] = t.w * scale;
                }
            } else if (vec_ok) {   // aligned operand, ragged edge of the matrix
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q][i] = (r4 + q < rows && k0 + i < k_end) ? base[(size_t)(k0 + i) * ld + r4 + q] * scale : 0.0f;
            } else {               // rows not 16-byte aligned (ld = 39, 217, 257, ...): thread w -> row w, segment q = k-chunk q; every
                                   // dword load of a wave covers 64 consecutive rows of one k (256 contiguous bytes)
                const int gr = row0 + w;
                const float* p = base + (size_t)kbase * ld + gr;
                if (gr < rows && kbase + 32 <= k_end) {     // the common case: no per-element checks, one running pointer
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[q][i] = p[(size_t)(8 * q + i) * ld] * scale;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[q][i] = (gr < rows && kbase + 8 * q + i < k_end) ? p[(size_t)(8 * q + i) * ld] * scale : 0.0f;
                }
            }
        } else {
            const int gr = row0 + w;
            const float* p = base + (size_t)gr * ld + kbase;
            if (vec_ok && gr < rows && kbase + 32 <= k_end) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 a = *reinterpret_cast<const float4*>(p + 8 * q), b = *reinterpret_cast<const float4*>(p + 8 * q + 4);
                    v[q][0] = a.x * scale; v[q][1] = a.y * scale; v[q][2] = a.z * scale; v[q][3] = a.w * scale;
                    v[q][4] = b.x * scale; v[q][5] = b.y * scale; v[q][6] = b.z * scale; v[q][7] = b.w * scale;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[q][i] = (gr < rows && kbase + 8 * q + i < k_end) ? p[8 * q + i] * scale : 0.0f;
            }
        }
    }
    // (row within the tile, k-chunk) of this thread's segment q
    __device__ __forceinline__ void seg_of(int w, int q, int& r, int& c) const {
        if (KSTRIDED && vec_ok) { r = 4 * (w & 31) + q; c = w >> 5; } else { r = w; c = q; }
    }
    // LDS byte offset of a segment inside an operand image: fragment (row tile, k-step, piece) x 1 KiB, lane (k-half, row % 32) x 16 B
    __device__ __forceinline__ int lds_offset(int w, int q) const {
        int r, c;
        seg_of(w, q, r, c);
        return (((r >> 5) * 2 + (c >> 1)) * 2) * 1024 + (((c & 1) * 32 + (r & 31)) * 16);
    }
};

template <bool A_KSTRIDED, bool B_KSTRIDED>
__global__ __launch_bounds__(256, 2) void k_gemm_split_f16(GemmArgs g) {   // <= 256 registers: two workgroups per CU cover each other's load phases
    extern __shared__ __attribute__((aligned(16))) char g_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;  // the wave's 64 x 64 quadrant of the tile
    const int m0 = blockIdx.x * kGemmTM, n0 = blockIdx.y * kGemmTN;
    const int k_begin = blockIdx.z * g.k_per_split;
    const int k_end = min(g.K, k_begin + g.k_per_split);
    const float a_scale = g.a_absmax ? gemm_scale_for(*g.a_absmax) : 1.0f;
    const bool loads_a = tid < 128;           // waves 0, 1 stage operand A, waves 2, 3 operand B
    const int w = tid & 127;

    GemmLoader<A_KSTRIDED> la;
    la.base = g.A; la.ld = g.lda; la.rows = g.M; la.row0 = m0;
    la.vec_ok = (g.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0);
    GemmLoader<B_KSTRIDED> lb;
    lb.base = g.B; lb.ld = g.ldb; lb.rows = g.N; lb.row0 = n0;
    lb.vec_ok = (g.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.B) & 15) == 0);

    g_f32x16 acc_hi[2][2], acc_lo[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc_hi[i][j][r] = 0.0f; acc_lo[i][j][r] = 0.0f; }

    float v[4][8];
    unsigned bad = 0;
    auto fetch = [&](int kbase) {
        if (loads_a) la.load(w, kbase, k_end, a_scale, v);
        else lb.load(w, kbase, k_end, 1.0f, v);
    };
    auto stage = [&](char* buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            g_u32x4 hi, lo;
            gemm_split8(v[q], hi, lo, bad);
            const int off = loads_a ? la.lds_offset(w, q) : kGemmOperandBytes + lb.lds_offset(w, q);
            *reinterpret_cast<g_u32x4*>(buf + off) = hi;
            *reinterpret_cast<g_u32x4*>(buf + off + 1024) = lo;
        }
    };

    int cur = 0;
    if (k_begin < k_end) {
        fetch(k_begin);
        stage(g_lds);
    }
    __syncthreads();
    for (int kb = k_begin; kb < k_end; kb += kGemmTK) {
        const bool more = kb + kGemmTK < k_end;
        if (more) fetch(kb + kGemmTK);  // global loads of the next slice fly under this slice's MFMAs
        const char* buf = g_lds + cur * kGemmStageBytes;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            g_half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int fa = (((wm * 2 + i) * 2 + ks) * 2) * 1024 + lane * 16;
                ah[i] = *reinterpret_cast<const g_half8*>(buf + fa);
                al[i] = *reinterpret_cast<const g_half8*>(buf + fa + 1024);
                const int fb = kGemmOperandBytes + (((wn * 2 + i) * 2 + ks) * 2) * 1024 + lane * 16;
                bh[i] = *reinterpret_cast<const g_half8*>(buf + fb);
                bl[i] = *reinterpret_cast<const g_half8*>(buf + fb + 1024);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc_lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc_lo[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc_hi[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc_hi[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc_lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc_lo[i][j], 0, 0, 0);
        }
        if (more) stage(g_lds + (cur ^ 1) * kGemmStageBytes);  // the other buffer: nobody reads it during this slice
        __syncthreads();
        cur ^= 1;
    }

    if (bad) atomicOr(&g_gemm_range_flag, 1);
    // epilogue: C/D layout of the 32x32 MFMA: lane l holds column n = l % 32, rows (r % 4) + 8 (r / 4) + 4 (l / 32)
    const float unscale = 1.0f / a_scale;
    float* C = g.C + (gridDim.z > 1 ? (size_t)blockIdx.z * g.M * g.N : 0);
    const int ldc = gridDim.z > 1 ? g.N : g.ldc;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + (wn * 2 + j) * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < g.M && n < g.N) {
                    const float v2 = fmaf(acc_lo[i][j][r], kGemmLoInv, acc_hi[i][j][r]) * unscale;
                    float* dst = C + (size_t)m * ldc + n;
                    *dst = (gridDim.z == 1 && g.beta != 0.0f) ? fmaf(g.beta, *dst, v2) : v2;
                }
            }
        }
}

// launch helper: a_kstrided / b_kstrided select the loader shapes; splits > 1 writes partial tiles to `C` = [splits, M*N]
template <bool AKS, bool BKS>
static inline hipError_t gemm_split_launch(const GemmArgs& g, int splits, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)k_gemm_split_f16<AKS, BKS>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kGemmStageBytes);
        attr = true;
    }
    const dim3 grid((g.M + kGemmTM - 1) / kGemmTM, (g.N + kGemmTN - 1) / kGemmTN, splits);
    hipLaunchKernelGGL((k_gemm_split_f16<AKS, BKS>), grid, dim3(256), 2 * kGemmStageBytes, st, g);
    return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------------------------------------------
// "Row" GEMM: C[R, N] = A[R, K] B, R = 10^5 rows, N and K <= 384 (one layer of an MLP applied to a batch, forward or backward).
// The minimum HBM traffic of such a product is A once in and C once out (K = 256: 43 FLOP per byte, below the split-fp16 ridge), so
// the kernel is built around exactly that: a workgroup owns 64 complete rows, splits them once into the LDS fragment image (all of
// K: <= 24 k-steps x 2 row tiles x {hi, lo} x 1 KiB = 96 KiB), and its four waves each produce a 64 x (N / 4) strip of C from it.
// The small operand B (the layer's weight matrix) is split ONCE per GEMM by k_gemm_pack_b into fragments in global memory
// ([col tile][k-step][hi, lo] x 1 KiB); every wave streams its own column tiles' fragments straight from L2 into registers -- each
// B byte is used by exactly one wave of the workgroup, so LDS would add nothing.
// Where its time goes (round 3, 131 072 x 256 x 256 on one box: 103 us = 2.6 TB/s of A in + C out; a device copy of the same bytes
// takes 37 us = 7.2 TB/s): with the B fragments served from L1 94 us, without the C stores 80, without the A loads 76, without all
// three 49 -- the phases of a workgroup (rows -> split -> LDS | MFMAs, B three k-steps deep from L2 | stores) run one after the other
// and two workgroups per CU cover each other only in part.  Neither a deeper loader (8 segments per thread: 105), a larger or
// smaller grid (256: 130, 512 / 2048: 101-103), the vgpr-form option (106), eight waves of half the accumulators (spills; C3 slower),
// nor a weight-stationary form (one workgroup per CU keeping all B fragments in 256 registers, rows by LDS-DMA under the previous
// block's MFMAs, split pass from LDS: 133 us -- split pass 52, DMA issue 25, epilogue 22 of it, nothing to cover them at four waves
// per CU) beat it; what would is a producer / consumer split of the workgroup's waves, not built.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int kRowsBM = 64;
constexpr int kRowsMaxKSteps = 24;   // K <= 384
constexpr int kRowsLdsBytes = 2 * kRowsMaxKSteps * 2 * 1024;

struct PackBArgs {
    const float* B;
    int ldb, N, K;        // operand indexed (n, k)
    int k_strided;        // 0: memory [n][k]; 1: memory [k][n]
    int n_tiles, k_steps;
    char* out;            // [n_tiles][k_steps][2] x 1 KiB
};

__global__ void k_gemm_pack_b(PackBArgs p) {
    const int frag = blockIdx.x;                 // (n tile, k-step)
    const int nt = frag / p.k_steps, ks = frag % p.k_steps;
    const int lane = threadIdx.x;                // 64 threads
    const int n = nt * 32 + (lane & 31), k0 = ks * 16 + 8 * (lane >> 5);
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = k0 + i;
        v[i] = (n < p.N && k < p.K) ? (p.k_strided ? p.B[(size_t)k * p.ldb + n] : p.B[(size_t)n * p.ldb + k]) : 0.0f;
    }
    g_u32x4 hi, lo;
    unsigned bad = 0;
    gemm_split8(v, hi, lo, bad);
    char* dst = p.out + (size_t)frag * 2048 + lane * 16;
    *reinterpret_cast<g_u32x4*>(dst) = hi;
    *reinterpret_cast<g_u32x4*>(dst + 1024) = lo;
    if (bad) atomicOr(&g_gemm_range_flag, 1);
}

// Fused epilogues of the row kernel (train.hip: the layer-wise backward of the SDF and material networks).  The activation and the
// activation-gradient passes over a layer's [R, 256] arrays are HBM-bound kernels of their own otherwise: one more read of what the
// GEMM has just written, one more write.
//   kEpiPlain     C = A B (+ beta C)
//   kEpiSdfAct    forward recompute of an SDF layer: z = A B + bias -> Z (kept for the reverse pass), next = (softplus_100(z),
//                 sigma'(z) zdot) * sc, written straight into the next layer's input
//   kEpiSdfBack   reverse of it: the product is dL/d(next layer input); with Z of THIS layer: dZ = (s1 abar + s2 zdot adotbar,
//                 s1 adotbar), the bias gradient (column sums of the value rows) and |dZ|_max on the way
//   kEpiReluAct / kEpiReluBack   the same two for a relu layer of a material network (no tangent rows)
// PAIRED rows (SDF net under a gradient loss): the rows are [values of m points | tangents of the same m points]; a block then
// takes 32 points and puts their value rows in row tile 0 and their tangent rows in row tile 1, so that z and zdot of one
// (point, feature) sit in the same lane and register index of the two accumulator tiles.
enum { kEpiPlain = 0, kEpiSdfAct = 1, kEpiSdfBack = 2, kEpiReluAct = 3, kEpiReluBack = 4 };

struct RowsEpi {
    int paired;            // rows are [m_pts values | m_pts tangents]
    int m_pts;
    int n_act;             // columns [0, n_act) carry the activation (the rest of the product is ignored / plain)
    float sc;
    const float* bias;     // Act: [n_act]
    float* Z;              // Act: out [R, ldz];  Back: in
    int ldz;
    float* out;            // Act: next layer's input [R, ld_out];  Back: dZ [R, ld_out]
    int ld_out;
    float* db;             // Back: [n_act] bias gradient (atomicAdd)
    float* amax_out;       // Back: |dZ|_max (atomicMax on the bit pattern), or null
};

struct RowsArgs {
    const float* A;        // [R, K], row stride lda
    const char* Bp;        // packed fragments (k_gemm_pack_b)
    float* C;              // [R, N], row stride ldc
    int lda, ldc, R, N, K, k_steps, n_tiles;
    const float* a_absmax; // device scalar or null
    float beta;
    RowsEpi e;
};

// F.softplus(beta = 100, threshold = 20) with its first two derivatives (models/fields.py:80)
__device__ __forceinline__ void softplus100(float z, float* a, float* s1, float* s2) {
    const float bz = 100.0f * z;
    if (bz > 20.0f) { *a = z; *s1 = 1.0f; *s2 = 0.0f; return; }
    const float e = expf(bz);
    *a = log1pf(e) / 100.0f;
    *s1 = e / (e + 1.0f);
    *s2 = 100.0f * (*s1) * (1.0f - *s1);
}

// The same on the hardware exp2 / log2 / rcp (1 ulp each), for the fused epilogues: there the activation runs at the row kernel's two
// waves per SIMD, and expf / log1pf (~60 instructions per element) would cost more than the tile's MFMAs.  u = exp(-|100 z|) never
// overflows; above the reference's threshold (100 z > 20) u < 2^-24, so 1 + u == 1 and (a, s1, s2) = (z, 1, 0) exactly as there.
__device__ __forceinline__ void softplus100_fast(float z, float* a, float* s1, float* s2) {
    const float u = __builtin_amdgcn_exp2f(fabsf(z) * -144.26950408889634f);
    const float w = 1.0f + u;
    const float r = __builtin_amdgcn_rcpf(w);
    *a = fmaf(__builtin_amdgcn_logf(w), 0.0069314718055994531f, fmaxf(z, 0.0f));
    const float sg = (z >= 0.0f ? 1.0f : u) * r;
    *s1 = sg;
    *s2 = 100.0f * sg * ((z >= 0.0f ? u : 1.0f) * r);   // 1 - sigmoid without cancellation
}

// wave maximum -> one atomicMax per wave on the float's bit pattern (values >= 0).  Called by EVERY lane of the wave.
__device__ __forceinline__ void publish_absmax(float m, float* out) {
    if (!out) return;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(m));
}

// global row of block row r (0..63) of block `blk`, or -1
__device__ __forceinline__ int rows_row_of(const RowsArgs& g, bool paired, int blk, int r) {
    if (!paired) { const int gr = blk * kRowsBM + r; return gr < g.R ? gr : -1; }
    const int p = blk * 32 + (r & 31);
    return p < g.e.m_pts ? (r < 32 ? p : g.e.m_pts + p) : -1;
}

// the block's accumulators -> C (plain) or the fused activation / activation-gradient outputs (see above)
template <int NTW, int EPI>
__device__ __forceinline__ void rows_epilogue(const RowsArgs& g, int blk, bool paired, float unscale, int nt0, int lane,
                                              g_f32x16 (&acc_hi)[2][NTW], g_f32x16 (&acc_lo)[2][NTW]) {
    auto row_of = [&](int b, int r) -> int { return rows_row_of(g, paired, b, r); };
    if constexpr (EPI == kEpiPlain) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const int n = (nt0 + j) * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = blk * kRowsBM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (m < g.R && n < g.N) {
                        const float v = fmaf(acc_lo[i][j][r], kGemmLoInv, acc_hi[i][j][r]) * unscale;
                        float* dst = g.C + (size_t)m * g.ldc + n;
                        *dst = g.beta != 0.0f ? fmaf(g.beta, *dst, v) : v;
                    }
                }
            }
    } else {
        const RowsEpi& e = g.e;
        float mx = 0.0f;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int n = (nt0 + j) * 32 + (lane & 31);
            const bool col_ok = n < e.n_act;
            const float bias = (col_ok && (EPI == kEpiSdfAct || EPI == kEpiReluAct)) ? e.bias[n] : 0.0f;
            float colsum = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float v0 = fmaf(acc_lo[0][j][r], kGemmLoInv, acc_hi[0][j][r]) * unscale;
                const float v1 = fmaf(acc_lo[1][j][r], kGemmLoInv, acc_hi[1][j][r]) * unscale;
                const int m0 = row_of(blk, rr), m1 = row_of(blk, 32 + rr);
                if (!col_ok) continue;
                if constexpr (EPI == kEpiSdfAct) {
                    if (paired) {   // (z, zdot) of one point
                        if (m0 >= 0) {
                            const float z = v0 + bias;
                            float a, s1, s2;
                            softplus100_fast(z, &a, &s1, &s2);
                            e.Z[(size_t)m0 * e.ldz + n] = z;
                            e.Z[(size_t)m1 * e.ldz + n] = v1;
                            e.out[(size_t)m0 * e.ld_out + n] = a * e.sc;
                            e.out[(size_t)m1 * e.ld_out + n] = s1 * v1 * e.sc;
                        }
                    } else {
                        float a, s1, s2;
                        if (m0 >= 0) { const float z = v0 + bias; softplus100_fast(z, &a, &s1, &s2); e.Z[(size_t)m0 * e.ldz + n] = z; e.out[(size_t)m0 * e.ld_out + n] = a * e.sc; }
                        if (m1 >= 0) { const float z = v1 + bias; softplus100_fast(z, &a, &s1, &s2); e.Z[(size_t)m1 * e.ldz + n] = z; e.out[(size_t)m1 * e.ld_out + n] = a * e.sc; }
                    }
                } else if constexpr (EPI == kEpiSdfBack) {
                    if (paired) {
                        if (m0 >= 0) {
                            float a, s1, s2;
                            softplus100_fast(e.Z[(size_t)m0 * e.ldz + n], &a, &s1, &s2);
                            const float abar = v0 * e.sc, adotbar = v1 * e.sc;
                            const float zd = e.Z[(size_t)m1 * e.ldz + n];
                            const float zbar = s1 * abar + s2 * zd * adotbar, tbar = s1 * adotbar;
                            e.out[(size_t)m0 * e.ld_out + n] = zbar;
                            e.out[(size_t)m1 * e.ld_out + n] = tbar;
                            colsum += zbar;
                            mx = fmaxf(mx, fmaxf(fabsf(zbar), fabsf(tbar)));
                        }
                    } else {
                        float a, s1, s2;
                        if (m0 >= 0) { softplus100_fast(e.Z[(size_t)m0 * e.ldz + n], &a, &s1, &s2); const float zb = s1 * (v0 * e.sc); e.out[(size_t)m0 * e.ld_out + n] = zb; colsum += zb; mx = fmaxf(mx, fabsf(zb)); }
                        if (m1 >= 0) { softplus100_fast(e.Z[(size_t)m1 * e.ldz + n], &a, &s1, &s2); const float zb = s1 * (v1 * e.sc); e.out[(size_t)m1 * e.ld_out + n] = zb; colsum += zb; mx = fmaxf(mx, fabsf(zb)); }
                    }
                } else if constexpr (EPI == kEpiReluAct) {
                    if (m0 >= 0) { const float z = v0 + bias; e.Z[(size_t)m0 * e.ldz + n] = z; e.out[(size_t)m0 * e.ld_out + n] = fmaxf(z, 0.0f) * e.sc; }
                    if (m1 >= 0) { const float z = v1 + bias; e.Z[(size_t)m1 * e.ldz + n] = z; e.out[(size_t)m1 * e.ld_out + n] = fmaxf(z, 0.0f) * e.sc; }
                } else {   // kEpiReluBack
                    if (m0 >= 0) { const float zb = e.Z[(size_t)m0 * e.ldz + n] > 0.0f ? v0 * e.sc : 0.0f; e.out[(size_t)m0 * e.ld_out + n] = zb; colsum += zb; mx = fmaxf(mx, fabsf(zb)); }
                    if (m1 >= 0) { const float zb = e.Z[(size_t)m1 * e.ldz + n] > 0.0f ? v1 * e.sc : 0.0f; e.out[(size_t)m1 * e.ld_out + n] = zb; colsum += zb; mx = fmaxf(mx, fabsf(zb)); }
                }
            }
            if constexpr (EPI == kEpiSdfBack || EPI == kEpiReluBack) {
                colsum += __shfl_xor(colsum, 32, 64);   // the two lane halves hold the other rows of the same column
                if (lane < 32 && col_ok && e.db) atomicAdd(&e.db[n], colsum);
            }
        }
        if constexpr (EPI == kEpiSdfBack || EPI == kEpiReluBack) publish_absmax(mx, e.amax_out);
    }
}

// NTW = column tiles (of 32) per wave: the workgroup covers 4 * NTW * 32 columns
template <int NTW, int EPI = kEpiPlain>
__global__ __launch_bounds__(256, (NTW <= 2 ? 2 : 1)) void k_gemm_rows(RowsArgs g) {   // NTW <= 2: <= 256 registers, two workgroups per CU
    extern __shared__ __attribute__((aligned(16))) char g_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float a_scale = g.a_absmax ? gemm_scale_for(*g.a_absmax) : 1.0f;
    const float unscale = 1.0f / a_scale;
    const bool vec_ok = (g.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0);
    const int k_chunks = g.k_steps * 2;                    // 8-wide chunks per row
    const int n_seg = kRowsBM * k_chunks;                  // segments of the 64-row block
    const bool paired = EPI != kEpiPlain && g.e.paired != 0;
    const int n_blocks = paired ? (g.e.m_pts + 31) / 32 : (g.R + kRowsBM - 1) / kRowsBM;
    auto row_of = [&](int blk, int r) -> int { return rows_row_of(g, paired, blk, r); };
    unsigned bad = 0;
    for (int blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        // ---- the block's rows -> split -> LDS fragment image: four segments per thread at a time, all their loads issued before
        // the first split (one HBM latency per batch instead of one per segment)
        for (int s0 = tid; s0 < n_seg; s0 += 4 * 256) {
            float v[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = s0 + 256 * u;
                const int r = s / k_chunks, c = s - r * k_chunks;   // consecutive threads: consecutive chunks of one row (coalesced)
                const int gr = s < n_seg ? row_of(blk, r) : -1, k0 = c * 8;
                const float* p = g.A + (size_t)(gr < 0 ? 0 : gr) * g.lda + k0;
                if (gr >= 0 && vec_ok && k0 + 8 <= g.K) {
                    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
                    v[u][0] = a.x; v[u][1] = a.y; v[u][2] = a.z; v[u][3] = a.w; v[u][4] = b.x; v[u][5] = b.y; v[u][6] = b.z; v[u][7] = b.w;
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[u][i] = (gr >= 0 && k0 + i < g.K) ? p[i] : 0.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = s0 + 256 * u;
                if (s >= n_seg) break;
                const int r = s / k_chunks, c = s - r * k_chunks;
#pragma unroll
                for (int i = 0; i < 8; ++i) v[u][i] *= a_scale;
                g_u32x4 hi, lo;
                gemm_split8(v[u], hi, lo, bad);
                // slot of (row, k-half) inside its 1-KiB fragment, rotated by the k-step within each half: a wave writes 32 chunks of ONE
                // row at a time (that is what keeps the global loads coalesced), i.e. the same row slot of 16 different fragments --
                // 2 KiB apart, all on the same banks; the rotation spreads those 16 over the 16 bank groups.  The reader applies the
                // same rotation (a bijection of the fragment's 64 slots, so ds_read_b128 stays conflict-free).
                const int ks = c >> 1, kh = c & 1;
                const int off = (((r >> 5) * g.k_steps + ks) * 2) * 1024 + ((kh * 32 + ((r + ks + 8 * kh) & 31)) * 16);
                *reinterpret_cast<g_u32x4*>(g_lds + off) = hi;
                *reinterpret_cast<g_u32x4*>(g_lds + off + 1024) = lo;
            }
        }
        __syncthreads();
        // ---- this wave's strip: column tiles wave * NTW .. + NTW
        g_f32x16 acc_hi[2][NTW], acc_lo[2][NTW];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NTW; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc_hi[i][j][r] = 0.0f; acc_lo[i][j][r] = 0.0f; }
        const int nt0 = wave * NTW;
        // B fragments of k-step ks for this wave's column tiles, straight from L2 into registers.  Two register sets in ping-pong
        // with STATIC names (an index computed at run time would make hipcc scalarise the half vectors into selects and permutes)
        g_u32x4 bh0[NTW], bl0[NTW], bh1[NTW], bl1[NTW], bh2[NTW], bl2[NTW];
        auto load_b = [&](int ks, g_u32x4 (&bh)[NTW], g_u32x4 (&bl)[NTW]) {
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const int nt = (nt0 + j) < g.n_tiles ? (nt0 + j) : (g.n_tiles - 1);   // a tile beyond N: any valid fragment (its columns are not stored)
                const char* src = g.Bp + ((size_t)nt * g.k_steps + ks) * 2048 + lane * 16;
                bh[j] = *reinterpret_cast<const g_u32x4*>(src);
                bl[j] = *reinterpret_cast<const g_u32x4*>(src + 1024);
            }
        };
        auto mma = [&](int ks, const g_u32x4 (&bh)[NTW], const g_u32x4 (&bl)[NTW]) {
            g_half8 ah[2], al[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int fa = ((i * g.k_steps + ks) * 2) * 1024 + (((lane & 32) + ((lane + ks + (lane >> 5) * 8) & 31)) * 16);
                ah[i] = *reinterpret_cast<const g_half8*>(g_lds + fa);
                al[i] = *reinterpret_cast<const g_half8*>(g_lds + fa + 1024);
            }
            // the two products into acc_lo of one tile are kept apart (hi products of all tiles between them): no back-to-back dependent MFMAs
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j) acc_lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], __builtin_bit_cast(g_half8, bl[j]), acc_lo[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j) acc_hi[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], __builtin_bit_cast(g_half8, bh[j]), acc_hi[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j) acc_lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], __builtin_bit_cast(g_half8, bh[j]), acc_lo[i][j], 0, 0, 0);
        };
        // three register sets in rotation, loads two k-steps (24 MFMAs) ahead of their use: an L2 round trip fits under them
        load_b(0, bh0, bl0);
        if (1 < g.k_steps) load_b(1, bh1, bl1);
        for (int ks = 0; ks < g.k_steps; ks += 3) {
            if (ks + 2 < g.k_steps) load_b(ks + 2, bh2, bl2);
            mma(ks, bh0, bl0);
            if (ks + 1 < g.k_steps) {
                if (ks + 3 < g.k_steps) load_b(ks + 3, bh0, bl0);
                mma(ks + 1, bh1, bl1);
            }
            if (ks + 2 < g.k_steps) {
                if (ks + 4 < g.k_steps) load_b(ks + 4, bh1, bl1);
                mma(ks + 2, bh2, bl2);
            }
        }
        // ---- epilogue
        rows_epilogue<NTW, EPI>(g, blk, paired, unscale, nt0, lane, acc_hi, acc_lo);
        __syncthreads();   // the image is rewritten by the next block
    }
    if (bad) atomicOr(&g_gemm_range_flag, 1);
}

template <int NTW, int EPI = kEpiPlain>
static inline hipError_t gemm_rows_launch(const RowsArgs& g, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)k_gemm_rows<NTW, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, kRowsLdsBytes);
        attr = true;
    }
    const int n_blocks = (EPI != kEpiPlain && g.e.paired) ? (g.e.m_pts + 31) / 32 : (g.R + kRowsBM - 1) / kRowsBM;
    const int lds = 2 * g.k_steps * 2 * 1024;
    hipLaunchKernelGGL((k_gemm_rows<NTW, EPI>), dim3(n_blocks < 1024 ? n_blocks : 1024), dim3(256), lds, st, g);
    return hipGetLastError();
}


}  // namespace iron_train
