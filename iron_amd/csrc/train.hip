// Backward passes of the stage-2 render operators: see include/iron_train.h.
// Layer-wise batched formulation: activations recomputed in fp32 and kept in the caller's workspace, per-layer products
// (plain GEMMs with K = number of points) on the hand-written split-fp16 MFMA GEMM of gemm_h2.h -- no BLAS library --, all glue
// hand-written below.  One translation unit, its own shared library (libiron_train.so).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/iron_train.h"
#include "gemm_h2.h"

namespace iron_train {

thread_local int g_hip_error = 0;
thread_local int g_blas_status = 0;

#define TR_HIP(expr)                                 \
    do {                                             \
        hipError_t _e = (expr);                      \
        if (_e != hipSuccess) {                      \
            g_hip_error = (int)_e;                   \
            return IRON_ERR_HIP;                     \
        }                                            \
    } while (0)
#define TR_TRY(expr)                \
    do {                            \
        int _r = (expr);            \
        if (_r != IRON_OK) return _r; \
    } while (0)

// What every GEMM of one backward call shares: the caller's stream and a 256-byte device scratch (the |dZ|_max of the GEMM in flight).
struct GemmCtx {
    hipStream_t st;
    float* scratch;
    float* pack;        // room for the weight operand's fragment image (the dW partial buffer: never in use at the same time)
    size_t pack_floats;
};

// IRON_TRAIN_FUSE=0 keeps the activation passes as kernels of their own (A/B switch; the default fuses them into the row GEMMs)
static bool fused_epilogues() {
    static const bool v = [] { const char* e = getenv("IRON_TRAIN_FUSE"); return !(e && e[0] == '0'); }();
    return v;
}

// row-major C[m,n] = op(A) op(B) + beta C;  A is [m,k] (or [k,m] when ta), B is [k,n] (or [n,k] when tb).
// Three shapes occur: Z = X W^T (ta = 0, tb = 1: forward recompute, operands as they are), dX = dZ W (ta = 0, tb = 0) and the
// small-K leftovers of dW = dZ^T X (ta = 1, tb = 0): in the last two A is a gradient and carries a power-of-two scale from its
// absolute maximum (gemm_h2.h).
// `amax`: device scalar already holding |A|_max (written by the kernel that produced A: publish_absmax), or null -> one pass over A.
// `epi` (with `epi_mode` != kEpiPlain): the activation / activation-gradient pass fused into the row kernel's epilogue (gemm_h2.h);
// *fused tells the caller whether it was (the shapes the row kernel does not take run the plain product into C).
static int gemm_rm(const GemmCtx& h, bool ta, bool tb, int m, int n, int k, const float* A, int lda, const float* B, int ldb, float beta,
                   float* C, int ldc, const float* amax = nullptr, int epi_mode = kEpiPlain, const RowsEpi* epi = nullptr, bool* fused = nullptr) {
    if (fused) *fused = false;
    if (m == 0 || n == 0) return IRON_OK;
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = m; g.N = n; g.K = k; g.beta = beta;
    g.k_per_split = (k + kGemmTK - 1) / kGemmTK * kGemmTK;
    g.a_absmax = nullptr;
    const bool a_is_gradient = !(tb && !ta);
    if (a_is_gradient && amax) {
        g.a_absmax = amax;
    } else if (a_is_gradient && k > 0) {
        TR_HIP(hipMemsetAsync(h.scratch, 0, sizeof(float), h.st));
        const int64_t count = (int64_t)(ta ? k : m) * lda;  // the rows in use are contiguous: lda == row length at every call site
        hipLaunchKernelGGL(k_absmax, dim3((unsigned)((count + 1023) / 1024 < 1024 ? (count + 1023) / 1024 : 1024)), dim3(256), 0, h.st, A, count, h.scratch);
        g.a_absmax = h.scratch;
    }
    // a layer applied to a batch of rows (forward recompute, dX): the row kernel reads A once and writes C once
    const int k_steps = (k + 15) / 16, n_tiles = (n + 31) / 32;
    if (!ta && k_steps <= kRowsMaxKSteps && n_tiles <= 12 && (size_t)n_tiles * k_steps * 512 <= h.pack_floats && m >= 256) {
        PackBArgs pb;
        pb.B = B; pb.ldb = ldb; pb.N = n; pb.K = k; pb.k_strided = tb ? 0 : 1; pb.n_tiles = n_tiles; pb.k_steps = k_steps; pb.out = (char*)h.pack;
        hipLaunchKernelGGL(k_gemm_pack_b, dim3(n_tiles * k_steps), dim3(64), 0, h.st, pb);
        RowsArgs r;
        r.A = A; r.Bp = (const char*)h.pack; r.C = C; r.lda = lda; r.ldc = ldc; r.R = m; r.N = n; r.K = k; r.k_steps = k_steps; r.n_tiles = n_tiles;
        r.a_absmax = g.a_absmax; r.beta = beta;
        memset(&r.e, 0, sizeof(r.e));
        hipError_t er;
        if (epi_mode != kEpiPlain && epi && n_tiles > 4 && n_tiles <= 8 && fused_epilogues()) {   // the 256- and 217-wide layers
            r.e = *epi;
            if (epi_mode == kEpiSdfAct) er = gemm_rows_launch<2, kEpiSdfAct>(r, h.st);
            else if (epi_mode == kEpiSdfBack) er = gemm_rows_launch<2, kEpiSdfBack>(r, h.st);
            else if (epi_mode == kEpiReluAct) er = gemm_rows_launch<2, kEpiReluAct>(r, h.st);
            else er = gemm_rows_launch<2, kEpiReluBack>(r, h.st);
            if (fused) *fused = true;
        } else {
            er = n_tiles <= 4 ? gemm_rows_launch<1>(r, h.st) : (n_tiles <= 8 ? gemm_rows_launch<2>(r, h.st) : gemm_rows_launch<3>(r, h.st));
        }
        TR_HIP(er);
        return IRON_OK;
    }
    hipError_t e;
    if (!ta && tb) e = gemm_split_launch<false, false>(g, 1, h.st);
    else if (!ta && !tb) e = gemm_split_launch<false, true>(g, 1, h.st);
    else if (ta && !tb) e = gemm_split_launch<true, true>(g, 1, h.st);
    else return IRON_ERR_UNSUPPORTED;
    TR_HIP(e);
    return IRON_OK;
}

// dW[out,in] = beta dW + dZ[R,out]^T X[R,in].  The output is one or two tiles while K = R is 10^5, so K is split over
// blockIdx.z into partial tiles ([splits, out*in] in `partial`) that fill the chip and are then summed in a fixed order.
constexpr int kSplitK = 128;  // capacity of the partial buffer; the number of splits in use is split_k()
static int split_k() {
    static const int v = [] {
        const char* e = getenv("IRON_TRAIN_SPLITK");  // tuning knob (tools/train_step.py): in [2, 128]
        const int x = e ? atoi(e) : 128;
        return x >= 2 && x <= kSplitK ? x : 128;
    }();
    return v;
}

__global__ void k_reduce_partials(const float* __restrict__ partial, int splits, int count, float beta, float* __restrict__ dst) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.0f;
        for (int k = 0; k < splits; ++k) s += partial[(size_t)k * count + i];
        dst[i] = beta != 0.0f ? beta * dst[i] + s : s;
    }
}

static int gemm_dw(const GemmCtx& h, hipStream_t st, int out, int in, int R, const float* dZ, const float* X, float beta, float* dW, float* partial,
                   const float* amax = nullptr) {
    if (out == 0 || in == 0) return IRON_OK;
    int S = split_k();
    int kps = ((R + S - 1) / S + kGemmTK - 1) / kGemmTK * kGemmTK;      // rows per split, a multiple of the K tile
    if (kps < 8 * kGemmTK) kps = 8 * kGemmTK;                           // no split shorter than 256 rows
    S = (R + kps - 1) / kps;
    if (S <= 1) return gemm_rm(h, true, false, out, in, R, dZ, out, X, in, beta, dW, in, amax);
    GemmArgs g;
    g.A = dZ; g.B = X; g.C = partial; g.lda = out; g.ldb = in; g.ldc = in; g.M = out; g.N = in; g.K = R; g.beta = 0.0f;
    g.k_per_split = kps;
    if (amax) {
        g.a_absmax = amax;
    } else {
        TR_HIP(hipMemsetAsync(h.scratch, 0, sizeof(float), st));
        const int64_t count = (int64_t)R * out;
        hipLaunchKernelGGL(k_absmax, dim3((unsigned)((count + 1023) / 1024 < 1024 ? (count + 1023) / 1024 : 1024)), dim3(256), 0, st, dZ, count, h.scratch);
        g.a_absmax = h.scratch;
    }
    TR_HIP((gemm_split_launch<true, true>(g, S, st)));
    const int cnt = out * in;
    hipLaunchKernelGGL(k_reduce_partials, dim3((cnt + 255) / 256), dim3(256), 0, st, partial, S, cnt, beta, dW);
    return IRON_OK;
}

struct Bump {
    char* base;
    size_t off = 0;
    explicit Bump(void* b) : base((char*)b) {}
    float* take(size_t count) {
        off = (off + 255) & ~(size_t)255;
        float* p = base ? (float*)(base + off) : nullptr;
        off += count * sizeof(float);
        return p;
    }
};

static inline dim3 grid1(int64_t total, int block = 256) {
    int64_t b = (total + block - 1) / block;
    if (b < 1) b = 1;
    if (b > 65536) b = 65536;
    return dim3((unsigned)b);
}
#define GRID_STRIDE(i, total) for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (total); i += (int64_t)gridDim.x * blockDim.x)

// ---- positional encoding (models/embedder.py:6-54): [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)] ------------
__device__ __forceinline__ int pe_width(int L) { return 3 + 6 * L; }
// value of PE column c at x[3]; *dv = derivative of that column w.r.t. its coordinate x[comp]; *comp = which coordinate
__device__ __forceinline__ float pe_col(const float* x, int c, float* dv, int* comp) {
    if (c < 3) { *dv = 1.0f; *comp = c; return x[c]; }
    const int k = (c - 3) / 6, r = (c - 3) % 6;
    const float f = (float)(1 << k);
    *comp = r % 3;
    const float a = x[*comp] * f;
    float s, co;
    sincosf(a, &s, &co);
    if (r < 3) { *dv = f * co; return s; }
    *dv = -f * s;
    return co;
}

// ---- weight norm (torch._weight_norm, dim 0) ------------------------------------------------------------------------------------
__global__ void k_wn_fold(const float* __restrict__ v, const float* __restrict__ g, int out, int in, float* __restrict__ W) {
    const int r = blockIdx.x;
    const int lane = threadIdx.x;
    float nrm = 1.0f, gr = 1.0f;
    if (g) {
        float s = 0.0f;
        for (int c = lane; c < in; c += 64) { const float t = v[(size_t)r * in + c]; s += t * t; }
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        nrm = sqrtf(s);
        gr = g[r];
    }
    for (int c = lane; c < in; c += 64) W[(size_t)r * in + c] = g ? v[(size_t)r * in + c] * (gr / nrm) : v[(size_t)r * in + c];
}

__global__ void k_wn_back(const float* __restrict__ v, const float* __restrict__ g, const float* __restrict__ dW, int out, int in,
                          float* __restrict__ dv, float* __restrict__ dg) {
    const int r = blockIdx.x;
    const int lane = threadIdx.x;
    if (!g) {
        for (int c = lane; c < in; c += 64) dv[(size_t)r * in + c] = dW[(size_t)r * in + c];
        return;
    }
    float s = 0.0f, d = 0.0f;
    for (int c = lane; c < in; c += 64) {
        const float t = v[(size_t)r * in + c];
        s += t * t;
        d += dW[(size_t)r * in + c] * t;
    }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); d += __shfl_xor(d, o, 64); }
    const float nrm = sqrtf(s), gr = g[r];
    if (lane == 0 && dg) dg[r] = d / nrm;
    const float k1 = gr / nrm, k2 = d / s;
    for (int c = lane; c < in; c += 64) dv[(size_t)r * in + c] = k1 * (dW[(size_t)r * in + c] - v[(size_t)r * in + c] * k2);
}

// ---- SDF network ----------------------------------------------------------------------------------------------------------------
// rows [0,m): value; rows [m,2m): tangent along v (only when v != NULL)
__global__ void k_sdf_in(const float* __restrict__ x, const float* __restrict__ v, int m, int L, float* __restrict__ in0) {
    const int D = pe_width(L);
    GRID_STRIDE(i, (int64_t)m * D) {
        const int p = (int)(i / D), c = (int)(i % D);
        float dv; int comp;
        const float val = pe_col(x + 3 * (size_t)p, c, &dv, &comp);
        in0[i] = val;
        if (v) in0[(size_t)m * D + i] = dv * v[3 * (size_t)p + comp];
    }
}

// Z [R, out] (R = m or 2m): add bias to the value rows in place; next[:, 0:out) = (softplus(z), sigma'(z) zdot) * sc
__global__ void k_sdf_act(float* __restrict__ Z, const float* __restrict__ bias, int m, int out, int tangent, float sc, float* __restrict__ next,
                          int ld_next) {
    GRID_STRIDE(i, (int64_t)m * out) {
        const int p = (int)(i / out), c = (int)(i % out);
        const float z = Z[i] + bias[c];
        Z[i] = z;
        float a, s1, s2;
        softplus100(z, &a, &s1, &s2);
        next[(size_t)p * ld_next + c] = a * sc;
        if (tangent) next[(size_t)(m + p) * ld_next + c] = s1 * Z[(size_t)m * out + i] * sc;
    }
}

// dst[:, col0:col0+w) = src[:, 0:w) * sc   over `rows` rows
__global__ void k_copy_cols(const float* __restrict__ src, int ld_src, int rows, int w, float sc, float* __restrict__ dst, int ld_dst, int col0) {
    GRID_STRIDE(i, (int64_t)rows * w) {
        const int r = (int)(i / w), c = (int)(i % w);
        dst[(size_t)r * ld_dst + col0 + c] = src[(size_t)r * ld_src + c] * sc;
    }
}

// dst[:, 0:w) += src[:, col0:col0+w) * sc
__global__ void k_add_cols(const float* __restrict__ src, int ld_src, int col0, int rows, int w, float sc, float* __restrict__ dst, int ld_dst) {
    GRID_STRIDE(i, (int64_t)rows * w) {
        const int r = (int)(i / w), c = (int)(i % w);
        dst[(size_t)r * ld_dst + c] += src[(size_t)r * ld_src + col0 + c] * sc;
    }
}

// Column-strip form shared by the kernels that WRITE a layer's dZ: a thread owns column c of kStrip consecutive rows, so the
// bias gradient (the column sum of the value rows of dZ) is accumulated on the way and costs one atomic per strip instead of
// a second pass over dZ.
constexpr int kStrip = 64;
static inline dim3 strip_grid(int m, int out) { return dim3((out + 255) / 256, (m + kStrip - 1) / kStrip); }
#define STRIP_PROLOGUE(m, out)                                  \
    const int c = blockIdx.x * blockDim.x + threadIdx.x;        \
    if (c >= (out)) return;                                     \
    const int r0 = blockIdx.y * kStrip, r1 = min((m), r0 + kStrip); \
    float colsum = 0.0f;

// |dZ|_max as a by-product of the kernels that WRITE a layer's dZ (the split-fp16 GEMMs scale their gradient operand by it,
// gemm_h2.h): wave maximum, one atomicMax per wave.  Called by EVERY lane of the wave (STRIP_PROLOGUE_LIVE keeps the lanes of a
// partial last block alive with `live` = false and m = 0).
#define STRIP_PROLOGUE_LIVE(m, out)                             \
    const int c = blockIdx.x * blockDim.x + threadIdx.x;        \
    const bool live = c < (out);                                \
    const int r0 = blockIdx.y * kStrip, r1 = live ? min((m), r0 + kStrip) : r0; \
    float colsum = 0.0f;

__global__ void k_sdf_seed(const float* __restrict__ d_sdf, const float* __restrict__ d_feat, int m, int out, int tangent, float* __restrict__ dZ,
                           float* __restrict__ db, float* __restrict__ amax) {
    STRIP_PROLOGUE_LIVE(m, out)
    float mx = (live && tangent && c == 0) ? 1.0f : 0.0f;
    for (int p = r0; p < r1; ++p) {
        const size_t i = (size_t)p * out + c;
        const float g = c == 0 ? (d_sdf ? d_sdf[p] : 0.0f) : (d_feat ? d_feat[(size_t)p * (out - 1) + c - 1] : 0.0f);
        dZ[i] = g;
        colsum += g;
        mx = fmaxf(mx, fabsf(g));
        if (tangent) dZ[(size_t)m * out + i] = c == 0 ? 1.0f : 0.0f;  // d<v, grad sdf>/d(tangent output 0) = 1
    }
    if (live) atomicAdd(&db[c], colsum);
    publish_absmax(mx, amax);
}

// reverse of k_sdf_act: dX [R, ld_dx] holds dL/d(next input) for the columns [0,out); writes dZ [R, out]
__global__ void k_sdf_act_back(const float* __restrict__ dX, int ld_dx, const float* __restrict__ Z, int m, int out, int tangent, float sc,
                               float* __restrict__ dZ, float* __restrict__ db, float* __restrict__ amax) {
    STRIP_PROLOGUE_LIVE(m, out)
    float mx = 0.0f;
    for (int p = r0; p < r1; ++p) {
        const size_t i = (size_t)p * out + c;
        float a, s1, s2;
        softplus100(Z[i], &a, &s1, &s2);
        const float abar = dX[(size_t)p * ld_dx + c] * sc;
        float zbar = s1 * abar;
        if (tangent) {
            const float adotbar = dX[(size_t)(m + p) * ld_dx + c] * sc;
            zbar += s2 * Z[(size_t)m * out + i] * adotbar;
            dZ[(size_t)m * out + i] = s1 * adotbar;
            mx = fmaxf(mx, fabsf(s1 * adotbar));
        }
        dZ[i] = zbar;
        colsum += zbar;
        mx = fmaxf(mx, fabsf(zbar));
    }
    if (live) atomicAdd(&db[c], colsum);
    publish_absmax(mx, amax);
}

constexpr int kSdfChunk = 65536;
constexpr int kMaxLayers = 16;

struct SdfPlan {
    int L, m_max;
    float *W[kMaxLayers], *dW[kMaxLayers], *db[kMaxLayers], *IN[kMaxLayers], *Z[kMaxLayers];
    float *dZ, *dX, *partial, *scratch;
    size_t bytes, partial_floats;
};

static int sdf_plan(const iron_sdf_train_desc* d, int64_t n, void* ws, SdfPlan& P) {
    if (!d || !d->layers || d->n_linear < 2 || d->n_linear > kMaxLayers || d->multires < 0 || d->multires > 12) return IRON_ERR_BAD_ARG;
    const int L = d->n_linear;
    const int D0 = 3 + 6 * d->multires;
    if (d->layers[0].in_dim != D0) return IRON_ERR_UNSUPPORTED;
    for (int l = 1; l < L; ++l) {
        const int expect = d->layers[l - 1].out_dim + (l == d->skip_layer ? D0 : 0);
        if (d->layers[l].in_dim != expect) return IRON_ERR_UNSUPPORTED;
    }
    P.L = L;
    P.m_max = (int)(n < kSdfChunk ? (n > 0 ? n : 1) : kSdfChunk);
    const size_t R = 2 * (size_t)P.m_max;
    Bump b(ws);
    int maxw = 0;
    for (int l = 0; l < L; ++l) {
        const size_t wn = (size_t)d->layers[l].out_dim * d->layers[l].in_dim;
        P.W[l] = b.take(wn);
        P.dW[l] = b.take(wn);
        P.db[l] = b.take(d->layers[l].out_dim);
        P.IN[l] = b.take(R * d->layers[l].in_dim);
        P.Z[l] = l + 1 < L ? b.take(R * d->layers[l].out_dim) : nullptr;
        maxw = max(maxw, max(d->layers[l].out_dim, d->layers[l].in_dim));
    }
    P.dZ = b.take(R * maxw);
    P.dX = b.take(R * maxw);
    P.partial = b.take((size_t)kSplitK * maxw * maxw + 64);  // + 64 floats behind it: the GEMMs' scratch (GemmCtx)
    P.scratch = P.partial + (size_t)kSplitK * maxw * maxw;
    P.partial_floats = (size_t)kSplitK * maxw * maxw;
    P.bytes = b.off + 256;
    return IRON_OK;
}

static int sdf_backward(const iron_sdf_train_desc* d, const float* x, int64_t n, const float* d_sdf, const float* d_feat, const float* d_grad,
                        void* ws, size_t ws_bytes, hipStream_t st) {
    SdfPlan P;
    TR_TRY(sdf_plan(d, n, ws, P));
    if (P.bytes > ws_bytes) return IRON_ERR_WORKSPACE;
    const GemmCtx h{st, P.scratch, P.partial, P.partial_floats};
    const int L = P.L, D0 = 3 + 6 * d->multires;
    const iron_train_layer* ly = d->layers;
    const float rs2 = 0.70710678118654752440f;
    for (int l = 0; l < L; ++l) {
        hipLaunchKernelGGL(k_wn_fold, dim3(ly[l].out_dim), dim3(64), 0, st, ly[l].weight_v, ly[l].weight_g, ly[l].out_dim, ly[l].in_dim, P.W[l]);
        TR_HIP(hipMemsetAsync(P.db[l], 0, sizeof(float) * ly[l].out_dim, st));
        if (n == 0) TR_HIP(hipMemsetAsync(P.dW[l], 0, sizeof(float) * ly[l].out_dim * ly[l].in_dim, st));
    }
    const int tangent = d_grad ? 1 : 0;
    for (int64_t p0 = 0; p0 < n; p0 += P.m_max) {
        const int m = (int)((n - p0) < P.m_max ? (n - p0) : P.m_max);
        const int R = tangent ? 2 * m : m;
        // forward, keeping every layer's input and pre-activation
        hipLaunchKernelGGL(k_sdf_in, grid1((int64_t)m * D0), dim3(256), 0, st, x + 3 * p0, d_grad ? d_grad + 3 * p0 : nullptr, m, d->multires, P.IN[0]);
        for (int l = 0; l + 1 < L; ++l) {
            const int out = ly[l].out_dim, in = ly[l].in_dim, in_next = ly[l + 1].in_dim;
            const bool to_skip = (l + 1 == d->skip_layer);
            RowsEpi ea;
            memset(&ea, 0, sizeof(ea));
            ea.paired = tangent; ea.m_pts = m; ea.n_act = out; ea.sc = to_skip ? rs2 : 1.0f; ea.bias = ly[l].bias;
            ea.Z = P.Z[l]; ea.ldz = out; ea.out = P.IN[l + 1]; ea.ld_out = in_next;
            bool fused = false;
            TR_TRY(gemm_rm(h, false, true, R, out, in, P.IN[l], in, P.W[l], in, 0.0f, P.Z[l], out, nullptr, kEpiSdfAct, &ea, &fused));
            if (!fused)
                hipLaunchKernelGGL(k_sdf_act, grid1((int64_t)m * out), dim3(256), 0, st, P.Z[l], ly[l].bias, m, out, tangent, to_skip ? rs2 : 1.0f,
                                   P.IN[l + 1], in_next);
            if (to_skip)
                hipLaunchKernelGGL(k_copy_cols, grid1((int64_t)R * D0), dim3(256), 0, st, P.IN[0], D0, R, D0, rs2, P.IN[l + 1], in_next, out);
        }
        // reverse
        const int out_last = ly[L - 1].out_dim;
        // |dZ|_max of a layer is written by the kernel that writes its dZ; two scalars in turn, because the fused row GEMM reads the
        // maximum of its operand (this layer's dZ) while its epilogue publishes the next one's.  dZ itself alternates between P.dZ and
        // P.dX for the same reason (a block's output rows are other blocks' input rows when the widths differ).
        float* dz_max_cur = h.scratch + 1;
        float* dz_max_nxt = h.scratch + 2;
        float* dz_cur = P.dZ;
        float* dz_nxt = P.dX;
        TR_HIP(hipMemsetAsync(dz_max_cur, 0, sizeof(float), st));
        hipLaunchKernelGGL(k_sdf_seed, strip_grid(m, out_last), dim3(256), 0, st, d_sdf ? d_sdf + p0 : nullptr,
                           d_feat ? d_feat + (size_t)p0 * (out_last - 1) : nullptr, m, out_last, tangent, dz_cur, P.db[L - 1], dz_max_cur);
        for (int l = L - 1; l >= 0; --l) {
            const int out = ly[l].out_dim, in = ly[l].in_dim;
            TR_TRY(gemm_dw(h, st, out, in, R, dz_cur, P.IN[l], p0 == 0 ? 0.0f : 1.0f, P.dW[l], P.partial, dz_max_cur));
            if (l == 0) break;
            const int outp = ly[l - 1].out_dim;
            const float sc = l == d->skip_layer ? rs2 : 1.0f;
            TR_HIP(hipMemsetAsync(dz_max_nxt, 0, sizeof(float), st));
            RowsEpi eb;
            memset(&eb, 0, sizeof(eb));
            eb.paired = tangent; eb.m_pts = m; eb.n_act = outp; eb.sc = sc; eb.Z = P.Z[l - 1]; eb.ldz = outp;
            eb.out = dz_nxt; eb.ld_out = outp; eb.db = P.db[l - 1]; eb.amax_out = dz_max_nxt;
            bool fused = false;
            // unfused: the product goes to dz_nxt (as dX, row length `in`), k_sdf_act_back turns it into dZ of layer l - 1 in dz_cur
            TR_TRY(gemm_rm(h, false, false, R, in, out, dz_cur, out, P.W[l], in, 0.0f, dz_nxt, in, dz_max_cur, kEpiSdfBack, &eb, &fused));
            if (fused) {
                float* t = dz_cur; dz_cur = dz_nxt; dz_nxt = t;
            } else {
                hipLaunchKernelGGL(k_sdf_act_back, strip_grid(m, outp), dim3(256), 0, st, dz_nxt, in, P.Z[l - 1], m, outp, tangent, sc, dz_cur,
                                   P.db[l - 1], dz_max_nxt);
            }
            float* tm = dz_max_cur; dz_max_cur = dz_max_nxt; dz_max_nxt = tm;
        }
    }
    for (int l = 0; l < L; ++l) {
        hipLaunchKernelGGL(k_wn_back, dim3(ly[l].out_dim), dim3(64), 0, st, ly[l].weight_v, ly[l].weight_g, P.dW[l], ly[l].out_dim, ly[l].in_dim,
                           ly[l].d_weight_v, ly[l].d_weight_g);
        TR_HIP(hipMemcpyAsync(ly[l].d_bias, P.db[l], sizeof(float) * ly[l].out_dim, hipMemcpyDeviceToDevice, st));
    }
    TR_HIP(hipGetLastError());
    return IRON_OK;
}

// ---- RenderingNetwork -----------------------------------------------------------------------------------------------------------
struct RenderIn {
    int np, nv, nn, nf, D0, Lp, Lv;
};
static int render_in(const iron_render_train_desc* d, RenderIn& r) {
    const bool use_v = d->mode == IRON_MODE_IDR || d->mode == IRON_MODE_NO_NORMAL;
    const bool use_n = d->mode == IRON_MODE_IDR || d->mode == IRON_MODE_NO_VIEW_DIR;
    if (d->mode < 0 || d->mode > 3) return IRON_ERR_BAD_ARG;
    r.Lp = d->multires > 0 ? d->multires : 0;
    r.Lv = d->multires_view > 0 ? d->multires_view : 0;
    r.np = 3 + 6 * r.Lp;
    r.nv = use_v ? 3 + 6 * r.Lv : 0;
    r.nn = use_n ? 3 : 0;
    r.nf = d->d_feature;
    r.D0 = r.np + r.nv + r.nn + r.nf;
    return IRON_OK;
}

__global__ void k_render_in(RenderIn r, const float* __restrict__ pts, const float* __restrict__ nrm, const float* __restrict__ view,
                            const float* __restrict__ feat, int m, float* __restrict__ in0) {
    GRID_STRIDE(i, (int64_t)m * r.D0) {
        const int p = (int)(i / r.D0);
        int c = (int)(i % r.D0);
        float dv; int comp;
        float val;
        if (c < r.np) val = pe_col(pts + 3 * (size_t)p, c, &dv, &comp);
        else if ((c -= r.np) < r.nv) val = pe_col(view + 3 * (size_t)p, c, &dv, &comp);
        else if ((c -= r.nv) < r.nn) val = nrm[3 * (size_t)p + c];
        else val = feat[(size_t)p * r.nf + (c - r.nn)];
        in0[i] = val;
    }
}

// one thread per point: fold dL/d(in0) back onto points / view_dirs / normals
__global__ void k_render_in_back(RenderIn r, const float* __restrict__ pts, const float* __restrict__ view, int m, const float* __restrict__ din0,
                                 float* __restrict__ d_pts, float* __restrict__ d_view, float* __restrict__ d_nrm) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= m) return;
    const float* g = din0 + (size_t)p * r.D0;
    if (d_pts) {
        float acc[3] = {0.f, 0.f, 0.f};
        for (int c = 0; c < r.np; ++c) { float dv; int comp; pe_col(pts + 3 * (size_t)p, c, &dv, &comp); acc[comp] += g[c] * dv; }
        for (int k = 0; k < 3; ++k) d_pts[3 * (size_t)p + k] = acc[k];
    }
    if (d_view) {
        float acc[3] = {0.f, 0.f, 0.f};
        for (int c = 0; c < r.nv; ++c) { float dv; int comp; pe_col(view + 3 * (size_t)p, c, &dv, &comp); acc[comp] += g[r.np + c] * dv; }
        for (int k = 0; k < 3; ++k) d_view[3 * (size_t)p + k] = acc[k];
    }
    if (d_nrm)
        for (int k = 0; k < 3; ++k) d_nrm[3 * (size_t)p + k] = r.nn ? g[r.np + r.nv + k] : 0.0f;
}

__global__ void k_relu_act(float* __restrict__ Z, const float* __restrict__ bias, int m, int out, float sc, float* __restrict__ next, int ld_next) {
    GRID_STRIDE(i, (int64_t)m * out) {
        const int p = (int)(i / out), c = (int)(i % out);
        const float z = Z[i] + bias[c];
        Z[i] = z;
        next[(size_t)p * ld_next + c] = fmaxf(z, 0.0f) * sc;
    }
}

__global__ void k_relu_back(const float* __restrict__ dX, int ld_dx, const float* __restrict__ Z, int m, int out, float sc, float* __restrict__ dZ,
                            float* __restrict__ db, float* __restrict__ amax) {
    STRIP_PROLOGUE_LIVE(m, out)
    float mx = 0.0f;
    for (int p = r0; p < r1; ++p) {
        const size_t i = (size_t)p * out + c;
        const float g = Z[i] > 0.0f ? dX[(size_t)p * ld_dx + c] * sc : 0.0f;
        dZ[i] = g;
        colsum += g;
        mx = fmaxf(mx, fabsf(g));
    }
    if (live) atomicAdd(&db[c], colsum);
    publish_absmax(mx, amax);
}

// last layer: z = Z + b; y = os (z + ob); optionally sq * sigmoid(y); dZ = dL/dz
__global__ void k_render_out_back(const float* __restrict__ Z, const float* __restrict__ bias, const float* __restrict__ d_out, int m, int out, float ob,
                                  float os, int squeeze, float sq, float* __restrict__ dZ, float* __restrict__ db) {
    STRIP_PROLOGUE(m, out)
    for (int p = r0; p < r1; ++p) {
        const size_t i = (size_t)p * out + c;
        float g = d_out[i];
        if (squeeze) {
            const float y = os * ((Z[i] + bias[c]) + ob);
            const float sg = 1.0f / (1.0f + expf(-y));
            g *= sq * sg * (1.0f - sg);
        }
        g *= os;
        dZ[i] = g;
        colsum += g;
    }
    atomicAdd(&db[c], colsum);
}

constexpr int kRenderChunk = 131072;

struct RenderPlan {
    int L, m_max;
    RenderIn in;
    float *W[kMaxLayers], *dW[kMaxLayers], *db[kMaxLayers], *X[kMaxLayers], *Z[kMaxLayers];
    float *dZ, *dX, *dIN0, *partial, *scratch;
    size_t bytes, partial_floats;
};

static int render_plan(const iron_render_train_desc* d, int64_t n, void* ws, RenderPlan& P) {
    if (!d || !d->layers || d->n_linear < 1 || d->n_linear > kMaxLayers || d->d_feature < 0) return IRON_ERR_BAD_ARG;
    TR_TRY(render_in(d, P.in));
    const int L = d->n_linear;
    if (d->layers[0].in_dim != P.in.D0 || d->layers[L - 1].out_dim != d->d_out) return IRON_ERR_UNSUPPORTED;
    for (int l = 1; l < L; ++l)
        if (d->layers[l].in_dim != d->layers[l - 1].out_dim + (l == d->skip_layer ? P.in.D0 : 0)) return IRON_ERR_UNSUPPORTED;
    if (d->skip_layer == 0) return IRON_ERR_UNSUPPORTED;
    P.L = L;
    P.m_max = (int)(n < kRenderChunk ? (n > 0 ? n : 1) : kRenderChunk);
    const size_t R = (size_t)P.m_max;
    Bump b(ws);
    int maxw = P.in.D0;
    for (int l = 0; l < L; ++l) {
        const size_t wn = (size_t)d->layers[l].out_dim * d->layers[l].in_dim;
        P.W[l] = b.take(wn);
        P.dW[l] = b.take(wn);
        P.db[l] = b.take(d->layers[l].out_dim);
        P.X[l] = b.take(R * d->layers[l].in_dim);
        P.Z[l] = b.take(R * d->layers[l].out_dim);
        maxw = max(maxw, max(d->layers[l].out_dim, d->layers[l].in_dim));
    }
    P.dZ = b.take(R * maxw);
    P.dX = b.take(R * maxw);
    P.dIN0 = b.take(R * P.in.D0);
    P.partial = b.take((size_t)kSplitK * maxw * maxw + 64);  // + 64 floats behind it: the GEMMs' scratch (GemmCtx)
    P.scratch = P.partial + (size_t)kSplitK * maxw * maxw;
    P.partial_floats = (size_t)kSplitK * maxw * maxw;
    P.bytes = b.off + 256;
    return IRON_OK;
}

static int render_backward(const iron_render_train_desc* d, const float* pts, const float* nrm, const float* view, const float* feat, int64_t n,
                           const float* d_out, float* d_pts, float* d_nrm, float* d_view, float* d_feat, void* ws, size_t ws_bytes,
                           hipStream_t st) {
    RenderPlan P;
    TR_TRY(render_plan(d, n, ws, P));
    if (P.bytes > ws_bytes) return IRON_ERR_WORKSPACE;
    const RenderIn& I = P.in;
    if (n > 0 && (!pts || !d_out || (I.nf && !feat) || (I.nn && !nrm) || (I.nv && !view))) return IRON_ERR_BAD_ARG;
    const GemmCtx h{st, P.scratch, P.partial, P.partial_floats};
    const int L = P.L;
    const iron_train_layer* ly = d->layers;
    const float rs2 = 0.70710678118654752440f;
    for (int l = 0; l < L; ++l) {
        hipLaunchKernelGGL(k_wn_fold, dim3(ly[l].out_dim), dim3(64), 0, st, ly[l].weight_v, ly[l].weight_g, ly[l].out_dim, ly[l].in_dim, P.W[l]);
        TR_HIP(hipMemsetAsync(P.db[l], 0, sizeof(float) * ly[l].out_dim, st));
        if (n == 0) TR_HIP(hipMemsetAsync(P.dW[l], 0, sizeof(float) * ly[l].out_dim * ly[l].in_dim, st));
    }
    for (int64_t p0 = 0; p0 < n; p0 += P.m_max) {
        const int m = (int)((n - p0) < P.m_max ? (n - p0) : P.m_max);
        const float* cp = pts + 3 * p0;
        const float* cv = view ? view + 3 * p0 : nullptr;
        hipLaunchKernelGGL(k_render_in, grid1((int64_t)m * I.D0), dim3(256), 0, st, I, cp, nrm ? nrm + 3 * p0 : nullptr, cv,
                           feat ? feat + (size_t)p0 * I.nf : nullptr, m, P.X[0]);
        for (int l = 0; l < L; ++l) {
            const int out = ly[l].out_dim, in = ly[l].in_dim;
            if (l + 1 == L) {
                TR_TRY(gemm_rm(h, false, true, m, out, in, P.X[l], in, P.W[l], in, 0.0f, P.Z[l], out));
                break;
            }
            const bool to_skip = (l + 1 == d->skip_layer);
            const int in_next = ly[l + 1].in_dim;
            RowsEpi ea;
            memset(&ea, 0, sizeof(ea));
            ea.m_pts = m; ea.n_act = out; ea.sc = to_skip ? rs2 : 1.0f; ea.bias = ly[l].bias; ea.Z = P.Z[l]; ea.ldz = out;
            ea.out = P.X[l + 1]; ea.ld_out = in_next;
            bool fused = false;
            TR_TRY(gemm_rm(h, false, true, m, out, in, P.X[l], in, P.W[l], in, 0.0f, P.Z[l], out, nullptr, kEpiReluAct, &ea, &fused));
            if (!fused)
                hipLaunchKernelGGL(k_relu_act, grid1((int64_t)m * out), dim3(256), 0, st, P.Z[l], ly[l].bias, m, out, to_skip ? rs2 : 1.0f, P.X[l + 1], in_next);
            if (to_skip)
                hipLaunchKernelGGL(k_copy_cols, grid1((int64_t)m * I.D0), dim3(256), 0, st, P.X[0], I.D0, m, I.D0, rs2, P.X[l + 1], in_next, out);
        }
        const int out_last = ly[L - 1].out_dim;
        hipLaunchKernelGGL(k_render_out_back, strip_grid(m, out_last), dim3(256), 0, st, P.Z[L - 1], ly[L - 1].bias, d_out + (size_t)p0 * out_last, m,
                           out_last, d->output_bias, d->output_scale, d->squeeze_out, d->squeeze_out_scale, P.dZ, P.db[L - 1]);
        TR_HIP(hipMemsetAsync(P.dIN0, 0, sizeof(float) * (size_t)m * I.D0, st));
        // |dZ|_max of the layer in flight, written by the kernel that writes its dZ (the 3-wide seed of the last layer: one tiny pass);
        // scalars and dZ buffers in turn as in sdf_backward
        float* dz_max_cur = h.scratch + 1;
        float* dz_max_nxt = h.scratch + 2;
        float* dz_cur = P.dZ;
        float* dz_nxt = P.dX;
        for (int l = L - 1; l >= 0; --l) {
            const int out = ly[l].out_dim, in = ly[l].in_dim;
            const float* known = l == L - 1 ? nullptr : dz_max_cur;
            TR_TRY(gemm_dw(h, st, out, in, m, dz_cur, P.X[l], p0 == 0 ? 0.0f : 1.0f, P.dW[l], P.partial, known));
            if (l == 0) {
                TR_TRY(gemm_rm(h, false, false, m, in, out, dz_cur, out, P.W[l], in, 0.0f, dz_nxt, in, known));
                hipLaunchKernelGGL(k_add_cols, grid1((int64_t)m * I.D0), dim3(256), 0, st, dz_nxt, in, 0, m, I.D0, 1.0f, P.dIN0, I.D0);
                break;
            }
            const int outp = ly[l - 1].out_dim;
            const bool is_skip = (l == d->skip_layer);
            TR_HIP(hipMemsetAsync(dz_max_nxt, 0, sizeof(float), st));
            RowsEpi eb;
            memset(&eb, 0, sizeof(eb));
            eb.m_pts = m; eb.n_act = outp; eb.sc = 1.0f; eb.Z = P.Z[l - 1]; eb.ldz = outp; eb.out = dz_nxt; eb.ld_out = outp;
            eb.db = P.db[l - 1]; eb.amax_out = dz_max_nxt;
            bool fused = false;
            // (the skip layer also needs the plain product's input columns: it keeps the separate pass)
            TR_TRY(gemm_rm(h, false, false, m, in, out, dz_cur, out, P.W[l], in, 0.0f, dz_nxt, in, known, is_skip ? kEpiPlain : kEpiReluBack,
                           is_skip ? nullptr : &eb, &fused));
            if (fused) {
                float* t = dz_cur; dz_cur = dz_nxt; dz_nxt = t;
            } else {
                if (is_skip) hipLaunchKernelGGL(k_add_cols, grid1((int64_t)m * I.D0), dim3(256), 0, st, dz_nxt, in, outp, m, I.D0, rs2, P.dIN0, I.D0);
                hipLaunchKernelGGL(k_relu_back, strip_grid(m, outp), dim3(256), 0, st, dz_nxt, in, P.Z[l - 1], m, outp, is_skip ? rs2 : 1.0f, dz_cur,
                                   P.db[l - 1], dz_max_nxt);
            }
            float* tm = dz_max_cur; dz_max_cur = dz_max_nxt; dz_max_nxt = tm;
        }
        hipLaunchKernelGGL(k_render_in_back, dim3((m + 255) / 256), dim3(256), 0, st, I, cp, cv, m, P.dIN0, d_pts ? d_pts + 3 * p0 : nullptr,
                           (d_view && I.nv) ? d_view + 3 * p0 : nullptr, d_nrm ? d_nrm + 3 * p0 : nullptr);
        if (d_view && !I.nv) TR_HIP(hipMemsetAsync(d_view + 3 * p0, 0, sizeof(float) * 3 * m, st));
        if (d_feat && I.nf)
            hipLaunchKernelGGL(k_copy_cols, grid1((int64_t)m * I.nf), dim3(256), 0, st, P.dIN0 + (I.np + I.nv + I.nn), I.D0, m, I.nf, 1.0f,
                               d_feat + (size_t)p0 * I.nf, I.nf, 0);
    }
    for (int l = 0; l < L; ++l) {
        hipLaunchKernelGGL(k_wn_back, dim3(ly[l].out_dim), dim3(64), 0, st, ly[l].weight_v, ly[l].weight_g, P.dW[l], ly[l].out_dim, ly[l].in_dim,
                           ly[l].d_weight_v, ly[l].d_weight_g);
        TR_HIP(hipMemcpyAsync(ly[l].d_bias, P.db[l], sizeof(float) * ly[l].out_dim, hipMemcpyDeviceToDevice, st));
    }
    TR_HIP(hipGetLastError());
    return IRON_OK;
}

// ---- GGX (models/renderer_ggx.py:82-146): first-order dual numbers in (cos, alpha, distance) -------------------------------------
struct D3 {
    float v, d[3];
};
__device__ __forceinline__ D3 cst(float c) { return {c, {0.f, 0.f, 0.f}}; }
__device__ __forceinline__ D3 var(float x, int k) { D3 r = {x, {0.f, 0.f, 0.f}}; r.d[k] = 1.0f; return r; }
__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return {a.v + b.v, {a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2]}}; }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return {a.v - b.v, {a.d[0] - b.d[0], a.d[1] - b.d[1], a.d[2] - b.d[2]}}; }
__device__ __forceinline__ D3 operator*(D3 a, D3 b) {
    return {a.v * b.v, {a.d[0] * b.v + a.v * b.d[0], a.d[1] * b.v + a.v * b.d[1], a.d[2] * b.v + a.v * b.d[2]}};
}
__device__ __forceinline__ D3 operator/(D3 a, D3 b) {
    const float q = a.v / b.v, ib = 1.0f / b.v;
    return {q, {(a.d[0] - q * b.d[0]) * ib, (a.d[1] - q * b.d[1]) * ib, (a.d[2] - q * b.d[2]) * ib}};
}
__device__ __forceinline__ D3 dsqrt(D3 a) {
    const float s = sqrtf(a.v), k = 0.5f / s;
    return {s, {a.d[0] * k, a.d[1] * k, a.d[2] * k}};
}

__global__ void k_ggx_back(float light, const float* __restrict__ dist, const float* __restrict__ nrm, const float* __restrict__ view,
                           const float* __restrict__ kd, const float* __restrict__ ks, const float* __restrict__ rough,
                           const float* __restrict__ tab_trans, const float* __restrict__ tab_diff, int n, const float* __restrict__ g_diff,
                           const float* __restrict__ g_spec, const float* __restrict__ g_rgb, float* __restrict__ d_light,
                           float* __restrict__ d_dist, float* __restrict__ d_nrm, float* __restrict__ d_view, float* __restrict__ d_kd,
                           float* __restrict__ d_ks, float* __restrict__ d_rough) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    float light_acc = 0.0f;
    if (p < n) {
        const float* nn = nrm + 3 * (size_t)p;
        const float* vv = view + 3 * (size_t)p;
        const float raw_dot = (vv[0] * nn[0] + vv[1] * nn[1]) + vv[2] * nn[2];
        const float cdot = fminf(fmaxf(raw_dot, 0.00001f), 0.99999f);
        const bool dot_live = raw_dot >= 0.00001f && raw_dot <= 0.99999f;  // torch.clamp passes the gradient on [min, max]
        const float r0 = rough[p];
        const float alpha0 = fmaxf(r0, 0.0001f);
        const bool alpha_live = r0 >= 0.0001f;
        const float pi_f = 3.14159274101257324219f;
        const float m_inv_eta2 = (float)(1.0 / (1.48958738 * 1.48958738));
        // piecewise-constant table factors
        const long long tx = (long long)floorf(powf(cdot, 0.25f) * 100.0f);
        const long long ty = (long long)floorf(powf(alpha0 / 4.0f, 0.25f) * 50.0f);
        long long ti = ty * 100 + tx;
        ti = ti < 0 ? 0 : (ti > 4999 ? 4999 : ti);
        const float T12 = fminf(fmaxf(tab_trans[ti], 0.0f), 1.0f);
        const long long ai = ty < 0 ? 0 : (ty > 49 ? 49 : ty);
        const float Fdr = fminf(fmaxf(1.0f - tab_diff[ai], 0.0f), 1.0f);
        const float fd = 1.0f - Fdr + 1e-10f;

        const D3 c = var(cdot, 0), a = var(alpha0, 1), ds = var(dist[p], 2);
        const D3 unit = cst(1.0f) / (ds * ds + cst(1e-10f));  // intensity per unit light
        const D3 c2 = c * c;
        const D3 a2 = a * a;
        const D3 root = c2 + (cst(1.0f) - c2) / (a2 + cst(1e-10f));
        const D3 Dm = cst(1.0f) / (cst(pi_f) * a2 * root * root + cst(1e-10f));
        const D3 tan_t = dsqrt(cst(1.0f) - c2) / (c + cst(1e-10f));
        const D3 rt = a * tan_t;
        const D3 g1 = cst(2.0f) / (cst(1.0f) + dsqrt(rt * rt + cst(1.0f)));
        const D3 Ks = unit * cst(0.03867f) * Dm * (g1 * g1) / (cst(4.0f) * c + cst(1e-10f));  // specular per unit light and albedo
        const D3 Kd = unit * c * cst(T12 * T12 * m_inv_eta2 / (fd * pi_f));                 // diffuse  per unit light and albedo
        float A = 0.0f, B = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float gd = (g_diff ? g_diff[3 * (size_t)p + k] : 0.0f) + (g_rgb ? g_rgb[3 * (size_t)p + k] : 0.0f);
            const float gs = (g_spec ? g_spec[3 * (size_t)p + k] : 0.0f) + (g_rgb ? g_rgb[3 * (size_t)p + k] : 0.0f);
            if (d_kd) d_kd[3 * (size_t)p + k] = gd * light * Kd.v;
            if (d_ks) d_ks[3 * (size_t)p + k] = gs * light * Ks.v;
            A += gd * kd[3 * (size_t)p + k];
            B += gs * ks[3 * (size_t)p + k];
        }
        light_acc = A * Kd.v + B * Ks.v;
        const float dc = dot_live ? light * (A * Kd.d[0] + B * Ks.d[0]) : 0.0f;
        if (d_rough) d_rough[p] = alpha_live ? light * (A * Kd.d[1] + B * Ks.d[1]) : 0.0f;
        if (d_dist) d_dist[p] = light * (A * Kd.d[2] + B * Ks.d[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (d_nrm) d_nrm[3 * (size_t)p + k] = dc * vv[k];
            if (d_view) d_view[3 * (size_t)p + k] = dc * nn[k];
        }
    }
    if (d_light) {
        for (int o = 32; o > 0; o >>= 1) light_acc += __shfl_xor(light_acc, o, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(d_light, light_acc);
    }
}

// ---- CompositeRenderer (models/renderer_ggx.py:781-858, point-light branch): duals in (cos, roughness, distance, metallic eta,
// metallic k, dielectric eta); linear in the two albedos and the light.  Same expressions as csrc/ggx_core.h:composite_point.
template <int N>
struct Dual {
    float v, d[N];
};
template <int N> __device__ __forceinline__ Dual<N> dc(float c) { Dual<N> r; r.v = c; for (int i = 0; i < N; ++i) r.d[i] = 0.f; return r; }
template <int N> __device__ __forceinline__ Dual<N> dvar(float x, int k) { Dual<N> r = dc<N>(x); r.d[k] = 1.f; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator+(Dual<N> a, Dual<N> b) { Dual<N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator-(Dual<N> a, Dual<N> b) { Dual<N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator*(Dual<N> a, Dual<N> b) { Dual<N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator/(Dual<N> a, Dual<N> b) {
    Dual<N> r; r.v = a.v / b.v; const float ib = 1.0f / b.v;
    for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib;
    return r;
}
template <int N> __device__ __forceinline__ Dual<N> dsqrt(Dual<N> a) {
    Dual<N> r; r.v = sqrtf(a.v); const float k = 0.5f / r.v;
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * k;
    return r;
}

struct CompBackArgs {
    const float *dist, *nrm, *view, *kd, *ks, *rough, *m_eta, *m_k, *d_eta, *env, *tab_trans, *tab_diff;
    const float *g_rgb, *g_spec, *g_met, *g_die, *g_env;
    float *d_light, *d_dist, *d_nrm, *d_view, *d_kd, *d_ks, *d_rough, *d_m_eta, *d_m_k, *d_d_eta, *d_env;
    float light;
    int n;
};

__global__ void k_composite_back(CompBackArgs a) {
    typedef Dual<6> D6;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    float light_acc = 0.0f;
    if (p < a.n) {
        const float* nn = a.nrm + 3 * (size_t)p;
        const float* vv = a.view + 3 * (size_t)p;
        const float raw_dot = (vv[0] * nn[0] + vv[1] * nn[1]) + vv[2] * nn[2];
        const float cdot = fminf(fmaxf(raw_dot, 0.00001f), 0.99999f);
        const bool dot_live = raw_dot >= 0.00001f && raw_dot <= 0.99999f;
        const float r_in = a.rough[p], me_in = a.m_eta[p], mk_in = a.m_k[p], de_in = a.d_eta[p];
        const float rough0 = fmaxf(r_in, 0.00001f);
        const float de0 = fminf(fmaxf(de_in, 1.000001f), 1.999999f);
        const float me0 = fminf(fmaxf(me_in, 0.099999f), 4.999999f);
        const float mk0 = fminf(fmaxf(mk_in, 0.099999f), 9.999999f);
        const bool use_env = a.env != nullptr;  // intensity = clamp(env_light, 1e-6, 20) instead of light / (d^2 + 1e-10)
        const float env_in = use_env ? a.env[p] : 0.0f;
        const bool live[6] = {dot_live, r_in >= 0.00001f, use_env ? (env_in >= 0.000001f && env_in <= 20.0f) : true,
                              me_in >= 0.099999f && me_in <= 4.999999f,
                              mk_in >= 0.099999f && mk_in <= 9.999999f, de_in >= 1.000001f && de_in <= 1.999999f};
        // piecewise-constant diffuse tables (alpha := max(rough, 1e-4) as CompositeRenderer.diffuse_reflection_ggx does)
        const float alpha_t = fmaxf(rough0, 0.0001f);
        const long long tx = (long long)floorf(powf(cdot, 0.25f) * 100.0f);
        const long long ty = (long long)floorf(powf(alpha_t / 4.0f, 0.25f) * 50.0f);
        long long ti = ty * 100 + tx;
        ti = ti < 0 ? 0 : (ti > 4999 ? 4999 : ti);
        const float T12 = fminf(fmaxf(a.tab_trans[ti], 0.0f), 1.0f);
        const long long ai = ty < 0 ? 0 : (ty > 49 ? 49 : ty);
        const float Fdr = fminf(fmaxf(1.0f - a.tab_diff[ai], 0.0f), 1.0f);
        const float fd = 1.0f - Fdr + 1e-10f;
        const float pi_f = 3.14159274101257324219f;
        const float inv_eta2 = (float)(1.0 / (1.48958738 * 1.48958738));
        const float eta2 = (float)(1.48958738 * 1.48958738 + 1e-10);
        const float pi_eta2 = (float)(3.141592653589793 * 1.48958738 * 1.48958738);

        const D6 c = dvar<6>(cdot, 0), rg = dvar<6>(rough0, 1), me = dvar<6>(me0, 3), mk = dvar<6>(mk0, 4), de = dvar<6>(de0, 5);
        const D6 one = dc<6>(1.0f);
        D6 U;  // intensity per unit `lightf`: variable 2 is the distance (point light) or the env-light value
        if (use_env) {
            U = dvar<6>(fminf(fmaxf(env_in, 0.000001f), 20.0f), 2);
        } else {
            const D6 ds = dvar<6>(a.dist[p], 2);
            U = one / (ds * ds + dc<6>(1e-10f));
        }
        const float lightf = use_env ? 1.0f : a.light;
        const D6 c2 = c * c, s2 = one - c2;
        // GGX NDF with alpha := eta (the reference's quirk), Smith G1 with the roughness
        const D6 root = c2 + s2 / dc<6>(eta2);
        const D6 Dn = one / (dc<6>(pi_eta2) * root * root + dc<6>(1e-10f));
        const D6 tan_t = dsqrt(s2) / (c + dc<6>(1e-10f));
        const D6 rt = rg * tan_t;
        const D6 g1 = dc<6>(2.0f) / (one + dsqrt(rt * rt + one));
        // conductor Fresnel
        const D6 s4 = s2 * s2;
        const D6 temp1 = me * me - mk * mk - s2;
        const D6 a2pb2 = dsqrt(temp1 * temp1 + dc<6>(4.0f) * mk * mk * me * me);
        const D6 aa = dsqrt(dc<6>(0.5f) * (a2pb2 + temp1));
        const D6 term1 = a2pb2 + c2, term2 = dc<6>(2.0f) * aa * c;
        const D6 rs2 = (term1 - term2) / (term1 + term2);
        const D6 term3 = a2pb2 * c2 + s4, term4 = term2 * s2;
        const D6 rp2 = rs2 * (term3 - term4) / (term3 + term4);
        const D6 Fm = dc<6>(0.5f) * (rp2 + rs2);
        // dielectric Fresnel (cos > 0)
        const D6 sc = one / de;
        const D6 cos_t = dsqrt(one - s2 * (sc * sc));
        const D6 rs = (c - de * cos_t) / (c + de * cos_t);
        const D6 rp = (de * c - cos_t) / (de * c + cos_t);
        const D6 Fd = dc<6>(0.5f) * (rs * rs + rp * rp);

        const D6 M = Fm * U;                                            // metallic   per unit light and specular albedo
        const D6 Dl = Fd * Dn * (g1 * g1) / (dc<6>(4.0f) * c) * U;      // dielectric per unit light and specular albedo
        const D6 Df = U * c * dc<6>(T12 * T12 * inv_eta2 / (fd * pi_f)); // diffuse    per unit light and diffuse albedo
        float Am = 0.f, Ad = 0.f, Af = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const size_t q = 3 * (size_t)p + k;
            const float g_r = a.g_rgb ? a.g_rgb[q] : 0.f, g_s = a.g_spec ? a.g_spec[q] : 0.f;
            const float gm = (a.g_met ? a.g_met[q] : 0.f) + g_s + g_r;
            const float gd = (a.g_die ? a.g_die[q] : 0.f) + g_s + g_r;
            const float ks_in = a.ks[q], kd_in = a.kd[q];
            const float ks = fmaxf(ks_in, 0.00001f), kd = fmaxf(kd_in, 0.00001f);
            if (a.d_ks) a.d_ks[q] = ks_in >= 0.00001f ? lightf * (gm * M.v + gd * Dl.v) : 0.f;
            if (a.d_kd) a.d_kd[q] = kd_in >= 0.00001f ? lightf * (g_r * Df.v) : 0.f;
            Am += gm * ks; Ad += gd * ks; Af += g_r * kd;
        }
        light_acc = use_env ? 0.0f : (Am * M.v + Ad * Dl.v) + Af * Df.v;
        float dv[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) dv[k] = live[k] ? lightf * ((Am * M.d[k] + Ad * Dl.d[k]) + Af * Df.d[k]) : 0.f;
        if (a.d_rough) a.d_rough[p] = dv[1];
        if (a.d_dist) a.d_dist[p] = use_env ? 0.0f : dv[2];
        if (a.d_env) a.d_env[p] = use_env ? dv[2] + ((a.g_env && live[2]) ? a.g_env[p] : 0.0f) : 0.0f;  // "env_light" output = the clamp itself
        if (a.d_m_eta) a.d_m_eta[p] = dv[3];
        if (a.d_m_k) a.d_m_k[p] = dv[4];
        if (a.d_d_eta) a.d_d_eta[p] = dv[5];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (a.d_nrm) a.d_nrm[3 * (size_t)p + k] = dv[0] * vv[k];
            if (a.d_view) a.d_view[3 * (size_t)p + k] = dv[0] * nn[k];
        }
    }
    if (a.d_light) {
        for (int o = 32; o > 0; o >>= 1) light_acc += __shfl_xor(light_acc, o, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(a.d_light, light_acc);
    }
}

// ---- the four simple co-located heads (models/renderer_ggx.py:149-395; kinds as iron_coloc_head) --------------------------------
struct HeadBackArgs {
    const float *dist, *nrm, *view, *kd, *ks, *rough, *g_diff, *g_spec, *g_rgb;
    float *d_light, *d_dist, *d_nrm, *d_view, *d_kd, *d_ks, *d_rough;
    float light, eta, k;
    int kind, n;
};

__global__ void k_coloc_head_back(HeadBackArgs a) {
    typedef Dual<3> D3d;  // (cos, roughness, distance)
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    float light_acc = 0.0f;
    if (p < a.n) {
        const float* nn = a.nrm + 3 * (size_t)p;
        const float* vv = a.view + 3 * (size_t)p;
        const float raw_dot = (vv[0] * nn[0] + vv[1] * nn[1]) + vv[2] * nn[2];
        const float cdot = fminf(fmaxf(raw_dot, 0.00001f), 0.99999f);
        const bool dot_live = raw_dot >= 0.00001f && raw_dot <= 0.99999f;
        const float r_in = a.rough ? a.rough[p] : 1.0f;
        const bool rough_live = a.kind == 3 && r_in >= 0.0001f;
        const D3d c = dvar<3>(cdot, 0), al = dvar<3>(fmaxf(r_in, 0.0001f), 1), ds = dvar<3>(a.dist[p], 2);
        const D3d one = dc<3>(1.0f);
        const D3d U = one / (ds * ds + dc<3>(1e-10f));
        D3d S;  // specular per unit light and specular albedo
        if (a.kind == 0) {
            S = U * dc<3>(0.04f);
        } else if (a.kind == 1) {
            S = U * dc<3>((float)(0.04 + 0.96 * 0.96 * 0.04 / (1.0 - 0.04 * 0.04)));
        } else {
            const D3d c2 = c * c, s2 = one - c2, s4 = s2 * s2;
            const D3d me = dc<3>(a.eta), mk = dc<3>(a.k);
            const D3d temp1 = me * me - mk * mk - s2;
            const D3d a2pb2 = dsqrt(temp1 * temp1 + dc<3>(4.0f) * mk * mk * me * me);
            const D3d aa = dsqrt(dc<3>(0.5f) * (a2pb2 + temp1));
            const D3d term1 = a2pb2 + c2, term2 = dc<3>(2.0f) * aa * c;
            const D3d rs2 = (term1 - term2) / (term1 + term2);
            const D3d term3 = a2pb2 * c2 + s4, term4 = term2 * s2;
            const D3d rp2 = rs2 * (term3 - term4) / (term3 + term4);
            S = U * dc<3>(0.5f) * (rp2 + rs2);
            if (a.kind == 3) {
                const D3d a2 = al * al;
                const D3d root = c2 + s2 / (a2 + dc<3>(1e-10f));
                const D3d Dn = one / (dc<3>(3.14159274101257324219f) * a2 * root * root + dc<3>(1e-10f));
                const D3d rt = al * (dsqrt(s2) / (c + dc<3>(1e-10f)));
                const D3d g1 = dc<3>(2.0f) / (one + dsqrt(rt * rt + one));
                S = S * Dn * (g1 * g1) / (dc<3>(4.0f) * c + dc<3>(1e-10f));
            }
        }
        const D3d Df = U * dc<3>(0.0001f);  // diffuse per unit light and diffuse albedo
        float As = 0.f, Ad = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const size_t q = 3 * (size_t)p + k;
            const float g_r = a.g_rgb ? a.g_rgb[q] : 0.f;
            const float gd = (a.g_diff ? a.g_diff[q] : 0.f) + g_r, gs = (a.g_spec ? a.g_spec[q] : 0.f) + g_r;
            if (a.d_kd) a.d_kd[q] = gd * a.light * Df.v;
            if (a.d_ks) a.d_ks[q] = gs * a.light * S.v;
            Ad += gd * a.kd[q];
            As += gs * a.ks[q];
        }
        light_acc = As * S.v + Ad * Df.v;
        const float dcos = dot_live ? a.light * (As * S.d[0] + Ad * Df.d[0]) : 0.f;
        if (a.d_rough) a.d_rough[p] = rough_live ? a.light * (As * S.d[1]) : 0.f;
        if (a.d_dist) a.d_dist[p] = a.light * (As * S.d[2] + Ad * Df.d[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (a.d_nrm) a.d_nrm[3 * (size_t)p + k] = dcos * vv[k];
            if (a.d_view) a.d_view[3 * (size_t)p + k] = dcos * nn[k];
        }
    }
    if (a.d_light) {
        for (int o = 32; o > 0; o >>= 1) light_acc += __shfl_xor(light_acc, o, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(a.d_light, light_acc);
    }
}

// ---- NeRF background field (models/fields.py:243-327), parameter gradients --------------------------------------------------------
// PE of a `dims`-wide input: [x, sin(2^0 x), cos(2^0 x), ...], each block `dims` wide
__device__ __forceinline__ float pe_val(const float* x, int c, int dims) {
    if (c < dims) return x[c];
    const int k = (c - dims) / (2 * dims), r = (c - dims) % (2 * dims);
    const float a = x[r % dims] * (float)(1 << k);
    return r < dims ? sinf(a) : cosf(a);
}

// dst[p, col0 + c] = PE(src[p, :dims])[c]
__global__ void k_pe_rows(const float* __restrict__ src, int dims, int L, int m, float* __restrict__ dst, int ld, int col0) {
    const int D = dims + 2 * dims * L;
    GRID_STRIDE(i, (int64_t)m * D) {
        const int p = (int)(i / D), c = (int)(i % D);
        dst[(size_t)p * ld + col0 + c] = pe_val(src + (size_t)p * dims, c, dims);
    }
}

// dst[p, c] = src[p, c] (+ bias[c]); db[c] += column sums of src when db != NULL  (strip form)
__global__ void k_bias_copy_colsum(const float* __restrict__ src, int ld_src, const float* __restrict__ bias, int m, int out, float* __restrict__ dst,
                                   int ld_dst, float* __restrict__ db) {
    STRIP_PROLOGUE(m, out)
    for (int p = r0; p < r1; ++p) {
        const float v = src[(size_t)p * ld_src + c];
        if (dst) dst[(size_t)p * ld_dst + c] = bias ? v + bias[c] : v;
        colsum += v;
    }
    if (db) atomicAdd(&db[c], colsum);
}

// dst[p, 0:w) += a[p] * row[0:w)   (the alpha head's contribution to dL/dh: one output row)
__global__ void k_add_outer(const float* __restrict__ a, const float* __restrict__ row, int m, int w, float* __restrict__ dst, int ld) {
    GRID_STRIDE(i, (int64_t)m * w) {
        const int p = (int)(i / w), c = (int)(i % w);
        dst[(size_t)p * ld + c] += a[p] * row[c];
    }
}

constexpr int kNerfChunk = 131072;

struct NerfPlan {
    int D, W, in_p, in_v, nl, m_max;
    float *Wt[kMaxLayers], *dW[kMaxLayers], *db[kMaxLayers], *X[kMaxLayers], *Z[kMaxLayers];
    float *HV, *ZV, *AV, *dA, *dB, *partial, *scratch;
    size_t bytes, partial_floats;
};

static int nerf_plan(const iron_nerf_train_desc* d, int64_t n, void* ws, NerfPlan& P) {
    if (!d || !d->layers || d->D < 2 || d->D + 4 > kMaxLayers || d->W < 1 || d->d_in < 1 || d->d_in_view < 1) return IRON_ERR_BAD_ARG;
    P.D = d->D; P.W = d->W; P.nl = d->D + 4;
    P.in_p = d->d_in + 2 * d->d_in * (d->multires > 0 ? d->multires : 0);
    P.in_v = d->d_in_view + 2 * d->d_in_view * (d->multires_view > 0 ? d->multires_view : 0);
    const iron_train_layer* ly = d->layers;
    for (int i = 0; i < d->D; ++i) {
        const int expect = i == 0 ? P.in_p : (d->W + ((i - 1) == d->skip ? P.in_p : 0));
        if (ly[i].in_dim != expect || ly[i].out_dim != d->W) return IRON_ERR_UNSUPPORTED;
    }
    const int hD = d->W + ((d->D - 1) == d->skip ? P.in_p : 0);
    if (ly[d->D].in_dim != hD || ly[d->D].out_dim != 1) return IRON_ERR_UNSUPPORTED;                       // alpha
    if (ly[d->D + 1].in_dim != hD || ly[d->D + 1].out_dim != d->W) return IRON_ERR_UNSUPPORTED;            // feature
    if (ly[d->D + 2].in_dim != d->W + P.in_v) return IRON_ERR_UNSUPPORTED;                                  // view layer
    if (ly[d->D + 3].in_dim != ly[d->D + 2].out_dim) return IRON_ERR_UNSUPPORTED;                           // rgb
    P.m_max = (int)(n < kNerfChunk ? (n > 0 ? n : 1) : kNerfChunk);
    const size_t R = (size_t)P.m_max;
    Bump b(ws);
    int maxw = 0;
    for (int l = 0; l < P.nl; ++l) {
        const size_t wn = (size_t)ly[l].out_dim * ly[l].in_dim;
        P.Wt[l] = b.take(wn);
        P.dW[l] = b.take(wn);
        P.db[l] = b.take(ly[l].out_dim);
        maxw = max(maxw, max(ly[l].out_dim, ly[l].in_dim));
    }
    for (int i = 0; i <= d->D; ++i) P.X[i] = b.take(R * (i < d->D ? ly[i].in_dim : hD));  // X[D] = h_D
    for (int i = 0; i < d->D; ++i) P.Z[i] = b.take(R * d->W);
    P.HV = b.take(R * ly[d->D + 2].in_dim);
    P.ZV = b.take(R * ly[d->D + 2].out_dim);
    P.AV = b.take(R * ly[d->D + 2].out_dim);
    P.dA = b.take(R * maxw);
    P.dB = b.take(R * maxw);
    P.partial = b.take((size_t)kSplitK * maxw * maxw + 64);  // + 64 floats behind it: the GEMMs' scratch (GemmCtx)
    P.scratch = P.partial + (size_t)kSplitK * maxw * maxw;
    P.partial_floats = (size_t)kSplitK * maxw * maxw;
    P.bytes = b.off + 256;
    return IRON_OK;
}

static int nerf_backward(const iron_nerf_train_desc* d, const float* pts, const float* views, int64_t n, const float* d_alpha, const float* d_rgb,
                         void* ws, size_t ws_bytes, hipStream_t st) {
    NerfPlan P;
    TR_TRY(nerf_plan(d, n, ws, P));
    if (P.bytes > ws_bytes) return IRON_ERR_WORKSPACE;
    const GemmCtx h{st, P.scratch, P.partial, P.partial_floats};
    const iron_train_layer* ly = d->layers;
    const int D = P.D, W = P.W, iA = D, iF = D + 1, iV = D + 2, iC = D + 3;
    const int wv = ly[iV].out_dim, hD = ly[iA].in_dim;
    for (int l = 0; l < P.nl; ++l) {
        hipLaunchKernelGGL(k_wn_fold, dim3(ly[l].out_dim), dim3(64), 0, st, ly[l].weight_v, ly[l].weight_g, ly[l].out_dim, ly[l].in_dim, P.Wt[l]);
        TR_HIP(hipMemsetAsync(P.db[l], 0, sizeof(float) * ly[l].out_dim, st));
        if (n == 0 || (l == iA && !d_alpha) || ((l == iF || l == iV || l == iC) && !d_rgb))
            TR_HIP(hipMemsetAsync(P.dW[l], 0, sizeof(float) * ly[l].out_dim * ly[l].in_dim, st));
    }
    for (int64_t p0 = 0; p0 < n; p0 += P.m_max) {
        const int m = (int)((n - p0) < P.m_max ? (n - p0) : P.m_max);
        const float beta = p0 == 0 ? 0.0f : 1.0f;
        // ---- forward, keeping every layer's input and pre-activation
        hipLaunchKernelGGL(k_pe_rows, grid1((int64_t)m * P.in_p), dim3(256), 0, st, pts + (size_t)p0 * d->d_in, d->d_in, d->multires > 0 ? d->multires : 0, m,
                           P.X[0], P.in_p, 0);
        for (int i = 0; i < D; ++i) {
            const int in = ly[i].in_dim;
            TR_TRY(gemm_rm(h, false, true, m, W, in, P.X[i], in, P.Wt[i], in, 0.0f, P.Z[i], W));
            const bool skip = (i == d->skip);
            const int ld_next = i + 1 < D ? ly[i + 1].in_dim : hD;
            const int off = skip ? P.in_p : 0;
            hipLaunchKernelGGL(k_relu_act, grid1((int64_t)m * W), dim3(256), 0, st, P.Z[i], ly[i].bias, m, W, 1.0f, P.X[i + 1] + off, ld_next);
            if (skip) hipLaunchKernelGGL(k_copy_cols, grid1((int64_t)m * P.in_p), dim3(256), 0, st, P.X[0], P.in_p, m, P.in_p, 1.0f, P.X[i + 1], ld_next, 0);
        }
        if (d_rgb) {
            const int inv = ly[iV].in_dim;
            TR_TRY(gemm_rm(h, false, true, m, W, hD, P.X[D], hD, P.Wt[iF], hD, 0.0f, P.dA, W));  // feature (pre-bias) in dA
            hipLaunchKernelGGL(k_bias_copy_colsum, strip_grid(m, W), dim3(256), 0, st, P.dA, W, ly[iF].bias, m, W, P.HV, inv, (float*)nullptr);
            hipLaunchKernelGGL(k_pe_rows, grid1((int64_t)m * P.in_v), dim3(256), 0, st, views + (size_t)p0 * d->d_in_view, d->d_in_view,
                               d->multires_view > 0 ? d->multires_view : 0, m, P.HV, inv, W);
            TR_TRY(gemm_rm(h, false, true, m, wv, inv, P.HV, inv, P.Wt[iV], inv, 0.0f, P.ZV, wv));
            hipLaunchKernelGGL(k_relu_act, grid1((int64_t)m * wv), dim3(256), 0, st, P.ZV, ly[iV].bias, m, wv, 1.0f, P.AV, wv);
        }
        // ---- reverse
        TR_HIP(hipMemsetAsync(P.dA, 0, sizeof(float) * (size_t)m * hD, st));  // dA = dL/dh_D
        if (d_rgb) {
            const float* g = d_rgb + (size_t)p0 * 3;
            const int inv = ly[iV].in_dim;
            TR_TRY(gemm_dw(h, st, 3, wv, m, g, P.AV, beta, P.dW[iC], P.partial));
            hipLaunchKernelGGL(k_bias_copy_colsum, strip_grid(m, 3), dim3(256), 0, st, g, 3, (const float*)nullptr, m, 3, (float*)nullptr, 0, P.db[iC]);
            TR_TRY(gemm_rm(h, false, false, m, wv, 3, g, 3, P.Wt[iC], wv, 0.0f, P.dB, wv));                               // dL/d av
            hipLaunchKernelGGL(k_relu_back, strip_grid(m, wv), dim3(256), 0, st, P.dB, wv, P.ZV, m, wv, 1.0f, P.AV, P.db[iV], (float*)nullptr);  // AV <- dL/d zv
            TR_TRY(gemm_dw(h, st, wv, inv, m, P.AV, P.HV, beta, P.dW[iV], P.partial));
            TR_TRY(gemm_rm(h, false, false, m, inv, wv, P.AV, wv, P.Wt[iV], inv, 0.0f, P.dB, inv));                        // dL/d hv; [:, :W] = dL/d feature
            hipLaunchKernelGGL(k_bias_copy_colsum, strip_grid(m, W), dim3(256), 0, st, P.dB, inv, (const float*)nullptr, m, W, P.HV, W, P.db[iF]);                                                                                // HV[:, :W] (ld W) <- dL/d feature
            TR_TRY(gemm_dw(h, st, W, hD, m, P.HV, P.X[D], beta, P.dW[iF], P.partial));
            TR_TRY(gemm_rm(h, false, false, m, hD, W, P.HV, W, P.Wt[iF], hD, 0.0f, P.dA, hD));
        }
        if (d_alpha) {
            const float* g = d_alpha + p0;
            TR_TRY(gemm_dw(h, st, 1, hD, m, g, P.X[D], beta, P.dW[iA], P.partial));
            hipLaunchKernelGGL(k_bias_copy_colsum, strip_grid(m, 1), dim3(256), 0, st, g, 1, (const float*)nullptr, m, 1, (float*)nullptr, 0, P.db[iA]);
            hipLaunchKernelGGL(k_add_outer, grid1((int64_t)m * hD), dim3(256), 0, st, g, P.Wt[iA], m, hD, P.dA, hD);
        }
        for (int i = D - 1; i >= 0; --i) {
            const int in = ly[i].in_dim;
            const int ld = i + 1 < D ? ly[i + 1].in_dim : hD;
            const int off = (i == d->skip) ? P.in_p : 0;
            hipLaunchKernelGGL(k_relu_back, strip_grid(m, W), dim3(256), 0, st, P.dA + off, ld, P.Z[i], m, W, 1.0f, P.dB, P.db[i], (float*)nullptr);  // dB = dL/d z_i
            TR_TRY(gemm_dw(h, st, W, in, m, P.dB, P.X[i], beta, P.dW[i], P.partial));
            if (i > 0) TR_TRY(gemm_rm(h, false, false, m, in, W, P.dB, W, P.Wt[i], in, 0.0f, P.dA, in));
        }
    }
    for (int l = 0; l < P.nl; ++l) {
        hipLaunchKernelGGL(k_wn_back, dim3(ly[l].out_dim), dim3(64), 0, st, ly[l].weight_v, ly[l].weight_g, P.dW[l], ly[l].out_dim, ly[l].in_dim,
                           ly[l].d_weight_v, ly[l].d_weight_g);
        TR_HIP(hipMemcpyAsync(ly[l].d_bias, P.db[l], sizeof(float) * ly[l].out_dim, hipMemcpyDeviceToDevice, st));
    }
    TR_HIP(hipGetLastError());
    return IRON_OK;
}

// ---- NeuS compositing (models/renderer.py:279-344 + :174-178), reverse pass ------------------------------------------------------
// One thread per ray: the forward is replayed keeping alpha_j and the transmittance T_j of the row (<= 192 samples), then a
// reverse scan turns dL/dw_j into dL/dalpha_j (w_j = alpha_j T_j, T_{j+1} = T_j (1 - alpha_j + 1e-7)) and from there into
// dL/d(sdf, gradient, sample colour, 1/s, background density / colour).
constexpr int kNeusMax = 192;
__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

struct NeusBackArgs {
    iron_neus_composite_args f;  // the forward's inputs (outputs unused)
    const float *d_color, *d_weight_sum, *d_weights, *d_gradient_error, *relax_count;
    float *d_sdf, *d_grad, *d_sample_color, *d_inv_s, *d_bg_density, *d_bg_color;
};

__global__ void k_neus_composite_back(NeusBackArgs a) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const iron_neus_composite_args& f = a.f;
    float dinv_acc = 0.0f;
    if (r < f.n) {
        const bool bg = f.bg_density != nullptr;
        const int m = f.m, mt = bg ? f.mo : f.m;
        float al[kNeusMax], T[kNeusMax];
        float trans = 1.0f;
        for (int j = 0; j < mt; ++j) {  // forward replay: composited alpha and transmittance
            float alpha, bga = 0.0f;
            if (bg) {
                const size_t q = (size_t)r * f.mo + j;
                const float d = f.bg_density[q];
                bga = 1.0f - expf(-(d > 20.0f ? d : log1pf(expf(d))) * f.bg_dists[q]);
            }
            if (j < m) {
                const size_t q = (size_t)r * m + j;
                const float tc = (f.dirs[3 * q] * f.grad[3 * q] + f.dirs[3 * q + 1] * f.grad[3 * q + 1]) + f.dirs[3 * q + 2] * f.grad[3 * q + 2];
                const float ic = -(fmaxf(-tc * 0.5f + 0.5f, 0.0f) * (1.0f - f.cos_anneal_ratio) + fmaxf(-tc, 0.0f) * f.cos_anneal_ratio);
                const float s = f.sdf[q], dist = f.dists[q];
                const float pc = sigm((s - ic * dist * 0.5f) * f.inv_s), nc = sigm((s + ic * dist * 0.5f) * f.inv_s);
                alpha = fminf(fmaxf((pc - nc + 1e-5f) / (pc + 1e-5f), 0.0f), 1.0f);
                if (bg) {
                    const float px = f.pts[3 * q], py = f.pts[3 * q + 1], pz = f.pts[3 * q + 2];
                    const float ins = sqrtf((px * px + py * py) + pz * pz) < 1.0f ? 1.0f : 0.0f;
                    alpha = alpha * ins + bga * (1.0f - ins);
                }
            } else {
                alpha = bga;
            }
            al[j] = alpha;
            T[j] = trans;
            trans = trans * (1.0f - alpha + 1e-7f);
        }
        const float dcol[3] = {a.d_color ? a.d_color[3 * (size_t)r] : 0.f, a.d_color ? a.d_color[3 * (size_t)r + 1] : 0.f,
                               a.d_color ? a.d_color[3 * (size_t)r + 2] : 0.f};
        const float dws = a.d_weight_sum ? a.d_weight_sum[r] : 0.0f;
        float bgr[3] = {0.f, 0.f, 0.f};
        if (f.background_rgb) { bgr[0] = f.background_rgb[0]; bgr[1] = f.background_rgb[1]; bgr[2] = f.background_rgb[2]; }
        const float dge = a.d_gradient_error ? a.d_gradient_error[0] : 0.0f;
        const float inv_cnt = a.relax_count ? 1.0f / (a.relax_count[0] + 1e-5f) : 0.0f;
        float suffix = 0.0f;  // sum over k > j of Gw_k w_k
        for (int j = mt - 1; j >= 0; --j) {
            const float w = al[j] * T[j];
            // the colour this sample contributed
            float c[3], ins = 1.0f;
            size_t q = 0, qb = (size_t)r * (bg ? f.mo : 1) + j;
            if (j < m) {
                q = (size_t)r * m + j;
                if (bg) {
                    const float px = f.pts[3 * q], py = f.pts[3 * q + 1], pz = f.pts[3 * q + 2];
                    ins = sqrtf((px * px + py * py) + pz * pz) < 1.0f ? 1.0f : 0.0f;
                }
                for (int k = 0; k < 3; ++k) c[k] = bg ? f.color[3 * q + k] * ins + f.bg_color[3 * qb + k] * (1.0f - ins) : f.color[3 * q + k];
            } else {
                ins = 0.0f;
                for (int k = 0; k < 3; ++k) c[k] = f.bg_color[3 * qb + k];
            }
            float Gw = dws + (a.d_weights ? a.d_weights[(size_t)r * mt + j] : 0.0f);
            for (int k = 0; k < 3; ++k) Gw += dcol[k] * (c[k] - bgr[k]);
            const float dalpha = Gw * T[j] - suffix / (1.0f - al[j] + 1e-7f);
            suffix += Gw * w;
            // colours
            if (j < m && a.d_sample_color)
                for (int k = 0; k < 3; ++k) a.d_sample_color[3 * q + k] = dcol[k] * w * ins;
            if (bg && a.d_bg_color)
                for (int k = 0; k < 3; ++k) a.d_bg_color[3 * qb + k] = dcol[k] * w * (1.0f - ins);
            // background alpha -> density
            if (bg && a.d_bg_density) {
                const float d = f.bg_density[qb], bd = f.bg_dists[qb];
                const float sp = d > 20.0f ? d : log1pf(expf(d));
                const float dsp = d > 20.0f ? 1.0f : sigm(d);
                a.d_bg_density[qb] = dalpha * (1.0f - ins) * expf(-sp * bd) * bd * dsp;
            }
            if (j < m) {
                const float gx = f.grad[3 * q], gy = f.grad[3 * q + 1], gz = f.grad[3 * q + 2];
                const float dx = f.dirs[3 * q], dy = f.dirs[3 * q + 1], dz = f.dirs[3 * q + 2];
                const float tc = (dx * gx + dy * gy) + dz * gz;
                const float ca = f.cos_anneal_ratio;
                const float a1 = -tc * 0.5f + 0.5f, b1 = -tc;
                const float ic = -(fmaxf(a1, 0.0f) * (1.0f - ca) + fmaxf(b1, 0.0f) * ca);
                const float s = f.sdf[q], dist = f.dists[q];
                const float ep = s - ic * dist * 0.5f, en = s + ic * dist * 0.5f;
                const float pc = sigm(ep * f.inv_s), nc = sigm(en * f.inv_s);
                const float araw = (pc - nc + 1e-5f) / (pc + 1e-5f);
                const float da = (araw >= 0.0f && araw <= 1.0f) ? dalpha * ins : 0.0f;
                const float ipc = 1.0f / (pc + 1e-5f);
                const float dpc = da * (ipc - araw * ipc), dnc = -da * ipc;
                const float dup = dpc * pc * (1.0f - pc), dun = dnc * nc * (1.0f - nc);
                dinv_acc += dup * ep + dun * en;
                const float dep = dup * f.inv_s, den = dun * f.inv_s;
                if (a.d_sdf) a.d_sdf[q] = dep + den;
                const float dic = (den - dep) * dist * 0.5f;
                const float dtc = dic * ((a1 > 0.0f ? 0.5f * (1.0f - ca) : 0.0f) + (b1 > 0.0f ? ca : 0.0f));
                float dg[3] = {dtc * dx, dtc * dy, dtc * dz};
                if (dge != 0.0f) {  // eikonal statistic: sum relax (|g|-1)^2 / (sum relax + 1e-5)
                    const float px = f.pts[3 * q], py = f.pts[3 * q + 1], pz = f.pts[3 * q + 2];
                    if (sqrtf((px * px + py * py) + pz * pz) < 1.2f) {
                        const float gn = sqrtf((gx * gx + gy * gy) + gz * gz);
                        const float k = dge * inv_cnt * 2.0f * (gn - 1.0f) / gn;
                        dg[0] += k * gx; dg[1] += k * gy; dg[2] += k * gz;
                    }
                }
                if (a.d_grad) { a.d_grad[3 * q] = dg[0]; a.d_grad[3 * q + 1] = dg[1]; a.d_grad[3 * q + 2] = dg[2]; }
            }
        }
    }
    if (a.d_inv_s) {
        for (int o = 32; o > 0; o >>= 1) dinv_acc += __shfl_xor(dinv_acc, o, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(a.d_inv_s, dinv_acc);
    }
}

}  // namespace iron_train

using namespace iron_train;

extern "C" int iron_coloc_head_backward(int32_t kind, float light, float eta, float k, const float* distance, const float* normal,
                                        const float* viewdir, const float* diffuse_albedo, const float* specular_albedo, const float* roughness,
                                        int64_t n, const float* d_diffuse_rgb, const float* d_specular_rgb, const float* d_rgb, float* d_light,
                                        float* d_distance, float* d_normal, float* d_viewdir, float* d_diffuse_albedo, float* d_specular_albedo,
                                        float* d_roughness, void* stream) {
    if (n < 0 || kind < 0 || kind > 3) return IRON_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (d_light) TR_HIP(hipMemsetAsync(d_light, 0, sizeof(float), st));
    if (n == 0) return IRON_OK;
    if (!distance || !normal || !viewdir || !diffuse_albedo || !specular_albedo || (kind == 3 && !roughness)) return IRON_ERR_BAD_ARG;
    HeadBackArgs a;
    a.dist = distance; a.nrm = normal; a.view = viewdir; a.kd = diffuse_albedo; a.ks = specular_albedo; a.rough = kind == 3 ? roughness : nullptr;
    a.g_diff = d_diffuse_rgb; a.g_spec = d_specular_rgb; a.g_rgb = d_rgb;
    a.d_light = d_light; a.d_dist = d_distance; a.d_nrm = d_normal; a.d_view = d_viewdir; a.d_kd = d_diffuse_albedo; a.d_ks = d_specular_albedo;
    a.d_rough = d_roughness;
    a.light = light; a.eta = eta; a.k = k; a.kind = kind; a.n = (int)n;
    hipLaunchKernelGGL(k_coloc_head_back, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, a);
    TR_HIP(hipGetLastError());
    return IRON_OK;
}

extern "C" size_t iron_nerf_backward_workspace_bytes(const iron_nerf_train_desc* desc, int64_t n) {
    NerfPlan P;
    if (n < 0 || nerf_plan(desc, n, nullptr, P) != IRON_OK) return 0;
    return P.bytes;
}

extern "C" int iron_nerf_backward(const iron_nerf_train_desc* desc, const float* pts, const float* views, int64_t n, const float* d_alpha,
                                  const float* d_rgb, void* workspace, size_t workspace_bytes, void* stream) {
    if (n < 0 || !workspace || (n > 0 && (!pts || !views))) return IRON_ERR_BAD_ARG;
    return nerf_backward(desc, pts, views, n, d_alpha, d_rgb, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int iron_neus_composite_backward(const iron_neus_composite_args* fwd, const iron_neus_composite_grads* g, void* stream) {
    if (!fwd || !g || fwd->n < 0 || fwd->m < 1 || fwd->m > kNeusMax) return IRON_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (g->d_inv_s) TR_HIP(hipMemsetAsync(g->d_inv_s, 0, sizeof(float), st));
    if (fwd->n == 0) return IRON_OK;
    if (!fwd->dists || !fwd->pts || !fwd->dirs || !fwd->sdf || !fwd->grad || !fwd->color) return IRON_ERR_BAD_ARG;
    if (fwd->bg_density && (!fwd->bg_dists || !fwd->bg_color || fwd->mo < fwd->m || fwd->mo > kNeusMax)) return IRON_ERR_BAD_ARG;
    if (g->d_gradient_error && !g->relax_count) return IRON_ERR_BAD_ARG;
    NeusBackArgs a;
    a.f = *fwd;
    a.d_color = g->d_color; a.d_weight_sum = g->d_weight_sum; a.d_weights = g->d_weights; a.d_gradient_error = g->d_gradient_error;
    a.relax_count = g->relax_count;
    a.d_sdf = g->d_sdf; a.d_grad = g->d_grad; a.d_sample_color = g->d_sample_color; a.d_inv_s = g->d_inv_s;
    a.d_bg_density = g->d_bg_density; a.d_bg_color = g->d_bg_color;
    hipLaunchKernelGGL(k_neus_composite_back, dim3((unsigned)((fwd->n + 63) / 64)), dim3(64), 0, st, a);
    TR_HIP(hipGetLastError());
    return IRON_OK;
}


extern "C" int iron_composite_colocated_backward(float light, const float* distance, const float* normal, const float* viewdir,
                                                 const iron_composite_params* p, const float* tab_trans, const float* tab_diff, int64_t n,
                                                 const iron_composite_grads_in* g, const iron_composite_grads_out* o, void* stream) {
    if (n < 0 || !p || !g || !o) return IRON_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (o->d_light) TR_HIP(hipMemsetAsync(o->d_light, 0, sizeof(float), st));
    if (n == 0) return IRON_OK;
    if ((!distance && !p->env_light) || !normal || !viewdir || !p->diffuse_albedo || !p->specular_albedo || !p->specular_roughness || !p->metallic_eta ||
        !p->metallic_k || !p->dielectric_eta || !tab_trans || !tab_diff)
        return IRON_ERR_BAD_ARG;
    CompBackArgs a;
    a.dist = distance; a.nrm = normal; a.view = viewdir; a.kd = p->diffuse_albedo; a.ks = p->specular_albedo; a.rough = p->specular_roughness;
    a.m_eta = p->metallic_eta; a.m_k = p->metallic_k; a.d_eta = p->dielectric_eta; a.env = p->env_light; a.tab_trans = tab_trans;
    a.tab_diff = tab_diff;
    a.g_rgb = g->d_rgb; a.g_spec = g->d_specular_rgb; a.g_met = g->d_metallic_rgb; a.g_die = g->d_dielectric_rgb; a.g_env = g->d_env_light_out;
    a.d_light = o->d_light; a.d_dist = o->d_distance; a.d_nrm = o->d_normal; a.d_view = o->d_viewdir; a.d_kd = o->d_diffuse_albedo;
    a.d_ks = o->d_specular_albedo; a.d_rough = o->d_specular_roughness; a.d_m_eta = o->d_metallic_eta; a.d_m_k = o->d_metallic_k;
    a.d_d_eta = o->d_dielectric_eta; a.d_env = o->d_env_light;
    a.light = light; a.n = (int)n;
    hipLaunchKernelGGL(k_composite_back, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, a);
    TR_HIP(hipGetLastError());
    return IRON_OK;
}


// Sticky operand-range flag of the split-fp16 GEMMs (gemm_h2.h: g_gemm_range_flag).  Synchronises `stream`.
extern "C" int iron_train_numeric_status(int32_t reset, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    int v = 0;
    TR_HIP(hipMemcpyFromSymbolAsync(&v, HIP_SYMBOL(g_gemm_range_flag), sizeof(int), 0, hipMemcpyDeviceToHost, st));
    TR_HIP(hipStreamSynchronize(st));
    if (v && reset) {
        const int z = 0;
        TR_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_gemm_range_flag), &z, sizeof(int), 0, hipMemcpyHostToDevice, st));
        TR_HIP(hipStreamSynchronize(st));
    }
    return v ? IRON_ERR_RANGE : IRON_OK;
}

extern "C" int iron_train_last_hip_error(void) { return g_hip_error; }
extern "C" int iron_train_last_blas_status(void) { return g_blas_status; }  // kept for ABI stability: there is no BLAS any more, always 0

// The layer product of the backward passes on its own (tests, micro-benchmarks): row-major C[m,n] = op(A) op(B) + beta C.
// op_a / op_b: 0 = as stored ([m,k] / [k,n]), 1 = transposed ([k,m] / [n,k]).  Supported like inside the library: (0,1) forward
// recompute, (0,0) dX, (1,0) dW (split over K into `workspace` partial tiles when k is large).
extern "C" size_t iron_train_gemm_workspace_bytes(int32_t op_a, int32_t m, int32_t n) {
    const size_t partial = op_a ? (size_t)kSplitK * (size_t)m * (size_t)n : 0;   // split-K partial tiles: the dW shape only
    return (partial + 64 + 12 * 24 * 512) * sizeof(float) + 512;
}

extern "C" int iron_train_gemm(int32_t op_a, int32_t op_b, int32_t m, int32_t n, int32_t k, const float* A, int32_t lda, const float* B, int32_t ldb,
                               float beta, float* C, int32_t ldc, void* workspace, size_t workspace_bytes, void* stream) {
    if (m < 0 || n < 0 || k < 0 || (op_a && op_b)) return IRON_ERR_BAD_ARG;
    if (m == 0 || n == 0) return IRON_OK;
    if (!A || !B || !C || !workspace || workspace_bytes < iron_train_gemm_workspace_bytes(op_a, m, n)) return IRON_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)workspace;
    float* scratch = partial + (op_a ? (size_t)kSplitK * (size_t)m * (size_t)n : 0);
    const GemmCtx h{st, scratch, scratch + 64, (size_t)12 * 24 * 512};
    int rc;
    if (op_a && !op_b) rc = gemm_dw(h, st, m, n, k, A, B, beta, C, partial);   // A stored [k,m] (ld = m), B [k,n] (ld = n), C [m,n] (ld = n)
    else rc = gemm_rm(h, op_a != 0, op_b != 0, m, n, k, A, lda, B, ldb, beta, C, ldc);
    if (rc != IRON_OK) return rc;
    TR_HIP(hipGetLastError());
    return IRON_OK;
}

extern "C" size_t iron_sdf_backward_workspace_bytes(const iron_sdf_train_desc* desc, int64_t n) {
    SdfPlan P;
    if (n < 0 || sdf_plan(desc, n, nullptr, P) != IRON_OK) return 0;
    return P.bytes;
}

extern "C" int iron_sdf_backward(const iron_sdf_train_desc* desc, const float* x, int64_t n, const float* d_sdf, const float* d_feature,
                                 const float* d_gradient, void* workspace, size_t workspace_bytes, void* stream) {
    if (n < 0 || !workspace || (n > 0 && !x)) return IRON_ERR_BAD_ARG;
    return sdf_backward(desc, x, n, d_sdf, d_feature, d_gradient, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" size_t iron_render_backward_workspace_bytes(const iron_render_train_desc* desc, int64_t n) {
    RenderPlan P;
    if (n < 0 || render_plan(desc, n, nullptr, P) != IRON_OK) return 0;
    return P.bytes;
}

extern "C" int iron_render_backward(const iron_render_train_desc* desc, const float* points, const float* normals, const float* view_dirs,
                                    const float* features, int64_t n, const float* d_out, float* d_points, float* d_normals, float* d_view_dirs,
                                    float* d_features, void* workspace, size_t workspace_bytes, void* stream) {
    if (n < 0 || !workspace) return IRON_ERR_BAD_ARG;
    return render_backward(desc, points, normals, view_dirs, features, n, d_out, d_points, d_normals, d_view_dirs, d_features, workspace,
                           workspace_bytes, (hipStream_t)stream);
}

extern "C" int iron_ggx_colocated_backward(float light, const float* distance, const float* normal, const float* viewdir,
                                           const float* diffuse_albedo, const float* specular_albedo, const float* roughness,
                                           const float* tab_trans, const float* tab_diff, int64_t n, const float* d_diffuse_rgb,
                                           const float* d_specular_rgb, const float* d_rgb, float* d_light, float* d_distance, float* d_normal,
                                           float* d_viewdir, float* d_diffuse_albedo, float* d_specular_albedo, float* d_roughness, void* stream) {
    if (n < 0) return IRON_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (d_light) TR_HIP(hipMemsetAsync(d_light, 0, sizeof(float), st));
    if (n == 0) return IRON_OK;
    if (!distance || !normal || !viewdir || !diffuse_albedo || !specular_albedo || !roughness || !tab_trans || !tab_diff) return IRON_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_ggx_back, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, light, distance, normal, viewdir, diffuse_albedo, specular_albedo,
                       roughness, tab_trans, tab_diff, (int)n, d_diffuse_rgb, d_specular_rgb, d_rgb, d_light, d_distance, d_normal, d_viewdir,
                       d_diffuse_albedo, d_specular_albedo, d_roughness);
    TR_HIP(hipGetLastError());
    return IRON_OK;
}
