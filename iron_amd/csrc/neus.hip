// Per-ray stages of the stage-1 NeuS volume renderer (models/renderer.py:45-75, 151-453): sample
// placement, hierarchical up-sampling (inverse-CDF), sorted merge, logistic-CDF alpha and compositing.  One thread per
// ray, rows of <= kMaxSamples samples walked sequentially in the reference's order (cumsum / cumprod are sequential scans
// there too); the MLP evaluations in between are the batched kernels of sdf_forward.hip / shade.hip / nerf.hip.
// HBM-bound and tiny next to the MLPs (4096 rays x 160 samples), so no LDS staging: rows are contiguous per ray.
#include "iron_common.h"

namespace iron {

constexpr int kMaxSamples = 192;

static inline int ray_grid(int64_t n) {
    const int64_t b = (n + 63) / 64;
    return (int)(b < 16384 ? (b > 0 ? b : 1) : 16384);
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float softplusf_(float x) { return x > 20.0f ? x : log1pf(expf(x)); }  // F.softplus(beta=1, threshold=20)

// z[r][j] = near[r] + (far[r] - near[r]) * lin[j]   (renderer.py:357-358)
__global__ void k_neus_linspace(const float* __restrict__ near, const float* __restrict__ far, const float* __restrict__ lin, int n,
                                int m, float* __restrict__ z) {
    const int64_t total = (int64_t)n * m;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / m), j = (int)(i % m);
        z[i] = near[r] + (far[r] - near[r]) * lin[j];
    }
}

// z[r][j] = far[r] / rev[j] + offset   (renderer.py:380-381: the depths of the outside samples, rev = flipped linspace)
__global__ void k_neus_outside_z(const float* __restrict__ far, const float* __restrict__ rev, int n, int m, float offset,
                                 float* __restrict__ z) {
    const int64_t total = (int64_t)n * m;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        z[i] = far[i / m] / rev[i % m] + offset;
}

// lattice points of extract_fields (renderer.py:9-31): pts[((i*ny)+j)*nz+k] = (xs[i], ys[j], zs[k])  ('ij' meshgrid order)
__global__ void k_grid_points(const float* __restrict__ xs, const float* __restrict__ ys, const float* __restrict__ zs, int nx, int ny, int nz,
                              float* __restrict__ pts) {
    const int64_t total = (int64_t)nx * ny * nz;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % nz), j = (int)((i / nz) % ny), a = (int)(i / ((int64_t)nz * ny));
        pts[3 * i] = xs[a]; pts[3 * i + 1] = ys[j]; pts[3 * i + 2] = zs[k];
    }
}

// pts[r][j] = o[r] + d[r] * z[r][j]
__global__ void k_neus_points(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ z, int n, int m,
                              float* __restrict__ pts) {
    const int64_t total = (int64_t)n * m;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / m);
        const float t = z[i];
#pragma unroll
        for (int c = 0; c < 3; ++c) pts[3 * i + c] = o[3 * r + c] + d[3 * r + c] * t;
    }
}

// NeuSRenderer.up_sample (renderer.py:189-232) + sample_pdf(det=True) (:45-75) for one ray per thread
__global__ void k_neus_up_sample(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ z,
                                 const float* __restrict__ sdf, int n, int m, int n_imp, float inv_s, float* __restrict__ new_z) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const float* zr = z + (size_t)r * m;
    const float* sr = sdf + (size_t)r * m;
    float cdf[kMaxSamples];  // cdf[0] = 0, cdf[j+1] = cumulative pdf of section j (m-1 sections)
    float w[kMaxSamples];
    const float ox = o[3 * r], oy = o[3 * r + 1], oz = o[3 * r + 2], dx = d[3 * r], dy = d[3 * r + 1], dz = d[3 * r + 2];
    float prev_cos = 0.0f, trans = 1.0f, wsum = 0.0f;
    float rad_prev;
    {
        const float px = ox + dx * zr[0], py = oy + dy * zr[0], pz = oz + dz * zr[0];
        rad_prev = sqrtf((px * px + py * py) + pz * pz);
    }
    for (int j = 0; j < m - 1; ++j) {
        const float zn = zr[j + 1], zp = zr[j];
        const float px = ox + dx * zn, py = oy + dy * zn, pz = oz + dz * zn;
        const float rad_next = sqrtf((px * px + py * py) + pz * pz);
        const bool inside = (rad_prev < 1.0f) || (rad_next < 1.0f);
        rad_prev = rad_next;
        const float mid_sdf = (sr[j] + sr[j + 1]) * 0.5f;
        const float cos_val = (sr[j + 1] - sr[j]) / (zn - zp + 1e-5f);
        float c = fminf(prev_cos, cos_val);
        prev_cos = cos_val;
        c = fminf(fmaxf(c, -1e3f), 0.0f) * (inside ? 1.0f : 0.0f);
        const float dist = zn - zp;
        const float prev_esti = mid_sdf - c * dist * 0.5f, next_esti = mid_sdf + c * dist * 0.5f;
        const float prev_cdf = sigmoidf_(prev_esti * inv_s), next_cdf = sigmoidf_(next_esti * inv_s);
        const float alpha = (prev_cdf - next_cdf + 1e-5f) / (prev_cdf + 1e-5f);
        w[j] = alpha * trans + 1e-5f;  // sample_pdf: weights + 1e-5
        trans = trans * (1.0f - alpha + 1e-7f);
        wsum += w[j];
    }
    cdf[0] = 0.0f;
    float acc = 0.0f;
    for (int j = 0; j < m - 1; ++j) {
        acc += w[j] / wsum;
        cdf[j + 1] = acc;
    }
    // inverse CDF at u_k = (k + 0.5) / n_imp; bins = z (m entries), cdf has m entries
    int ind = 0;
    for (int k = 0; k < n_imp; ++k) {
        // torch.linspace(0.5/n, 1 - 0.5/n, n): start + k * step (the upper half is computed from the end, like torch)
        const float start = 0.5f / n_imp, end = 1.0f - 0.5f / n_imp;
        const float step = (end - start) / (float)(n_imp - 1);
        const float u = n_imp == 1 ? start : (k < n_imp / 2 ? start + step * k : end - step * (n_imp - 1 - k));
        while (ind < m && cdf[ind] <= u) ++ind;  // searchsorted(right=True): first index with cdf > u (cdf is non-decreasing)
        const int below = ind - 1 > 0 ? ind - 1 : 0;
        const int above = ind < m - 1 ? ind : m - 1;
        float denom = cdf[above] - cdf[below];
        if (denom < 1e-5f) denom = 1.0f;
        const float t = (u - cdf[below]) / denom;
        new_z[(size_t)r * n_imp + k] = zr[below] + t * (zr[above] - zr[below]);
    }
}

// sorted merge of two ascending rows (torch.sort of their concatenation, renderer.py:238-239) with optional payload
__global__ void k_neus_merge(const float* __restrict__ za, const float* __restrict__ sa, int ma, const float* __restrict__ zb,
                             const float* __restrict__ sb, int mb, int n, float* __restrict__ zo, float* __restrict__ so) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const float* a = za + (size_t)r * ma;
    const float* b = zb + (size_t)r * mb;
    int i = 0, j = 0;
    for (int k = 0; k < ma + mb; ++k) {
        const bool take_a = j >= mb || (i < ma && a[i] <= b[j]);
        zo[(size_t)r * (ma + mb) + k] = take_a ? a[i] : b[j];
        if (so) so[(size_t)r * (ma + mb) + k] = take_a ? sa[(size_t)r * ma + i] : sb[(size_t)r * mb + j];
        if (take_a) ++i; else ++j;
    }
}

// section lengths, mid points (+ the outside parametrisation) of render_core / render_core_outside (renderer.py:157-172, 265-277)
__global__ void k_neus_mid_points(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ z, int n, int m,
                                  float sample_dist, int outside, float* __restrict__ dists, float* __restrict__ pts,
                                  float* __restrict__ dirs) {
    const int64_t total = (int64_t)n * m;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / m), j = (int)(i % m);
        const float dist = j + 1 < m ? z[i + 1] - z[i] : sample_dist;
        const float mid = z[i] + dist * 0.5f;
        dists[i] = dist;
        float p[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) p[c] = o[3 * r + c] + d[3 * r + c] * mid;
        if (outside) {
            const float nr = fminf(fmaxf(sqrtf((p[0] * p[0] + p[1] * p[1]) + p[2] * p[2]), 1.0f), 1e10f);
            pts[4 * i] = p[0] / nr; pts[4 * i + 1] = p[1] / nr; pts[4 * i + 2] = p[2] / nr; pts[4 * i + 3] = 1.0f / nr;
        } else {
            pts[3 * i] = p[0]; pts[3 * i + 1] = p[1]; pts[3 * i + 2] = p[2];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) dirs[3 * i + c] = d[3 * r + c];
    }
}

// need[r][j] = 1 where the compositing reads the background field at sample j of the fed row: every outside sample (j >= m), and an
// inside sample only when its section mid point is NOT inside the unit sphere (renderer.py:300-312 blends with (1 - inside_sphere));
// the same expression as k_neus_composite's `inside`, so the two can never disagree.
__global__ void k_neus_need_background(const float* __restrict__ pts, int n, int m, int mo, uint8_t* __restrict__ need) {
    const int64_t total = (int64_t)n * mo;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / mo), j = (int)(i % mo);
        uint8_t v = 1;
        if (j < m) {
            const size_t q = (size_t)r * m + j;
            const float px = pts[3 * q], py = pts[3 * q + 1], pz = pts[3 * q + 2];
            v = sqrtf((px * px + py * py) + pz * pz) < 1.0f ? 0 : 1;
        }
        need[i] = v;
    }
}

struct NeusCompositeArgs {
    const float *dists, *pts, *dirs, *sdf, *grad, *color;  // [n*m], [n*m,3], [n*m,3], [n*m], [n*m,3], [n*m,3]
    const float *bg_dists, *bg_density, *bg_color;          // outside: [n*mo], [n*mo], [n*mo,3] or null
    const float* background_rgb;                            // [3] or null
    const float* bg_alpha;                                  // [n*mo] or null: the outside pass's alpha given directly (render_core's own signature)
    int n, m, mo;
    float inv_s, cos_anneal;
    float *out_color, *weights, *cdf, *inside, *weight_sum, *weight_max, *gerr_acc;  // gerr_acc[2]: sum relax*err, sum relax
};

// render_core (renderer.py:279-344) with the background of render_core_outside (:174-178) folded in
__global__ void k_neus_composite(NeusCompositeArgs a) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    float e_sum = 0.0f, e_cnt = 0.0f;
    if (r < a.n) {
        const bool has_bg = a.bg_density || a.bg_alpha;
        const int mt = has_bg ? a.mo : a.m;  // total samples of the composited row
        float trans = 1.0f, wsum = 0.0f, wmax = 0.0f, col[3] = {0.f, 0.f, 0.f};
        for (int j = 0; j < mt; ++j) {
            float alpha = 0.0f, c[3] = {0.f, 0.f, 0.f};
            float bg_alpha = 0.0f;
            if (a.bg_alpha) {
                bg_alpha = a.bg_alpha[(size_t)r * a.mo + j];
            } else if (a.bg_density) {
                const size_t q = (size_t)r * a.mo + j;
                bg_alpha = 1.0f - expf(-softplusf_(a.bg_density[q]) * a.bg_dists[q]);
            }
            if (j < a.m) {
                const size_t q = (size_t)r * a.m + j;
                const float gx = a.grad[3 * q], gy = a.grad[3 * q + 1], gz = a.grad[3 * q + 2];
                const float true_cos = (a.dirs[3 * q] * gx + a.dirs[3 * q + 1] * gy) + a.dirs[3 * q + 2] * gz;
                const float iter_cos = -(fmaxf(-true_cos * 0.5f + 0.5f, 0.0f) * (1.0f - a.cos_anneal) + fmaxf(-true_cos, 0.0f) * a.cos_anneal);
                const float s = a.sdf[q], dist = a.dists[q];
                const float est_next = s + iter_cos * dist * 0.5f, est_prev = s - iter_cos * dist * 0.5f;
                const float prev_cdf = sigmoidf_(est_prev * a.inv_s), next_cdf = sigmoidf_(est_next * a.inv_s);
                alpha = fminf(fmaxf((prev_cdf - next_cdf + 1e-5f) / (prev_cdf + 1e-5f), 0.0f), 1.0f);
                const float px = a.pts[3 * q], py = a.pts[3 * q + 1], pz = a.pts[3 * q + 2];
                const float pn = sqrtf((px * px + py * py) + pz * pz);
                const float inside = pn < 1.0f ? 1.0f : 0.0f, relax = pn < 1.2f ? 1.0f : 0.0f;
                if (a.cdf) a.cdf[q] = prev_cdf;
                if (a.inside) a.inside[q] = inside;
                const float gn = sqrtf((gx * gx + gy * gy) + gz * gz) - 1.0f;
                e_sum += relax * gn * gn;
                e_cnt += relax;
#pragma unroll
                for (int k = 0; k < 3; ++k) c[k] = a.color[3 * q + k];
                if (has_bg) {
                    const size_t qb = (size_t)r * a.mo + j;
                    alpha = alpha * inside + bg_alpha * (1.0f - inside);
#pragma unroll
                    for (int k = 0; k < 3; ++k) c[k] = c[k] * inside + a.bg_color[3 * qb + k] * (1.0f - inside);
                }
            } else {
                const size_t qb = (size_t)r * a.mo + j;
                alpha = bg_alpha;
#pragma unroll
                for (int k = 0; k < 3; ++k) c[k] = a.bg_color[3 * qb + k];
            }
            const float w = alpha * trans;
            trans = trans * (1.0f - alpha + 1e-7f);
            a.weights[(size_t)r * mt + j] = w;
            wsum += w;
            wmax = fmaxf(wmax, w);
#pragma unroll
            for (int k = 0; k < 3; ++k) col[k] += c[k] * w;
        }
        if (a.background_rgb) {
#pragma unroll
            for (int k = 0; k < 3; ++k) col[k] += a.background_rgb[k] * (1.0f - wsum);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) a.out_color[3 * (size_t)r + k] = col[k];
        a.weight_sum[r] = wsum;
        a.weight_max[r] = wmax;
    }
    // eikonal statistics: wave reduction, one atomic pair per wave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        e_sum += __shfl_xor(e_sum, off, 64);
        e_cnt += __shfl_xor(e_cnt, off, 64);
    }
    if ((threadIdx.x & 63) == 0 && a.gerr_acc) {
        atomicAdd(&a.gerr_acc[0], e_sum);
        atomicAdd(&a.gerr_acc[1], e_cnt);
    }
}

// render_core_outside, compositing half (renderer.py:174-187): alpha = 1 - exp(-softplus(density) * dists), transmittance
// scan, colour; one thread per ray
__global__ void k_neus_outside_composite(const float* __restrict__ density, const float* __restrict__ dists, const float* __restrict__ rgb,
                                         const float* __restrict__ background_rgb, int n, int mo, float* __restrict__ alpha_out,
                                         float* __restrict__ weights, float* __restrict__ color) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    float trans = 1.0f, wsum = 0.0f, col[3] = {0.f, 0.f, 0.f};
    for (int j = 0; j < mo; ++j) {
        const size_t q = (size_t)r * mo + j;
        const float alpha = 1.0f - expf(-softplusf_(density[q]) * dists[q]);
        const float w = alpha * trans;
        trans = trans * (1.0f - alpha + 1e-7f);
        alpha_out[q] = alpha;
        weights[q] = w;
        wsum += w;
#pragma unroll
        for (int k = 0; k < 3; ++k) col[k] += rgb[3 * q + k] * w;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) color[3 * (size_t)r + k] = col[k] + (background_rgb ? background_rgb[k] * (1.0f - wsum) : 0.0f);
}

// sample_pdf (renderer.py:45-75): inverse-CDF samples of a piecewise-constant density.  bins [n,mb], weights [n,mb-1];
// u [n,k] uniform numbers, or null for det=True (u = linspace(0.5/k, 1 - 0.5/k, k)).  One thread per row.
__global__ void k_neus_sample_pdf(const float* __restrict__ bins, const float* __restrict__ weights, const float* __restrict__ u_in,
                                  int n, int mb, int k_samples, float* __restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const float* b = bins + (size_t)r * mb;
    const float* w = weights + (size_t)r * (mb - 1);
    float cdf[kMaxSamples];
    float wsum = 0.0f;
    for (int j = 0; j < mb - 1; ++j) wsum += w[j] + 1e-5f;
    cdf[0] = 0.0f;
    float acc = 0.0f;
    for (int j = 0; j < mb - 1; ++j) {
        acc += (w[j] + 1e-5f) / wsum;
        cdf[j + 1] = acc;
    }
    const float start = 0.5f / k_samples, end = 1.0f - 0.5f / k_samples;
    const float step = k_samples > 1 ? (end - start) / (float)(k_samples - 1) : 0.0f;
    for (int k = 0; k < k_samples; ++k) {
        const float u = u_in ? u_in[(size_t)r * k_samples + k]
                             : (k_samples == 1 ? start : (k < k_samples / 2 ? start + step * k : end - step * (k_samples - 1 - k)));
        int lo = 0, hi = mb;  // searchsorted(cdf, u, right=True): first index with cdf > u
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
        }
        const int below = lo - 1 > 0 ? lo - 1 : 0;
        const int above = lo < mb - 1 ? lo : mb - 1;
        float denom = cdf[above] - cdf[below];
        if (denom < 1e-5f) denom = 1.0f;
        const float t = (u - cdf[below]) / denom;
        out[(size_t)r * k_samples + k] = b[below] + t * (b[above] - b[below]);
    }
}

}  // namespace iron

using namespace iron;

extern "C" int iron_neus_linspace(const float* near, const float* far, const float* lin, int64_t n, int32_t m, float* z, void* stream) {
    if (n < 0 || m < 1) return IRON_ERR_BAD_ARG;
    if (n == 0) return IRON_OK;
    if (!near || !far || !lin || !z) return IRON_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_neus_linspace, dim3(ray_grid(n * m)), dim3(64), 0, (hipStream_t)stream, near, far, lin, (int)n, m, z);
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}

extern "C" int iron_neus_outside_z(const float* far, const float* rev, int64_t n, int32_t m, float offset, float* z, void* stream) {
    if (n < 0 || m < 1) return IRON_ERR_BAD_ARG;
    if (n == 0) return IRON_OK;
    if (!far || !rev || !z) return IRON_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_neus_outside_z, dim3(ray_grid(n * m)), dim3(64), 0, (hipStream_t)stream, far, rev, (int)n, m, offset, z);
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}

extern "C" int iron_grid_points(const float* xs, const float* ys, const float* zs, int32_t nx, int32_t ny, int32_t nz, float* pts, void* stream) {
    if (nx < 0 || ny < 0 || nz < 0) return IRON_ERR_BAD_ARG;
    if ((int64_t)nx * ny * nz == 0) return IRON_OK;
    if (!xs || !ys || !zs || !pts) return IRON_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_grid_points, dim3(ray_grid((int64_t)nx * ny * nz)), dim3(64), 0, (hipStream_t)stream, xs, ys, zs, nx, ny, nz, pts);
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}

extern "C" int iron_neus_points(const float* rays_o, const float* rays_d, const float* z, int64_t n, int32_t m, float* pts, void* stream) {
    if (n < 0 || m < 1) return IRON_ERR_BAD_ARG;
    if (n == 0) return IRON_OK;
    if (!rays_o || !rays_d || !z || !pts) return IRON_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_neus_points, dim3(ray_grid(n * m)), dim3(64), 0, (hipStream_t)stream, rays_o, rays_d, z, (int)n, m, pts);
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}

extern "C" int iron_neus_up_sample(const float* rays_o, const float* rays_d, const float* z, const float* sdf, int64_t n, int32_t m,
                                   int32_t n_importance, float inv_s, float* new_z, void* stream) {
    if (n < 0 || m < 2 || m > kMaxSamples || n_importance < 1) return IRON_ERR_BAD_ARG;
    if (n == 0) return IRON_OK;
    if (!rays_o || !rays_d || !z || !sdf || !new_z) return IRON_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_neus_up_sample, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, rays_o, rays_d, z, sdf, (int)n, m,
                       n_importance, inv_s, new_z);
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}

extern "C" int iron_neus_merge(const float* z_a, const float* s_a, int32_t m_a, const float* z_b, const float* s_b, int32_t m_b, int64_t n,
                               float* z_out, float* s_out, void* stream) {
    if (n < 0 || m_a < 0 || m_b < 0) return IRON_ERR_BAD_ARG;
    if (n == 0 || m_a + m_b == 0) return IRON_OK;
    if (!z_a || !z_b || !z_out || (s_out && (!s_a || !s_b))) return IRON_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_neus_merge, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, z_a, s_a, m_a, z_b, s_b, m_b, (int)n,
                       z_out, s_out);
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}

extern "C" int iron_neus_mid_points(const float* rays_o, const float* rays_d, const float* z, int64_t n, int32_t m, float sample_dist,
                                    int32_t outside, float* dists, float* pts, float* dirs, void* stream) {
    if (n < 0 || m < 1) return IRON_ERR_BAD_ARG;
    if (n == 0) return IRON_OK;
    if (!rays_o || !rays_d || !z || !dists || !pts || !dirs) return IRON_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_neus_mid_points, dim3(ray_grid(n * m)), dim3(64), 0, (hipStream_t)stream, rays_o, rays_d, z, (int)n, m, sample_dist,
                       outside, dists, pts, dirs);
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}

extern "C" int iron_neus_need_background(const float* pts, int64_t n, int32_t m, int32_t mo, uint8_t* need, void* stream) {
    if (n < 0 || m < 1 || mo < m) return IRON_ERR_BAD_ARG;
    if (n == 0) return IRON_OK;
    if (!pts || !need) return IRON_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_neus_need_background, dim3(ray_grid(n * mo)), dim3(64), 0, (hipStream_t)stream, pts, (int)n, m, mo, need);
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}

static int neus_composite_launch(const iron_neus_composite_args* p, const float* bg_alpha, void* stream);

extern "C" int iron_neus_composite(const iron_neus_composite_args* p, void* stream) { return neus_composite_launch(p, nullptr, stream); }

extern "C" int iron_neus_composite_alpha(const iron_neus_composite_args* p, const float* background_alpha, void* stream) {
    if (!background_alpha) return IRON_ERR_BAD_ARG;
    return neus_composite_launch(p, background_alpha, stream);
}

extern "C" int iron_neus_outside_composite(const float* density, const float* dists, const float* sampled_color, const float* background_rgb,
                                           int64_t n, int32_t mo, float* alpha, float* weights, float* color, void* stream) {
    if (n < 0 || mo < 1) return IRON_ERR_BAD_ARG;
    if (n == 0) return IRON_OK;
    if (!density || !dists || !sampled_color || !alpha || !weights || !color) return IRON_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_neus_outside_composite, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, density, dists, sampled_color,
                       background_rgb, (int)n, mo, alpha, weights, color);
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}

extern "C" int iron_neus_sample_pdf(const float* bins, const float* weights, const float* u, int64_t n, int32_t n_bins, int32_t n_samples,
                                    float* samples, void* stream) {
    if (n < 0 || n_bins < 2 || n_bins > kMaxSamples || n_samples < 1) return IRON_ERR_BAD_ARG;
    if (n == 0) return IRON_OK;
    if (!bins || !weights || !samples) return IRON_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_neus_sample_pdf, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, bins, weights, u, (int)n, n_bins,
                       n_samples, samples);
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}

static int neus_composite_launch(const iron_neus_composite_args* p, const float* bg_alpha, void* stream) {
    if (!p || p->n < 0 || p->m < 1 || p->m > kMaxSamples) return IRON_ERR_BAD_ARG;
    if (p->n == 0) return IRON_OK;
    if (!p->dists || !p->pts || !p->dirs || !p->sdf || !p->grad || !p->color || !p->out_color || !p->weights || !p->weight_sum ||
        !p->weight_max)
        return IRON_ERR_BAD_ARG;
    if (p->bg_density && (!p->bg_dists || !p->bg_color || p->mo < p->m || p->mo > kMaxSamples)) return IRON_ERR_BAD_ARG;
    if (bg_alpha && (!p->bg_color || p->mo < p->m || p->mo > kMaxSamples)) return IRON_ERR_BAD_ARG;
    NeusCompositeArgs a;
    a.bg_alpha = bg_alpha;
    a.dists = p->dists; a.pts = p->pts; a.dirs = p->dirs; a.sdf = p->sdf; a.grad = p->grad; a.color = p->color;
    a.bg_dists = p->bg_dists; a.bg_density = p->bg_density; a.bg_color = p->bg_color; a.background_rgb = p->background_rgb;
    a.n = (int)p->n; a.m = p->m; a.mo = p->mo; a.inv_s = p->inv_s; a.cos_anneal = p->cos_anneal_ratio;
    a.out_color = p->out_color; a.weights = p->weights; a.cdf = p->cdf; a.inside = p->inside_sphere; a.weight_sum = p->weight_sum;
    a.weight_max = p->weight_max; a.gerr_acc = p->gradient_error_acc;
    hipStream_t st = (hipStream_t)stream;
    if (a.gerr_acc) IRON_HIP_TRY(hipMemsetAsync(a.gerr_acc, 0, 2 * sizeof(float), st));
    hipLaunchKernelGGL(k_neus_composite, dim3((unsigned)((p->n + 63) / 64)), dim3(64), 0, st, a);
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}
