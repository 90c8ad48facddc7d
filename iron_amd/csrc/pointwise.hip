// HBM-bound pointwise entry points: camera rays, unit-sphere intersection, standalone GGX.
#include "iron_common.h"
#include "ggx_core.h"

namespace iron {

struct CamMat {
    float kinv[9];
    float c2w[12];
};

// Camera.get_rays (models/raytracer.py:254-286)
__global__ void k_camera_rays(CamMat m, const float* __restrict__ uv, int64_t n, float* __restrict__ ray_o,
                              float* __restrict__ ray_d, float* __restrict__ ray_d_norm) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float u = uv[2 * i], v = uv[2 * i + 1];
        float c[3], w[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) c[r] = fmaf(1.0f, m.kinv[3 * r + 2], fmaf(v, m.kinv[3 * r + 1], u * m.kinv[3 * r]));
#pragma unroll
        for (int r = 0; r < 3; ++r)
            w[r] = fmaf(c[2], m.c2w[4 * r + 2], fmaf(c[1], m.c2w[4 * r + 1], c[0] * m.c2w[4 * r]));
        const float nrm = sqrtf((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]);
        ray_d[3 * i] = w[0] / nrm;
        ray_d[3 * i + 1] = w[1] / nrm;
        ray_d[3 * i + 2] = w[2] / nrm;
        ray_d_norm[i] = nrm;
        ray_o[3 * i] = m.c2w[3];
        ray_o[3 * i + 1] = m.c2w[7];
        ray_o[3 * i + 2] = m.c2w[11];
    }
}

__global__ void k_intersect_sphere(const float* __restrict__ ray_o, const float* __restrict__ ray_d, int64_t n,
                                   float r, uint8_t* __restrict__ mask, float* __restrict__ near,
                                   float* __restrict__ far) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float o[3] = {ray_o[3 * i], ray_o[3 * i + 1], ray_o[3 * i + 2]};
        const float d[3] = {ray_d[3 * i], ray_d[3 * i + 1], ray_d[3 * i + 2]};
        bool hit;
        float a, b;
        intersect_sphere_ray(o, d, r, hit, a, b);
        mask[i] = hit ? 1 : 0;
        near[i] = a;
        far[i] = b;
    }
}

__global__ void k_ggx(float light, const float* __restrict__ distance, const float* __restrict__ normal,
                      const float* __restrict__ viewdir, const float* __restrict__ kd, const float* __restrict__ ks,
                      const float* __restrict__ rough, const float* __restrict__ tab_trans,
                      const float* __restrict__ tab_diff, int64_t n, float* __restrict__ diffuse_rgb,
                      float* __restrict__ specular_rgb, float* __restrict__ rgb) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float nn[3] = {normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]};
        const float vv[3] = {viewdir[3 * i], viewdir[3 * i + 1], viewdir[3 * i + 2]};
        const float a[3] = {kd[3 * i], kd[3 * i + 1], kd[3 * i + 2]};
        const float s[3] = {ks[3 * i], ks[3 * i + 1], ks[3 * i + 2]};
        GgxOut o;
        ggx_colocated_point(light, distance[i], nn, vv, a, s, rough[i], tab_trans, tab_diff, o);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (diffuse_rgb) diffuse_rgb[3 * i + c] = o.diffuse[c];
            if (specular_rgb) specular_rgb[3 * i + c] = o.specular[c];
            if (rgb) rgb[3 * i + c] = o.rgb[c];
        }
    }
}

struct CompositeArgs {
    const float *distance, *normal, *viewdir, *kd, *ks, *rough, *m_eta, *m_k, *d_eta, *env_light, *tab_trans, *tab_diff;
    float *specular, *metallic, *dielectric, *rgb, *env_out;
    float light;
    int64_t n;
};

__global__ void k_composite(CompositeArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
        const float nn[3] = {a.normal[3 * i], a.normal[3 * i + 1], a.normal[3 * i + 2]};
        const float vv[3] = {a.viewdir[3 * i], a.viewdir[3 * i + 1], a.viewdir[3 * i + 2]};
        const float kd[3] = {a.kd[3 * i], a.kd[3 * i + 1], a.kd[3 * i + 2]};
        const float ks[3] = {a.ks[3 * i], a.ks[3 * i + 1], a.ks[3 * i + 2]};
        float intensity;
        if (a.env_light) {
            intensity = fminf(fmaxf(a.env_light[i], 0.000001f), 20.0f);
            if (a.env_out) a.env_out[i] = intensity;
        } else {
            const float d = a.distance[i];
            intensity = a.light / (d * d + 1e-10f);
        }
        CompositeOut o;
        composite_point(intensity, nn, vv, kd, ks, a.rough[i], a.m_eta[i], a.m_k[i], a.d_eta[i], a.tab_trans, a.tab_diff, o);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (a.specular) a.specular[3 * i + c] = o.specular[c];
            if (a.metallic) a.metallic[3 * i + c] = o.metallic[c];
            if (a.dielectric) a.dielectric[3 * i + c] = o.dielectric[c];
            a.rgb[3 * i + c] = o.rgb[c];
        }
    }
}

__global__ void k_coloc_head(int kind, float light, float eta, float kk, const float* __restrict__ distance,
                             const float* __restrict__ normal, const float* __restrict__ viewdir, const float* __restrict__ kd,
                             const float* __restrict__ ks, const float* __restrict__ rough, int64_t n,
                             float* __restrict__ diffuse_rgb, float* __restrict__ specular_rgb, float* __restrict__ rgb) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float nn[3] = {normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]};
        const float vv[3] = {viewdir[3 * i], viewdir[3 * i + 1], viewdir[3 * i + 2]};
        const float a[3] = {kd[3 * i], kd[3 * i + 1], kd[3 * i + 2]};
        const float s[3] = {ks[3 * i], ks[3 * i + 1], ks[3 * i + 2]};
        GgxOut o;
        coloc_head_point(kind, light, distance[i], nn, vv, a, s, rough ? rough[i] : 0.0f, eta, kk, o);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (diffuse_rgb) diffuse_rgb[3 * i + c] = o.diffuse[c];
            if (specular_rgb) specular_rgb[3 * i + c] = o.specular[c];
            if (rgb) rgb[3 * i + c] = o.rgb[c];
        }
    }
}

// 3x3 max (SIGN=+1) / min (SIGN=-1) filter with a border that never wins: the two passes of kornia's
// morphology.closing(x, ones(3,3)) with its default 'geodesic' border (models/raytracer.py:554-557).
template <int SIGN>
__global__ void k_minmax3x3(const float* __restrict__ in, int H, int W, float* __restrict__ out) {
    const int64_t n = (int64_t)H * W;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int y = (int)(i / W), x = (int)(i % W);
        float v = in[i];
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = y + dy, xx = x + dx;
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                const float u = in[(int64_t)yy * W + xx];
                v = SIGN > 0 ? fmaxf(v, u) : fminf(v, u);
            }
        out[i] = v;
    }
}

// kornia.filters.sobel(x) (normalized, eps = 1e-6; models/raytracer.py:569): Sobel kernels / 8, replicate border,
// sqrt(gx^2 + gy^2 + 1e-6)
__global__ void k_sobel(const float* __restrict__ in, int H, int W, float* __restrict__ out) {
    const int64_t n = (int64_t)H * W;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int y = (int)(i / W), x = (int)(i % W);
        float p[3][3];
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                int yy = y + dy, xx = x + dx;
                yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
                xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
                p[dy + 1][dx + 1] = in[(int64_t)yy * W + xx];
            }
        const float gx = ((p[0][2] - p[0][0]) + 2.0f * (p[1][2] - p[1][0]) + (p[2][2] - p[2][0])) / 8.0f;
        const float gy = ((p[2][0] - p[0][0]) + 2.0f * (p[2][1] - p[0][1]) + (p[2][2] - p[0][2])) / 8.0f;
        out[i] = sqrtf(gx * gx + gy * gy + 1e-6f);
    }
}

// fill_holes branch of raytrace_camera (models/raytracer.py:554-564) without the host round trip of `update_mask.any()`:
// pass 1 raises *flag when the closed depth image turns any non-convergent pixel into a hit, pass 2 applies the
// reference's whole-image update (depth at the new hits, mask := closed > 1e-2, distance and points recomputed from the
// depth for EVERY pixel) only if the flag is up.
__global__ void k_fill_holes_mark(const float* __restrict__ closed, const uint8_t* __restrict__ conv, int64_t n, int* __restrict__ flag) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    bool any = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) any = any || (closed[i] > 1e-2f && conv[i] == 0);
    if (__ballot(any) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

__global__ void k_fill_holes_apply(const float* __restrict__ closed, const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                   const float* __restrict__ ray_d_norm, int64_t n, const int* __restrict__ flag,
                                   float* __restrict__ depth, uint8_t* __restrict__ conv, float* __restrict__ distance,
                                   float* __restrict__ points) {
    if (*flag == 0) return;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const bool hit = closed[i] > 1e-2f;
        const float z = (hit && conv[i] == 0) ? closed[i] : depth[i];
        const float t = z * ray_d_norm[i];
        depth[i] = z;
        conv[i] = hit ? 1 : 0;
        distance[i] = t;
#pragma unroll
        for (int c = 0; c < 3; ++c) points[3 * i + c] = ray_o[3 * i + c] + ray_d[3 * i + c] * t;
    }
}

// render_edge_pixels, geometry half (models/raytracer.py:680-698): per edge pixel the image-plane normal of the
// silhouette, the two side samples on a circle of radius 0.707 px around the pixel centre and the area weight of the
// positive side.  side_uv is [2n,2]: the n positive-side samples first, then the n negative-side ones.
__global__ void k_edge_sides(const float* __restrict__ edge_uv, const float* __restrict__ edge_grad, CamMat m /* c2w unused; kinv = W2C[:3,:3] */,
                             int64_t n, float* __restrict__ side_uv, float* __restrict__ pos_weight) {
    const float radius = 0.707f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float g[3] = {edge_grad[3 * i], edge_grad[3 * i + 1], edge_grad[3 * i + 2]};
        const float gl = sqrtf((g[0] * g[0] + g[1] * g[1]) + g[2] * g[2]) + 1e-10f;
        const float nx = g[0] / gl, ny = g[1] / gl, nz = g[2] / gl;
        float ex = fmaf(nz, m.kinv[2], fmaf(ny, m.kinv[1], nx * m.kinv[0]));
        float ey = fmaf(nz, m.kinv[5], fmaf(ny, m.kinv[4], nx * m.kinv[3]));
        const float el = sqrtf(ex * ex + ey * ey) + 1e-10f;
        ex /= el;
        ey /= el;
        const float u = edge_uv[2 * i], v = edge_uv[2 * i + 1];
        const float cu = floorf(u) + 0.5f, cv = floorf(v) + 0.5f;
        side_uv[2 * i] = cu - radius * ex;
        side_uv[2 * i + 1] = cv - radius * ey;
        side_uv[2 * (n + i)] = cu + radius * ex;
        side_uv[2 * (n + i) + 1] = cv + radius * ey;
        const float off = (u - cu) * ex + (v - cv) * ey;
        const float alpha = 2.0f * acosf(fminf(fmaxf(off / radius, 0.0f), 1.0f));
        pos_weight[i] = 1.0f - (alpha - sinf(alpha)) / 6.2831855f;
    }
}

// render_edge_pixels, blend half (models/raytracer.py:706-729): colour of an edge pixel = area-weighted mix of its two
// side samples; its normal / uv / point are those of the edge point itself.
__global__ void k_edge_blend(const float* __restrict__ side_color /*[2n,3]*/, const float* __restrict__ pos_weight,
                             const float* __restrict__ edge_grad, const float* __restrict__ edge_uv,
                             const float* __restrict__ edge_points, const int64_t* __restrict__ pixel, int64_t n, int64_t n_pixels,
                             float* __restrict__ color, float* __restrict__ normal, float* __restrict__ uv, float* __restrict__ points) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t p = pixel[i];
        if (p < 0 || p >= n_pixels) continue;
        const float w = pos_weight[i];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            color[3 * p + c] = side_color[3 * i + c] * w + side_color[3 * (n + i) + c] * (1.0f - w);
            normal[3 * p + c] = edge_grad[3 * i + c];
            points[3 * p + c] = edge_points[3 * i + c];
        }
        uv[2 * p] = edge_uv[2 * i];
        uv[2 * p + 1] = edge_uv[2 * i + 1];
    }
}

struct Mat44x2 {
    float w2c[16];
    float k[16];
};

// Tail of locate_edge_points (models/raytracer.py:481-500): project the found points (Camera.project: p_h . W2C^T . K^T, then
// the perspective division), take the pixel they fall in, and keep for every pixel the FIRST found point in candidate order
// (what unique() / scatter_ of the reversed permutation select): first[pixel] = min candidate index, by atomicMin.
// The pixel index is y * W + x and only IT is range-checked, as in the reference (:489-490).
__global__ void k_edge_project_first(const float* __restrict__ points, const uint8_t* __restrict__ found, int64_t n, Mat44x2 m, int H, int W,
                                     float* __restrict__ uv, int* __restrict__ first) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float x = points[3 * i], y = points[3 * i + 1], z = points[3 * i + 2];
        float c[4], q[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r] = fmaf(1.0f, m.w2c[4 * r + 3], fmaf(z, m.w2c[4 * r + 2], fmaf(y, m.w2c[4 * r + 1], x * m.w2c[4 * r])));
#pragma unroll
        for (int r = 0; r < 3; ++r) q[r] = fmaf(c[3], m.k[4 * r + 3], fmaf(c[2], m.k[4 * r + 2], fmaf(c[1], m.k[4 * r + 1], c[0] * m.k[4 * r])));
        const float u = q[0] / q[2], v = q[1] / q[2];
        uv[2 * i] = u;
        uv[2 * i + 1] = v;
        if (!found[i]) continue;
        const long long pix = (long long)floorf(v) * (long long)W + (long long)floorf(u);
        if (pix < 0 || pix >= (long long)H * W) continue;
        atomicMin(&first[pix], (int)i);
    }
}

// smithG1 (models/renderer_ggx.py:12-16) as a standalone operator
__global__ void k_smith_g1(const float* __restrict__ cos_theta, const float* __restrict__ alpha, int64_t n, float* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = smith_g1(cos_theta[i], alpha[i]);
}

static inline int pw_grid(int64_t n) {
    const int64_t b = (n + 255) / 256;
    return (int)(b < 2048 ? (b > 0 ? b : 1) : 2048);
}

}  // namespace iron

using namespace iron;

extern "C" int iron_camera_rays(const float* k_inv3, const float* c2w34, const float* uv, int64_t n, float* ray_o,
                                float* ray_d, float* ray_d_norm, void* stream) {
    if (!k_inv3 || !c2w34 || n < 0 || (n > 0 && (!uv || !ray_o || !ray_d || !ray_d_norm))) return IRON_ERR_BAD_ARG;
    if (n == 0) return IRON_OK;
    CamMat m;
    for (int i = 0; i < 9; ++i) m.kinv[i] = k_inv3[i];
    for (int i = 0; i < 12; ++i) m.c2w[i] = c2w34[i];
    hipLaunchKernelGGL(k_camera_rays, dim3(pw_grid(n)), dim3(256), 0, (hipStream_t)stream, m, uv, n, ray_o, ray_d,
                       ray_d_norm);
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}

extern "C" int iron_intersect_sphere(const float* ray_o, const float* ray_d, int64_t n, float r, uint8_t* mask,
                                     float* near, float* far, void* stream) {
    if (n < 0 || (n > 0 && (!ray_o || !ray_d || !mask || !near || !far))) return IRON_ERR_BAD_ARG;
    if (n == 0) return IRON_OK;
    hipLaunchKernelGGL(k_intersect_sphere, dim3(pw_grid(n)), dim3(256), 0, (hipStream_t)stream, ray_o, ray_d, n, r,
                       mask, near, far);
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}

extern "C" int iron_ggx_colocated(float light, const float* distance, const float* normal, const float* viewdir,
                                  const float* diffuse_albedo, const float* specular_albedo, const float* roughness,
                                  const float* tab_trans, const float* tab_diff_trans, int64_t n, float* diffuse_rgb,
                                  float* specular_rgb, float* rgb, void* stream) {
    if (n < 0) return IRON_ERR_BAD_ARG;
    if (n == 0) return IRON_OK;
    if (!distance || !normal || !viewdir || !diffuse_albedo || !specular_albedo || !roughness || !tab_trans ||
        !tab_diff_trans)
        return IRON_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_ggx, dim3(pw_grid(n)), dim3(256), 0, (hipStream_t)stream, light, distance, normal, viewdir,
                       diffuse_albedo, specular_albedo, roughness, tab_trans, tab_diff_trans, n, diffuse_rgb,
                       specular_rgb, rgb);
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}

extern "C" int iron_morph_closing3x3(const float* depth, int32_t H, int32_t W, float* tmp, float* out, void* stream) {
    if (H < 0 || W < 0) return IRON_ERR_BAD_ARG;
    const int64_t n = (int64_t)H * W;
    if (n == 0) return IRON_OK;
    if (!depth || !tmp || !out || tmp == out || tmp == depth) return IRON_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_minmax3x3<1>, dim3(pw_grid(n)), dim3(256), 0, (hipStream_t)stream, depth, H, W, tmp);
    hipLaunchKernelGGL(k_minmax3x3<-1>, dim3(pw_grid(n)), dim3(256), 0, (hipStream_t)stream, tmp, H, W, out);
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}

extern "C" int iron_sobel_magnitude(const float* depth, int32_t H, int32_t W, float* out, void* stream) {
    if (H < 0 || W < 0) return IRON_ERR_BAD_ARG;
    const int64_t n = (int64_t)H * W;
    if (n == 0) return IRON_OK;
    if (!depth || !out || depth == out) return IRON_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_sobel, dim3(pw_grid(n)), dim3(256), 0, (hipStream_t)stream, depth, H, W, out);
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}

extern "C" int iron_composite_colocated(float light, const float* distance, const float* normal, const float* viewdir,
                                        const iron_composite_params* p, const float* tab_trans, const float* tab_diff_trans,
                                        int64_t n, float* specular_rgb, float* metallic_rgb, float* dielectric_rgb, float* rgb,
                                        float* env_light_out, void* stream) {
    if (n < 0 || !p) return IRON_ERR_BAD_ARG;
    if (n == 0) return IRON_OK;
    if (!normal || !viewdir || !p->diffuse_albedo || !p->specular_albedo || !p->specular_roughness || !p->metallic_eta ||
        !p->metallic_k || !p->dielectric_eta || !tab_trans || !tab_diff_trans || !rgb)
        return IRON_ERR_BAD_ARG;
    if (!p->env_light && !distance) return IRON_ERR_BAD_ARG;
    CompositeArgs a;
    a.distance = distance; a.normal = normal; a.viewdir = viewdir; a.kd = p->diffuse_albedo; a.ks = p->specular_albedo;
    a.rough = p->specular_roughness; a.m_eta = p->metallic_eta; a.m_k = p->metallic_k; a.d_eta = p->dielectric_eta;
    a.env_light = p->env_light; a.tab_trans = tab_trans; a.tab_diff = tab_diff_trans; a.specular = specular_rgb;
    a.metallic = metallic_rgb; a.dielectric = dielectric_rgb; a.rgb = rgb; a.env_out = env_light_out; a.light = light; a.n = n;
    hipLaunchKernelGGL(k_composite, dim3(pw_grid(n)), dim3(256), 0, (hipStream_t)stream, a);
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}

extern "C" int iron_coloc_head(int32_t kind, float light, float eta, float k, const float* distance, const float* normal,
                               const float* viewdir, const float* diffuse_albedo, const float* specular_albedo,
                               const float* roughness, int64_t n, float* diffuse_rgb, float* specular_rgb, float* rgb,
                               void* stream) {
    if (n < 0 || kind < 0 || kind > 3) return IRON_ERR_BAD_ARG;
    if (n == 0) return IRON_OK;
    if (!distance || !normal || !viewdir || !diffuse_albedo || !specular_albedo) return IRON_ERR_BAD_ARG;
    if (kind == kHeadRoughConductor && !roughness) return IRON_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_coloc_head, dim3(pw_grid(n)), dim3(256), 0, (hipStream_t)stream, (int)kind, light, eta, k, distance,
                       normal, viewdir, diffuse_albedo, specular_albedo, roughness, n, diffuse_rgb, specular_rgb, rgb);
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}

extern "C" int iron_fill_holes(const float* depth_closed, const float* ray_o, const float* ray_d, const float* ray_d_norm, int64_t n,
                               float* depth, uint8_t* conv, float* distance, float* points, int32_t* flag, void* stream) {
    if (n < 0) return IRON_ERR_BAD_ARG;
    if (n == 0) return IRON_OK;
    if (!depth_closed || !ray_o || !ray_d || !ray_d_norm || !depth || !conv || !distance || !points || !flag) return IRON_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    IRON_HIP_TRY(hipMemsetAsync(flag, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(k_fill_holes_mark, dim3(pw_grid(n)), dim3(256), 0, st, depth_closed, conv, n, (int*)flag);
    hipLaunchKernelGGL(k_fill_holes_apply, dim3(pw_grid(n)), dim3(256), 0, st, depth_closed, ray_o, ray_d, ray_d_norm, n, (const int*)flag,
                       depth, conv, distance, points);
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}

extern "C" int iron_edge_sides(const float* edge_uv, const float* edge_grad, const float* w2c_rot9, int64_t n, float* side_uv,
                               float* pos_weight, void* stream) {
    if (n < 0 || !w2c_rot9) return IRON_ERR_BAD_ARG;
    if (n == 0) return IRON_OK;
    if (!edge_uv || !edge_grad || !side_uv || !pos_weight) return IRON_ERR_BAD_ARG;
    CamMat m;
    for (int i = 0; i < 9; ++i) m.kinv[i] = w2c_rot9[i];
    for (int i = 0; i < 12; ++i) m.c2w[i] = 0.0f;
    hipLaunchKernelGGL(k_edge_sides, dim3(pw_grid(n)), dim3(256), 0, (hipStream_t)stream, edge_uv, edge_grad, m, n, side_uv, pos_weight);
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}

extern "C" int iron_edge_blend(const float* side_color, const float* pos_weight, const float* edge_grad, const float* edge_uv,
                               const float* edge_points, const int64_t* pixel_idx, int64_t n, int64_t n_pixels, float* color,
                               float* normal, float* uv, float* points, void* stream) {
    if (n < 0 || n_pixels < 0) return IRON_ERR_BAD_ARG;
    if (n == 0) return IRON_OK;
    if (!side_color || !pos_weight || !edge_grad || !edge_uv || !edge_points || !pixel_idx || !color || !normal || !uv || !points)
        return IRON_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_edge_blend, dim3(pw_grid(n)), dim3(256), 0, (hipStream_t)stream, side_color, pos_weight, edge_grad, edge_uv,
                       edge_points, pixel_idx, n, n_pixels, color, normal, uv, points);
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}

extern "C" int iron_smith_g1(const float* cos_theta, const float* alpha, int64_t n, float* out, void* stream) {
    if (n < 0) return IRON_ERR_BAD_ARG;
    if (n == 0) return IRON_OK;
    if (!cos_theta || !alpha || !out) return IRON_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_smith_g1, dim3(pw_grid(n)), dim3(256), 0, (hipStream_t)stream, cos_theta, alpha, n, out);
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}

extern "C" int iron_edge_pixels(const float* points, const uint8_t* found, int64_t n, const float* w2c16, const float* k16, int32_t H, int32_t W,
                                float* uv, int32_t* first, void* stream) {
    if (n < 0 || H < 0 || W < 0 || !w2c16 || !k16) return IRON_ERR_BAD_ARG;
    if (n > 0x7fffffffLL) return IRON_ERR_BAD_ARG;
    if (n == 0) return IRON_OK;
    if (!points || !found || !uv || !first) return IRON_ERR_BAD_ARG;
    Mat44x2 m;
    for (int i = 0; i < 16; ++i) { m.w2c[i] = w2c16[i]; m.k[i] = k16[i]; }
    hipLaunchKernelGGL(k_edge_project_first, dim3(pw_grid(n)), dim3(256), 0, (hipStream_t)stream, points, found, n, m, (int)H, (int)W, uv,
                       (int*)first);
    IRON_HIP_TRY(hipGetLastError());
    return IRON_OK;
}
