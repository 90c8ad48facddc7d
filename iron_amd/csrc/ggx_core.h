// Pointwise geometry + co-located GGX BRDF device functions (fp32, one rounding per reference op;
// the translation unit is compiled with -ffp-contract=off so a*b+c stays two roundings like the
// reference's separate torch kernels).
#pragma once
#include <hip/hip_runtime.h>

namespace iron {

struct GgxOut {
    float diffuse[3];
    float specular[3];
    float rgb[3];
};

// smithG1 (models/renderer_ggx.py:12-16)
__device__ __forceinline__ float smith_g1(float cos_theta, float alpha) {
    const float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    const float tan_theta = sin_theta / (cos_theta + 1e-10f);
    const float root = alpha * tan_theta;
    return 2.0f / (1.0f + hypotf(root, 1.0f));
}

// GGXColocatedRenderer.forward (models/renderer_ggx.py:82-146) for one point.
// tab_trans: 5000 floats (100 theta x 50 alpha), tab_diff: 50 floats.
__device__ __forceinline__ void ggx_colocated_point(float light, float distance, const float n[3], const float v[3],
                                                    const float kd[3], const float ks[3], float rough,
                                                    const float* __restrict__ tab_trans,
                                                    const float* __restrict__ tab_diff, GgxOut& o) {
    const float intensity = light / (distance * distance + 1e-10f);
    float dot = (v[0] * n[0] + v[1] * n[1]) + v[2] * n[2];
    dot = fminf(fmaxf(dot, 0.00001f), 0.99999f);
    const float m_inv_eta2 = (float)(1.0 / (1.48958738 * 1.48958738));
    const float alpha = fmaxf(rough, 0.0001f);
    const float pi_f = 3.14159274101257324219f;  // float32(np.pi)

    const float c2 = dot * dot;
    const float root = c2 + (1.0f - c2) / (alpha * alpha + 1e-10f);
    const float D = 1.0f / (pi_f * alpha * alpha * root * root + 1e-10f);
    const float Fr = 0.03867f;
    const float g1 = smith_g1(dot, alpha);
    const float G = g1 * g1;
    const float denom = 4.0f * dot + 1e-10f;

    const float warped_cos = powf(dot, 0.25f);
    const float warped_alpha = powf(alpha / 4.0f, 0.25f);
    const long long tx = (long long)floorf(warped_cos * 100.0f);
    const long long ty = (long long)floorf(warped_alpha * 50.0f);
    long long ti = ty * 100 + tx;
    ti = ti < 0 ? 0 : (ti > 4999 ? 4999 : ti);
    const float T12 = fminf(fmaxf(tab_trans[ti], 0.0f), 1.0f);
    long long ai = ty < 0 ? 0 : (ty > 49 ? 49 : ty);
    const float Fdr = fminf(fmaxf(1.0f - tab_diff[ai], 0.0f), 1.0f);
    const float fd = 1.0f - Fdr + 1e-10f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        o.specular[c] = intensity * ks[c] * Fr * D * G / denom;
        o.diffuse[c] = intensity * (kd[c] / fd / pi_f) * dot * T12 * T12 * m_inv_eta2;
        o.rgb[c] = o.diffuse[c] + o.specular[c];
    }
}

// intersect_sphere (models/raytracer.py:223-237) for one ray
__device__ __forceinline__ void intersect_sphere_ray(const float o[3], const float d[3], float r, bool& hit,
                                                     float& near, float& far) {
    const float dd = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2];
    const float d1 = -((d[0] * o[0] + d[1] * o[1]) + d[2] * o[2]) / dd;
    const float px = o[0] + d1 * d[0], py = o[1] + d1 * d[1], pz = o[2] + d1 * d[2];
    const float tmp = r * r - ((px * px + py * py) + pz * pz);
    hit = tmp > 0.0f;
    const float d2 = sqrtf(fmaxf(tmp, 0.0f)) / sqrtf(dd);
    near = fmaxf(d1 - d2, 0.0f);
    far = d1 + d2;
}

}  // namespace iron
