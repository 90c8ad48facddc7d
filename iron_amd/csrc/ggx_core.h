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

// ---- the fork's other co-located heads (models/renderer_ggx.py:149-517, 520-858) -----------------

// fresnel_conductor_exact (renderer_ggx.py:419-432 = CompositeRenderer.fresnel_conductor_exact :592-605)
__device__ __forceinline__ float fresnel_conductor_exact(float cos_i, float eta, float k) {
    const float c2 = cos_i * cos_i;
    const float s2 = 1.0f - c2;
    const float s4 = s2 * s2;
    const float temp1 = eta * eta - k * k - s2;
    const float a2pb2 = sqrtf(temp1 * temp1 + 4.0f * k * k * eta * eta);
    const float a = sqrtf(0.5f * (a2pb2 + temp1));
    const float term1 = a2pb2 + c2;
    const float term2 = 2.0f * a * cos_i;
    const float rs2 = (term1 - term2) / (term1 + term2);
    const float term3 = a2pb2 * c2 + s4;
    const float term4 = term2 * s2;
    const float rp2 = rs2 * (term3 - term4) / (term3 + term4);
    return 0.5f * (rp2 + rs2);
}

// fresnel_dielectric (renderer_ggx.py:398-416) for cos_i > 0 (the callers clamp it to [1e-5, 0.99999])
__device__ __forceinline__ float fresnel_dielectric_pos(float cos_i, float eta) {
    const float scale = 1.0f / eta;
    const float cos_t_sqr = 1.0f - (1.0f - cos_i * cos_i) * (scale * scale);
    const float cos_t = sqrtf(cos_t_sqr);
    const float rs = (cos_i - eta * cos_t) / (cos_i + eta * cos_t);
    const float rp = (eta * cos_i - cos_t) / (eta * cos_i + cos_t);
    return 0.5f * (rs * rs + rp * rp);
}

// the Mitsuba rough-plastic diffuse term through the two tables (CompositeRenderer.diffuse_reflection_ggx :654-681);
// returns the factor that multiplies intensity * kd: 1 / (1 - Fdr + 1e-10) / pi * cos * T12^2 / eta^2 is applied by the caller
__device__ __forceinline__ void rtrans_lookup(float cos_theta, float alpha, const float* __restrict__ tab_trans,
                                              const float* __restrict__ tab_diff, float& T12, float& fd) {
    const float warped_cos = powf(cos_theta, 0.25f);
    const float warped_alpha = powf(alpha / 4.0f, 0.25f);
    const long long tx = (long long)floorf(warped_cos * 100.0f);
    const long long ty = (long long)floorf(warped_alpha * 50.0f);
    long long ti = ty * 100 + tx;
    ti = ti < 0 ? 0 : (ti > 4999 ? 4999 : ti);
    T12 = fminf(fmaxf(tab_trans[ti], 0.0f), 1.0f);
    const long long ai = ty < 0 ? 0 : (ty > 49 ? 49 : ty);
    const float Fdr = fminf(fmaxf(1.0f - tab_diff[ai], 0.0f), 1.0f);
    fd = 1.0f - Fdr + 1e-10f;
}

struct CompositeOut {
    float specular[3], metallic[3], dielectric[3], rgb[3];
};

// CompositeRenderer.forward (renderer_ggx.py:781-858) for one point, quirks included: the NDF takes alpha := 1.48958738
// (:806), the metallic / dielectric weights are unused (:829-831), diffuse_rgb is rgb (in-place alias :847-853).
// `intensity` = light / (d^2 + 1e-10) or the clamped env light.
__device__ __forceinline__ void composite_point(float intensity, const float n[3], const float v[3], const float kd_in[3],
                                                const float ks_in[3], float rough_in, float m_eta_in, float m_k_in,
                                                float d_eta_in, const float* __restrict__ tab_trans,
                                                const float* __restrict__ tab_diff, CompositeOut& o) {
    const float rough = fmaxf(rough_in, 0.00001f);
    const float d_eta = fminf(fmaxf(d_eta_in, 1.000001f), 1.999999f);
    const float m_eta = fminf(fmaxf(m_eta_in, 0.099999f), 4.999999f);
    const float m_k = fminf(fmaxf(m_k_in, 0.099999f), 9.999999f);
    float cos_i = (v[0] * n[0] + v[1] * n[1]) + v[2] * n[2];
    cos_i = fminf(fmaxf(cos_i, 0.00001f), 0.99999f);
    const float eta2 = (float)(1.48958738 * 1.48958738 + 1e-10);
    const float pi_eta2 = (float)(3.141592653589793 * 1.48958738 * 1.48958738);
    const float c2 = cos_i * cos_i;
    const float root = c2 + (1.0f - c2) / eta2;
    const float D = 1.0f / (pi_eta2 * root * root + 1e-10f);
    const float g1 = smith_g1(cos_i, rough);
    const float G = g1 * g1;
    const float Fm = fresnel_conductor_exact(cos_i, m_eta, m_k);
    const float Fd = fresnel_dielectric_pos(cos_i, d_eta);
    const float denom = 4.0f * fabsf(cos_i);
    float T12, fd;
    rtrans_lookup(cos_i, fmaxf(rough, 0.0001f), tab_trans, tab_diff, T12, fd);
    const float pi_f = 3.14159274101257324219f;
    const float inv_eta2 = (float)(1.0 / (1.48958738 * 1.48958738));
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float ks = fmaxf(ks_in[c], 0.00001f);
        const float kd = fmaxf(kd_in[c], 0.00001f);
        o.metallic[c] = (ks * Fm) * intensity;
        o.dielectric[c] = (ks * Fd * D * G / denom) * intensity;
        o.specular[c] = o.dielectric[c] + o.metallic[c];
        const float diffuse = intensity * (kd / fd / pi_f) * cos_i * T12 * T12 * inv_eta2;
        o.rgb[c] = diffuse + o.specular[c];
    }
}

// SmoothDielectric (:171-204), ThinDielectric (:229-267), SmoothConductorCoLoc (:299-319), RoughConductorCoLoc (:351-395)
enum { kHeadSmoothDielectric = 0, kHeadThinDielectric = 1, kHeadSmoothConductor = 2, kHeadRoughConductor = 3 };

__device__ __forceinline__ void coloc_head_point(int kind, float light, float distance, const float n[3], const float v[3],
                                                 const float kd[3], const float ks[3], float rough, float eta, float k,
                                                 GgxOut& o) {
    const float intensity = light / (distance * distance + 1e-10f);
    float dot = (v[0] * n[0] + v[1] * n[1]) + v[2] * n[2];
    dot = fminf(fmaxf(dot, 0.00001f), 0.99999f);
    float spec_scale;  // specular_rgb = intensity * ks * spec_scale  (left-to-right products below keep the reference order)
    float D = 1.0f, G = 1.0f, denom = 1.0f;
    if (kind == kHeadSmoothDielectric) {
        spec_scale = 0.04f;
    } else if (kind == kHeadThinDielectric) {
        spec_scale = (float)(0.04 + 0.96 * 0.96 * 0.04 / (1.0 - 0.04 * 0.04));
    } else {
        spec_scale = fresnel_conductor_exact(dot, eta, k);
        if (kind == kHeadRoughConductor) {
            const float alpha = fmaxf(rough, 0.0001f);
            const float pi_f = 3.14159274101257324219f;
            const float c2 = dot * dot;
            const float root = c2 + (1.0f - c2) / (alpha * alpha + 1e-10f);
            D = 1.0f / (pi_f * alpha * alpha * root * root + 1e-10f);
            const float g1 = smith_g1(dot, alpha);
            G = g1 * g1;
            denom = 4.0f * dot + 1e-10f;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        o.specular[c] = kind == kHeadRoughConductor ? intensity * ks[c] * spec_scale * D * G / denom : intensity * ks[c] * spec_scale;
        o.diffuse[c] = intensity * kd[c] * 0.0001f;
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
