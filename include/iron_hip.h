/*
 * iron_hip.h -- C ABI of libiron_hip.so, the MI355X (gfx950) implementation of IRON's
 * stage-2 forward render path (sphere-trace + normal/material MLPs + co-located GGX).
 *
 * The reference (arthurlirui/IRON) has no FFI layer: its boundary for this path is the Python
 * operator surface of models/raytracer.py, models/renderer_ggx.py, models/rendering_func.py
 * and models/fields.py.  Each entry point below names the reference function it replaces
 * (file:line, relative to the reference root); iron_amd/*.py binds them with ctypes and
 * re-exposes the reference's names.
 *
 * Conventions
 *   - plain C, no C++/torch types; every pointer is a DEVICE pointer unless marked HOST.
 *   - all arrays are contiguous row-major fp32 unless noted; masks are uint8 (0/1), i.e. the
 *     storage of a torch.bool tensor.
 *   - all buffers are caller-allocated and caller-owned; the library keeps nothing past the call
 *     except the packed weight copy made by iron_net_create().
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises
 *     the host, nothing allocates (graph-capturable) except iron_net_create/destroy.
 *   - return value: IRON_OK (0) or a negative iron_status; no exceptions / aborts cross the ABI.
 *   - handles are immutable after create and may be shared between host threads.
 */
#ifndef IRON_HIP_H
#define IRON_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IRON_ABI_VERSION 1

typedef enum iron_status {
    IRON_OK = 0,
    IRON_ERR_BAD_ARG = -1,      /* null pointer, negative size, misaligned buffer              */
    IRON_ERR_UNSUPPORTED = -2,  /* network shape / mode the kernels are not built for          */
    IRON_ERR_HIP = -3,          /* a HIP runtime call failed; see iron_last_hip_error()        */
    IRON_ERR_NO_DEVICE = -4,    /* no gfx950 device visible                                    */
    IRON_ERR_WORKSPACE = -5,    /* workspace too small                                         */
    IRON_ERR_RANGE = -6         /* IRON_H2_OVERFLOW=error: the previous call on this network left the fp16 range of the h2 core */
} iron_status;

int iron_version(void);
const char* iron_strerror(int status);
/* hipError_t (as int) of the last failing HIP call on this host thread, 0 if none. */
int iron_last_hip_error(void);

/* ---------------------------------------------------------------------------------------------
 * Networks.  Replaces the parameter side of models/fields.py:9-98 (SDFNetwork) and :141-239
 * (RenderingNetwork).  Input format = the reference state_dict: per linear layer `weight_v`
 * [out,in], `weight_g` [out] (old-style weight_norm, fields.py:75-76; NULL for a plain layer,
 * then weight_v is the weight itself) and `bias` [out].  The library folds the weight norm
 * (W = v * g/||v||_row) and re-packs into its MFMA fragment layout on the device.
 * ------------------------------------------------------------------------------------------- */
typedef struct iron_net iron_net_t; /* opaque */

typedef struct iron_linear {
    const float* weight_v; /* [out_dim, in_dim] */
    const float* weight_g; /* [out_dim] or NULL  */
    const float* bias;     /* [out_dim]          */
    int32_t out_dim;
    int32_t in_dim;
} iron_linear;

enum { IRON_NET_SDF = 0, IRON_NET_RENDER = 1, IRON_NET_NERF = 2 };
enum { IRON_MODE_IDR = 0, IRON_MODE_NO_VIEW_DIR = 1, IRON_MODE_NO_NORMAL = 2, IRON_MODE_POINTS_ONLY = 3 };

typedef struct iron_net_desc {
    int32_t kind;          /* IRON_NET_SDF | IRON_NET_RENDER | IRON_NET_NERF                        */
    int32_t n_linear;      /* number of linear layers (= n_layers + 1)                              */
    int32_t d_hidden;      /* hidden width (256)                                                    */
    int32_t d_out;         /* SDF: 257 (sdf + feature); RENDER: 1..3                                */
    int32_t multires;      /* PE levels on points (SDF: 6)                                          */
    int32_t multires_view; /* RENDER: PE levels on view_dirs (<=0: none)                            */
    int32_t skip_layer;    /* index of the skip-concat layer, -1 if none (SDF: 4)                   */
    int32_t mode;          /* RENDER: IRON_MODE_*                                                   */
    int32_t d_feature;     /* RENDER: width of the feature vector input (256)                       */
    int32_t squeeze_out;   /* RENDER: apply squeeze_out_scale*sigmoid()                             */
    float squeeze_out_scale;
    float output_bias;     /* RENDER: x = output_scale * (x + output_bias)                          */
    float output_scale;
    float scale;           /* SDF: input scale (fields.py:83,98)                                    */
} iron_net_desc;

/* Build a packed network on the current device.  `layers` is a HOST array of n_linear entries
 * whose pointers are DEVICE pointers.  Synchronises `stream` before returning (the source
 * tensors may be freed afterwards).  Must be re-run when parameters change. */
int iron_net_create(iron_net_t** out, const iron_net_desc* desc, const iron_linear* layers, void* stream);
int iron_net_destroy(iron_net_t* net);

/* Numeric envelope of the default ("h2", split-fp16) core: weights are checked at create (a network with a folded |w| >= 65 504
 * runs on the exact-fp32 core); activations and features are guarded at run time: every entry that ran a network on the h2 core
 * scans the values it returns, a non-finite one raises the network's flag, and the NEXT entry on that handle moves the network to
 * the exact-fp32 MFMA core for good (IRON_H2_OVERFLOW=error: returns IRON_ERR_RANGE instead).  The reference is plain fp32
 * (models/fields.py:82-98, 203-239), which the exact core reproduces over the whole fp32 range.
 *   iron_net_numeric_status: synchronises `stream`; *status_out = bit 0: an overflow was seen, bit 1: the network runs on the exact
 *                            core, bit 2: a flag is pending (the call just finished overflowed).
 *   iron_net_force_exact:    on != 0 pins the network to the exact core; 0 returns it to the default core and clears the status. */
int iron_net_numeric_status(const iron_net_t* net, int32_t* status_out, void* stream);
int iron_net_force_exact(iron_net_t* net, int32_t on);

/* ---------------------------------------------------------------------------------------------
 * Batched field queries
 * ------------------------------------------------------------------------------------------- */
/* SDFNetwork.forward / .sdf (models/fields.py:82-104).  x [n,3] -> out [n,out_cols],
 * out_cols = 1 (sdf only) or d_out (sdf + feature). */
int iron_sdf_forward(const iron_net_t* sdf, const float* x, int64_t n, float* out, int32_t out_cols, void* stream);

/* SDFNetwork.get_all(x, is_training=False) (models/fields.py:120-137): sdf [n], feature
 * [n,d_out-1], grad = d sdf/dx [n,3] (closed form, no autograd graph).  Any output may be NULL.
 * `workspace` (iron_sdf_get_all_workspace_bytes(sdf, n) bytes, 16-byte aligned, caller-owned, contents undefined afterwards) is
 * the tape of the reverse-mode kernel: forward pass, then ONE reverse sweep over transposed weights (csrc/getall_rev.hip) -- what
 * autograd.grad does in the reference.  The query returns 0 for a network that has no reverse stream (any shape other than the
 * reference's 8 x 256 / skip 4 / PE-6 / 257 outputs, or IRON_GETALL=fwd); with workspace == NULL the gradient is evaluated as
 * three forward-mode tangents instead (same results to rounding, 2.3x the matrix work). */
size_t iron_sdf_get_all_workspace_bytes(const iron_net_t* sdf, int64_t n);
int iron_sdf_get_all(const iron_net_t* sdf, const float* x, int64_t n, float* sdf_out, float* feature,
                     float* grad, void* workspace, size_t workspace_bytes, void* stream);

/* RenderingNetwork.forward (models/fields.py:203-239).  view_dirs may be NULL for modes that do
 * not read it.  out [n,d_out].
 * A net with a skip connection (skip_layer >= 1: the stage-1 colour net) parks the skip layer's partial sums in a scratch buffer
 * owned by the handle: calls on ONE such handle must be ordered on one stream (different handles are independent). */
int iron_render_forward(const iron_net_t* net, const float* points, const float* normals,
                        const float* view_dirs, const float* features, int64_t n, float* out, void* stream);

/* NeRF.forward (models/fields.py:299-327, use_viewdirs=True): pts [n,4] (the stage-1 background
 * parametrisation (x/r, 1/r)), view_dirs [n,3] -> alpha [n] (raw density), rgb [n,3] (raw).  The net is created with
 * kind IRON_NET_NERF, d_hidden 256, multires / multires_view = PE levels, skip_layer = the layer after which the input is
 * concatenated again (4), n_linear = D + 4 layers in the order pts_linears[0..D-1], alpha_linear, feature_linear,
 * views_linears[0], rgb_linear (plain nn.Linear: weight_g = NULL). */
int iron_nerf_forward(const iron_net_t* nerf, const float* pts4, const float* view_dirs, int64_t n, float* alpha, float* rgb,
                      void* stream);

/* Per-ray stages of NeuSRenderer.render (models/renderer.py:346-453; rows of at most 192 samples, one thread per ray):
 *   iron_neus_linspace    z = near + (far - near) * linspace(0,1,m)                              (:357-358)
 *   iron_neus_outside_z   z = far / rev[j] + offset: depths of the n_outside background samples            (:380-381)
 *   iron_neus_points      pts = o + d * z                                                        (:389, :236)
 *   iron_neus_up_sample   up_sample (:189-232) + sample_pdf(det=True) (:45-75): n_importance new depths per ray
 *   iron_neus_merge       the sort of cat([z, new_z]) as a merge of two ascending rows, sdf carried along (:238-246)
 *   iron_neus_mid_points  section lengths, mid points and per-sample dirs of render_core (:265-277); outside != 0: the
 *                         (x/r, 1/r) parametrisation of render_core_outside (:163-172), pts [n*m,4]
 *   iron_neus_need_background  which samples of the fed row the compositing takes from the background field: all outside samples
 *                         and the inside samples whose mid point lies outside the unit sphere (the rest are multiplied by
 *                         (1 - inside_sphere) = 0, :300-312); lets the caller evaluate the NeRF field on those only
 *   iron_neus_composite   render_core's alpha (logistic CDF), inside-sphere blend with the NeRF background, weights,
 *                         colour, weight_sum / weight_max, cdf, inside_sphere, eikonal statistics (:279-344, :174-178). */
/* extract_fields' lattice (models/renderer.py:9-31): pts [nx*ny*nz,3] = meshgrid(xs, ys, zs) in 'ij' order; the field values
 * are then one iron_sdf_forward (or any other batched query) over pts. */
int iron_grid_points(const float* xs, const float* ys, const float* zs, int32_t nx, int32_t ny, int32_t nz, float* pts, void* stream);
int iron_neus_linspace(const float* near, const float* far, const float* lin, int64_t n, int32_t m, float* z, void* stream);
int iron_neus_outside_z(const float* far, const float* rev, int64_t n, int32_t m, float offset, float* z, void* stream);
int iron_neus_points(const float* rays_o, const float* rays_d, const float* z, int64_t n, int32_t m, float* pts, void* stream);
int iron_neus_up_sample(const float* rays_o, const float* rays_d, const float* z, const float* sdf, int64_t n, int32_t m,
                        int32_t n_importance, float inv_s, float* new_z, void* stream);
int iron_neus_merge(const float* z_a, const float* s_a, int32_t m_a, const float* z_b, const float* s_b, int32_t m_b, int64_t n,
                    float* z_out, float* s_out, void* stream);
int iron_neus_mid_points(const float* rays_o, const float* rays_d, const float* z, int64_t n, int32_t m, float sample_dist,
                         int32_t outside, float* dists, float* pts, float* dirs, void* stream);
int iron_neus_need_background(const float* pts, int64_t n, int32_t m, int32_t mo, uint8_t* need, void* stream);
typedef struct iron_neus_composite_args {
    const float* dists;            /* [n,m]   section lengths of the inside samples                      */
    const float* pts;              /* [n*m,3] section mid points                                         */
    const float* dirs;             /* [n*m,3]                                                            */
    const float* sdf;              /* [n*m]                                                              */
    const float* grad;             /* [n*m,3] d sdf / dx                                                 */
    const float* color;            /* [n*m,3] colour network output                                      */
    const float* bg_dists;         /* [n,mo]  outside pass (NULL: no background model)                   */
    const float* bg_density;       /* [n*mo]  NeRF alpha output                                          */
    const float* bg_color;         /* [n*mo,3] NeRF rgb output                                           */
    const float* background_rgb;   /* [3] or NULL                                                        */
    int64_t n;
    int32_t m, mo;                 /* mo = m + n_outside when the background model is used               */
    float inv_s, cos_anneal_ratio;
    float* out_color;              /* [n,3]                                                              */
    float* weights;                /* [n, mo or m]                                                       */
    float* cdf;                    /* [n,m] or NULL                                                      */
    float* inside_sphere;          /* [n,m] or NULL                                                      */
    float* weight_sum;             /* [n]                                                                */
    float* weight_max;             /* [n]                                                                */
    float* gradient_error_acc;     /* [2]: sum(relax * (|g|-1)^2), sum(relax); or NULL                   */
} iron_neus_composite_args;
int iron_neus_composite(const iron_neus_composite_args* args, void* stream);
/* The same with the outside pass given as its ALPHA [n,mo] (what render_core itself takes, models/renderer.py:259-260,
 * 312-321) instead of the NeRF density: args->bg_density / bg_dists are ignored, args->bg_color [n*mo,3] is the outside
 * pass's sampled colour. */
int iron_neus_composite_alpha(const iron_neus_composite_args* args, const float* background_alpha, void* stream);
/* render_core_outside's compositing (models/renderer.py:174-187): density [n*mo] (the NeRF field's first output), dists
 * [n,mo], sampled_color [n*mo,3], background_rgb [3] or NULL -> alpha [n,mo] = 1 - exp(-softplus(density) dists), weights
 * [n,mo], color [n,3]. */
int iron_neus_outside_composite(const float* density, const float* dists, const float* sampled_color, const float* background_rgb,
                                int64_t n, int32_t mo, float* alpha, float* weights, float* color, void* stream);
/* sample_pdf (models/renderer.py:45-75): bins [n,n_bins], weights [n,n_bins-1] -> samples [n,n_samples] by inverting the
 * CDF of the piecewise-constant density at u [n,n_samples]; u == NULL is det=True (u = linspace(0.5/k, 1-0.5/k, k)). */
int iron_neus_sample_pdf(const float* bins, const float* weights, const float* u, int64_t n, int32_t n_bins, int32_t n_samples,
                         float* samples, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Pointwise geometry
 * ------------------------------------------------------------------------------------------- */
/* Camera.get_rays (models/raytracer.py:254-286).  k_inv3 = K^-1[:3,:3], c2w34 = C2W[:3,:4], both
 * HOST row-major.  uv [n,2] -> ray_o [n,3], ray_d [n,3] (normalised), ray_d_norm [n]. */
int iron_camera_rays(const float* k_inv3, const float* c2w34, const float* uv, int64_t n, float* ray_o,
                     float* ray_d, float* ray_d_norm, void* stream);

/* intersect_sphere (models/raytracer.py:223-237). */
int iron_intersect_sphere(const float* ray_o, const float* ray_d, int64_t n, float r, uint8_t* mask,
                          float* near, float* far, void* stream);

/* GGXColocatedRenderer.forward (models/renderer_ggx.py:82-146).  distance [n], normal/viewdir
 * [n,3], albedos [n,3], roughness [n]; tab_trans [5000], tab_diff_trans [50] (models/ggx/*.txt). */
int iron_ggx_colocated(float light, const float* distance, const float* normal, const float* viewdir,
                       const float* diffuse_albedo, const float* specular_albedo, const float* roughness,
                       const float* tab_trans, const float* tab_diff_trans, int64_t n, float* diffuse_rgb,
                       float* specular_rgb, float* rgb, void* stream);

/* smithG1 (models/renderer_ggx.py:12-16) on its own: cos_theta, alpha, out all [n]. */
int iron_smith_g1(const float* cos_theta, const float* alpha, int64_t n, float* out, void* stream);

/* SURVEY 8 row f-4 -- the fork's other co-located heads.
 * CompositeRenderer.forward (models/renderer_ggx.py:781-858), quirks included: the GGX NDF is evaluated with
 * alpha := 1.48958738 (:806), the `metallic` / `dielectric` weight maps are clamped and then unused (:829-831), and
 * "diffuse_rgb" is the same tensor as "rgb" (in-place alias, :847-853) -- hence no separate diffuse output.  All
 * parameter maps are raw network outputs (the clamps of :790-797 happen inside); [n,3] albedos, [n] scalars.
 * env_light != NULL selects use_env_light=True (intensity = clamp(env_light, 1e-6, 20), `distance` unused) and
 * env_light_out (optional) receives that intensity. */
typedef struct iron_composite_params {
    const float* diffuse_albedo;
    const float* specular_albedo;
    const float* specular_roughness;
    const float* metallic_eta;
    const float* metallic_k;
    const float* dielectric_eta;
    const float* env_light; /* NULL: point light */
} iron_composite_params;
int iron_composite_colocated(float light, const float* distance, const float* normal, const float* viewdir,
                             const iron_composite_params* p, const float* tab_trans, const float* tab_diff_trans, int64_t n,
                             float* specular_rgb, float* metallic_rgb, float* dielectric_rgb, float* rgb,
                             float* env_light_out, void* stream);

/* kind 0 SmoothDielectricRenderer (:171-204), 1 ThinDielectricRenderer (:229-267), 2 SmoothConductorCoLocRenderer
 * (:299-319), 3 RoughConductorCoLocRenderer (:351-395); eta, k: the conductor's constants (ignored by kinds 0, 1);
 * roughness [n] is read by kind 3 only.  (RoughPlasticCoLocRenderer / CoLocRenderer raise TypeError in the reference
 * -- a float is indexed at :404 -- and have no entry.) */
int iron_coloc_head(int32_t kind, float light, float eta, float k, const float* distance, const float* normal,
                    const float* viewdir, const float* diffuse_albedo, const float* specular_albedo, const float* roughness,
                    int64_t n, float* diffuse_rgb, float* specular_rgb, float* rgb, void* stream);

/* Image-space passes of raytrace_camera's silhouette handling (models/raytracer.py:554-570), [H,W] fp32:
 * iron_morph_closing3x3 = kornia.morphology.closing(depth, ones(3,3)) (erosion of the dilation, border never wins;
 * `tmp` is an [H,W] scratch image); iron_sobel_magnitude = kornia.filters.sobel(depth) (kernels / 8, replicate
 * border, sqrt(gx^2+gy^2+1e-6)).  kornia is not installable offline: both are restated from its documented
 * semantics and are parity-unpinned. */
int iron_morph_closing3x3(const float* depth, int32_t H, int32_t W, float* tmp, float* out, void* stream);
int iron_sobel_magnitude(const float* depth, int32_t H, int32_t W, float* out, void* stream);

/* Tail of locate_edge_points (models/raytracer.py:481-500) in one launch: points [n,3] are projected like Camera.project
 * (w2c16 = W2C, k16 = K, both 4x4 row-major HOST floats) into uv [n,2]; for every point with found[i] != 0 whose pixel index
 * floor(v) * W + floor(u) lies in [0, H*W) the pixel's entry of first [H*W] (int32, pre-set by the caller to INT32_MAX) is
 * lowered to i: afterwards first[p] is the first found candidate of pixel p -- the one unique() keeps -- or INT32_MAX. */
int iron_edge_pixels(const float* points, const uint8_t* found, int64_t n, const float* w2c16, const float* k16, int32_t H, int32_t W,
                     float* uv, int32_t* first, void* stream);

/* The fill_holes update of raytrace_camera (models/raytracer.py:558-564) on the device: where the closed depth image
 * `depth_closed` (iron_morph_closing3x3) makes a hit of a non-convergent pixel, the reference rewrites depth at those
 * pixels, sets the mask to depth_closed > 1e-2 and recomputes distance = depth * ray_d_norm and points = ray_o + ray_d *
 * distance for EVERY pixel -- and does nothing at all when no pixel changes.  `flag` (device int32, scratch) carries that
 * any() between the two launches, so there is no host synchronisation.  All arrays [n] / [n,3]; conv is uint8. */
int iron_fill_holes(const float* depth_closed, const float* ray_o, const float* ray_d, const float* ray_d_norm, int64_t n,
                    float* depth, uint8_t* conv, float* distance, float* points, int32_t* flag, void* stream);

/* render_edge_pixels (models/raytracer.py:665-729), inference form, as two launches around the side-ray trace + shade:
 * iron_edge_sides: edge point gradients [n,3] and projections edge_uv [n,2], w2c_rot9 = W2C[:3,:3] row-major (HOST
 * float[9]) -> side_uv [2n,2] (the n positive-side samples centre - 0.707 n2d first, then the n negative-side ones) and
 * pos_weight [n] = 1 - (a - sin a) / 2pi, a = 2 acos(clamp(((uv - centre) . n2d) / 0.707, 0, 1)) (:680-698).
 * iron_edge_blend: side_color [2n,3] (same order) -> color[p] = pos * w + neg * (1 - w), normal[p] = edge_grad,
 * uv[p] = edge_uv, points[p] = edge_points at p = pixel_idx[i] (int64, flat pixel index; out-of-range entries skipped)
 * of the [n_pixels, .] image buffers (:706-729). */
int iron_edge_sides(const float* edge_uv, const float* edge_grad, const float* w2c_rot9, int64_t n, float* side_uv,
                    float* pos_weight, void* stream);
int iron_edge_blend(const float* side_color, const float* pos_weight, const float* edge_grad, const float* edge_uv,
                    const float* edge_points, const int64_t* pixel_idx, int64_t n, int64_t n_pixels, float* color, float* normal,
                    float* uv, float* points, void* stream);

/* The surface walk of locate_edge_points (models/raytracer.py:441-478) in one launch: every start point walks along
 * the surface (step_size per step, at most max_step steps) until |n.v| <= dot_threshold seen from cam_origin3 (HOST
 * float[3]); points [n,3] receives the final positions, found [n] whether the silhouette was reached.  Needs the h2
 * core for this network (IRON_ERR_UNSUPPORTED otherwise: walk with iron_sdf_get_all instead). */
int iron_edge_walk(const iron_net_t* sdf, const float* start, int64_t n, const float* cam_origin3, int32_t max_step,
                   float step_size, float dot_threshold, float* points, uint8_t* found, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Sphere tracer.  Replaces RayTracer.forward = sphere_tracing + ray_sampler + rootfind
 * (models/raytracer.py:45-220) for a batch of n rays, with the per-call chunking of
 * raytrace_pixels (:378-392) expressed as `chunk`: rays [k*chunk, (k+1)*chunk) share the
 * bisection iteration count exactly as one reference RayTracer.forward call does.
 * ------------------------------------------------------------------------------------------- */
typedef struct iron_trace_params {
    float sdf_threshold;          /* 5e-5 */
    int32_t sphere_tracing_iters; /* 16   */
    int32_t n_steps;              /* 128  */
    int64_t chunk;                /* rays per reference call (<=0: all n in one chunk) */
} iron_trace_params;

typedef struct iron_trace_stats { /* written to DEVICE memory, all int64 */
    int64_t n_evals;       /* SDF point evaluations actually performed                          */
    int64_t n_sphere_conv; /* rays convergent after sphere tracing                              */
    int64_t n_sampler;     /* rays handed to the dense sampler                                  */
    int64_t n_bisect;      /* rays bisected                                                     */
    int64_t n_conv;        /* convergent rays at the end                                        */
    int64_t n_evals_ref;   /* evaluations the REFERENCE algorithm makes on the same rays: sphere-trace
                              evals + 128 per sampled ray + (chunk count + 1) per bisected ray
                              (SURVEY 8d's E; the HIP sampler stops early, so n_evals <= n_evals_ref) */
    int64_t n_evals_sphere;/* evaluations made by the sphere-tracing kernel                     */
    int64_t reserved;          /* 0; non-zero = k_sampler workgroups that left their work queue by the poll bound (a bug: report it) */
} iron_trace_stats;

size_t iron_trace_workspace_bytes(int64_t n, const iron_trace_params* p);

/* lin_steps: n_steps floats = torch.linspace(0,1,n_steps) (raytracer.py:144-146), DEVICE.
 * Outputs: conv uint8[n], points [n,3], sdf [n], dist [n] (state of non-convergent rays as the
 * reference leaves it).  stats may be NULL.  chunk_iters_io (int32[n_chunks], DEVICE) may be NULL;
 * see iron_trace_phase for the multi-rank protocol. */
int iron_trace(const iron_net_t* sdf, const iron_trace_params* p, const float* lin_steps, const float* ray_o,
               const float* ray_d, const float* near, const float* far, const uint8_t* work, int64_t n,
               uint8_t* conv, float* points, float* sdf_out, float* dist, iron_trace_stats* stats,
               void* workspace, size_t workspace_bytes, void* stream);

/* The three stages of RayTracer.forward as separate calls, for callers that use the reference's methods directly (models/raytracer.py
 * :105-140 sphere_tracing, :142-197 ray_sampler, :199-220 rootfind; tests/test_raytracer.py).  One call = one reference call: the
 * bisection count is global to its n rays.  `workspace` as for iron_trace (iron_trace_workspace_bytes(n, p)).
 *   stage 0 sphere_tracing: in0 = min_dis, in1 = max_dis, work  ->  mask_out = convergent, unfinished_out, points, sdf_out, dist
 *   stage 1 ray_sampler:    in0 = min_dis, in1 = max_dis        ->  mask_out = rays with a bracketed root, points, sdf_out, dist
 *                           (zeros for the others, as the reference returns them)
 *   stage 2 rootfind:       in0 = f_low, in1 = f_high, in2 = d_low, in3 = d_high  ->  points = p_mid, dist = d_mid, sdf_out = f_mid
 *                           (mask_out: scratch, n bytes; the reference's in-place update of its four bracket arguments is not made) */
int iron_trace_stage(int32_t stage, const iron_net_t* sdf, const iron_trace_params* p, const float* lin_steps, const float* ray_o,
                     const float* ray_d, const float* in0, const float* in1, const float* in2, const float* in3, const uint8_t* work,
                     int64_t n, uint8_t* mask_out, uint8_t* unfinished_out, float* points, float* sdf_out, float* dist,
                     void* workspace, size_t workspace_bytes, void* stream);

/* Multi-rank form: rays of one reference chunk may live on several ranks, so the chunk-global
 * bisection count needs one MAX all-reduce between the two halves.  phase 0 = sphere trace +
 * sampler + per-ray bisection, writing each local chunk's own count to chunk_iters[n_chunks];
 * the caller all-reduces (MAX) that array, then phase 1 finishes the bisection with it.
 * ray_index [n] (int64, may be NULL = identity) gives each ray's position in the full image so
 * that chunk = ray_index / p->chunk. */
int iron_trace_phase(int32_t phase, const iron_net_t* sdf, const iron_trace_params* p, const float* lin_steps,
                     const float* ray_o, const float* ray_d, const float* near, const float* far,
                     const uint8_t* work, const int64_t* ray_index, int64_t n, int32_t* chunk_iters,
                     int64_t n_chunks, uint8_t* conv, float* points, float* sdf_out, float* dist,
                     iron_trace_stats* stats, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Shading.  Replaces render_normal_and_color (models/raytracer.py:593-662) with the driver's GGX
 * render_fn (render_surface.py:117-156): for every ray with conv != 0: get_all -> normalise ->
 * get_materials (models/rendering_func.py:5-16) -> GGXColocatedRenderer; other rays get zeros.
 * Outputs are full-size [n,3] / [n]; any may be NULL.
 * ------------------------------------------------------------------------------------------- */
typedef struct iron_shade_nets {
    const iron_net_t* sdf;
    const iron_net_t* diffuse_albedo;
    const iron_net_t* specular_albedo;
    const iron_net_t* specular_roughness;
} iron_shade_nets;

typedef struct iron_shade_out {
    float* color;              /* [n,3] rgb                    */
    float* diffuse_color;      /* [n,3]                        */
    float* specular_color;     /* [n,3]                        */
    float* diffuse_albedo;     /* [n,3]                        */
    float* specular_albedo;    /* [n,3]                        */
    float* specular_roughness; /* [n]                          */
    float* normal;             /* [n,3] normalised             */
} iron_shade_out;

size_t iron_shade_workspace_bytes(int64_t n);

int iron_shade_ggx(const iron_shade_nets* nets, float light, int32_t is_metal, const float* tab_trans,
                   const float* tab_diff_trans, const float* ray_o, const float* ray_d, const float* points,
                   const uint8_t* conv, int64_t n, const iron_shade_out* out, void* workspace,
                   size_t workspace_bytes, void* stream);

/* The same with the composite render_fn (render_surface.py:159-234): get_all -> normalise ->
 * get_materials_comp (models/rendering_func.py:19-49: eight material networks) -> CompositeRenderer.forward.
 * Scalar maps are [n]; "diffuse_color" receives the same values as "color" (the reference's in-place alias). */
typedef struct iron_shade_comp_nets {
    const iron_net_t* sdf;
    const iron_net_t* diffuse_albedo;
    const iron_net_t* specular_albedo;
    const iron_net_t* specular_roughness;
    const iron_net_t* metallic;
    const iron_net_t* dielectric;
    const iron_net_t* metallic_eta;
    const iron_net_t* metallic_k;
    const iron_net_t* dielectric_eta;
} iron_shade_comp_nets;

typedef struct iron_shade_comp_out {
    float* color;              /* [n,3] */
    float* diffuse_color;      /* [n,3] == color */
    float* specular_color;     /* [n,3] */
    float* diffuse_albedo;     /* [n,3] */
    float* specular_albedo;    /* [n,3] */
    float* specular_roughness; /* [n]   */
    float* metallic_eta;       /* [n]   */
    float* metallic_k;         /* [n]   */
    float* dielectric_eta;     /* [n]   */
    float* normal;             /* [n,3] normalised */
    float* metallic_rgb;       /* [n,3] */
    float* metallic;           /* [n]   */
    float* dielectric_rgb;     /* [n,3] */
    float* dielectric;         /* [n]   */
} iron_shade_comp_out;

size_t iron_shade_composite_workspace_bytes(int64_t n);

int iron_shade_composite(const iron_shade_comp_nets* nets, float light, const float* tab_trans, const float* tab_diff_trans,
                         const float* ray_o, const float* ray_d, const float* points, const uint8_t* conv, int64_t n,
                         const iron_shade_comp_out* out, void* workspace, size_t workspace_bytes, void* stream);

/* CU budget of the launches that follow (all streams, process-wide): the persistent kernels fill `n_cus` compute units instead of
 * the whole device; n_cus <= 0 removes the limit.  Returns the device's CU count.  For callers that run two launch sequences side by
 * side on two streams (iron_amd.raytracer.render_camera: hit shading beside the silhouette pass; no counterpart in the reference,
 * whose render_camera, models/raytracer.py:778-814, is one sequence). */
int32_t iron_set_cu_limit(int32_t n_cus);

/* Number of independent parts (1..4) the tracer cuts the rays of a call into, each part's kernel chain on its own stream (the
 * caller's + library-owned side streams, forked / joined with events); results do not depend on it.  parts <= 0 restores the
 * default (IRON_TRACE_SPLIT, else 1: measured no faster on MI355X, csrc/trace.hip).  Returns the previous setting.  Process-wide;
 * no counterpart in the reference (RayTracer.forward, models/raytracer.py:45-103, is one sequence of masked torch ops). */
int32_t iron_set_trace_split(int32_t parts);

/* ---------------------------------------------------------------------------------------------
 * Diagnostics (no reference counterpart): per-kernel device time from hipEvents recorded on the
 * caller's stream around each compute kernel.  Off by default.  iron_profile_read blocks on the
 * recorded events, adds their elapsed milliseconds / launch counts into the caller's arrays
 * (IRON_PROF_KINDS entries each) and clears the pending list.
 * ------------------------------------------------------------------------------------------- */
enum {
    IRON_PROF_SPHERE = 0, IRON_PROF_SAMPLER = 1, IRON_PROF_BISECT_A = 2, IRON_PROF_BISECT_B = 3,
    IRON_PROF_SDF_GRAD = 4, IRON_PROF_MATERIAL = 5, IRON_PROF_GGX = 6, IRON_PROF_SDF_FORWARD = 7,
    IRON_PROF_KINDS = 8
};
int iron_profile_enable(int32_t on);
int iron_profile_read(double* ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* IRON_HIP_H */
