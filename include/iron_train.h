/* iron_train.h -- C ABI of iron_amd/csrc/libiron_train.so: the BACKWARD passes of the stage-2 render operators
 * (BASELINE config C3).  gfx950 only.
 *
 * The reference trains through torch.autograd: `SDFNetwork.get_all(is_training=True)` builds a second-order graph
 * (models/fields.py:120-137), `RenderingNetwork.forward` and `GGXColocatedRenderer.forward` are ordinary differentiable
 * torch code (models/fields.py:203-239, models/renderer_ggx.py:82-146), and `loss.backward()` in render_surface.py:533-653
 * walks that graph.  Each entry below is the backward of ONE of those operators, evaluated in closed form on the GPU; the
 * Python side (iron_amd/autograd.py) wraps forward (libiron_hip.so) + backward (this library) as torch.autograd.Function
 * so the reference's own render_fn / reparam_points / loss code runs unchanged on top.
 *
 * Method: the hit points of one call are processed as a batch, layer by layer, in fp32.  The forward activations are
 * recomputed from the inputs and kept in the caller's workspace (HBM is 288 GB: 37 KB per point for the SDF net), the
 * per-layer products (Z = X W^T, dX = dZ W, dW = dZ^T X with K = number of points) are this library's own split-fp16 MFMA GEMMs
 * (csrc/gemm_h2.h, exported as iron_train_gemm; no BLAS library is linked),
 * everything between them (positional encoding and its derivative, softplus-100 first and second derivative, the
 * forward-mode tangent rows that carry d(normal)/d(theta), weight-norm fold and its backward, column sums, the GGX
 * derivative) is hand-written HIP.
 *
 * Conventions as iron_hip.h: device pointers, fp32 row-major contiguous, caller-owned buffers and workspace, work enqueued
 * on `stream`, IRON_OK or a negative iron_status.  Parameter gradients are WRITTEN (not accumulated) to the d_* pointers
 * of each layer.  Threading: the two GEMM-based entries (iron_sdf_backward, iron_render_backward) serialise their host-side
 * enqueue per process; the work itself runs asynchronously on the caller's stream.
 * Operand range: every GEMM operand is split into two fp16 pieces; gradient operands are rescaled by a power of two from their
 * absolute maximum, all other operands (weights, recomputed activations, tangent rows) must stay below 65 504 in magnitude.
 */
#ifndef IRON_TRAIN_H
#define IRON_TRAIN_H

#include "iron_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* One linear layer as the reference stores it (nn.utils.weight_norm: W = weight_g * weight_v / ||weight_v||_row). */
typedef struct iron_train_layer {
    const float* weight_v; /* [out,in]                                        */
    const float* weight_g; /* [out] or NULL (plain nn.Linear: W = weight_v)   */
    const float* bias;     /* [out]                                           */
    float* d_weight_v;     /* [out,in] out                                    */
    float* d_weight_g;     /* [out] out (ignored when weight_g is NULL)       */
    float* d_bias;         /* [out] out                                       */
    int32_t out_dim, in_dim;
} iron_train_layer;

/* SDFNetwork (models/fields.py:9-98): PE(multires) -> n_linear layers, softplus(beta=100) between them, one skip layer
 * whose input is cat[x, PE]/sqrt(2); scale must be 1. */
typedef struct iron_sdf_train_desc {
    int32_t n_linear;   /* 9 for the 8x256 network */
    int32_t multires;   /* 6 */
    int32_t skip_layer; /* 4, or -1 */
    const iron_train_layer* layers;
} iron_sdf_train_desc;

/* Backward of get_all (fields.py:120-137): given dL/d sdf [n], dL/d feature [n, d_out-1], dL/d gradient [n,3] (each may
 * be NULL = zero) at the points x [n,3], writes dL/d(weight_v, weight_g, bias) of every layer.  The second-order part
 * (dL/d gradient) is the reverse pass over a forward-mode tangent along v = dL/d gradient: <v, grad_x sdf> is the
 * directional derivative of the network along v, so its parameter gradient is one more (value, tangent) reverse sweep. */
size_t iron_sdf_backward_workspace_bytes(const iron_sdf_train_desc* desc, int64_t n);
int iron_sdf_backward(const iron_sdf_train_desc* desc, const float* x, int64_t n, const float* d_sdf, const float* d_feature,
                      const float* d_gradient, void* workspace, size_t workspace_bytes, void* stream);

/* RenderingNetwork (models/fields.py:141-239): input cat per `mode` (IRON_MODE_*), ReLU MLP with an optional skip layer,
 * y = output_scale * (z + output_bias), optional squeeze_out_scale * sigmoid(y). */
typedef struct iron_render_train_desc {
    int32_t n_linear;
    int32_t mode;
    int32_t multires, multires_view;
    int32_t d_feature, d_out;
    int32_t skip_layer; /* -1: none */
    int32_t squeeze_out;
    float output_bias, output_scale, squeeze_out_scale;
    const iron_train_layer* layers;
} iron_render_train_desc;

/* Backward of RenderingNetwork.forward: d_out [n,d_out] -> parameter gradients and (each nullable) d_points [n,3],
 * d_normals [n,3], d_view_dirs [n,3], d_features [n,d_feature].  normals / view_dirs may be NULL where the mode ignores
 * them. */
size_t iron_render_backward_workspace_bytes(const iron_render_train_desc* desc, int64_t n);
int iron_render_backward(const iron_render_train_desc* desc, const float* points, const float* normals, const float* view_dirs,
                         const float* features, int64_t n, const float* d_out, float* d_points, float* d_normals, float* d_view_dirs,
                         float* d_features, void* workspace, size_t workspace_bytes, void* stream);

/* Backward of GGXColocatedRenderer.forward (models/renderer_ggx.py:82-146).  Upstream d_diffuse_rgb / d_specular_rgb /
 * d_rgb [n,3] (each nullable); outputs (each nullable): d_light [1] (device, written), d_distance [n], d_normal [n,3],
 * d_viewdir [n,3], d_diffuse_albedo [n,3], d_specular_albedo [n,3], d_roughness [n].  The two Mitsuba tables are
 * piecewise constant in (cos, alpha), so they carry no gradient (torch gives none either: integer indexing). */
int iron_ggx_colocated_backward(float light, const float* distance, const float* normal, const float* viewdir,
                                const float* diffuse_albedo, const float* specular_albedo, const float* roughness,
                                const float* tab_trans, const float* tab_diff, int64_t n, const float* d_diffuse_rgb,
                                const float* d_specular_rgb, const float* d_rgb, float* d_light, float* d_distance, float* d_normal,
                                float* d_viewdir, float* d_diffuse_albedo, float* d_specular_albedo, float* d_roughness, void* stream);

/* Backward of CompositeRenderer.forward (models/renderer_ggx.py:781-858), both branches (p->env_light != NULL: use_env_light,
 * intensity = clamp(env_light, 1e-6, 20); d_light / d_distance are then zero and d_env_light receives the gradient, including
 * that of the returned "env_light" key, d_env_light_out).  Upstream: d_rgb (the reference returns the SAME tensor as "rgb" and "diffuse_rgb"; pass
 * the sum of both keys' gradients), d_specular_rgb, d_metallic_rgb, d_dielectric_rgb, each [n,3] or NULL.  Outputs, each
 * nullable: d_light [1]; d_distance, the four scalar maps [n]; d_normal, d_viewdir, the two albedos [n,3].  Clamped inputs
 * carry gradient on the closed clamp interval, the diffuse tables none. */
typedef struct iron_composite_grads_in {
    const float* d_rgb;
    const float* d_specular_rgb;
    const float* d_metallic_rgb;
    const float* d_dielectric_rgb;
    const float* d_env_light_out; /* [n] or NULL (env-light branch: upstream of the returned clamped env light) */
} iron_composite_grads_in;
typedef struct iron_composite_grads_out {
    float* d_light;
    float* d_distance;
    float* d_normal;
    float* d_viewdir;
    float* d_diffuse_albedo;
    float* d_specular_albedo;
    float* d_specular_roughness;
    float* d_metallic_eta;
    float* d_metallic_k;
    float* d_dielectric_eta;
    float* d_env_light; /* [n] or NULL */
} iron_composite_grads_out;
int iron_composite_colocated_backward(float light, const float* distance, const float* normal, const float* viewdir,
                                      const iron_composite_params* p, const float* tab_trans, const float* tab_diff, int64_t n,
                                      const iron_composite_grads_in* upstream, const iron_composite_grads_out* out, void* stream);

/* Backward of iron_coloc_head (SmoothDielectric / ThinDielectric / SmoothConductorCoLoc / RoughConductorCoLoc renderers,
 * models/renderer_ggx.py:149-395; same `kind`, eta, k).  Upstream d_diffuse_rgb / d_specular_rgb / d_rgb [n,3], each nullable;
 * outputs as iron_ggx_colocated_backward (d_roughness is non-zero for kind 3 only). */
int iron_coloc_head_backward(int32_t kind, float light, float eta, float k, const float* distance, const float* normal,
                             const float* viewdir, const float* diffuse_albedo, const float* specular_albedo, const float* roughness,
                             int64_t n, const float* d_diffuse_rgb, const float* d_specular_rgb, const float* d_rgb, float* d_light,
                             float* d_distance, float* d_normal, float* d_viewdir, float* d_diffuse_albedo, float* d_specular_albedo,
                             float* d_roughness, void* stream);

/* NeRF (models/fields.py:243-327, use_viewdirs=True): D ReLU layers of width W on PE(input_pts), `skip` = the layer after
 * whose activation the encoded input is concatenated IN FRONT (fields.py:309-310; -1: none), then alpha (1), feature (W), one
 * view layer on cat[feature, PE(views)] and rgb (3).  layers = the D point layers followed by alpha, feature, view, rgb; plain
 * nn.Linear (weight_g NULL).  Backward: d_alpha [n,1] / d_rgb [n,3] (each nullable) -> parameter gradients; the inputs get
 * none (the reference feeds sample positions computed without grad, renderer.py:163-172). */
typedef struct iron_nerf_train_desc {
    int32_t D, W;
    int32_t d_in, d_in_view;
    int32_t multires, multires_view;
    int32_t skip;
    const iron_train_layer* layers;
} iron_nerf_train_desc;
size_t iron_nerf_backward_workspace_bytes(const iron_nerf_train_desc* desc, int64_t n);
int iron_nerf_backward(const iron_nerf_train_desc* desc, const float* pts, const float* views, int64_t n, const float* d_alpha,
                       const float* d_rgb, void* workspace, size_t workspace_bytes, void* stream);

/* Backward of iron_neus_composite (the compositing of NeuSRenderer.render_core, models/renderer.py:279-344, with the
 * background blend of :174-178).  `fwd` = the forward call's argument block (its output pointers are ignored).  Upstream,
 * each nullable: d_color [n,3], d_weight_sum [n], d_weights [n, mo or m], d_gradient_error [1] (device; needs relax_count =
 * the forward's gradient_error_acc + 1, i.e. the sum of the relax mask).  Outputs, each nullable: d_sdf [n*m], d_grad
 * [n*m,3] (through the annealed cosine and the eikonal statistic), d_sample_color [n*m,3], d_inv_s [1], d_bg_density
 * [n*mo], d_bg_color [n*mo,3]. */
typedef struct iron_neus_composite_grads {
    const float* d_color;
    const float* d_weight_sum;
    const float* d_weights;
    const float* d_gradient_error;
    const float* relax_count;
    float* d_sdf;
    float* d_grad;
    float* d_sample_color;
    float* d_inv_s;
    float* d_bg_density;
    float* d_bg_color;
} iron_neus_composite_grads;
int iron_neus_composite_backward(const iron_neus_composite_args* fwd, const iron_neus_composite_grads* grads, void* stream);

/* The layer product every backward pass above is built from, on its own (tests / micro-benchmarks): row-major
 * C[m,n] = op(A) op(B) + beta C in fp32, computed by the library's hand-written split-fp16 MFMA GEMMs (csrc/gemm_h2.h; no BLAS
 * library).  op = 0: operand as stored ([m,k] / [k,n]); 1: transposed ([k,m] / [n,k]).  (0,1) = Z = X W^T, (0,0) = dX = dZ W,
 * (1,0) = dW = dZ^T X (there lda = m, ldb = n, ldc = n are implied: tightly packed, as inside the library); (1,1) is refused.
 * In (0,0) and (1,0) A is treated as a gradient: scaled by a power of two from its absolute maximum before the fp16 split. */
size_t iron_train_gemm_workspace_bytes(int32_t op_a, int32_t m, int32_t n);
int iron_train_gemm(int32_t op_a, int32_t op_b, int32_t m, int32_t n, int32_t k, const float* A, int32_t lda, const float* B, int32_t ldb,
                    float beta, float* C, int32_t ldc, void* workspace, size_t workspace_bytes, void* stream);

/* Operand range of the backward passes.  Their layer products split every fp32 operand into two fp16 pieces (csrc/gemm_h2.h); a
 * gradient operand is scaled by a power of two from its absolute maximum first, the other operands (activations, tangent rows,
 * weights) are split as they are, so an element beyond fp16's largest number (|x| > 65 504) or a non-finite one yields inf / NaN
 * gradients where the reference's fp32 autograd (render_surface.py:533-653) has ordinary numbers.  Every splitting kernel raises
 * a sticky device flag when it meets such an element; this call synchronises `stream` and returns IRON_ERR_RANGE if the flag is
 * up (and lowers it when `reset` is non-zero), IRON_OK otherwise.  The networks of the reference's scenes stay far inside the
 * range (weight_g up to 137, pre-activations about 10). */
int iron_train_numeric_status(int32_t reset, void* stream);

/* Diagnostics: last hipError_t seen by this library on the calling thread (iron_train_last_blas_status: kept for ABI
 * stability, always 0: the library links no BLAS). */
int iron_train_last_hip_error(void);
int iron_train_last_blas_status(void);

#ifdef __cplusplus
}
#endif
#endif
