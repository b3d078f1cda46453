/* nrf.h -- C ABI of libnrf.so, the MI355X (gfx950) implementation of the NeRF per-ray rendering path.
 *
 * The reference (duudual/NeRF, pure Python/PyTorch) has no FFI: the boundary it offers is the set of
 * Python callables of positional_encoding.py / network.py / sampling.py / render.py.
 * Each entry point below names the reference lines it replaces; the nerf_amd modules keep the reference's
 * Python signatures and calls these through ctypes (see INTEGRATION.md for the binding a reference
 * maintainer would add).
 *
 * Conventions
 *  - extern "C", plain C types.  Every tensor argument is a raw DEVICE pointer to a contiguous
 *    row-major f32 buffer (tensor.data_ptr()) with explicit sizes.
 *  - The caller allocates and owns every input, output and workspace buffer.  The library never
 *    allocates, frees or retains device memory.
 *  - Every call enqueues work on `stream` (a hipStream_t passed as void*; NULL = default stream) and
 *    returns without synchronising.
 *  - Return value: 0 on success, otherwise a nrf_status; nrf_last_error_string() describes the last
 *    failure on the calling thread.  No C++ exception crosses the boundary.
 */
#ifndef NRF_H_
#define NRF_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bumped whenever a struct layout or a signature changes (200: nrf_net_t carries the LayerNorm parameters); a binding
 * compares nrf_version() with the NRF_VERSION it was written against before passing structs */
#define NRF_VERSION 230
#define NRF_MAX_D 16

typedef enum {
    NRF_OK = 0,
    NRF_ERR_INVALID = 1,     /* bad argument (null pointer, negative size, ...) */
    NRF_ERR_UNSUPPORTED = 2, /* shape outside what the kernels are built for */
    NRF_ERR_LAUNCH = 3       /* HIP reported an error at launch */
} nrf_status;

int nrf_version(void);
const char *nrf_last_error_string(void);

/* ---- network description: the nn.Linear parameters of network.py:29-44 (weight [out,in], bias [out]) */
typedef struct {
    int32_t D, W, input_ch, input_ch_views, output_ch, use_viewdirs;
    int32_t skips_mask;   /* bit i set <=> i in skips (network.py:52) */
    int32_t reserved;     /* input_ch_time of a DNeRF_Straightforward descriptor (nrf_dnerf_straight_*), else 0 */
    const float *pts_w[NRF_MAX_D];
    const float *pts_b[NRF_MAX_D];
    const float *views_w, *views_b, *feature_w, *feature_b, *alpha_w, *alpha_b, *rgb_w, *rgb_b;
    const float *out_w, *out_b; /* output_linear (use_viewdirs == 0) */
    /* LayerNorm variant (vanilla_dynamic/network.py:46-48, :75-76, :88-89, :276-300): nn.LayerNorm(W) after every trunk
     * Linear (weight / bias of pts_layernorms[i], or of _time_layernorms[i] for a time-net descriptor) and nn.LayerNorm(W/2)
     * after the view layer.  use_layernorm == 0: the pointers are ignored.  eps = 1e-5.  Inference and (220) training in f32 for
     * W = 64, 128, 256: nrf_query_fwd_train saves the normalised values, nrf_query_bwd / nrf_query_bwd_input / nrf_dnerf_time_bwd
     * run the LayerNorm backward inside the dgrad chain and write the affine parameters' gradients through the same fields of
     * `grads`. */
    int32_t use_layernorm, reserved2;
    const float *ln_w[NRF_MAX_D];
    const float *ln_b[NRF_MAX_D];
    const float *views_ln_w, *views_ln_b;
} nrf_net_t;

/* how the query kernel obtains the network input */
typedef struct {
    int32_t mode;            /* 0: raw points/directions, positional encoding fused into the kernel;
                                1: rows already embedded by the caller (NeRF.forward(x), network.py:46-47) */
    int32_t multires;        /* L of get_embedder (positional_encoding.py:39-64); 3+6L == net.input_ch */
    int32_t multires_views;  /* L for directions; 3+6L == net.input_ch_views (0 if !use_viewdirs)     */
    int32_t multires_time;   /* D-NeRF time embedding L (vanilla_dynamic/positional_encoding.py:71-97) */
} nrf_embed_t;

/* K0  ray matrix of render() from (rows of the image, intrinsics, pose) -- rays.py:5-27 (get_rays), render.py:283-290
 * (viewdirs; c2w_static = the camera when c2w_staticcam is given, may be NULL), rays.py:52-83 (ndc_rays),
 * render.py:298-304 (bounds).  c2w / c2w_static: DEVICE pointers to 3 rows of >= 4 floats, `stride` floats apart.
 * rays [nrows * W, use_viewdirs ? 11 : 8], rows row0 .. row0+nrows-1 of the H x W image. */
int nrf_gen_rays(int32_t H, int32_t W, float fx, float fy, float cx, float cy, const float *c2w, int32_t c2w_stride,
                 const float *c2w_static, int32_t static_stride, int32_t row0, int32_t nrows, float near, float far,
                 int32_t use_viewdirs, int32_t ndc, float ndc_focal, float ndc_near, float *rays, void *stream);

/* K0b  one batch of train.py's shuffled ray table without materialising it -- train.py:147-153 (rays_rgb: get_rays_np
 * rays.py:30-49 of every training pose, stacked with the image), :183-185 (batch = rows i_batch .. i_batch+N_rand).
 * rows [n]: DEVICE int64 rows of the UNSHUFFLED table, row = slot * H*W + y*W + x (the caller owns the permutation);
 * img_index [n_slots] (i_train; NULL = identity); poses: image i at poses + i*pose_pitch, 3 rows of >= 4 floats
 * `pose_stride` apart; images [n_images, H, W, 3] f32.  f64 = 1: float64 arithmetic rounded once (what NumPy >= 2 computes for a
 * float64 K), 0: float32 (NumPy 1.x / get_rays order).  rays [2, n, 3] (origins, directions), rgb [n, 3].
 * Rows outside the table, and slots whose img_index entry is not in [0, n_images), produce NaN. */
int nrf_gather_ray_batch(const int64_t *rows, int64_t n, const int32_t *img_index, int32_t n_slots, const float *poses,
                         int32_t pose_stride, int64_t pose_pitch, const float *images, int32_t n_images, int32_t H,
                         int32_t W, double fx, double fy, double cx, double cy, int32_t f64, float *rays, float *rgb,
                         void *stream);

/* K1  coarse depths  -- render.py:186-208 (linspace, near/far lerp or lindisp, optional stratified jitter)
 * rays [N, ncol] = o(3) d(3) near far [viewdirs(3)]  (render.py:181-184); t_rand: NULL (perturb == 0) or
 * [N, S] uniform randoms (render.py:200-206).  z_vals [N, S].  Bit-exact with the reference for perturb==0. */
int nrf_sample_coarse(const float *rays, int64_t n_rays, int32_t ncol, int32_t n_samples, int32_t lindisp,
                      const float *t_rand, float *z_vals, void *stream);

/* pts = o + d * z -- render.py:210, :224.  pts [N, S, 3].  Bit-exact (separate multiply and add). */
int nrf_points(const float *rays, int64_t n_rays, int32_t ncol, const float *z_vals, int32_t n_samples,
               float *pts, void *stream);

/* A1  Embedder.embed -- positional_encoding.py:5-36.  x [P, dims] -> out [P, dims*(include_input+2*num_freqs)] */
int nrf_pe_fwd(const float *x, int64_t n_points, int32_t dims, int32_t num_freqs, float max_freq_log2,
               int32_t log_sampling, int32_t include_input, float *out, void *stream);

/* autograd of A1 w.r.t. x -- what loss.backward() computes through embed_fn(pts + dx) on the D-NeRF path
 * (vanilla_dynamic/network.py:331-333).  g_out [P, out_dim] -> g_x [P, dims]. */
int nrf_pe_bwd(const float *x, const float *g_out, int64_t n_points, int32_t dims, int32_t num_freqs,
               float max_freq_log2, int32_t log_sampling, int32_t include_input, float *g_x, void *stream);

/* K2+K3 weights: nn.Linear layout -> the MFMA-fragment-ordered layout the query kernels stream.
 * nrf_mlp_packed_floats returns the number of floats `packed` must hold (<= 0: unsupported shape). */
int64_t nrf_mlp_packed_floats(const nrf_net_t *net, const nrf_embed_t *emb);
int nrf_mlp_pack_weights(const nrf_net_t *net, const nrf_embed_t *emb, float *packed, void *stream);

/* K2+K3 network query, fused PE + MLP -- render.py:31-56 (run_network) + network.py:46-69 (NeRF.forward).
 * mode 0: pts [P, 3]; viewdirs [P / pts_per_ray, 3] (one per ray, render.py:49) or NULL when !use_viewdirs.
 * mode 1: pts is x [P, input_ch + input_ch_views] (already embedded), viewdirs ignored.
 * raw [P, 4] (use_viewdirs) or [P, output_ch].  `net` supplies the dimensions only (pointers unused);
 * `packed` must have been produced by nrf_mlp_pack_weights with the same net/emb. */
int nrf_query_fwd(const nrf_net_t *net, const nrf_embed_t *emb, const float *packed, const float *pts,
                  const float *viewdirs, int64_t n_points, int32_t pts_per_ray, float *raw, void *stream);

/* ---- training (train.py:226-243: forward with saved activations, then autograd's backward) ----
 * Workspaces are caller-allocated; sizes in floats from the *_floats queries (<= 0: unsupported shape).
 *   acts : inputs and per-layer activations saved by the forward        (~10.1 KB / point at W=256, D=8)
 *   gbuf : per-layer gradients w.r.t. pre-activations (backward scratch) (~10.0 KB / point)
 *   slab : per-chunk partial weight gradients, one region per nn.Linear (backward scratch; <= 1.1 GB at W=256, D=8) */
int64_t nrf_train_acts_floats(const nrf_net_t *net, int64_t n_points);
int64_t nrf_train_grads_floats(const nrf_net_t *net, int64_t n_points);
int64_t nrf_train_slab_floats(const nrf_net_t *net, int64_t n_points);

/* nrf_query_fwd that also fills `acts` (same results bit for bit). */
int nrf_query_fwd_train(const nrf_net_t *net, const nrf_embed_t *emb, const float *packed, const float *pts,
                        const float *viewdirs, int64_t n_points, int32_t pts_per_ray, float *raw, float *acts,
                        void *stream);

/* K3b  backward of the network query w.r.t. every nn.Linear parameter -- autograd of network.py:46-69.
 * g_raw [P, 4 | output_ch] = d(loss)/d(raw).  `grads` has the layout of nrf_net_t; its pointers are the
 * DESTINATION gradient buffers (same shapes as the parameters), which are overwritten, not accumulated. */
int nrf_query_bwd(const nrf_net_t *net, const nrf_embed_t *emb, const float *packed, const float *g_raw,
                  const float *acts, float *gbuf, float *slab, int64_t n_points, const nrf_net_t *grads,
                  void *stream);

/* nrf_query_bwd that ALSO returns d(loss)/d(pts) [P,3] -- needed when the query points depend on parameters, as in
 * the D-NeRF chain x + dx (vanilla_dynamic/network.py:331-339; autograd through embed_fn).  Embed mode 0 only. */
int nrf_query_bwd_input(const nrf_net_t *net, const nrf_embed_t *emb, const float *packed, const float *g_raw,
                        const float *acts, float *gbuf, float *slab, int64_t n_points, const nrf_net_t *grads,
                        const float *pts, float *g_pts, void *stream);

/* Allocation-free variant of nrf_query_bwd.  NO counterpart in the reference: it exists for callers that cannot allocate
 * device memory (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  Same inputs, outputs
 * and kernels as nrf_query_bwd, but `gbuf` and `slab` live in a 64 MiB file-scope device array inside the library (each GPU
 * the library runs on holds its copy until the process ends) and the wgrad reduction uses <= 8 partial slabs per GEMM
 * instead of <= 256, so the results agree with nrf_query_bwd's to f32 summation order (1e-5 of a gradient's scale), not
 * bit for bit.  Limit: nrf_train_grads_floats(net, n) + 8-chunk slabs must fit the array -- 3040 points (95 tiles) at D = 8,
 * W = 256 (the error string of a refused call states the limit for the given network); larger passes use nrf_query_bwd.
 * Calls no hipMalloc / hipFree.  All calls on one GPU share the array: calls that overlap on one GPU (two streams, two
 * threads) race; serialise them. */
int nrf_query_bwd_static(const nrf_net_t *net, const nrf_embed_t *emb, const float *packed, const float *g_raw,
                         const float *acts, int64_t n_points, const nrf_net_t *grads, void *stream);

/* ---- D-NeRF deformation chain (vanilla_dynamic/) ----
 * The time ("deformation") net of DNeRF_Deformation (vanilla_dynamic/network.py:254-305) is described with
 * nrf_net_t as: pts_w/pts_b = the `_time` layers, out_w/out_b = `_time_out` (W -> 3), input_ch = position
 * channels, input_ch_views = TIME channels (layer 0 reads input_ch + time channels), use_viewdirs = 0,
 * skips_mask as the canonical net (network.py:245).  emb->multires_time = L of get_time_embedder. */
int64_t nrf_dnerf_time_packed_floats(const nrf_net_t *time_net, const nrf_embed_t *emb);
int nrf_dnerf_pack_time_weights(const nrf_net_t *time_net, const nrf_embed_t *emb, float *packed, void *stream);

/* K6  fused deformation query -- vanilla_dynamic/render.py:49-124 (run_network_dnerf) +
 * vanilla_dynamic/network.py:307-341 (DNeRF_Deformation.forward):  dx = time_net(gamma(x), gamma(t));
 * raw = occ_net(gamma(x + dx), gamma(d)).  pts [P,3]; viewdirs [P/pts_per_ray, 3]; times [P/pts_per_ray]
 * (one per ray) or [P] when times_per_point != 0; raw [P,4]; dx [P,3].  packed_occ from
 * nrf_mlp_pack_weights(occ_net), packed_time from nrf_dnerf_pack_time_weights.  `netchunk` and
 * `zero_canonical` reproduce the reference's per-netchunk "t == 0 -> skip the time net" decision
 * (network.py:315-322, taken from the first point of each chunk of `netchunk` points). */
int nrf_dnerf_query_fwd(const nrf_net_t *time_net, const nrf_net_t *occ_net, const nrf_embed_t *emb,
                        const float *packed_time, const float *packed_occ, const float *pts, const float *viewdirs,
                        const float *times, int64_t n_points, int32_t pts_per_ray, int32_t times_per_point,
                        int64_t netchunk, int32_t zero_canonical, float *raw, float *dx, void *stream);

/* DNeRF_Straightforward -- vanilla_dynamic/network.py:100-196: NeRF.forward with cat[gamma(x), gamma(t)] as trunk input and
 * as skip (layer 0 reads input_ch + input_ch_time columns, the layer after a skip input_ch + input_ch_time + W), view head
 * as network.py:179-189.  net->reserved = input_ch_time; emb->multires_time = L of get_time_embedder.
 * mode 0: pts [P,3], viewdirs [P/pts_per_ray,3], times [P/pts_per_ray] (or [P] when times_per_point).
 * mode 1 (what forward(x, ts) receives): pts = x [P, input_ch + input_ch_views] embedded, times = ts [P, input_ch_time]
 * embedded.  raw [P,4].  Inference (the reference's own run_network_dnerf cannot call this network type at HEAD). */
int64_t nrf_dnerf_straight_packed_floats(const nrf_net_t *net, const nrf_embed_t *emb);
int nrf_dnerf_straight_pack_weights(const nrf_net_t *net, const nrf_embed_t *emb, float *packed, void *stream);
int nrf_dnerf_straight_query_fwd(const nrf_net_t *net, const nrf_embed_t *emb, const float *packed, const float *pts,
                                 const float *viewdirs, const float *times, int64_t n_points, int32_t pts_per_ray,
                                 int32_t times_per_point, float *raw, void *stream);

/* DNeRF_Straightforward training (vanilla_dynamic/network.py:151-196 under autograd; the reference trains this network
 * with loss.backward(), vanilla_dynamic/train.py:266-277): the forward above on raw points (embed mode 0) that also saves
 * inputs and activations (acts: nrf_train_acts_floats), and the parameter gradients from that workspace (gbuf / slab:
 * nrf_train_grads_floats / nrf_train_slab_floats; `grads` as in nrf_query_bwd, pts_w[l] with the widths of this network:
 * [W, input_ch + input_ch_time] for layer 0, [W, input_ch + input_ch_time + W] for the layer after a skip). */
int nrf_dnerf_straight_fwd_train(const nrf_net_t *net, const nrf_embed_t *emb, const float *packed, const float *pts,
                                 const float *viewdirs, const float *times, int64_t n_points, int32_t pts_per_ray,
                                 int32_t times_per_point, float *raw, float *acts, void *stream);
int nrf_dnerf_straight_bwd(const nrf_net_t *net, const nrf_embed_t *emb, const float *packed, const float *g_raw,
                           const float *acts, float *gbuf, float *slab, int64_t n_points, const nrf_net_t *grads,
                           void *stream);

/* The same query on the bf16x3 arithmetic (three bf16 MFMAs per f32 product on hi/residual operands: f32-grade results,
 * no counterpart in the reference): `packed` of nrf_dnerf_straight_bf16x3_packed_bytes bytes from
 * nrf_dnerf_straight_bf16x3_pack_weights; raw points only (embed mode 0), W = 128 or 256, one skip layer. */
int64_t nrf_dnerf_straight_bf16x3_packed_bytes(const nrf_net_t *net, const nrf_embed_t *emb);
int nrf_dnerf_straight_bf16x3_pack_weights(const nrf_net_t *net, const nrf_embed_t *emb, void *packed, void *stream);
int nrf_dnerf_straight_query_fwd_bf16x3(const nrf_net_t *net, const nrf_embed_t *emb, const void *packed, const float *pts,
                                        const float *viewdirs, const float *times, int64_t n_points, int32_t pts_per_ray,
                                        int32_t times_per_point, float *raw, void *stream);

/* ... and on plain bf16 MFMAs (the north star's optional bf16 variant, tolerance 1e-2): same arguments and limits. */
int64_t nrf_dnerf_straight_bf16_packed_bytes(const nrf_net_t *net, const nrf_embed_t *emb);
int nrf_dnerf_straight_bf16_pack_weights(const nrf_net_t *net, const nrf_embed_t *emb, void *packed, void *stream);
int nrf_dnerf_straight_query_fwd_bf16(const nrf_net_t *net, const nrf_embed_t *emb, const void *packed, const float *pts,
                                      const float *viewdirs, const float *times, int64_t n_points, int32_t pts_per_ray,
                                      int32_t times_per_point, float *raw, void *stream);

/* ---- bf16 MFMA variant of the network query (BASELINE north star: "optional bf16 variant"; no counterpart in the
 * reference, which is fp32 only).  Same arguments and results as nrf_query_fwd with embed mode 0, computed with
 * bf16 weights / activations and f32 accumulation (v_mfma_f32_32x32x16_bf16); tolerance 1e-2 on rendered values.
 * `packed` is a separate bf16 weight stream of nrf_mlp_bf16_packed_bytes() bytes.  Inference only. */
int64_t nrf_mlp_bf16_packed_bytes(const nrf_net_t *net, const nrf_embed_t *emb);
int nrf_mlp_bf16_pack_weights(const nrf_net_t *net, const nrf_embed_t *emb, void *packed, void *stream);
int nrf_query_fwd_bf16(const nrf_net_t *net, const nrf_embed_t *emb, const void *packed, const float *pts,
                       const float *viewdirs, int64_t n_points, int32_t pts_per_ray, float *raw, void *stream);

/* The same 16-bit variant with f16 instead of bf16 for the hidden activations and their weights as well (the encoded inputs
 * are f16 in both): same MFMA rate and f32 accumulation (v_mfma_f32_32x32x16_f16), 11 mantissa bits instead of 8.  Range is
 * what bf16 buys, so this variant saturates: activations and weights beyond +-65504 are clamped to it (round-to-nearest
 * conversion + packed clamp).  Same stream size (nrf_mlp_bf16_packed_bytes) and arguments; `packed` must come from
 * nrf_mlp_f16_pack_weights.  No counterpart in the reference.  Inference only. */
int nrf_mlp_f16_pack_weights(const nrf_net_t *net, const nrf_embed_t *emb, void *packed, void *stream);
int nrf_query_fwd_f16(const nrf_net_t *net, const nrf_embed_t *emb, const void *packed, const float *pts,
                      const float *viewdirs, int64_t n_points, int32_t pts_per_ray, float *raw, void *stream);

/* D-NeRF training (autograd of vanilla_dynamic/network.py:307-341, the loss.backward() of vanilla_dynamic/train.py):
 *   forward : dx  = nrf_dnerf_time_fwd_train(time net; saves acts_t);   x' = x + dx   (caller; dx := 0 where the
 *             reference's t == 0 shortcut applies);   raw = nrf_query_fwd_train(canonical net, x'; saves acts_c)
 *   backward: nrf_query_bwd_input(canonical net, g_raw) -> canonical parameter grads + g_x';
 *             nrf_dnerf_time_bwd(time net, g_dx = g_x' [+ d(loss)/d(dx)]) -> time-net parameter grads.
 * `grads` of nrf_dnerf_time_bwd uses the time-net descriptor layout (pts_w/pts_b = _time layers, out_w/out_b = _time_out).
 * Workspace sizes: nrf_train_acts_floats / nrf_train_grads_floats / nrf_train_slab_floats with the same descriptor. */
int nrf_dnerf_time_fwd_train(const nrf_net_t *time_net, const nrf_embed_t *emb, const float *packed_time, const float *pts,
                             const float *times, int64_t n_points, int32_t pts_per_ray, int32_t times_per_point,
                             float *dx, float *acts, void *stream);
int nrf_dnerf_time_bwd(const nrf_net_t *time_net, const nrf_embed_t *emb, const float *packed_time, const float *g_dx,
                       const float *acts, float *gbuf, float *slab, int64_t n_points, const nrf_net_t *grads,
                       void *stream);

/* nrf_dnerf_time_bwd with the weight-gradient GEMMs on the bf16x3 arithmetic (see nrf_query_bwd_bf16x3 below) and, when
 * `packed_bwd_x3` (nrf_mlp_bf16x3_bwd_pack_weights of this descriptor; may be NULL) is given, the deformation net's
 * activation-gradient chain too: it ends at the net's own parameters, so nothing downstream amplifies the arithmetic's
 * 1e-5 (the canonical net's chain, which carries d/dx into this one, stays exact f32). */
int nrf_dnerf_time_bwd_bf16x3(const nrf_net_t *time_net, const nrf_embed_t *emb, const float *packed_time,
                              const void *packed_bwd_x3, const float *g_dx, const float *acts, float *gbuf, float *slab,
                              int64_t n_points, const nrf_net_t *grads, void *stream);

/* ---- bf16x3: the same query at f32-grade accuracy on the bf16 matrix pipe (no counterpart in the reference).
 * Every f32 operand is carried as hi = bf16(v), lo = bf16(v - hi); a product is the three MFMAs Wh*xh + Wh*xl + Wl*xh
 * accumulated in f32.  Raw outputs agree with the f32 kernel to ~1e-5 of their scale (bf16: 5e-3), rendered values to the
 * f32 bar of 1e-4; about 3x the f32-MFMA kernel's rate.  `packed` = nrf_mlp_bf16x3_packed_bytes() bytes.  Inference only. */
int64_t nrf_mlp_bf16x3_packed_bytes(const nrf_net_t *net, const nrf_embed_t *emb);
int nrf_mlp_bf16x3_pack_weights(const nrf_net_t *net, const nrf_embed_t *emb, void *packed, void *stream);
int nrf_query_fwd_bf16x3(const nrf_net_t *net, const nrf_embed_t *emb, const void *packed, const float *pts,
                         const float *viewdirs, int64_t n_points, int32_t pts_per_ray, float *raw, void *stream);

/* Training forward on the bf16x3 arithmetic (no counterpart in the reference): nrf_query_fwd_train with every product as
 * three bf16 MFMAs on hi/residual operands.  `packed` is the bf16x3 stream of nrf_mlp_bf16x3_pack_weights; raw agrees with
 * nrf_query_fwd_train's to ~1e-5 of its scale; `acts` is filled in the SAME f32 layout (nrf_train_acts_floats), so
 * nrf_query_bwd / nrf_query_bwd_input / nrf_query_bwd_static take it as it is (with the f32 `packed` of nrf_mlp_pack_weights). */
int nrf_query_fwd_train_bf16x3(const nrf_net_t *net, const nrf_embed_t *emb, const void *packed, const float *pts,
                               const float *viewdirs, int64_t n_points, int32_t pts_per_ray, float *raw, float *acts,
                               void *stream);

/* Backward on the bf16x3 arithmetic (no counterpart in the reference): nrf_query_bwd (pts == g_pts == NULL) or
 * nrf_query_bwd_input with every GEMM as three bf16 MFMAs per product on hi/residual operands; same workspaces (f32
 * layout) and result layout.  Weight-gradient GEMMs: operands split as they leave LDS.  Activation-gradient chain: W^T
 * blocks streamed from `packed_bwd_x3` (nrf_mlp_bf16x3_bwd_pack_weights; NULL, input gradients requested, or a shape
 * without a bf16x3 kernel: that chain runs on the exact f32 kernel instead).  `packed` is the f32 copy of
 * nrf_mlp_pack_weights (head vectors; the f32 chain).  Gradients agree with nrf_query_bwd's to ~1e-5 of their scale. */
int64_t nrf_mlp_bf16x3_bwd_packed_bytes(const nrf_net_t *net, const nrf_embed_t *emb);
int nrf_mlp_bf16x3_bwd_pack_weights(const nrf_net_t *net, const nrf_embed_t *emb, void *packed, void *stream);
int nrf_query_bwd_bf16x3(const nrf_net_t *net, const nrf_embed_t *emb, const float *packed, const void *packed_bwd_x3,
                         const float *g_raw, const float *acts, float *gbuf, float *slab, int64_t n_points,
                         const nrf_net_t *grads, const float *pts, float *g_pts, void *stream);

/* The bf16x3 training pass with the saved activations in f16 ("bf16x3_a16"; no counterpart in the reference): the step is
 * bound by the bytes of its workspaces, and the post-ReLU activations are what the forward writes and
 * the weight-gradient GEMMs read back.  nrf_query_fwd_train_bf16x3_a16 saves them (and the encoded inputs) as f16 --
 * round-to-nearest, saturating at 65504; two bytes a value, each section point-major ("P16": per 32-channel tile two
 * 1 KiB blocks j of [point 32][lane half 2][8 values], value e = accumulator register 8j + e of that lane half; DESIGN.md
 * section 3), inside a workspace of the usual nrf_train_acts_floats size; the layout is private to this pair of calls -- and nrf_query_bwd_bf16x3_a16 (parameter gradients; `packed_bwd_x3` required) reads them as
 * exact hi + lo bf16 pairs.  Gradient blocks stay f32.  Forward results are those of nrf_query_fwd_train_bf16x3 bit for bit;
 * weight gradients carry the activations' f16 rounding (2^-12 per value: ~2e-4 of a gradient's norm against 2e-3, the
 * bar every training arithmetic is tested to).  Static nets, W = 128 / 256, embed mode 0. */
int nrf_query_fwd_train_bf16x3_a16(const nrf_net_t *net, const nrf_embed_t *emb, const void *packed, const float *pts,
                                   const float *viewdirs, int64_t n_points, int32_t pts_per_ray, float *raw, float *acts,
                                   void *stream);
int nrf_query_bwd_bf16x3_a16(const nrf_net_t *net, const nrf_embed_t *emb, const float *packed, const void *packed_bwd_x3,
                             const float *g_raw, const float *acts, float *gbuf, float *slab, int64_t n_points,
                             const nrf_net_t *grads, void *stream);

/* "bf16x3_w16" (no counterpart in the reference): the gradient blocks saved as f16 as well.  The activation-gradient chain
 * (bf16x3 arithmetic, unchanged) runs on gradients scaled by the power of two that brings the pass's largest |g_raw| into
 * [2^10, 2^11) -- the chain is linear, so this is what mixed-precision training calls loss scaling, decided per pass on the device --
 * and saves its blocks as f16 (saturating at 65504); the weight-gradient GEMMs are then ONE f16 MFMA per tile pair on the two
 * f16 workspaces (a third of the matrix work, no operand split, half the bytes on both sides) and the slab reduction undoes
 * the scale exactly.  Forward: nrf_query_fwd_train_bf16x3_a16.  Weight and bias gradients carry a 2^-12 rounding of both
 * operands: within ~1e-3 of their scale of the exact pass (the gradient tests' bar is 2e-3).  `slab` as nrf_train_slab_floats
 * (its last 64 floats hold the scale's scalar). */
int nrf_query_bwd_bf16x3_w16(const nrf_net_t *net, const nrf_embed_t *emb, const float *packed, const void *packed_bwd_x3,
                             const float *g_raw, const float *acts, float *gbuf, float *slab, int64_t n_points,
                             const nrf_net_t *grads, void *stream);

/* bf16 variant of K6 (BASELINE configs[4]: "deformation network ... bf16 MFMA path").  ONE weight stream holds both
 * nets (time net first); dx and x + dx stay in f32.  Same arguments / results as nrf_dnerf_query_fwd otherwise. */
int64_t nrf_dnerf_bf16_packed_bytes(const nrf_net_t *time_net, const nrf_net_t *occ_net, const nrf_embed_t *emb);
int nrf_dnerf_bf16_pack_weights(const nrf_net_t *time_net, const nrf_net_t *occ_net, const nrf_embed_t *emb,
                                void *packed, void *stream);
int nrf_dnerf_query_fwd_bf16(const nrf_net_t *time_net, const nrf_net_t *occ_net, const nrf_embed_t *emb,
                             const void *packed, const float *pts, const float *viewdirs, const float *times,
                             int64_t n_points, int32_t pts_per_ray, int32_t times_per_point, int64_t netchunk,
                             int32_t zero_canonical, float *raw, float *dx, void *stream);

/* bf16x3 variant of K6: the deformation chain with every product as three bf16 MFMAs on hi/residual operands (see the
 * bf16x3 block above); dx, x + dx and the re-embedding stay in f32.  Same arguments / results as nrf_dnerf_query_fwd_bf16;
 * `packed` = nrf_dnerf_bf16x3_packed_bytes() bytes. */
int64_t nrf_dnerf_bf16x3_packed_bytes(const nrf_net_t *time_net, const nrf_net_t *occ_net, const nrf_embed_t *emb);
int nrf_dnerf_bf16x3_pack_weights(const nrf_net_t *time_net, const nrf_net_t *occ_net, const nrf_embed_t *emb,
                                  void *packed, void *stream);
int nrf_dnerf_query_fwd_bf16x3(const nrf_net_t *time_net, const nrf_net_t *occ_net, const nrf_embed_t *emb,
                               const void *packed, const float *pts, const float *viewdirs, const float *times,
                               int64_t n_points, int32_t pts_per_ray, int32_t times_per_point, int64_t netchunk,
                               int32_t zero_canonical, float *raw, float *dx, void *stream);

/* Fused optimizer step: torch.optim.Adam as model.py:52 / vanilla_dynamic/model.py:115 configure it
 * (no weight decay, no amsgrad), all tensors in one launch.  The pointer tables and `numel` are HOST arrays of
 * n_tensors entries holding DEVICE pointers; steps[i] is tensor i's 1-based step count (torch counts per parameter); grad_scale: NULL or a DEVICE scalar the
 * gradients are multiplied by (the clip_grad_norm_ coefficient, vanilla_dynamic/train.py:273-274). */
int nrf_adam_step(int32_t n_tensors, float *const *params, const float *const *grads, float *const *exp_avg,
                  float *const *exp_avg_sq, const int64_t *numel, const int64_t *steps, float lr, float beta1, float beta2,
                  float eps, const float *grad_scale, void *stream);

/* K4  raw2outputs -- render.py:82-131.  raw [N,S,4], z_vals [N,S], rays_d rows of `d_stride` floats
 * (pass rays+3 with d_stride = ncol to read directions out of the ray matrix), noise NULL or [N,S]
 * (already multiplied by raw_noise_std).  Outputs rgb [N,3], disp [N], acc [N], weights [N,S], depth [N]. */
int nrf_composite_fwd(const float *raw, const float *z_vals, const float *rays_d, int32_t d_stride,
                      int64_t n_rays, int32_t n_samples, const float *noise, int32_t white_bkgd, float *rgb_map,
                      float *disp_map, float *acc_map, float *weights, float *depth_map, void *stream);

/* K4b backward of raw2outputs w.r.t. raw -- autograd of render.py:100-129.  g_rgb [N,3] is required;
 * g_disp, g_acc, g_depth [N] and g_weights [N,S] may be NULL (train.py:232-239 only uses rgb).  g_raw [N,S,4]. */
int nrf_composite_bwd(const float *raw, const float *z_vals, const float *rays_d, int32_t d_stride,
                      int64_t n_rays, int32_t n_samples, const float *noise, int32_t white_bkgd,
                      const float *g_rgb, const float *g_acc, const float *g_depth, const float *g_weights,
                      float *g_raw, void *stream);

/* K5  hierarchical resampling -- sampling.py:5-60 on weights[...,1:-1] and mid-points, then
 * sort(cat(z_vals, z_samples)) and std(z_samples) -- render.py:219-224, :238.
 * z_vals [N,S], weights [N,S] (full coarse rows); u NULL (det, perturb == 0) or [N, n_importance].
 * z_samples [N, n_importance] (may be NULL), z_merged [N, S+n_importance], z_std [N] (may be NULL).
 * Bit-exact with the reference CPU path on identical inputs. */
int nrf_sample_fine(const float *z_vals, const float *weights, int64_t n_rays, int32_t n_samples,
                    int32_t n_importance, const float *u, float *z_samples, float *z_merged, float *z_std,
                    void *stream);

/* sample_pdf alone -- sampling.py:5-60.  bins [N, nb], weights [N, nb-1] -> samples [N, n_samples]. */
int nrf_sample_pdf(const float *bins, const float *weights, int64_t n_rows, int32_t n_bins, int32_t n_samples,
                   const float *u, float *samples, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NRF_H_ */
