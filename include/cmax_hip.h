/*
 * cmax_hip.h -- C ABI of libcmax_hip.so: the MI355X (gfx950) contrast-maximization inner loop
 * (event warp -> image of warped events -> contrast objective + analytic gradient).
 *
 * The reference (tub-rip/event_based_optical_flow) is pure Python and has NO FFI; its "plugin API"
 * for this path is three Python classes.  Each entry point below names the
 * reference function it replaces (file:line relative to the reference root); the Python host
 * layer in event_based_optical_flow_amd/ mirrors those classes and calls this ABI through ctypes.
 * INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *  - every function returns int: 0 ok, <0 bad argument (CMAX_E*), >0 a hipError_t.
 *    cmax_last_error() returns a thread-local description of the last failure.
 *  - all pointers are DEVICE pointers borrowed from the caller (PyTorch's allocator) unless the
 *    name ends in _host; nothing is retained after the call returns except by cmax_set_events.
 *  - work is enqueued on `stream` (a hipStream_t, e.g. torch.cuda.current_stream().cuda_stream);
 *    no entry point synchronises the device.
 *  - events are row-major [n,4] = (x, y, t, p); x is the ROW coordinate, y the COLUMN
 *    (src/event_image_converter.py:344-345); images are row-major [H, W].
 *  - dtype: CMAX_F32 or CMAX_F64 (leaf ops compute in the dtype of their inputs, like the reference,
 *    whose outputs follow the events' dtype, src/event_image_converter.py:338).
 */
#ifndef CMAX_HIP_H
#define CMAX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CMAX_ABI_VERSION 3

/* dtypes */
#define CMAX_F32 0
#define CMAX_F64 1

/* motion models -- Warp.warp_event dispatcher, src/warp.py:156-199 */
#define CMAX_MODEL_2DOF 0  /* "2d-translation", "rigid-optical-flow"  src/warp.py:483-522 */
#define CMAX_MODEL_DENSE 1 /* "dense-flow"                            src/warp.py:263-313 */
#define CMAX_MODEL_VOXEL 2 /* "dense-flow-voxel"                      src/warp.py:315-396 */

/* reference-time modes -- Warp.calculate_reftime, src/warp.py:201-233 */
#define CMAX_REF_FIRST 0 /* t_min exactly */
#define CMAX_REF_LAST 1  /* t_max exactly */
#define CMAX_REF_FRAC 2  /* t_min + (t_max - t_min) * frac ("middle" = 0.5, "before" = -1, "after" = 2) */

/* contrast functions */
#define CMAX_COST_VARIANCE 0 /* ImageVariance      src/costs/image_variance.py:27-71 */
#define CMAX_COST_GRADMAG 1  /* GradientMagnitude  src/costs/gradient_magnitude.py:60-76 */

/* flow propagation schemes -- src/utils/flow_utils.py */
#define CMAX_SCHEME_BURGERS 0 /* inviscid_burger_flow_to_voxel_torch  567-639 */
#define CMAX_SCHEME_UPWIND 1  /* upwind_flow_to_voxel_torch           439-493 */

/* error codes */
#define CMAX_EINVAL -1   /* bad argument */
#define CMAX_ENOMEM -2   /* workspace allocation failed */
#define CMAX_ESTATE -3   /* call order (e.g. objective before set_events) */
#define CMAX_ENODEV -4   /* no usable gfx950 device / RCCL not loadable */
#define CMAX_ECOMM -5    /* an RCCL call failed (message in cmax_last_error) */

typedef void *cmax_stream_t; /* hipStream_t */
typedef struct cmax_handle_s *cmax_handle_t;

const char *cmax_last_error(void);
int cmax_abi_version(void);

/* =============================================================================================
 * Leaf operators (stateless).  One per reference leaf function; dtype-generic.
 * ============================================================================================= */

/* min / max of the event timestamps -> tminmax[2] (device doubles).
 * Replaces nt_min / nt_max over events[..., 2] in Warp.calculate_reftime (src/warp.py:216-224). */
int cmax_tminmax(const void *events, int dtype, int64_t n, double *tminmax, cmax_stream_t stream);

/* Warp.warp_event (src/warp.py:156-199) incl. calculate_reftime (201-233) and calculate_dt
 * (235-259): warped[n,4] = (x', y', dt, p).
 *   ref_mode/ref_frac : reference time (CMAX_REF_*), evaluated on the device from tminmax[2]
 *   normalize_t       : dt /= (max(dt) - min(dt))            (src/warp.py:254-259)
 *   motion            : theta[2] | flow[2,H,W] | voxel[T,2,H,W], same dtype as events
 *   dt_out            : optional [n] copy of dt (saved for the backward pass)
 *   bin_out           : optional int32[n], voxel model only: time bin per event (-1 = none)  */
int cmax_warp_events(const void *events, int dtype, int64_t n, int model, const void *motion, int T,
                     int H, int W, const double *tminmax, int ref_mode, double ref_frac,
                     int normalize_t, void *warped, void *dt_out, int32_t *bin_out,
                     cmax_stream_t stream);

/* adjoint of cmax_warp_events w.r.t. the motion (what torch autograd derives for warp.py:304-307,
 * 357-362, 506-520): given gwarped[n,4] (only columns 0,1 are read)
 *   2DOF : gmotion[2]        = sum_e dt_e * (gx_e, gy_e)        (dtype, overwritten)
 *   DENSE: gmotion[2,H,W]   += -dt_e * g_e at the source pixel  (zeroed inside)
 *   VOXEL: gmotion[T,2,H,W] likewise into bin(e)                                            */
int cmax_warp_events_bwd(const void *events, int dtype, int64_t n, int model, int T, int H, int W,
                         const void *dt, const int32_t *bin, const void *gwarped, void *gmotion,
                         cmax_stream_t stream);

/* EventImageConverter.bilinear_vote_tensor / count_event_tensor
 * (src/event_image_converter.py:316-374, 209-255).
 *   xy      : pointer to x' of event 0; `stride` elements between events (4 for an [n,4] array)
 *   weight  : optional per-event weight [n]; else the scalar `wscalar`   (330-331)
 *   Hp, Wp  : PADDED image size; ph, pw: padding offsets                  (28, 344-345)
 *   eps     : 1e-6 (torch branch, 340) or 1e-8 (numpy branch, 282)
 *   count   : !=0 -> every in-bounds corner += 1                          (251-254)
 *   img     : [Hp,Wp], overwritten                                                          */
int cmax_vote(const void *xy, int dtype, int64_t stride, int64_t n, const void *weight,
              double wscalar, int Hp, int Wp, int ph, int pw, double eps, int count, void *img,
              cmax_stream_t stream);

/* adjoint of cmax_vote (autograd of scatter_add_ = gather; floor has zero derivative):
 *   gxy[n,2] = (dL/dx', dL/dy'),  gw[n] = dL/dweight (optional).  G is [Hp,Wp].             */
int cmax_vote_bwd(const void *xy, int dtype, int64_t stride, int64_t n, const void *weight,
                  double wscalar, int Hp, int Wp, int ph, int pw, double eps, const void *G,
                  void *gxy, void *gw, cmax_stream_t stream);

/* Gaussian blur of create_image_from_events_tensor (src/event_image_converter.py:153-159):
 * 3 taps exp(-0.5 (k/sigma)^2) normalised, reflect-101 padding, separable.  adjoint != 0 applies
 * the transposed operator (backward pass).  in != out.                                       */
int cmax_blur3(const void *in, int dtype, int H, int W, double sigma, int adjoint, void *out,
               cmax_stream_t stream);

/* numpy-branch blur of create_image_from_events_numpy (src/event_image_converter.py:122-124):
 * scipy.ndimage.gaussian_filter(image, sigma) -- radius int(4 sigma + 0.5), edge-duplicating 'reflect'
 * boundary, axis 0 then axis 1.  tmp: scratch image of the same size.                          */
int cmax_gaussian_filter(const void *in, int dtype, int H, int W, double sigma, void *tmp, void *out,
                         cmax_stream_t stream);

/* ImageVariance.calculate / GradientMagnitude.calculate (src/costs/image_variance.py:27-71,
 * src/costs/gradient_magnitude.py:60-76 + SobelTorch src/utils/stat_utils.py:50-83).
 *   value  : device double[4]: value[0] = the RAW contrast (no sign), value[1..3] = scratch
 *            accumulators (zeroed inside); ddof 1 = torch.var, 0 = np.var
 *   G      : optional [H,W] = gscale * d value / d img, gscale read from the device double
 *            `gscale` (NULL = 1) -- lets the caller chain the upstream gradient on the device */
int cmax_contrast(const void *img, int dtype, int H, int W, int cost, int omit_boundary, int ddof,
                  double *value, void *G, const double *gscale, cmax_stream_t stream);

/* TotalVariation.calculate_torch (src/costs/total_variation.py:60-75, 110-126): flow [2,h,w];
 * value is a device double[4] as for cmax_contrast.                                           */
int cmax_total_variation(const void *flow, int dtype, int h, int w, int omit_boundary,
                         double *value, void *G, const double *gscale, cmax_stream_t stream);

/* One propagation step (src/utils/flow_utils.py:567-639 / 439-493) on F[2,H,W] and its adjoint
 * (gF += J^T gout).                                                                          */
int cmax_flow_step(const void *F, int dtype, int H, int W, double dt, int scheme, void *out,
                   cmax_stream_t stream);
int cmax_flow_step_adj(const void *F, int dtype, int H, int W, double dt, int scheme,
                       const void *gout, void *gF, cmax_stream_t stream);

/* construct_dense_flow_voxel_torch (src/utils/flow_utils.py:99-161): V[T,2,H,W] from F at bin
 * t0 (0 = "first", T/2 = "middle"); and its adjoint (gV is clobbered: the sweep leaves dL/dF in
 * its bin t0; gF[2,H,W] receives a copy, NULL: no copy).                                        */
int cmax_voxel_construct(const void *F, int dtype, int T, int t0, int H, int W, int scheme, void *V,
                         cmax_stream_t stream);
int cmax_voxel_construct_adj(const void *V, int dtype, int T, int t0, int H, int W, int scheme,
                             void *gV, void *gF, cmax_stream_t stream);

/* Second order of the same chain, for exact Hessian-vector products of time-aware objectives (what
 * torch.autograd.functional.vhp differentiates, src/solver/scipy_autograd/torch_wrapper.py:51-73):
 *   _tan      V[T,2,H,W] and its directional derivative dV along dF, in one sweep
 *   _adj_tan  on entry gV = dL/dV, dgV = its tangent (both clobbered: the sweep leaves the results in their bins
 *             t0); gF[2,H,W] = copy of dL/dF, dgF[2,H,W] = copy of d/d(eps) [dL/dF](F + eps dF)
 *             = J^T dgV + (dJ[dV])^T gV   (either may be NULL: no copy)
 * sign(), the selectors of maximum / minimum and |.|' are piecewise constant (torch's convention).          */
int cmax_voxel_construct_tan(const void *F, const void *dF, int dtype, int T, int t0, int H, int W, int scheme,
                             void *V, void *dV, cmax_stream_t stream);
int cmax_voxel_construct_adj_tan(const void *V, const void *dV, int dtype, int T, int t0, int H, int W, int scheme,
                                 void *gV, void *dgV, void *gF, void *dgF, cmax_stream_t stream);

/* interpolate_dense_flow_from_patch_tensor (src/solver/patch_contrast_base.py:462-506): patch motion
 * [2,ph,pw] -> dense flow [2,H,W] = centre-crop(bilinear x(sw_h,sw_w), align_corners=False, of the
 * replicate-padded NEGATED grid).  adjoint != 0: `motion` is dL/dflow [2,H,W], out = dL/dmotion
 * [2,ph,pw] (overwritten).                                                                     */
int cmax_patch_to_dense(const void *motion, int dtype, int ph, int pw, int pad_h, int pad_w, int sw_h,
                        int sw_w, int H, int W, int adjoint, void *out, cmax_stream_t stream);

/* =============================================================================================
 * Fused objective (the hot path): events are packed + sorted once per batch, then every
 * evaluation runs  warp+vote -> contrast -> gather-gradient  without materialising warped events.
 * Replaces one pass of PatchContrastMaximization.get_arg_for_cost + cost.calculate +
 * torch.autograd.grad (src/solver/patch_contrast_base.py:273-352,
 * src/solver/scipy_autograd/torch_wrapper.py:30-49).  Arithmetic: fp32 per event, fp64 reductions.
 * ============================================================================================= */

/* H, W: un-padded sensor size; ph, pw: outer padding (image is [H+2ph, W+2pw]).               */
int cmax_create(int H, int W, int ph, int pw, cmax_handle_t *out);
int cmax_destroy(cmax_handle_t h);

/* Pack (12-bit row | 12-bit col | 8-bit time bin ; fp32 normalised time ; optional fractional
 * residuals) and sort the batch: source tile (16 x 16 pixels) major, inside a tile by pixel
 * (n_time_bin == 0) or by time bin.  events: [n,4] dtype; events whose source pixel is outside
 * the sensor (or NaN) are dropped.  If have_tminmax, (tmin, tmax) are the GLOBAL batch extremes
 * (multi-GPU time slices); otherwise they are reduced from this call's events (all of them) on the
 * device.  n_time_bin > 0 precomputes the voxel bin of every event with the reference's fp64 edge
 * arithmetic (src/warp.py:342-345).  Blocks once (work list sized on the host); 0.10 ms per 1M events.
 * Device memory per event: 64 B (packed events, fp64 times, the sort's staging copy) + 4.5 B for the compact copy the hot kernels read
 * when the work list is cut into big segments (un-binned batches of >= 8M events, or by the work-list rule from ~4M). */
int cmax_set_events(cmax_handle_t h, const void *events, int dtype, int64_t n, int have_tminmax,
                    double tmin, double tmax, int n_time_bin, cmax_stream_t stream);
int cmax_set_time_bins(cmax_handle_t h, int n_time_bin, cmax_stream_t stream);
/* LARGE MOTIONS of a 2-DoF / dense objective (round 4): re-order the batch into n_slab time slabs, slab-major inside every source
 * tile row -- (tile row, slab, tile column) -- so that a segment's events span 1/n_slab of the batch's duration and its LDS window is
 * the tiles' extent plus 1/n_slab of the displacement range (a window that would overflow at 150 px over the batch fits again; see
 * DESIGN.md section 4 "large motions").  Results are those of the un-binned order (same events, same arithmetic; the sums are
 * associated differently).  Voxel objectives need cmax_set_time_bins order and refuse a slab handle (cmax_patch_search walks either
 * order since round 5); n_slab <= 1 returns to the un-binned order; the next cmax_set_events starts un-slabbed again.  Blocks once
 * like cmax_set_time_bins.                                                                                                        */
int cmax_set_time_slabs(cmax_handle_t h, int n_slab, cmax_stream_t stream);

/* Image of warped events for one reference time (fp32 [Hp,Wp], blurred if sigma > 0).
 * motion: fp32 theta[2] | flow[2,H,W] | voxel[T,2,H,W] in pixel per (normalised) time.
 * model < 0 builds the un-warped image ("orig_iwe", patch_contrast_base.py:295-301).          */
int cmax_iwe(cmax_handle_t h, int model, const float *motion, int T, int ref_mode, double ref_frac,
             int normalize_t, double sigma, float *iwe_out, cmax_stream_t stream);

/* Objective descriptor.  loss = sum_k mult_k * term(v_k) over the listed reference times, with
 *   normalized == 0 : term = sign * v_k                      (ImageVariance / GradientMagnitude,
 *                                                              sign = -1 for "minimize")
 *   normalized == 1 : term = v_orig / v_k  ("minimize")       (normalized_*.py, multi_focal_*.py)
 *                     or v_k / v_orig      (otherwise)
 * v = raw contrast (cost, omit_boundary) of the (blurred) IWE; v_orig of the un-warped IWE
 * (NOT boundary-cropped for the variance, normalized_image_variance.py:40-41).                */
typedef struct {
    int32_t model;        /* CMAX_MODEL_* */
    int32_t cost;         /* CMAX_COST_* */
    int32_t normalized;   /* 0 / 1 */
    int32_t minimize;     /* 1 = direction "minimize", 0 = "natural"/"maximize" value */
    int32_t negate;       /* 1 = return -loss (multi_focal_* with direction "maximize") */
    int32_t omit_boundary;
    int32_t normalize_t;
    int32_t n_ref;        /* 1..4 reference times */
    int32_t ref_mode[4];  /* CMAX_REF_* */
    double ref_frac[4];
    double mult[4];       /* e.g. forward 1, backward 1, middle 2 */
    double sigma;         /* blur (0 = none) */
    int32_t T;            /* voxel time bins */
    int32_t motion_dtype; /* CMAX_F32 (0): `motion` is fp32.  CMAX_F64: 2-DoF only -- `motion` points to double theta[2], the
                             optimiser's own fp64 parameters (src/solver/patch_contrast_pyramid.py:186): the bulk arithmetic stays
                             fp32, but the cell floor(x' + 1e-6) of an event that lies within fp32 rounding of a cell border is
                             decided in fp64 from this theta, like the reference does (src/event_image_converter.py:340) */
} cmax_objective_t;

/* One evaluation.  result[0] = loss, result[1..n_ref] = v_k, result[5] = v_orig (device
 * doubles, result has 8 entries).  grad: fp64 [2] for 2DOF, fp32 [2,H,W] / [T,2,H,W] otherwise
 * (overwritten); NULL skips the gradient pass.  motion: fp32 theta[2] | flow[2,H,W] | voxel[T,2,H,W]
 * (or double theta[2], see cmax_objective_t::motion_dtype).                                    */
int cmax_objective(cmax_handle_t h, const cmax_objective_t *desc_host, const void *motion,
                   double *result, void *grad, cmax_stream_t stream);

/* The same evaluation with its results delivered TO THE HOST -- what an optimiser written in C calls once per iteration
 * (the reference's TorchWrapper.get_value_and_grad ends in .cpu().numpy(), src/solver/scipy_autograd/torch_wrapper.py:46-49):
 * enqueues the evaluation and the copies on `stream` and returns when result_host[8] and grad_host (double[2] for 2DOF, else
 * fp32 [2,H,W] / [T,2,H,W]; NULL: value only) are filled.  Blocks (busy-waits).  For the 2-DoF image-variance objective
 * the one-wave finishing kernel writes loss and gradient straight into pinned host memory and a run counter behind them,
 * which this call polls: no copy engine and no driver call between the last kernel and the caller (23.7 us per sequential
 * evaluation of 1M events, profiles/r03_ablation.txt 10).  Other objectives: cmax_objective + copies through pinned staging. */
int cmax_objective_host(cmax_handle_t h, const cmax_objective_t *desc_host, const void *motion, double *result_host,
                        void *grad_host, cmax_stream_t stream);

/* K candidate motions in one call (the reference's gradient-free paths evaluate the objective for batches of sampled motions:
 * src/solver/base.py:738-758 run_optuna, src/solver/patch_contrast_pyramid.py:363-415; a line search asks for several steps along one
 * direction).  motions: K consecutive motions as cmax_objective takes one (fp32 theta[2] | double theta[2] with motion_dtype CMAX_F64 |
 * flow[2,H,W] | voxel[T,2,H,W]); results: device double[K][8]; grads: device double[K][2] (2DOF) or fp32 [K][...] (NULL: values only).
 * For the 2-DoF image-variance objective (sigma 0, not normalised, default mode, no communicator) the K evaluations share ONE
 * launch of each kernel: blockIdx.z is the candidate -- its own vote image, sums and windows -- so the launch structure that is 40 %
 * of a single 1M-event evaluation is paid once per batch.  Every other objective is evaluated candidate by candidate inside the call
 * (same results).  1 <= K <= 64; the handle keeps 2 K vote images for it.                                                       */
int cmax_objective_batch(cmax_handle_t h, const cmax_objective_t *desc_host, const void *motions, int K, double *results,
                         void *grads, cmax_stream_t stream);

/* Raw form of the 2-DoF objectives (default mode, non-empty batch; cmax_objective_has_raw says whether a descriptor on the
 * current batch qualifies -- everything 2-DoF except a NORMALISED plain variance, whose fold needs device-side statistics).
 * cmax_objective ends such an evaluation with a one-wave kernel that only adds up what the gathering kernel left and divides a
 * few numbers -- a third of the headline evaluation, as a launch; here that kernel is not launched.  `raw` (device,
 * CMAX_RAW_DOUBLES doubles per reference time, cleared by the library inside the evaluation) receives CMAX_RAW_LINES partial
 * sums, one 128-byte line each, added with fp64 atomics by the gathering workgroups:
 *   image variance, sigma 0 ("deferred statistics": no image kernel runs at all): doubles 0..5 of a line =
 *     (S1x, S1y, S2x, S2y, sum I, sum I^2), S1 = sum_e dt * bilinear-difference(1_Omega IWE), S2 = the same of 1_Omega;
 *   every other objective: doubles 0, 1 = sum_e dt * dL/d(x', y') (chain factors applied), and doubles 8..15 of the FIRST
 *     line = result[8] as cmax_objective would deliver it (written by the gathering kernel's first workgroup).
 * The CONSUMER finishes: copy the buffer to the host whenever it needs the numbers and call cmax_finalize_raw_host (pure host
 * arithmetic: sums the lines; deferred form: loss = -/+ var, dL/dtheta = c (S1 - mu S2)).  Asynchronous like cmax_objective. */
#define CMAX_RAW_LINES 32
#define CMAX_RAW_DOUBLES (CMAX_RAW_LINES * 16)
int cmax_objective_has_raw(cmax_handle_t h, const cmax_objective_t *desc_host);
int cmax_objective_raw(cmax_handle_t h, const cmax_objective_t *desc_host, const void *motion, double *raw,
                       cmax_stream_t stream);
/* raw_host: host copy of `raw` ([n_ref][CMAX_RAW_DOUBLES]).  result_host[8] as for cmax_objective; grad_host double[2]
 * (NULL: value only).                                                                            */
int cmax_finalize_raw_host(cmax_handle_t h, const cmax_objective_t *desc_host, const double *raw_host,
                           double *result_host, double *grad_host);

/* Exact Hessian-vector product of the objective w.r.t. the motion, H u -- what
 * torch.autograd.functional.vhp returns in the reference (src/solver/scipy_autograd/torch_wrapper.py:
 * 51-73; the Hessian is symmetric): derivative of the analytic gradient along `tangent` with the
 * bilinear cells held fixed.  tangent: fp32, same layout as motion, SCALED TO UNIT MAX-NORM by the
 * caller (H is linear in u; the derivative votes are accumulated in fixed point).  hv: fp64 [2]
 * (2DOF) or fp32 [2,H,W] / [T,2,H,W], overwritten.  Costs as in cmax_objective.                 */
int cmax_objective_hvp(cmax_handle_t h, const cmax_objective_t *desc_host, const void *motion,
                       const float *tangent, void *hv, cmax_stream_t stream);

/* Phase-split form for time-sliced multi-GPU runs (one handle per GPU, each holding a contiguous
 * time slice of the batch and the GLOBAL tmin/tmax):
 *   cmax_objective_vote    raw votes of this slice: images[k] for k < n_ref, plus images[n_ref] =
 *                          the un-warped image when a normalised cost needs it and it is not cached;
 *                          *n_images_host = number of images written (n_ref or n_ref + 1)
 *   -- caller all-reduces (sum) images[0 .. n_images) across GPUs (RCCL) --
 *   cmax_objective_finish  blur, contrast statistics, loss (identical on every GPU), then the
 *                          gradient contribution of THIS slice's events
 *   -- caller all-reduces (sum) grad --
 * images: fp32 [5, Hp, Wp] caller-owned.  cmax_objective == vote + finish on internal images.
 * `motion` of the finish call must be the buffer AND the values of the preceding vote call: the gather re-uses
 * the LDS windows the vote derived for every segment.                                          */
int cmax_objective_vote(cmax_handle_t h, const cmax_objective_t *desc_host, const void *motion,
                        float *images, int *n_images_host, cmax_stream_t stream);
int cmax_objective_finish(cmax_handle_t h, const cmax_objective_t *desc_host, const void *motion,
                          const float *images, int n_images, double *result, void *grad,
                          cmax_stream_t stream);

/* =============================================================================================
 * Time-sliced multi-GPU evaluation (the reference has no distributed code).
 * One process per GPU; every rank owns a handle holding a contiguous TIME SLICE of the batch
 * (cmax_set_events with have_tminmax = 1 and the batch-wide extremes) and one RCCL communicator.
 * The collectives are enqueued by the library on the caller's stream, between its own kernels:
 *   K1 votes of this slice -> all-reduce(sum) of [n_images, Hp, Wp] fp32 (all reference times and the
 *   un-warped image in ONE call) -> contrast statistics + loss (redundantly, identical on every rank)
 *   + K3 gather of this slice's events -> all-reduce(sum) of the gradient (fp64 [2] | fp32 [2,H,W] |
 *   fp32 [T,2,H,W]).  No other exchange; no host synchronisation.
 *   2-DoF with the plain variance cost (BASELINE's headline) needs ONE exchange: K1 votes the image and its two
 *   tangent images dI/dtheta (two more votes per event), one all-reduce carries the three of them, and every rank
 *   finishes loss and gradient <dL/dI, dI/dtheta> in image space -- no pass over the events for the gradient and no
 *   16-byte all-reduce whose cost is pure latency.
 * RCCL is bound at run time (the librccl already in the process, i.e. torch's, else ROCm's); a handle
 * without a communicator -- or with nranks == 1 -- never touches it.
 * ============================================================================================= */
/* 0 if RCCL can be bound in this process (and binds it), CMAX_ENODEV otherwise -- a LOCAL call, no rank waits for another:
 * the ranks agree on it (by whatever transport carries the rendezvous id) BEFORE any of them enters cmax_comm_init, which
 * blocks in ncclCommInitRank until every rank has called it.  path_host (optional, path_capacity bytes): the shared
 * object the entry points were resolved from -- torch's bundled librccl when the process already holds one.       */
int cmax_comm_available(char *path_host, int path_capacity);
#define CMAX_COMM_ID_BYTES 128
/* rank 0: a fresh rendezvous id (ncclGetUniqueId) into id_host[128]; ship it to the other ranks
 * by any means (torch.distributed broadcast, a file, MPI).                                      */
int cmax_comm_unique_id(void *id_host);
/* Collective over the ranks (blocks until all nranks processes have called): binds the handle's
 * device (the current HIP device must be the one the handle was created on) to rank `rank`.
 * nranks == 1 with id_host == NULL: no communicator, RCCL is never loaded; with an id a real 1-rank
 * communicator is made (cmax_objective_dist then runs the N > 1 enqueue sequence, RCCL included). */
int cmax_comm_init(cmax_handle_t h, const void *id_host, int nranks, int rank);
int cmax_comm_destroy(cmax_handle_t h);
/* nranks / rank of the handle's communicator (1 / 0 without one); rccl_version: ncclGetVersion, 0
 * if RCCL was never loaded.                                                                      */
int cmax_comm_info(cmax_handle_t h, int *nranks, int *rank, int *rccl_version);
/* In-place all-reduce of a device buffer on the handle's communicator (dtype CMAX_F32 / CMAX_F64;
 * op 0 sum, 1 min, 2 max) -- e.g. (t_min, -t_max) with op min to agree on the batch extremes.   */
int cmax_comm_allreduce(cmax_handle_t h, void *buf, int64_t count, int dtype, int op, cmax_stream_t stream);
/* Overlap of the gradient exchange with the gather (dense objectives).  With bands > 1 cmax_objective_dist exchanges the gradient
 * as `bands` grouped all-reduces of source-tile-row bands on a second stream of the handle (events order the two streams; the
 * caller's stream continues after the last band is reduced).  EVERY RANK MUST SET THE SAME VALUE: the bands follow from the
 * sensor's tile rows and this setting alone, so all ranks issue the same collectives whatever their own slice looks like; a
 * rank whose work list is group-aligned (cmax_batch_info's owned_groups) launches its gathering kernel band by band, so that a
 * band's rows are exchanged while the next band is gathered -- the others (no owned groups, deterministic mode, no events)
 * finish their gradient first and issue the same exchanges behind it.  bands = 1 (default): one all-reduce.  Same results. */
int cmax_comm_set_c2_bands(cmax_handle_t h, int bands);
/* One evaluation of the whole (time-sliced) batch: same arguments and results as cmax_objective,
 * the same on every rank BIT FOR BIT -- the gradient because it is all-reduced, result[8] because
 * it leaves as rank 0's (every other rank zeroes its eight doubles and they ride in the gradient's
 * all-reduce as one grouped call; a value-only evaluation exchanges the 64 bytes alone): replicated
 * optimisers (src/solver/scipy_autograd/scipy_minimize.py:100-117 on every rank) take identical
 * decisions.  A rank may hold zero events.  Without a communicator: == cmax_objective.          */
int cmax_objective_dist(cmax_handle_t h, const cmax_objective_t *desc_host, const void *motion,
                        double *result, void *grad, cmax_stream_t stream);

/* The exact Hessian-vector product of the whole (time-sliced) batch: cmax_objective_hvp with one grouped all-reduce of the images
 * and the tangent images (both are sums over events) and one of the product.  Without a communicator: == cmax_objective_hvp.   */
int cmax_objective_hvp_dist(cmax_handle_t h, const cmax_objective_t *desc_host, const void *motion,
                            const float *tangent, void *hv, cmax_stream_t stream);

/* Deterministic mode ("race detection"): bit-identical IWE, loss and gradient from run
 * to run for cmax_iwe / cmax_objective / cmax_objective_vote + _finish / cmax_objective_hvp.  By default the vote flush, the flow
 * gradient and the contrast statistics use floating-point atomics, whose order -- like the order of events inside
 * a sorted group -- varies between runs, so results differ in their last bits (~1e-7 relative).  With enable != 0
 * everything that depends on such an order is accumulated in INTEGERS: votes as 2^-20 fixed point in a 64-bit image
 * (rounded to fp32 once), per-event gradient terms as 64-bit fixed point (scale from max |image|) added per thread,
 * per workgroup and with 64-bit global atomics, statistics with one workgroup per accumulator and the unfused image
 * kernels.  Same arithmetic per event, same parity (1e-4 of the reference); slower: no fused image kernels, one pair
 * of global atomics per event for the flow gradient (dense 5M events: ~10x an evaluation), 40 B per pixel more HBM.
 * cmax_objective_hvp is covered as well (round 3: integer tangent-vote images, integer accumulation of the second-order gather).
 * A patch plan on a deterministic handle (cmax_patch_plan_evaluate / _hvp) is covered too: the interpolation and its adjoint are
 * gathers, the adjoint sweeps of the voxel chain run order-free step kernels (every destination pixel evaluates the scatter of
 * its five sources itself instead of LDS atomics), the tail is one workgroup.  The stand-alone leaf entries
 * cmax_flow_step_adj / cmax_voxel_construct_adj[_tan] have no handle to read the mode from: cmax_set_leaf_deterministic (round 5, a
 * process-wide switch) makes them take the same order-free step kernels.  Across ranks (cmax_objective_dist) the result is as
 * repeatable as RCCL's reduction order.                                                                                          */
int cmax_set_deterministic(cmax_handle_t h, int enable);
int cmax_get_deterministic(cmax_handle_t h, int *enabled);
/* Process-wide: the adjoint leaf operators of the propagation step and of the voxel chain (cmax_flow_step_adj,
 * cmax_voxel_construct_adj, cmax_voxel_construct_adj_tan) accumulate without atomics, in a fixed order -- bit-identical results from
 * run to run (default 0: global / LDS atomics, last-bit differences).  Returns the previous setting.                              */
int cmax_set_leaf_deterministic(int enable);

/* Per-kernel-class timing for bench.py's roofline: when enabled every launch of the four hot kernels
 * is bracketed by HIP events ON THE LAUNCH STREAM.  enable = 1: one launch per bracket (results stay
 * valid; the bracket adds ~2.5 us of marker/dispatch latency to a ~8 us kernel).  enable = R in 2..64:
 * every hot launch is issued R times back to back inside its bracket, which amortises that latency;
 * TIMING ONLY -- votes and gradients are accumulated R times, so the evaluation's numbers are meaningless.  cmax_read_profile synchronises on them, returns
 * total milliseconds and launch counts for class 0 = K1 warp+vote, 1 = K2 contrast statistics,
 * 2 = K2b gradient image, 3 = K3 per-event gradient (host arrays of 4), and resets.           */
int cmax_set_profiling(cmax_handle_t h, int enable);
int cmax_read_profile(cmax_handle_t h, double *total_ms_host, int64_t *count_host);
/* The same with every class (host arrays of CMAX_PROF_CLASSES): 0..3 as above, 4 = finishing kernels
 * (k_finish*, k_finalize), 5 = RCCL collectives (never repeated inside a bracket).                 */
#define CMAX_PROF_CLASSES 6
int cmax_read_profile_all(cmax_handle_t h, double *total_ms_host, int64_t *count_host);

/* sizeof(cmax_objective_t) as compiled into the library (binding self-check).                 */
int cmax_sizeof_objective(void);

/* Copy the fp32 IWE [Hp,Wp] of reference time k of the last cmax_objective call (the image the
 * contrast was evaluated on, i.e. blurred when sigma > 0) into iwe_out, on `stream`.          */
int cmax_copy_iwe(cmax_handle_t h, int k, float *iwe_out, cmax_stream_t stream);

/* What cmax_set_events made of the last batch: events packed; events DROPPED because their source pixel lies
 * outside the sensor or is NaN (the fused path indexes the flow field and the source tiles with it; for 2-DoF
 * objectives cmax_set_keep_outside below keeps them, as the reference does -- the leaf operators cmax_warp_events +
 * cmax_vote have no such filter either way); whether
 * any source coordinate is fractional; whether the work list gives every group to one segment (owned groups:
 * single-reference dense / voxel gradients are then stored, not added).  Any pointer may be NULL.            */
int cmax_batch_info(cmax_handle_t h, int64_t *n_packed, int64_t *n_dropped, int *has_fractional, int *owned_groups);
/* Events OFF THE SENSOR (round 4).  The reference's 2-DoF warp has no bounds test on the source (src/warp.py:506-515): an event from
 * outside the sensor votes wherever it warps into the padded image.  With cmax_set_keep_outside(h, 1) the next cmax_set_events packs
 * such events (finite coordinates) instead of dropping them: at the NEAREST sensor pixel, the rest of the way in the fp32 residuals
 * that fractional source coordinates use -- the event kernels then see x = ix + rx exactly as for any fractional source (a residual of
 * r px carries ulp(r) / 2 of rounding: 4e-6 px at 100 px).  cmax_batch_outside reports how many the last batch held; they count as
 * packed, not as dropped.  Only the 2-DoF model (and the un-warped image) is defined for them: a dense / voxel objective, cmax_iwe of
 * those models and cmax_patch_search refuse a batch that holds any (CMAX_EINVAL) -- the reference indexes its flow with the source
 * pixel there (negative indices wrap around in torch; nothing a caller can mean).  Default since round 5: ON -- a handle follows the
 * function cited above; cmax_set_keep_outside(h, 0) asks for such events to be dropped (and counted: cmax_batch_info) while packing,
 * which is what a caller of a dense / voxel objective on such a batch has to say explicitly.                                     */
int cmax_set_keep_outside(cmax_handle_t h, int on);
int cmax_batch_outside(cmax_handle_t h, int64_t *n_outside);
/* The work list cmax_set_events / cmax_set_time_bins cut for the event kernels (one workgroup per segment): number of
 * segments; segment_events = the most events a segment holds (2040, or 4088 for BIG segments: batches of >= 8M events, and
 * smaller ones whose group sizes ask for it -- DESIGN.md section 2); small_accumulators = 1 when a binned list was cut at
 * three (tile, bin) groups per segment for the voxel gradient kernel with the small LDS accumulator array.  Any pointer may
 * be NULL.  (Introspection for tests and tuning: results do not depend on the cut.)                                       */
int cmax_work_list_info(cmax_handle_t h, int *n_segments, int *segment_events, int *small_accumulators);

/* Introspection for tests / bench: number of packed events, HBM bytes held by the handle.     */
int cmax_handle_info(cmax_handle_t h, int64_t *n_events, int64_t *workspace_bytes);
/* Measurement aid (bench.py's roofline.launch_floor_us): enqueues `pairs` times two dependent EMPTY kernels with the grids of
 * the warp + vote kernel and of the gather kernel of a single-reference objective on the current batch -- what the headline
 * evaluation's launch structure costs when its kernels do nothing.  The caller brackets the call with events on `stream`.   */
int cmax_debug_launch_floor(cmax_handle_t h, int pairs, cmax_stream_t stream);
/* Tuning aid (tools/timeline.py; libraries built with -DCMAX_TIMELINE only, CMAX_ESTATE otherwise): device buffer of 2 x 4096 x 8
 * uint64 into which thread 0 of the event kernels' workgroups stamps the wall clock at its phase boundaries (NULL: off).       */
int cmax_debug_timeline(void *device_buffer);
/* Test aid (tests/test_gpu_fused.py::test_packed_order): copies the packed, sorted events of the current batch -- n_events x 2 uint32:
 * [row | col << 12 | (time bin, or the time's residual beyond fp32) << 24,  bits of the fp32 normalised time] -- and the group starts
 * the work list was cut from ([n_groups + 1] int32; NULL skips them; *n_groups_host receives their number: source tiles, x time bins
 * / slabs on binned handles) into device buffers, on `stream`.  The layout is what DESIGN.md section 2 describes, not a stable ABI.  */
int cmax_debug_packed_events(cmax_handle_t h, void *events_out, int *group_start_out, int *n_groups_host, cmax_stream_t stream);

/* =============================================================================================
 * The optimiser's objective for patch-based flow in one call: what
 * scipy.optimize calls through TorchWrapper.get_value_and_grad / get_hvp
 * (src/solver/scipy_autograd/torch_wrapper.py:30-73) on
 * PyramidalPatchContrastMaximization.objective_scipy + motion_to_dense_flow
 * (src/solver/patch_contrast_pyramid.py:430-516):
 *   x [2,ph,pw] patch motion (host fp64, pixel per time unit)
 *     -> dense flow (cmax_patch_to_dense) * t_scale  [-> voxel (cmax_voxel_construct)]  -> fp32
 *     -> loss = sum_i weight_i * cmax_objective(term_i)  + tv_weight * total_variation(x)
 *   and the gradient back through the adjoints, returned to host fp64.  One stream
 *   synchronisation per call; intermediates live in the plan.
 * ============================================================================================= */
typedef struct {
    int32_t n_terms;          /* 1..4 fused contrast terms of a hybrid cost (src/costs/hybrid.py:48-57) */
    int32_t time_aware;       /* 0: terms use CMAX_MODEL_DENSE; 1: CMAX_MODEL_VOXEL on the propagated flow */
    int32_t T, scheme, t0;    /* time bins, CMAX_SCHEME_*, bin that holds the patch flow (0 "first", T/2 "middle") */
    int32_t H, W;             /* sensor size of the handle */
    int32_t ph, pw;           /* patch grid */
    int32_t sw_h, sw_w;       /* sliding window = up-sampling factor */
    int32_t pad_h, pad_w;     /* replicate padding of the grid (patch_contrast_base.py:470-479) */
    int32_t tv_omit_boundary;
    double t_scale;           /* pixel per time unit -> pixel per normalised batch period */
    double weight[4];
    double tv_weight;         /* 0 = no total_variation term; sign of the cost direction included */
    cmax_objective_t term[4];
} cmax_patch_objective_t;
typedef struct cmax_patch_plan_s *cmax_patch_plan_t;

int cmax_sizeof_patch_objective(void);
int cmax_patch_plan_create(cmax_handle_t h, const cmax_patch_objective_t *desc_host, cmax_patch_plan_t *out);
int cmax_patch_plan_destroy(cmax_patch_plan_t plan);
/* A plan outlives a batch: the next batch behind the same handle (cmax_set_events) usually has another duration.  */
int cmax_patch_plan_set_t_scale(cmax_patch_plan_t plan, double t_scale);
/* Introspection for tests.  Default: every evaluation is launched eagerly (*graph_replay_enabled = 0).  With
 * CMAX_PLAN_GRAPHS=1 in the environment at plan creation, evaluations after the first few are replayed from
 * captured hipGraphs (one per distinct launch sequence; 0 again if a capture failed).                        */
int cmax_patch_plan_info(cmax_patch_plan_t plan, int *n_graphs, int *graph_replay_enabled);
/* On a handle that holds a communicator (cmax_comm_init: a time slice of the batch per rank) both calls below evaluate the WHOLE
 * batch, the same numbers on every rank: the fused terms exchange their images like cmax_objective_dist (and the product its
 * tangent images), but the flow gradient is not exchanged -- each rank carries its share through the (linear) adjoints of the voxel
 * chain and of the patch interpolation, and the ranks all-reduce 2 ph pw numbers (4 KB for a 16 x 16 grid instead of 7.4 MB of flow
 * gradient at 720p).  Every rank must make the same calls in the same order.
 * x_host [2*ph*pw] -> *loss_host, grad_host [2*ph*pw] (NULL: value only).  with_tv = 0 leaves the
 * total_variation term out (the smooth part, differenced for time-aware Hessian-vector products).
 * Blocks until the result is on the host.                                                      */
int cmax_patch_plan_evaluate(cmax_patch_plan_t plan, const double *x_host, int with_tv, double *loss_host,
                             double *grad_host, cmax_stream_t stream);
/* Exact Hessian-vector product w.r.t. x: t^2 P^T H_flow P v (cmax_objective_hvp inside); time-aware plans add the
 * voxel chain, t P^T [J^T H_VV J + (dJ[.])^T g_V] t P v (cmax_voxel_construct_tan / _adj_tan).  The
 * total_variation term has zero Hessian almost everywhere.                                      */
int cmax_patch_plan_hvp(cmax_patch_plan_t plan, const double *x_host, const double *v_host, double *hv_host,
                        cmax_stream_t stream);

/* =============================================================================================
 * Per-patch translation search: the re-initialisation the pyramid solver runs at every scale above
 * the coarsest.  Replaces the trial loop of initialize_guess_from_optuna_sampling / objective_initial /
 * calculate_cost_for_small_patch (src/solver/patch_contrast_pyramid.py:320-414): per patch, the events
 * with x_min <= x < x_max, y_min <= y < y_max (utils.crop_event) are shifted to the patch origin, warped
 * by a candidate translation to the MIDDLE of the patch's own time span, voted (bilinear) into an
 * img_h x img_w image, blurred with scipy.ndimage.gaussian_filter(sigma) and scored with the numpy
 * branch of GradientMagnitude: mean of (Sobel/8)^2 over the whole image, reflect-101 border
 * (src/costs/gradient_magnitude.py:78-95).  One workgroup per (patch, candidate); every pair of a scale
 * in one launch.
 *   boxes     device int32 [n_patch][4] = x_min, x_max, y_min, y_max (rows first), sensor coordinates
 *   cand      device fp32  [n_patch][n_cand][2] translations, pixel per unit of the RAW timestamps
 *   gm_out    device fp32  [n_patch][n_cand + 1]: gradient magnitude per candidate; the last column is
 *             the un-warped patch, so the reference's NormalizedGradientMagnitude loss of candidate c is
 *             gm_out[p][n_cand] / gm_out[p][c]   (normalized_gradient_magnitude.py:81-94)
 *   count_out device int32 [n_patch] events inside the box (the reference keeps the incoming motion
 *             when a patch holds <= 10 events, patch_contrast_pyramid.py:336)
 * Needs events set on the handle; 2 * img_h * img_w floats must fit 64 KB of LDS.  Asynchronous.  */
int cmax_patch_search(cmax_handle_t h, int n_patch, const int *boxes, int img_h, int img_w, int n_cand,
                      const float *cand, double sigma, float *gm_out, int *count_out, cmax_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CMAX_HIP_H */
