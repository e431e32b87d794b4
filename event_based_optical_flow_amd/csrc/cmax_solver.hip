// The optimiser's objective for patch-based flow in ONE library call: x[2 * n_patch] (host, fp64) ->
// loss, gradient (host, fp64), every stage on the device, one stream synchronisation per call.
//
// Equivalent of PyramidalPatchContrastMaximization.objective_scipy + motion_to_dense_flow
// (src/solver/patch_contrast_pyramid.py:430-516) under TorchWrapper.get_value_and_grad / get_hvp
// (src/solver/scipy_autograd/torch_wrapper.py:30-73):
//   patch motion -> dense flow (cmax_patch_to_dense) -> x t_scale [-> Burgers / upwind voxel] -> fp32
//   -> sum_i w_i * fused contrast objective_i (cmax_objective)  [+ w_tv * total_variation(patch motion)]
// and back through the hand-written adjoints.  The stages are the library's own C entry points (the same
// kernels the autograd wrappers of functional.py chain one Python call at a time); what this file adds is
// the plan that owns the intermediate buffers and the pinned staging, so that an evaluation costs one
// ctypes call instead of ~40 Python-level operations, and replays its launch sequence from a captured hipGraph
// (measured on the cfg1-shaped objective: 0.40 ms -> 0.09 ms per value + gradient).
// Pre/post stages run in fp64 like the reference's solver (patch_contrast_pyramid.py:186); the event
// path is fp32 per event with fp64 reductions as everywhere in cmax_fused.hip.
#include <cstring>
#include <new>
#include <atomic>
#include <vector>

#include "cmax_common.h"
#include "cmax_image_kernels.h"
#include "cmax_patch_kernels.h"

namespace cmax {

// out = in * scale [* *scale_dev]: scalars that change from call to call (the tangent's norm) live in device memory so
// that a captured graph can be replayed with new values
template <typename TO, typename TI>
__global__ void __launch_bounds__(256)
k_convert_scale(const TI *__restrict__ in, int64_t n, double scale, const double *__restrict__ scale_dev, TO *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const double sc = scale_dev ? scale * scale_dev[0] : scale;
    if (i < n) out[i] = (TO)((double)in[i] * sc);
}

// acc (=, +=) w [* *w_dev] * g
__global__ void __launch_bounds__(256)
k_accumulate(const float *__restrict__ g, int64_t n, double w, int first, double *__restrict__ acc, const double *__restrict__ w_dev = nullptr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w_dev) w *= w_dev[0];
    if (i < n) acc[i] = (first ? 0.0 : acc[i]) + w * (double)g[i];
}

// max |x| as the bit pattern of a non-negative double (order-preserving under unsigned compare); *bits zeroed by the caller
__global__ void __launch_bounds__(256) k_absmax(const double *__restrict__ x, int64_t n, unsigned long long *__restrict__ bits) {
    __shared__ double s_m[256 / kWave];
    double m = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {  // four independent loads in flight per thread
        const double a = x[i], b = x[i + stride], c = x[i + 2 * stride], d = x[i + 3 * stride];
        m = fmax(fmax(m, fmax(fabs(a), fabs(b))), fmax(fabs(c), fabs(d)));
    }
    for (; i < n; i += stride) m = fmax(m, fabs(x[i]));
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, kWave));
    if ((threadIdx.x & (kWave - 1)) == 0) s_m[threadIdx.x / kWave] = m;
    __syncthreads();
    // one atomic per workgroup (and few workgroups): equal-address atomics serialise at ~12 ns each
    if (threadIdx.x == 0) atomicMax(bits, (unsigned long long)__double_as_longlong(fmax(fmax(s_m[0], s_m[1]), fmax(s_m[2], s_m[3]))));
}

// scal[0] = max (1 if the field is all zero), scal[1] = 1 / scal[0]
__global__ void k_absmax_finish(const unsigned long long *__restrict__ bits, double *__restrict__ scal) {
    const double m = __longlong_as_double((long long)bits[0]);
    scal[0] = m > 0.0 ? m : 1.0;
    scal[1] = 1.0 / scal[0];
}

struct FinalParams {
    int n_terms, with_tv, nx;  // nx = 0: value only
    int flag_slot;             // index (in doubles) of the run counter in the output: 1 + the plan's nx, whatever this call's nx is
    int ph, pw, tv_crop;       // patch grid of the total-variation term
    double weight[4], tv_weight, gscale;
};

// Tail of an evaluation in ONE workgroup (every dependent launch costs ~4.5 us on this chip and the operands are a
// few hundred numbers): total variation of the patch motion x [2,ph,pw] -- mean |Sobel/8| over 4 channels,
// TotalVariation.calculate_torch, src/costs/total_variation.py:60-75, 110-126 -- and its sub-gradient, the weighted
// sum of the contrast terms, and the chain through t_scale:
//   out[0] = sum_i w_i result_i + w_tv TV(x),   out[1 + j] = gscale * gx[j] + w_tv dTV/dx[j]
// gx comes as fp64 (gx64) or fp32 (gx32); gscale_dev multiplies gscale (Hessian-vector products).  `out` may be
// pinned host memory.
constexpr int kTailLds = 4096;  // patch-grid values staged in LDS by k_patch_tail (2 x 45 x 45 patches)
__global__ void __launch_bounds__(256)
k_patch_tail(FinalParams fp, const double *__restrict__ results, const double *__restrict__ x, const double *__restrict__ gx64,
             const float *__restrict__ gx32, const double *__restrict__ gscale_dev, double *__restrict__ out,
             unsigned long long *__restrict__ seq_dev) {
    __shared__ double smem[4];
    __shared__ double s_tv;
    __shared__ double s_x[kTailLds];
    __shared__ signed char s_sx[kTailLds], s_sy[kTailLds];  // sign of the Sobel responses on Omega, 0 elsewhere
    const int h = fp.ph, w = fp.pw, hw = h * w;
    const int i0 = fp.tv_crop ? 1 : 0, hh = h - 2 * i0, ww = w - 2 * i0;
    const double ntv = 4.0 * (double)hh * (double)ww;
    const bool lds = 2 * hw <= kTailLds;
    const double *xs = x;
    if (fp.with_tv) {
        if (lds) {
            for (int p = threadIdx.x; p < 2 * hw; p += blockDim.x) s_x[p] = x[p];
            __syncthreads();
            xs = s_x;
        }
        double v[1] = {0.0};
        for (int p = threadIdx.x; p < 2 * hw; p += blockDim.x) {
            const int c = p / hw, r = p - c * hw, i = r / w, j = r - i * w;
            signed char gx = 0, gy = 0;
            if (i >= i0 && i < h - i0 && j >= i0 && j < w - i0) {
                double sx, sy;
                sobel8<double>(xs + c * hw, h, w, i, j, sx, sy);
                v[0] += fabs(sx) + fabs(sy);
                gx = (signed char)((sx > 0) - (sx < 0));
                gy = (signed char)((sy > 0) - (sy < 0));
            }
            if (lds) {
                s_sx[p] = gx;
                s_sy[p] = gy;
            }
        }
        block_sum<1>(v, smem);  // its barriers also publish s_sx / s_sy
        if (threadIdx.x == 0) s_tv = v[0] / ntv;
        __syncthreads();
    }
    double gscale = fp.gscale;
    if (gscale_dev) gscale *= gscale_dev[0];
    if (threadIdx.x == 0) {
        double loss = 0.0;
        for (int i = 0; i < fp.n_terms; ++i) loss += fp.weight[i] * results[8 * i];
        if (fp.with_tv) loss += fp.tv_weight * s_tv;
        out[0] = loss;
    }
    for (int p = threadIdx.x; p < fp.nx; p += blockDim.x) {
        double g = gscale * (gx64 ? gx64[p] : (double)gx32[p]);
        if (fp.with_tv) {
            const int c = p / hw, r = p - c * hw, i = r / w, j = r - i * w;
            double sgn = 0.0;
            for (int a = -1; a <= 1; ++a)
                for (int b = -1; b <= 1; ++b) {
                    const int qi = i - a, qj = j - b;  // output pixel that reads (i, j) with tap (a, b)
                    if (qi < i0 || qi >= h - i0 || qj < i0 || qj >= w - i0) continue;
                    double gsx, gsy;
                    if (lds) {
                        gsx = (double)s_sx[c * hw + qi * w + qj];
                        gsy = (double)s_sy[c * hw + qi * w + qj];
                    } else {
                        double sx, sy;
                        sobel8<double>(x + c * hw, h, w, qi, qj, sx, sy);
                        gsx = (double)((sx > 0) - (sx < 0));
                        gsy = (double)((sy > 0) - (sy < 0));
                    }
                    // SX[a+1][b+1] = a * (2 - |b|), SY[a+1][b+1] = b * (2 - |a|)
                    sgn += gsx * (double)(a * (2 - (b < 0 ? -b : b))) + gsy * (double)(b * (2 - (a < 0 ? -a : a)));
                }
            g += fp.tv_weight * sgn / 8.0 / ntv;
        }
        out[1 + p] = g;
    }
    // completion flag for a host that polls the (pinned) output instead of sleeping in hipStreamSynchronize: a run
    // counter, written after every result of this launch is visible system-wide
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long v = seq_dev[0] + 1ull;
        seq_dev[0] = v;
        reinterpret_cast<volatile unsigned long long *>(out + fp.flag_slot)[0] = v;
    }
}

}  // namespace cmax

struct cmax_patch_plan_s {
    cmax_handle_t handle = nullptr;
    cmax_patch_objective_t d;
    int nx = 0;            // 2 * ph * pw
    int64_t nflow = 0;     // 2 * H * W
    int64_t nmotion = 0;   // nflow or T * nflow
    double *x64 = nullptr, *v64 = nullptr, *flow64 = nullptr, *vox64 = nullptr, *gacc64 = nullptr, *gflow64 = nullptr;
    double *gx64 = nullptr, *results = nullptr;
    // time-aware Hessian-vector product (allocated on first use): tangent flow / voxel, tangent of dL/dvoxel, scalars
    double *dflow64 = nullptr, *dvox64 = nullptr, *dgacc64 = nullptr, *dgflow64 = nullptr, *scal = nullptr;
    float *motion32 = nullptr, *grad32 = nullptr, *tan32 = nullptr, *gx32 = nullptr;
    double *h_out_dev = nullptr;  // device view of the pinned output: the tail kernel writes the result there
    double *h_in = nullptr, *h_out = nullptr;  // pinned staging: x | v | 2 scalars, loss | grad | run counter
    unsigned long long *seq_dev = nullptr;     // device-side run counter of the tail kernel
    unsigned long long seq_seen = 0;           // its value after the last completed call
    // Captured launch sequences (hipGraph), replayed on the plan's own stream: an evaluation is ~25 small launches,
    // host-bound when issued one by one.  Keyed by everything that changes the sequence: the kind of call and the
    // handle's host-side state (which vote buffer is current, which images are already zero, whether the
    // un-warped image is cached, the generation of the packed events).
    struct GraphEntry {
        uint64_t key;
        uint64_t generation;
        hipGraphExec_t exec;
        cmax::HandleEvalState post;  // handle state after the sequence
    };
    std::vector<GraphEntry> graphs;
    hipStream_t own_stream = nullptr;
    hipEvent_t ev_caller = nullptr;
    bool graphs_ok = false;  // hipGraph replay: opt-in (CMAX_PLAN_GRAPHS=1), see run_sequence
    int eager_calls = 0;  // the first calls run eagerly (lazy allocations, caches)
};

using namespace cmax;

namespace {

template <typename T>
int plan_alloc(T **p, int64_t count) {
    hipError_t e = hipMalloc((void **)p, (size_t)(count > 0 ? count : 1) * sizeof(T));
    if (e != hipSuccess) {
        set_error("patch plan: hipMalloc(%lld bytes) failed: %s", (long long)(count * sizeof(T)), hipGetErrorString(e));
        return CMAX_ENOMEM;
    }
    return 0;
}

// Time-sliced batches (the handle holds a communicator, cmax_comm_init): the solver's objective is evaluated the way
// cmax_objective_dist evaluates a fused term -- K1 -> all-reduce(images) -> image kernels -> K3 on this rank's events -- but the
// flow gradient is NOT exchanged (2 H W [T] floats: 7.4 MB at 720p).  Every step from there to the optimiser's gradient is linear
// (weighted sum over the terms, adjoint sweep of the voxel chain, adjoint of the patch interpolation), so each rank carries its
// share through them and the ranks all-reduce 2 n_patch numbers (4 KB for a 16 x 16 grid) in front of the tail kernel.  Loss,
// total variation and everything derived from x are computed redundantly and identically on every rank.
int plan_objective(cmax_patch_plan_s *p, const cmax_objective_t *term, double *result, void *grad, hipStream_t s) {
    if (handle_has_comm(p->handle)) return objective_dist_local_grad(p->handle, term, p->motion32, result, grad, s);
    return cmax_objective(p->handle, term, p->motion32, result, grad, s);
}
int plan_objective_hvp(cmax_patch_plan_s *p, const cmax_objective_t *term, void *hv, hipStream_t s) {
    if (handle_has_comm(p->handle)) return objective_hvp_dist_local(p->handle, term, p->motion32, p->tan32, hv, s);
    return cmax_objective_hvp(p->handle, term, p->motion32, p->tan32, hv, s);
}
// C2 of the plan: the gradient (or product) w.r.t. the patch motion, summed over the ranks (a no-op without a communicator)
int plan_reduce_gx(cmax_patch_plan_s *p, const double *gx64, const float *gx32, hipStream_t s) {
    if (!handle_has_comm(p->handle)) return 0;
    if (gx64) return handle_allreduce_sum(p->handle, const_cast<double *>(gx64), (size_t)p->nx, true, s);
    return handle_allreduce_sum(p->handle, const_cast<float *>(gx32), (size_t)p->nx, false, s);
}
// ... of an EVALUATION: the terms' result[8] ride along and leave as rank 0's on every rank (each rank summed its statistics in its own
// order; replicated optimisers must see the same loss bit for bit).  Value-only evaluations exchange the scalars alone.
int plan_reduce_gx_and_results(cmax_patch_plan_s *p, const double *gx64, const float *gx32, hipStream_t s) {
    if (!handle_has_comm(p->handle)) return 0;
    const int n_scalars = 8 * p->d.n_terms;
    if (gx64) return handle_allreduce_sum_with_scalars(p->handle, const_cast<double *>(gx64), (size_t)p->nx, true, p->results, n_scalars, s);
    return handle_allreduce_sum_with_scalars(p->handle, const_cast<float *>(gx32), gx32 ? (size_t)p->nx : 0, false, p->results, n_scalars, s);
}

// x (host) -> fp32 motion of the fused objective: flow [2,H,W] or voxel [T,2,H,W] in pixel per normalised time.
// Leaves the fp64 flow (scaled) in flow64 and, when time-aware, the fp64 voxel in vox64.
int forward_motion(cmax_patch_plan_s *p, const double *src64, double scale, const double *scale_dev, float *dst32, hipStream_t s) {
    const cmax_patch_objective_t &d = p->d;
    const int grid = div_up(p->nflow, 256);
    if (!d.time_aware && !scale_dev) {  // patch grid -> fp32 flow * scale in one kernel
        hipLaunchKernelGGL((k_patch_to_dense<double, float>), dim3(grid), dim3(256), 0, s, src64, d.ph, d.pw, d.pad_h, d.pad_w, d.sw_h, d.sw_w,
                           d.H, d.W, dst32, scale);
        CMAX_CHECK_LAUNCH();
        return 0;
    }
    if (d.time_aware && !scale_dev) {
        // patch grid -> fp64 displacement field (x t_scale) in one kernel; the voxel is built on the displacement
        // field (patch_contrast_pyramid.py:452, 499-515) and leaves the chain kernel in fp64 and fp32 at once
        hipLaunchKernelGGL((k_patch_to_dense<double, double>), dim3(grid), dim3(256), 0, s, src64, d.ph, d.pw, d.pad_h, d.pad_w, d.sw_h, d.sw_w,
                           d.H, d.W, p->flow64, scale);
        CMAX_CHECK_LAUNCH();
        bool wrote32 = false;
        int rc = voxel_construct_f64_f32(p->flow64, d.T, d.t0, d.H, d.W, d.scheme, p->vox64, dst32, &wrote32, s);
        if (rc) return rc;
        if (!wrote32) {
            hipLaunchKernelGGL((k_convert_scale<float, double>), dim3(div_up(p->nmotion, 256)), dim3(256), 0, s, p->vox64, p->nmotion, 1.0, (const double *)nullptr, dst32);
            CMAX_CHECK_LAUNCH();
        }
        return 0;
    }
    int rc = cmax_patch_to_dense(src64, CMAX_F64, d.ph, d.pw, d.pad_h, d.pad_w, d.sw_h, d.sw_w, d.H, d.W, 0, p->flow64, s);
    if (rc) return rc;
    if (!d.time_aware) {
        hipLaunchKernelGGL((k_convert_scale<float, double>), dim3(grid), dim3(256), 0, s, p->flow64, p->nflow, scale, scale_dev, dst32);
        CMAX_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL((k_convert_scale<double, double>), dim3(grid), dim3(256), 0, s, p->flow64, p->nflow, scale, scale_dev, p->flow64);
    CMAX_CHECK_LAUNCH();
    rc = cmax_voxel_construct(p->flow64, CMAX_F64, d.T, d.t0, d.H, d.W, d.scheme, p->vox64, s);
    if (rc) return rc;
    hipLaunchKernelGGL((k_convert_scale<float, double>), dim3(div_up(p->nmotion, 256)), dim3(256), 0, s, p->vox64, p->nmotion, 1.0, (const double *)nullptr, dst32);
    CMAX_CHECK_LAUNCH();
    return 0;
}

// gradient of the fused terms w.r.t. the patch motion (without t_scale): gx64 or gx32 (returned through the pointers)
int backward_motion(cmax_patch_plan_s *p, const double **gx64, const float **gx32, double *weight_scale, hipStream_t s) {
    const cmax_patch_objective_t &d = p->d;
    *gx64 = nullptr;
    *gx32 = nullptr;
    *weight_scale = 1.0;
    if (d.n_terms == 1 && !d.time_aware) {
        // one term on the dense flow: the adjoint of the interpolation reads the fp32 flow gradient as it is
        // (fp64 accumulation inside), the term's weight is applied by the tail
        int rc = cmax_patch_to_dense(p->grad32, CMAX_F32, d.ph, d.pw, d.pad_h, d.pad_w, d.sw_h, d.sw_w, d.H, d.W, 1, p->gx32, s);
        if (rc) return rc;
        *gx32 = p->gx32;
        *weight_scale = d.weight[0];
        return 0;
    }
    const double *gflow = p->gacc64;
    if (d.time_aware) {  // the sweep leaves dL/dF in bin t0 of the gradient voxel: read it there
        int rc = voxel_construct_adj_f64(p->vox64, d.T, d.t0, d.H, d.W, d.scheme, p->gacc64, s, handle_is_deterministic(p->handle));
        if (rc) return rc;
        gflow = p->gacc64 + (int64_t)d.t0 * p->nflow;
    }
    int rc = cmax_patch_to_dense(gflow, CMAX_F64, d.ph, d.pw, d.pad_h, d.pad_w, d.sw_h, d.sw_w, d.H, d.W, 1, p->gx64, s);
    if (rc) return rc;
    *gx64 = p->gx64;
    return 0;
}

// Everything between the pinned input and the pinned output of one evaluation, enqueued on `s` (no synchronisation).
int enqueue_evaluate(cmax_patch_plan_s *p, bool tv, bool want_grad, hipStream_t s) {
    const cmax_patch_objective_t &d = p->d;
    CMAX_CHECK_HIP(hipMemcpyAsync(p->x64, p->h_in, (size_t)p->nx * sizeof(double), hipMemcpyHostToDevice, s));
    int rc = forward_motion(p, p->x64, d.t_scale, nullptr, p->motion32, s);
    if (rc) return rc;
    const bool accumulate = want_grad && !(d.n_terms == 1 && !d.time_aware);
    for (int i = 0; i < d.n_terms; ++i) {
        rc = plan_objective(p, &d.term[i], p->results + 8 * i, want_grad ? p->grad32 : nullptr, s);
        if (rc) return rc;
        if (accumulate) {
            hipLaunchKernelGGL(k_accumulate, dim3(div_up(p->nmotion, 256)), dim3(256), 0, s, p->grad32, p->nmotion, d.weight[i], i == 0 ? 1 : 0, p->gacc64);
            CMAX_CHECK_LAUNCH();
        }
    }
    const double *gx64 = nullptr;
    const float *gx32 = nullptr;
    double wscale = 1.0;
    if (want_grad) rc = backward_motion(p, &gx64, &gx32, &wscale, s);
    if (!rc && d.n_terms > 0) rc = plan_reduce_gx_and_results(p, gx64, gx32, s);
    else if (!rc && want_grad) rc = plan_reduce_gx(p, gx64, gx32, s);
    if (rc) return rc;
    FinalParams fp;
    fp.n_terms = d.n_terms;
    fp.flag_slot = 1 + (int)p->nx;
    fp.with_tv = tv ? 1 : 0;
    fp.nx = want_grad ? p->nx : 0;
    fp.ph = d.ph;
    fp.pw = d.pw;
    fp.tv_crop = d.tv_omit_boundary && d.ph > 2 && d.pw > 2;  // total_variation.py:123-125
    for (int i = 0; i < 4; ++i) fp.weight[i] = d.weight[i];
    fp.tv_weight = d.tv_weight;
    fp.gscale = d.t_scale * wscale;  // d(flow * t_scale) / d flow
    hipLaunchKernelGGL(k_patch_tail, dim3(1), dim3(256), 0, s, fp, p->results, p->x64, gx64, gx32, (const double *)nullptr, p->h_out_dev, p->seq_dev);
    CMAX_CHECK_LAUNCH();
    return 0;
}

// Hessian-vector product: h_in = x | v | 1/|v|_inf | |v|_inf
// Time-aware objective: x -> F = t P x -> voxel V(F) -> L(V).  With g_V = dL/dV, H_VV the Hessian of the fused
// terms (cmax_objective_hvp) and J = dV/dF:
//   H_x v = t P^T [ J^T H_VV J (t P v) + (dJ[t P v])^T g_V ]
// J (t P v) is the tangent voxel of the dual-number sweep; the bracket is what its adjoint sweep returns for the
// seeds (g_V, H_VV dV).  The tangent handed to cmax_objective_hvp must have max-norm <= 1, hence the device-side
// max |dV| (the propagation can amplify the tangent).
int enqueue_hvp_time_aware(cmax_patch_plan_s *p, hipStream_t s) {
    const cmax_patch_objective_t &d = p->d;
    CMAX_CHECK_HIP(hipMemcpyAsync(p->x64, p->h_in, (2 * (size_t)p->nx + 2) * sizeof(double), hipMemcpyHostToDevice, s));
    const double *inv_vmax = p->x64 + 2 * p->nx, *vmax = inv_vmax + 1;
    const int fgrid = div_up(p->nflow, 256), mgrid = div_up(p->nmotion, 256);
    // F = t P x,  dF = t P v / |v|_inf
    int rc = cmax_patch_to_dense(p->x64, CMAX_F64, d.ph, d.pw, d.pad_h, d.pad_w, d.sw_h, d.sw_w, d.H, d.W, 0, p->flow64, s);
    if (rc) return rc;
    rc = cmax_patch_to_dense(p->v64, CMAX_F64, d.ph, d.pw, d.pad_h, d.pad_w, d.sw_h, d.sw_w, d.H, d.W, 0, p->dflow64, s);
    if (rc) return rc;
    hipLaunchKernelGGL((k_convert_scale<double, double>), dim3(fgrid), dim3(256), 0, s, p->flow64, p->nflow, d.t_scale, (const double *)nullptr, p->flow64);
    hipLaunchKernelGGL((k_convert_scale<double, double>), dim3(fgrid), dim3(256), 0, s, p->dflow64, p->nflow, d.t_scale, inv_vmax, p->dflow64);
    CMAX_CHECK_LAUNCH();
    rc = cmax_voxel_construct_tan(p->flow64, p->dflow64, CMAX_F64, d.T, d.t0, d.H, d.W, d.scheme, p->vox64, p->dvox64, s);
    if (rc) return rc;
    // fp32 motion and unit-norm fp32 tangent of the fused terms
    unsigned long long *bits = reinterpret_cast<unsigned long long *>(p->scal + 2);
    CMAX_CHECK_HIP(hipMemsetAsync(bits, 0, sizeof(unsigned long long), s));
    hipLaunchKernelGGL(k_absmax, dim3(mgrid < 256 ? mgrid : 256), dim3(256), 0, s, p->dvox64, p->nmotion, bits);
    hipLaunchKernelGGL(k_absmax_finish, dim3(1), dim3(1), 0, s, bits, p->scal);
    hipLaunchKernelGGL((k_convert_scale<float, double>), dim3(mgrid), dim3(256), 0, s, p->vox64, p->nmotion, 1.0, (const double *)nullptr, p->motion32);
    hipLaunchKernelGGL((k_convert_scale<float, double>), dim3(mgrid), dim3(256), 0, s, p->dvox64, p->nmotion, 1.0, p->scal + 1, p->tan32);
    CMAX_CHECK_LAUNCH();
    for (int i = 0; i < d.n_terms; ++i) {
        rc = plan_objective(p, &d.term[i], p->results + 8 * i, p->grad32, s);  // g_V (time-sliced batch: this rank's share, like everything below)
        if (rc) return rc;
        hipLaunchKernelGGL(k_accumulate, dim3(mgrid), dim3(256), 0, s, p->grad32, p->nmotion, d.weight[i], i == 0 ? 1 : 0, p->gacc64, (const double *)nullptr);
        CMAX_CHECK_LAUNCH();
        rc = plan_objective_hvp(p, &d.term[i], p->grad32, s);  // H_VV dV / max|dV|
        if (rc) return rc;
        hipLaunchKernelGGL(k_accumulate, dim3(mgrid), dim3(256), 0, s, p->grad32, p->nmotion, d.weight[i], i == 0 ? 1 : 0, p->dgacc64, (const double *)p->scal);
        CMAX_CHECK_LAUNCH();
    }
    // the sweep leaves d(dL/dF) in bin t0 of the tangent gradient voxel: the interpolation adjoint reads it there
    rc = voxel_construct_adj_tan_f64(p->vox64, p->dvox64, d.T, d.t0, d.H, d.W, d.scheme, p->gacc64, p->dgacc64, s, handle_is_deterministic(p->handle));
    if (rc) return rc;
    rc = cmax_patch_to_dense(p->dgacc64 + (int64_t)d.t0 * p->nflow, CMAX_F64, d.ph, d.pw, d.pad_h, d.pad_w, d.sw_h, d.sw_w, d.H, d.W, 1, p->gx64, s);
    if (!rc) rc = plan_reduce_gx(p, p->gx64, nullptr, s);
    if (rc) return rc;
    FinalParams fp;
    fp.n_terms = 0;
    fp.flag_slot = 1 + (int)p->nx;
    fp.with_tv = 0;  // total_variation is piecewise linear: zero Hessian almost everywhere
    fp.nx = p->nx;
    fp.ph = d.ph;
    fp.pw = d.pw;
    fp.tv_crop = 0;
    for (int i = 0; i < 4; ++i) fp.weight[i] = 0.0;
    fp.tv_weight = 0.0;
    fp.gscale = d.t_scale;  // the outer t P^T; times |v|_inf from device memory
    hipLaunchKernelGGL(k_patch_tail, dim3(1), dim3(256), 0, s, fp, p->results, p->x64, p->gx64, (const float *)nullptr, vmax, p->h_out_dev, p->seq_dev);
    CMAX_CHECK_LAUNCH();
    return 0;
}

int enqueue_hvp(cmax_patch_plan_s *p, hipStream_t s) {
    const cmax_patch_objective_t &d = p->d;
    if (d.time_aware) return enqueue_hvp_time_aware(p, s);
    // x64 | v64 | scal64 are one allocation: a single copy brings x, v and the two scalars
    CMAX_CHECK_HIP(hipMemcpyAsync(p->x64, p->h_in, (2 * (size_t)p->nx + 2) * sizeof(double), hipMemcpyHostToDevice, s));
    const double *inv_vmax = p->x64 + 2 * p->nx, *vmax = inv_vmax + 1;
    int rc = forward_motion(p, p->x64, d.t_scale, nullptr, p->motion32, s);
    if (rc) return rc;
    // tangent of the flow, scaled to max-norm <= 1 (the interpolation is a negated convex combination, so
    // |P v|_inf <= |v|_inf): u = (t_scale * |v|_inf) * tan32, and H is linear in u
    rc = forward_motion(p, p->v64, 1.0, inv_vmax, p->tan32, s);
    if (rc) return rc;
    const bool accumulate = !(d.n_terms == 1 && !d.time_aware);
    for (int i = 0; i < d.n_terms; ++i) {
        rc = plan_objective_hvp(p, &d.term[i], p->grad32, s);
        if (rc) return rc;
        if (accumulate) {
            hipLaunchKernelGGL(k_accumulate, dim3(div_up(p->nmotion, 256)), dim3(256), 0, s, p->grad32, p->nmotion, d.weight[i], i == 0 ? 1 : 0, p->gacc64);
            CMAX_CHECK_LAUNCH();
        }
    }
    const double *gx64 = nullptr;
    const float *gx32 = nullptr;
    double wscale = 1.0;
    rc = backward_motion(p, &gx64, &gx32, &wscale, s);
    if (!rc) rc = plan_reduce_gx(p, gx64, gx32, s);
    if (rc) return rc;
    FinalParams fp;
    fp.n_terms = 0;
    fp.flag_slot = 1 + (int)p->nx;
    fp.with_tv = 0;  // total_variation is piecewise linear: zero Hessian almost everywhere
    fp.nx = p->nx;
    fp.ph = d.ph;
    fp.pw = d.pw;
    fp.tv_crop = 0;
    for (int i = 0; i < 4; ++i) fp.weight[i] = 0.0;
    fp.tv_weight = 0.0;
    fp.gscale = d.t_scale * d.t_scale * wscale;  // H_x = t^2 P^T H_flow P (times |v|_inf, from device memory)
    hipLaunchKernelGGL(k_patch_tail, dim3(1), dim3(256), 0, s, fp, p->results, p->x64, gx64, gx32, vmax, p->h_out_dev, p->seq_dev);
    CMAX_CHECK_LAUNCH();
    return 0;
}

uint64_t state_key(const HandleEvalState &st, int kind) { return handle_state_key(st, kind); }

void drop_graphs(cmax_patch_plan_s *p) {
    for (auto &g : p->graphs) (void)hipGraphExecDestroy(g.exec);
    p->graphs.clear();
}

// The tail kernel bumps a run counter in the pinned output after its results are visible.  Polling it returns as
// soon as the graph's last kernel has written, without the sleep / wake-up of hipStreamSynchronize.
static volatile unsigned long long *plan_flag(cmax_patch_plan_s *p) {
    return reinterpret_cast<volatile unsigned long long *>(p->h_out + 1 + p->nx);
}

static int wait_for_tail(cmax_patch_plan_s *p, hipStream_t s, bool poll) {
    const unsigned long long expected = p->seq_seen + 1ull;
    if (poll) {
        volatile unsigned long long *flag = plan_flag(p);
        for (long spin = 0; spin < 4000000L; ++spin) {  // a few ms at most, then sleep like everybody else
            if (*flag == expected) {
                std::atomic_thread_fence(std::memory_order_acquire);
                p->seq_seen = expected;
                return 0;
            }
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#elif defined(__aarch64__)
            asm volatile("yield" ::: "memory");
#endif
        }
    }
    CMAX_CHECK_HIP(hipStreamSynchronize(s));
    p->seq_seen = *plan_flag(p);
    return 0;
}

// Runs `enqueue(stream)` and waits for its result.  Default: eager launches on the caller's stream -- the first kernel
// starts while the host is still enqueueing the rest, and that beats replaying the same sequence from a captured
// hipGraph (measured, 30k-event YAML objective: 0.063 vs 0.075 ms per value+gradient, 0.154 vs 0.165 ms time-aware; a graph
// launch costs ~12 us before its first node runs, the nodes are dependent either way).  CMAX_PLAN_GRAPHS=1 (read at plan
// creation) switches the replay on: eager for the first calls (and whenever the handle is being profiled or a capture
// failed), afterwards a captured hipGraph per (kind, handle state) replayed on the plan's stream.
template <typename F>
int run_sequence(cmax_patch_plan_s *p, int kind, hipStream_t caller, F enqueue) {
    HandleEvalState pre;
    handle_get_eval_state(p->handle, &pre);
    if (!p->graphs.empty() && p->graphs[0].generation != pre.generation) {
        drop_graphs(p);  // new events behind the handle: other device pointers, another work list
        p->eager_calls = 0;
    }
    const bool eager = !p->graphs_ok || pre.profiling || p->eager_calls < 3 || handle_has_comm(p->handle);  // (collectives are never captured)
    if (eager) {
        ++p->eager_calls;
        int rc = enqueue(caller);
        if (rc) return rc;
        return wait_for_tail(p, caller, p->eager_calls > 1);  // (the first call may compile / allocate: sleep)
    }
    const uint64_t key = state_key(pre, kind);
    cmax_patch_plan_s::GraphEntry *entry = nullptr;
    for (auto &g : p->graphs)
        if (g.key == key) entry = &g;
    if (!entry) {
        if (p->graphs.size() >= 32) drop_graphs(p);
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        bool ok = hipStreamBeginCapture(p->own_stream, hipStreamCaptureModeRelaxed) == hipSuccess;
        int rc = 0;
        if (ok) {
            rc = enqueue(p->own_stream);  // host-side bookkeeping of the handle runs as in an eager call
            ok = hipStreamEndCapture(p->own_stream, &graph) == hipSuccess && rc == 0 && graph != nullptr;
        }
        if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
        if (graph) (void)hipGraphDestroy(graph);
        if (!ok) {  // never again: fall back to eager launches
            (void)hipGetLastError();
            p->graphs_ok = false;
            handle_set_eval_state(p->handle, &pre);
            rc = enqueue(caller);
            if (rc) return rc;
            return wait_for_tail(p, caller, false);
        }
        cmax_patch_plan_s::GraphEntry e;
        e.key = key;
        e.generation = pre.generation;
        e.exec = exec;
        handle_get_eval_state(p->handle, &e.post);
        p->graphs.push_back(e);
        entry = &p->graphs.back();
    } else {
        handle_set_eval_state(p->handle, &entry->post);
    }
    // order after whatever the caller has queued, run, wait
    CMAX_CHECK_HIP(hipEventRecord(p->ev_caller, caller));
    CMAX_CHECK_HIP(hipStreamWaitEvent(p->own_stream, p->ev_caller, 0));
    CMAX_CHECK_HIP(hipGraphLaunch(entry->exec, p->own_stream));
    return wait_for_tail(p, p->own_stream, true);
}

}  // namespace

extern "C" {

int cmax_sizeof_patch_objective(void) { return (int)sizeof(cmax_patch_objective_t); }

int cmax_patch_plan_create(cmax_handle_t h, const cmax_patch_objective_t *desc, cmax_patch_plan_t *out) {
    CMAX_REQUIRE(h && desc && out, "patch_plan_create: null pointer");
    const cmax_patch_objective_t &d = *desc;
    CMAX_REQUIRE(d.n_terms >= 1 && d.n_terms <= 4, "patch_plan_create: n_terms");
    CMAX_REQUIRE(d.H > 0 && d.W > 0 && d.ph > 0 && d.pw > 0 && d.sw_h > 0 && d.sw_w > 0 && d.pad_h >= 0 && d.pad_w >= 0, "patch_plan_create: sizes");
    CMAX_REQUIRE(!d.time_aware || (d.T > 0 && d.t0 >= 0 && d.t0 < d.T), "patch_plan_create: time bins");
    for (int i = 0; i < d.n_terms; ++i) {
        CMAX_REQUIRE(d.term[i].model == (d.time_aware ? CMAX_MODEL_VOXEL : CMAX_MODEL_DENSE), "patch_plan_create: term model must match time_aware");
        CMAX_REQUIRE(!d.time_aware || d.term[i].T == d.T, "patch_plan_create: term T");
    }
    cmax_patch_plan_s *p = new (std::nothrow) cmax_patch_plan_s();
    if (!p) {
        set_error("patch_plan_create: out of host memory");
        return CMAX_ENOMEM;
    }
    p->handle = h;
    p->d = d;
    p->nx = 2 * d.ph * d.pw;
    if (const char *e = getenv("CMAX_PLAN_GRAPHS")) p->graphs_ok = atoi(e) != 0;
    p->nflow = 2 * (int64_t)d.H * d.W;
    p->nmotion = d.time_aware ? (int64_t)d.T * p->nflow : p->nflow;
    int rc = 0;
    if (!rc) rc = plan_alloc(&p->x64, 2 * (int64_t)p->nx + 2);  // x | v | 2 scalars, filled by one copy
    if (!rc) p->v64 = p->x64 + p->nx;
    if (!rc) rc = plan_alloc(&p->flow64, p->nflow);
    if (!rc && d.time_aware) rc = plan_alloc(&p->vox64, p->nmotion);
    if (!rc) rc = plan_alloc(&p->gacc64, p->nmotion);
    if (!rc && d.time_aware) rc = plan_alloc(&p->gflow64, p->nflow);
    if (!rc) rc = plan_alloc(&p->gx64, p->nx);
    if (!rc) rc = plan_alloc(&p->results, 8 * 4);
    if (!rc) rc = plan_alloc(&p->motion32, p->nmotion);
    if (!rc) rc = plan_alloc(&p->grad32, p->nmotion);
    if (!rc) rc = plan_alloc(&p->tan32, p->nmotion);
    if (!rc) rc = plan_alloc(&p->gx32, p->nx);
    if (!rc && hipHostMalloc((void **)&p->h_in, (2 * (size_t)p->nx + 2) * sizeof(double)) != hipSuccess) rc = CMAX_ENOMEM;
    if (!rc && hipStreamCreateWithFlags(&p->own_stream, hipStreamNonBlocking) != hipSuccess) rc = CMAX_ENOMEM;
    if (!rc && hipEventCreateWithFlags(&p->ev_caller, hipEventDisableTiming) != hipSuccess) rc = CMAX_ENOMEM;
    if (!rc && hipHostMalloc((void **)&p->h_out, (2 + (size_t)p->nx) * sizeof(double), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) rc = CMAX_ENOMEM;
    if (!rc && hipHostGetDevicePointer((void **)&p->h_out_dev, p->h_out, 0) != hipSuccess) {
        (void)hipGetLastError();
        p->h_out_dev = p->h_out;  // unified addressing: pinned host memory is addressable from the device as it is
    }
    if (!rc) rc = plan_alloc(&p->seq_dev, 1);
    if (!rc && hipMemset(p->seq_dev, 0, sizeof(unsigned long long)) != hipSuccess) rc = CMAX_ENOMEM;
    if (!rc) reinterpret_cast<unsigned long long *>(p->h_out + 1 + p->nx)[0] = 0ull;
    if (rc) {
        if (rc == CMAX_ENOMEM) set_error("patch_plan_create: allocation failed");
        cmax_patch_plan_destroy(p);
        return rc;
    }
    *out = p;
    return 0;
}

int cmax_patch_plan_set_t_scale(cmax_patch_plan_t p, double t_scale) {
    CMAX_REQUIRE(p != nullptr, "patch_plan_set_t_scale");
    if (p->d.t_scale != t_scale) {
        p->d.t_scale = t_scale;
        drop_graphs(p);  // the scale is baked into captured launches
        p->eager_calls = 0;
    }
    return 0;
}

int cmax_patch_plan_info(cmax_patch_plan_t p, int *n_graphs, int *graph_replay_enabled) {
    CMAX_REQUIRE(p && n_graphs && graph_replay_enabled, "patch_plan_info");
    *n_graphs = (int)p->graphs.size();
    *graph_replay_enabled = p->graphs_ok ? 1 : 0;
    return 0;
}

int cmax_patch_plan_destroy(cmax_patch_plan_t p) {
    if (!p) return 0;
    drop_graphs(p);
    if (p->own_stream) (void)hipStreamDestroy(p->own_stream);
    if (p->ev_caller) (void)hipEventDestroy(p->ev_caller);
    double *d64[] = {p->x64, p->flow64, p->vox64, p->gacc64, p->gflow64, p->gx64, p->results, p->dflow64, p->dvox64, p->dgacc64, p->dgflow64, p->scal};
    for (double *q : d64)
        if (q) (void)hipFree(q);
    float *d32[] = {p->motion32, p->grad32, p->tan32, p->gx32};
    for (float *q : d32)
        if (q) (void)hipFree(q);
    if (p->h_in) (void)hipHostFree(p->h_in);
    if (p->h_out) (void)hipHostFree(p->h_out);
    if (p->seq_dev) (void)hipFree(p->seq_dev);
    delete p;
    return 0;
}

int cmax_patch_plan_evaluate(cmax_patch_plan_t p, const double *x_host, int with_tv, double *loss_host, double *grad_host,
                             cmax_stream_t stream) {
    CMAX_REQUIRE(p && x_host && loss_host, "patch_plan_evaluate: null pointer");
    const bool tv = with_tv && p->d.tv_weight != 0.0;
    std::memcpy(p->h_in, x_host, (size_t)p->nx * sizeof(double));
    const int kind = (tv ? 1 : 0) | (grad_host ? 2 : 0);
    int rc = run_sequence(p, kind, (hipStream_t)stream, [&](hipStream_t s) { return enqueue_evaluate(p, tv, grad_host != nullptr, s); });
    if (rc) return rc;
    *loss_host = p->h_out[0];
    if (grad_host) std::memcpy(grad_host, p->h_out + 1, (size_t)p->nx * sizeof(double));
    return 0;
}

int cmax_patch_plan_hvp(cmax_patch_plan_t p, const double *x_host, const double *v_host, double *hv_host, cmax_stream_t stream) {
    CMAX_REQUIRE(p && x_host && v_host && hv_host, "patch_plan_hvp: null pointer");
    const cmax_patch_objective_t &d = p->d;
    if (d.time_aware && !p->dvox64) {  // second-order buffers of the voxel chain, on first use
        int rc = plan_alloc(&p->dflow64, p->nflow);
        if (!rc) rc = plan_alloc(&p->dvox64, p->nmotion);
        if (!rc) rc = plan_alloc(&p->dgacc64, p->nmotion);
        if (!rc) rc = plan_alloc(&p->dgflow64, p->nflow);
        if (!rc) rc = plan_alloc(&p->scal, 4);
        if (rc) return rc;
    }
    double vmax = 0.0;
    for (int j = 0; j < p->nx; ++j) vmax = fabs(v_host[j]) > vmax ? fabs(v_host[j]) : vmax;
    if (!(vmax > 0.0)) {
        std::memset(hv_host, 0, (size_t)p->nx * sizeof(double));
        return 0;
    }
    std::memcpy(p->h_in, x_host, (size_t)p->nx * sizeof(double));
    std::memcpy(p->h_in + p->nx, v_host, (size_t)p->nx * sizeof(double));
    p->h_in[2 * p->nx] = 1.0 / vmax;  // the tangent of the flow is scaled to max-norm <= 1 ...
    p->h_in[2 * p->nx + 1] = vmax;    // ... and the product scaled back (H is linear in the tangent)
    int rc = run_sequence(p, 4, (hipStream_t)stream, [&](hipStream_t s) { return enqueue_hvp(p, s); });
    if (rc) return rc;
    std::memcpy(hv_host, p->h_out + 1, (size_t)p->nx * sizeof(double));
    return 0;
}

}  // extern "C"
