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
// ctypes call instead of ~40 Python-level operations (measured: 0.47 ms -> see DESIGN.md).
// Pre/post stages run in fp64 like the reference's solver (patch_contrast_pyramid.py:186); the event
// path is fp32 per event with fp64 reductions as everywhere in cmax_fused.hip.
#include <cstring>
#include <new>

#include "cmax_common.h"

namespace cmax {

template <typename TO, typename TI>
__global__ void __launch_bounds__(256) k_convert_scale(const TI *__restrict__ in, int64_t n, double scale, TO *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (TO)((double)in[i] * scale);
}

// acc (=, +=) w * g
__global__ void __launch_bounds__(256) k_accumulate(const float *__restrict__ g, int64_t n, double w, int first, double *__restrict__ acc) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) acc[i] = (first ? 0.0 : acc[i]) + w * (double)g[i];
}

struct FinalParams {
    int n_terms, with_tv, nx;
    double weight[4], tv_weight, gscale;
};

// out[0] = loss, out[1 + j] = d loss / d x[j]
__global__ void __launch_bounds__(256)
k_patch_final(FinalParams fp, const double *__restrict__ results, const double *__restrict__ tv_value, const double *__restrict__ gx,
              const double *__restrict__ gtv, double *__restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0) {
        double loss = 0.0;
        for (int i = 0; i < fp.n_terms; ++i) loss += fp.weight[i] * results[8 * i];
        if (fp.with_tv) loss += fp.tv_weight * tv_value[0];
        out[0] = loss;
    }
    if (j < fp.nx) out[1 + j] = fp.gscale * gx[j] + (fp.with_tv ? gtv[j] : 0.0);
}

}  // namespace cmax

struct cmax_patch_plan_s {
    cmax_handle_t handle = nullptr;
    cmax_patch_objective_t d;
    int nx = 0;            // 2 * ph * pw
    int64_t nflow = 0;     // 2 * H * W
    int64_t nmotion = 0;   // nflow or T * nflow
    double *x64 = nullptr, *v64 = nullptr, *flow64 = nullptr, *vox64 = nullptr, *gacc64 = nullptr, *gflow64 = nullptr;
    double *gx64 = nullptr, *gtv64 = nullptr, *tv_value = nullptr, *results = nullptr, *out64 = nullptr, *d_tvw = nullptr;
    float *motion32 = nullptr, *grad32 = nullptr, *tan32 = nullptr;
    double *h_in = nullptr, *h_out = nullptr;  // pinned staging: x | v, loss | grad
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

// x (host) -> fp32 motion of the fused objective: flow [2,H,W] or voxel [T,2,H,W] in pixel per normalised time.
// Leaves the fp64 flow (scaled) in flow64 and, when time-aware, the fp64 voxel in vox64.
int forward_motion(cmax_patch_plan_s *p, const double *src64, double scale, float *dst32, hipStream_t s) {
    const cmax_patch_objective_t &d = p->d;
    int rc = cmax_patch_to_dense(src64, CMAX_F64, d.ph, d.pw, d.pad_h, d.pad_w, d.sw_h, d.sw_w, d.H, d.W, 0, p->flow64, s);
    if (rc) return rc;
    const int grid = div_up(p->nflow, 256);
    if (!d.time_aware) {
        hipLaunchKernelGGL((k_convert_scale<float, double>), dim3(grid), dim3(256), 0, s, p->flow64, p->nflow, scale, dst32);
        CMAX_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL((k_convert_scale<double, double>), dim3(grid), dim3(256), 0, s, p->flow64, p->nflow, scale, p->flow64);
    CMAX_CHECK_LAUNCH();
    // the voxel is built on the displacement field (patch_contrast_pyramid.py:452, 499-515)
    rc = cmax_voxel_construct(p->flow64, CMAX_F64, d.T, d.t0, d.H, d.W, d.scheme, p->vox64, s);
    if (rc) return rc;
    hipLaunchKernelGGL((k_convert_scale<float, double>), dim3(div_up(p->nmotion, 256)), dim3(256), 0, s, p->vox64, p->nmotion, 1.0, dst32);
    CMAX_CHECK_LAUNCH();
    return 0;
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
    p->nflow = 2 * (int64_t)d.H * d.W;
    p->nmotion = d.time_aware ? (int64_t)d.T * p->nflow : p->nflow;
    int rc = 0;
    if (!rc) rc = plan_alloc(&p->x64, p->nx);
    if (!rc) rc = plan_alloc(&p->v64, p->nx);
    if (!rc) rc = plan_alloc(&p->flow64, p->nflow);
    if (!rc && d.time_aware) rc = plan_alloc(&p->vox64, p->nmotion);
    if (!rc) rc = plan_alloc(&p->gacc64, p->nmotion);
    if (!rc && d.time_aware) rc = plan_alloc(&p->gflow64, p->nflow);
    if (!rc) rc = plan_alloc(&p->gx64, p->nx);
    if (!rc) rc = plan_alloc(&p->gtv64, p->nx);
    if (!rc) rc = plan_alloc(&p->tv_value, 4);
    if (!rc) rc = plan_alloc(&p->results, 8 * 4);
    if (!rc) rc = plan_alloc(&p->out64, 1 + p->nx);
    if (!rc) rc = plan_alloc(&p->d_tvw, 1);
    if (!rc) rc = plan_alloc(&p->motion32, p->nmotion);
    if (!rc) rc = plan_alloc(&p->grad32, p->nmotion);
    if (!rc) rc = plan_alloc(&p->tan32, p->nmotion);
    if (!rc && hipHostMalloc((void **)&p->h_in, 2 * (size_t)p->nx * sizeof(double)) != hipSuccess) rc = CMAX_ENOMEM;
    if (!rc && hipHostMalloc((void **)&p->h_out, (1 + (size_t)p->nx) * sizeof(double)) != hipSuccess) rc = CMAX_ENOMEM;
    if (!rc && hipMemcpy(p->d_tvw, &d.tv_weight, sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = CMAX_ENOMEM;
    if (rc) {
        if (rc == CMAX_ENOMEM) set_error("patch_plan_create: allocation failed");
        cmax_patch_plan_destroy(p);
        return rc;
    }
    *out = p;
    return 0;
}

int cmax_patch_plan_destroy(cmax_patch_plan_t p) {
    if (!p) return 0;
    double *d64[] = {p->x64, p->v64, p->flow64, p->vox64, p->gacc64, p->gflow64, p->gx64, p->gtv64, p->tv_value, p->results, p->out64, p->d_tvw};
    for (double *q : d64)
        if (q) (void)hipFree(q);
    float *d32[] = {p->motion32, p->grad32, p->tan32};
    for (float *q : d32)
        if (q) (void)hipFree(q);
    if (p->h_in) (void)hipHostFree(p->h_in);
    if (p->h_out) (void)hipHostFree(p->h_out);
    delete p;
    return 0;
}

int cmax_patch_plan_evaluate(cmax_patch_plan_t p, const double *x_host, int with_tv, double *loss_host, double *grad_host,
                             cmax_stream_t stream) {
    CMAX_REQUIRE(p && x_host && loss_host, "patch_plan_evaluate: null pointer");
    const cmax_patch_objective_t &d = p->d;
    hipStream_t s = (hipStream_t)stream;
    const bool tv = with_tv && d.tv_weight != 0.0;
    std::memcpy(p->h_in, x_host, (size_t)p->nx * sizeof(double));
    CMAX_CHECK_HIP(hipMemcpyAsync(p->x64, p->h_in, (size_t)p->nx * sizeof(double), hipMemcpyHostToDevice, s));
    int rc = forward_motion(p, p->x64, d.t_scale, p->motion32, s);
    if (rc) return rc;
    for (int i = 0; i < d.n_terms; ++i) {
        rc = cmax_objective(p->handle, &d.term[i], p->motion32, p->results + 8 * i, grad_host ? p->grad32 : nullptr, s);
        if (rc) return rc;
        if (grad_host) {
            hipLaunchKernelGGL(k_accumulate, dim3(div_up(p->nmotion, 256)), dim3(256), 0, s, p->grad32, p->nmotion, d.weight[i], i == 0 ? 1 : 0, p->gacc64);
            CMAX_CHECK_LAUNCH();
        }
    }
    if (grad_host) {
        const double *gflow = p->gacc64;
        if (d.time_aware) {
            rc = cmax_voxel_construct_adj(p->vox64, CMAX_F64, d.T, d.t0, d.H, d.W, d.scheme, p->gacc64, p->gflow64, s);
            if (rc) return rc;
            gflow = p->gflow64;
        }
        rc = cmax_patch_to_dense(gflow, CMAX_F64, d.ph, d.pw, d.pad_h, d.pad_w, d.sw_h, d.sw_w, d.H, d.W, 1, p->gx64, s);
        if (rc) return rc;
    }
    if (tv) {
        rc = cmax_total_variation(p->x64, CMAX_F64, d.ph, d.pw, d.tv_omit_boundary, p->tv_value, grad_host ? p->gtv64 : nullptr, p->d_tvw, s);
        if (rc) return rc;
    }
    FinalParams fp;
    fp.n_terms = d.n_terms;
    fp.with_tv = tv ? 1 : 0;
    fp.nx = grad_host ? p->nx : 0;
    for (int i = 0; i < 4; ++i) fp.weight[i] = d.weight[i];
    fp.tv_weight = d.tv_weight;
    fp.gscale = d.t_scale;  // d(flow * t_scale) / d flow
    hipLaunchKernelGGL(k_patch_final, dim3(div_up(p->nx + 1, 256)), dim3(256), 0, s, fp, p->results, p->tv_value, p->gx64, p->gtv64, p->out64);
    CMAX_CHECK_LAUNCH();
    const size_t nout = grad_host ? 1 + (size_t)p->nx : 1;
    CMAX_CHECK_HIP(hipMemcpyAsync(p->h_out, p->out64, nout * sizeof(double), hipMemcpyDeviceToHost, s));
    CMAX_CHECK_HIP(hipStreamSynchronize(s));
    *loss_host = p->h_out[0];
    if (grad_host) std::memcpy(grad_host, p->h_out + 1, (size_t)p->nx * sizeof(double));
    return 0;
}

int cmax_patch_plan_hvp(cmax_patch_plan_t p, const double *x_host, const double *v_host, double *hv_host, cmax_stream_t stream) {
    CMAX_REQUIRE(p && x_host && v_host && hv_host, "patch_plan_hvp: null pointer");
    const cmax_patch_objective_t &d = p->d;
    CMAX_REQUIRE(!d.time_aware, "patch_plan_hvp: the Burgers voxel chain has no second-order adjoint (difference the gradient instead)");
    hipStream_t s = (hipStream_t)stream;
    double vmax = 0.0;
    for (int j = 0; j < p->nx; ++j) vmax = fabs(v_host[j]) > vmax ? fabs(v_host[j]) : vmax;
    if (!(vmax > 0.0)) {
        std::memset(hv_host, 0, (size_t)p->nx * sizeof(double));
        return 0;
    }
    std::memcpy(p->h_in, x_host, (size_t)p->nx * sizeof(double));
    std::memcpy(p->h_in + p->nx, v_host, (size_t)p->nx * sizeof(double));
    CMAX_CHECK_HIP(hipMemcpyAsync(p->x64, p->h_in, (size_t)p->nx * sizeof(double), hipMemcpyHostToDevice, s));
    CMAX_CHECK_HIP(hipMemcpyAsync(p->v64, p->h_in + p->nx, (size_t)p->nx * sizeof(double), hipMemcpyHostToDevice, s));
    int rc = forward_motion(p, p->x64, d.t_scale, p->motion32, s);
    if (rc) return rc;
    // tangent of the flow, scaled to max-norm <= 1 (the interpolation is a negated convex combination, so
    // |P v|_inf <= |v|_inf): u = (t_scale * vmax) * tan32, and H is linear in u
    rc = forward_motion(p, p->v64, 1.0 / vmax, p->tan32, s);
    if (rc) return rc;
    for (int i = 0; i < d.n_terms; ++i) {
        rc = cmax_objective_hvp(p->handle, &d.term[i], p->motion32, p->tan32, p->grad32, s);
        if (rc) return rc;
        hipLaunchKernelGGL(k_accumulate, dim3(div_up(p->nmotion, 256)), dim3(256), 0, s, p->grad32, p->nmotion, d.weight[i], i == 0 ? 1 : 0, p->gacc64);
        CMAX_CHECK_LAUNCH();
    }
    rc = cmax_patch_to_dense(p->gacc64, CMAX_F64, d.ph, d.pw, d.pad_h, d.pad_w, d.sw_h, d.sw_w, d.H, d.W, 1, p->gx64, s);
    if (rc) return rc;
    FinalParams fp;
    fp.n_terms = 0;
    fp.with_tv = 0;  // total_variation is piecewise linear: zero Hessian almost everywhere
    fp.nx = p->nx;
    for (int i = 0; i < 4; ++i) fp.weight[i] = 0.0;
    fp.tv_weight = 0.0;
    fp.gscale = d.t_scale * d.t_scale * vmax;  // H_x = t^2 P^T H_flow P
    hipLaunchKernelGGL(k_patch_final, dim3(div_up(p->nx + 1, 256)), dim3(256), 0, s, fp, p->results, p->tv_value, p->gx64, p->gtv64, p->out64);
    CMAX_CHECK_LAUNCH();
    CMAX_CHECK_HIP(hipMemcpyAsync(p->h_out, p->out64, (1 + (size_t)p->nx) * sizeof(double), hipMemcpyDeviceToHost, s));
    CMAX_CHECK_HIP(hipStreamSynchronize(s));
    std::memcpy(hv_host, p->h_out + 1, (size_t)p->nx * sizeof(double));
    return 0;
}

}  // extern "C"
