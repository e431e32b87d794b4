// Time-aware flow: explicit Burgers / upwind propagation of a dense flow field into a voxel of
// time bins, and the adjoints.  Restates src/utils/flow_utils.py:99-161 (voxel), 567-639 (Burgers),
// 439-493 (upwind) as one gather kernel per step: each thread owns one pixel and reads its
// 4-neighbourhood (the reference materialises ~30 intermediate ATen tensors per step).
#include "cmax_common.h"
#include "cmax_flow_dual.h"
#include "cmax_patch_kernels.h"

namespace cmax {

template <typename T>
__device__ __forceinline__ T sgn(T v) { return (T)((v > (T)0) - (v < (T)0)); }
template <typename T>
__device__ __forceinline__ T max0(T v) { return v > (T)0 ? v : (T)0; }
template <typename T>
__device__ __forceinline__ T min0(T v) { return v < (T)0 ? v : (T)0; }
// torch.maximum / minimum(x, 0) sub-gradients: 1 where selected, 1/2 at the tie
template <typename T>
__device__ __forceinline__ T dmax0(T v) { return v > (T)0 ? (T)1 : (v == (T)0 ? (T)0.5 : (T)0); }
template <typename T>
__device__ __forceinline__ T dmin0(T v) { return v < (T)0 ? (T)1 : (v == (T)0 ? (T)0.5 : (T)0); }

// One step.  s = sign(dt), tau = |dt|; f = s*F; out = s * f_new.   (flow_utils.py:582-639)
// Up to 3 independent jobs per launch (blockIdx.y): the voxel is propagated from bin t0 in both time directions, and
// the two chains advance in the same launch (a dependent launch costs ~4.5 us, a step on 2 x 260 x 346 about as much).
// s == 0: plain copy src -> dst.
template <typename T>
struct StepJobs {
    const T *src[3];
    T *dst[3];        // forward: output; adjoint: gradient to accumulate into
    const T *gout[3]; // adjoint only: upstream gradient
    T s[3];
};

template <typename T, int SCHEME>
__device__ __forceinline__ void flow_step_pixel(const T *__restrict__ F, int H, int W, T s, T tau, T *__restrict__ out) {
    const int64_t hw = (int64_t)H * W;
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hw) return;
    const int i = (int)(p / W), j = (int)(p % W);
    auto U = [&](int r, int c) { return s * F[(int64_t)r * W + c]; };
    auto V = [&](int r, int c) { return s * F[hw + (int64_t)r * W + c]; };
    const T u = U(i, j), v = V(i, j);
    T nu, nv;
    if (SCHEME == CMAX_SCHEME_BURGERS) {
        const int ip = i + 1 < H ? i + 1 : H - 1, im = i > 0 ? i - 1 : 0;  // replicate pad 598-601
        const int jp = j + 1 < W ? j + 1 : W - 1, jm = j > 0 ? j - 1 : 0;
        const T uf = U(ip, j), ub = U(im, j), vf = V(i, jp), vb = V(i, jm);
        const T bu = (u * u * sgn(u) + max0(sgn(ub)) * (-ub * ub) - min0(sgn(uf)) * (uf * uf)) / (T)2;  // 611-615
        const T bv = (v * v * sgn(v) + max0(sgn(vb)) * (-vb * vb) - min0(sgn(vf)) * (vf * vf)) / (T)2;
        const T u_dy_back = j > 0 ? u - U(i, j - 1) : (T)0, u_dy_forw = j + 1 < W ? U(i, j + 1) - u : (T)0;  // 618-625
        const T v_dx_back = i > 0 ? v - V(i - 1, j) : (T)0, v_dx_forw = i + 1 < H ? V(i + 1, j) - v : (T)0;
        nu = u - tau * (max0(v) * u_dy_back + min0(v) * u_dy_forw + bu);  // 628-638
        nv = v - tau * (max0(u) * v_dx_back + min0(u) * v_dx_forw + bv);
    } else {  // upwind, flow_utils.py:459-492
        const T u_dx_back = i > 0 ? u - U(i - 1, j) : (T)0, u_dx_forw = i + 1 < H ? U(i + 1, j) - u : (T)0;
        const T u_dy_back = j > 0 ? u - U(i, j - 1) : (T)0, u_dy_forw = j + 1 < W ? U(i, j + 1) - u : (T)0;
        const T v_dx_back = i > 0 ? v - V(i - 1, j) : (T)0, v_dx_forw = i + 1 < H ? V(i + 1, j) - v : (T)0;
        const T v_dy_back = j > 0 ? v - V(i, j - 1) : (T)0, v_dy_forw = j + 1 < W ? V(i, j + 1) - v : (T)0;
        nu = u - tau * (max0(u) * u_dx_back + min0(u) * u_dx_forw + max0(v) * u_dy_back + min0(v) * u_dy_forw);
        nv = v - tau * (max0(u) * v_dx_back + min0(u) * v_dx_forw + max0(v) * v_dy_back + min0(v) * v_dy_forw);
    }
    out[p] = nu * s;
    out[hw + p] = nv * s;
}

template <typename T, int SCHEME>
__global__ void __launch_bounds__(256) k_flow_step(const T *__restrict__ F, int H, int W, T s, T tau, T *__restrict__ out) {
    flow_step_pixel<T, SCHEME>(F, H, W, s, tau, out);
}

template <typename T, int SCHEME>
__global__ void __launch_bounds__(256) k_flow_step_jobs(StepJobs<T> jobs, int H, int W, T tau) {
    const int y = blockIdx.y;
    if (jobs.s[y] == (T)0) {  // copy
        const int64_t hw = (int64_t)H * W, p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (p < hw) {
            jobs.dst[y][p] = jobs.src[y][p];
            jobs.dst[y][hw + p] = jobs.src[y][hw + p];
        }
        return;
    }
    flow_step_pixel<T, SCHEME>(jobs.src[y], H, W, jobs.s[y], tau, jobs.dst[y]);
}

// Adjoint of one step, scatter form: the thread of output pixel (i,j) adds its contributions to
// the gradient of every input it read.  d out / d F = d f_new / d f because s*s = 1.
template <typename T, int SCHEME>
__device__ __forceinline__ void flow_step_adj_pixel(const T *__restrict__ F, int H, int W, T s, T tau, const T *__restrict__ gout,
                                                    T *__restrict__ gF) {
    const int64_t hw = (int64_t)H * W;
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hw) return;
    const int i = (int)(p / W), j = (int)(p % W);
    auto U = [&](int r, int c) { return s * F[(int64_t)r * W + c]; };
    auto V = [&](int r, int c) { return s * F[hw + (int64_t)r * W + c]; };
    auto GU = [&](int r, int c, T val) { atomic_add(&gF[(int64_t)r * W + c], val); };
    auto GV = [&](int r, int c, T val) { atomic_add(&gF[hw + (int64_t)r * W + c], val); };
    const T u = U(i, j), v = V(i, j);
    const T gnu = gout[p], gnv = gout[hw + p];
    const T mt = -tau;
    if (SCHEME == CMAX_SCHEME_BURGERS) {
        const int ip = i + 1 < H ? i + 1 : H - 1, im = i > 0 ? i - 1 : 0;
        const int jp = j + 1 < W ? j + 1 : W - 1, jm = j > 0 ? j - 1 : 0;
        const T uf = U(ip, j), ub = U(im, j), vf = V(i, jp), vb = V(i, jm);
        const T u_dy_back = j > 0 ? u - U(i, j - 1) : (T)0, u_dy_forw = j + 1 < W ? U(i, j + 1) - u : (T)0;
        const T v_dx_back = i > 0 ? v - V(i - 1, j) : (T)0, v_dx_forw = i + 1 < H ? V(i + 1, j) - v : (T)0;
        // channel u
        T self_u = gnu * ((T)1 - tau * fabs(u));  // d(u|u|/2)/du = |u|
        GU(im, j, gnu * mt * (-(ub > (T)0 ? ub : (T)0)));
        GU(ip, j, gnu * mt * ((uf < (T)0 ? uf : (T)0)));
        T self_v = gnu * mt * (dmax0(v) * u_dy_back + dmin0(v) * u_dy_forw);
        const T mvp = max0(v), mvn = min0(v);
        if (j > 0) {
            self_u += gnu * mt * mvp;
            GU(i, j - 1, -gnu * mt * mvp);
        }
        if (j + 1 < W) {
            GU(i, j + 1, gnu * mt * mvn);
            self_u -= gnu * mt * mvn;
        }
        // channel v
        self_v += gnv * ((T)1 - tau * fabs(v));
        GV(i, jm, gnv * mt * (-(vb > (T)0 ? vb : (T)0)));
        GV(i, jp, gnv * mt * ((vf < (T)0 ? vf : (T)0)));
        self_u += gnv * mt * (dmax0(u) * v_dx_back + dmin0(u) * v_dx_forw);
        const T mup = max0(u), mun = min0(u);
        if (i > 0) {
            self_v += gnv * mt * mup;
            GV(i - 1, j, -gnv * mt * mup);
        }
        if (i + 1 < H) {
            GV(i + 1, j, gnv * mt * mun);
            self_v -= gnv * mt * mun;
        }
        GU(i, j, self_u);
        GV(i, j, self_v);
    } else {
        const T mup = max0(u), mun = min0(u), mvp = max0(v), mvn = min0(v);
        T self[2] = {(T)0, (T)0};
        for (int c = 0; c < 2; ++c) {
            const T g = c == 0 ? gnu : gnv;
            auto Cc = [&](int r, int q) { return c == 0 ? U(r, q) : V(r, q); };
            auto GC = [&](int r, int q, T val) { if (c == 0) GU(r, q, val); else GV(r, q, val); };
            const T f = c == 0 ? u : v;
            const T dx_back = i > 0 ? f - Cc(i - 1, j) : (T)0, dx_forw = i + 1 < H ? Cc(i + 1, j) - f : (T)0;
            const T dy_back = j > 0 ? f - Cc(i, j - 1) : (T)0, dy_forw = j + 1 < W ? Cc(i, j + 1) - f : (T)0;
            self[c] += g;
            self[0] += g * mt * (dmax0(u) * dx_back + dmin0(u) * dx_forw);
            self[1] += g * mt * (dmax0(v) * dy_back + dmin0(v) * dy_forw);
            if (i > 0) { self[c] += g * mt * mup; GC(i - 1, j, -g * mt * mup); }
            if (i + 1 < H) { GC(i + 1, j, g * mt * mun); self[c] -= g * mt * mun; }
            if (j > 0) { self[c] += g * mt * mvp; GC(i, j - 1, -g * mt * mvp); }
            if (j + 1 < W) { GC(i, j + 1, g * mt * mvn); self[c] -= g * mt * mvn; }
        }
        GU(i, j, self[0]);
        GV(i, j, self[1]);
    }
}

template <typename T, int SCHEME>
__global__ void __launch_bounds__(256)
k_flow_step_adj(const T *__restrict__ F, int H, int W, T s, T tau, const T *__restrict__ gout, T *__restrict__ gF) {
    flow_step_adj_pixel<T, SCHEME>(F, H, W, s, tau, gout, gF);
}

template <typename T, int SCHEME>
__global__ void __launch_bounds__(256) k_flow_step_adj_jobs(StepJobs<T> jobs, int H, int W, T tau) {
    const int y = blockIdx.y;
    flow_step_adj_pixel<T, SCHEME>(jobs.src[y], H, W, jobs.s[y], tau, jobs.gout[y], jobs.dst[y]);
}

template <typename T>
__global__ void __launch_bounds__(256) k_axpy1(int64_t n, const T *__restrict__ x, T *__restrict__ y) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) y[p] += x[p];
}

template <typename T>
static int flow_step(const T *F, int H, int W, double dt, int scheme, T *out, hipStream_t s) {
    const int64_t hw = (int64_t)H * W;
    if (dt == 0.0) {  // flow_utils.py:582-583
        CMAX_CHECK_HIP(hipMemcpyAsync(out, F, 2 * hw * sizeof(T), hipMemcpyDeviceToDevice, s));
        return 0;
    }
    const T sg = dt > 0 ? (T)1 : (T)-1, tau = (T)fabs(dt);
    const int grid = div_up(hw, 256);
    if (scheme == CMAX_SCHEME_BURGERS)
        hipLaunchKernelGGL((k_flow_step<T, CMAX_SCHEME_BURGERS>), dim3(grid), dim3(256), 0, s, F, H, W, sg, tau, out);
    else
        hipLaunchKernelGGL((k_flow_step<T, CMAX_SCHEME_UPWIND>), dim3(grid), dim3(256), 0, s, F, H, W, sg, tau, out);
    CMAX_CHECK_LAUNCH();
    return 0;
}

template <typename T>
static int flow_step_adj(const T *F, int H, int W, double dt, int scheme, const T *gout, T *gF, hipStream_t s) {
    const int64_t hw = (int64_t)H * W;
    if (dt == 0.0) {
        hipLaunchKernelGGL(k_axpy1<T>, dim3(div_up(2 * hw, 256)), dim3(256), 0, s, 2 * hw, gout, gF);
        CMAX_CHECK_LAUNCH();
        return 0;
    }
    const T sg = dt > 0 ? (T)1 : (T)-1, tau = (T)fabs(dt);
    const int grid = div_up(hw, 256);
    if (scheme == CMAX_SCHEME_BURGERS)
        hipLaunchKernelGGL((k_flow_step_adj<T, CMAX_SCHEME_BURGERS>), dim3(grid), dim3(256), 0, s, F, H, W, sg, tau, gout, gF);
    else
        hipLaunchKernelGGL((k_flow_step_adj<T, CMAX_SCHEME_UPWIND>), dim3(grid), dim3(256), 0, s, F, H, W, sg, tau, gout, gF);
    CMAX_CHECK_LAUNCH();
    return 0;
}

// V[t0] = F; backward steps -1/T down to 0, forward steps +1/T up to T-1 (flow_utils.py:68-79;
// the torch loop's stray extra backward iteration, 138-139, is not reproduced).  Step j of both time directions
// (and, with j = 1, the copy of F into bin t0) share a launch.
template <typename T>
int voxel_construct(const T *F, int Tn, int t0, int H, int W, int scheme, T *V, hipStream_t s) {
    const int64_t sz = 2 * (int64_t)H * W;
    const T tau = (T)(1.0 / (double)Tn);
    const int grid = div_up((int64_t)H * W, 256);
    const int nb = t0, nf = Tn - 1 - t0, nstep = nb > nf ? nb : nf;
    if (nstep == 0) {
        CMAX_CHECK_HIP(hipMemcpyAsync(V + (int64_t)t0 * sz, F, sz * sizeof(T), hipMemcpyDeviceToDevice, s));
        return 0;
    }
    for (int j = 1; j <= nstep; ++j) {
        StepJobs<T> jobs = {};
        int n = 0;
        if (j <= nb) {  // bin t0-j+1 -> t0-j
            jobs.src[n] = j == 1 ? F : V + (int64_t)(t0 - j + 1) * sz;
            jobs.dst[n] = V + (int64_t)(t0 - j) * sz;
            jobs.s[n++] = (T)-1;
        }
        if (j <= nf) {  // bin t0+j-1 -> t0+j
            jobs.src[n] = j == 1 ? F : V + (int64_t)(t0 + j - 1) * sz;
            jobs.dst[n] = V + (int64_t)(t0 + j) * sz;
            jobs.s[n++] = (T)1;
        }
        if (j == 1) {
            jobs.src[n] = F;
            jobs.dst[n] = V + (int64_t)t0 * sz;
            jobs.s[n++] = (T)0;
        }
        if (scheme == CMAX_SCHEME_BURGERS)
            hipLaunchKernelGGL((k_flow_step_jobs<T, CMAX_SCHEME_BURGERS>), dim3(grid, n), dim3(256), 0, s, jobs, H, W, tau);
        else
            hipLaunchKernelGGL((k_flow_step_jobs<T, CMAX_SCHEME_UPWIND>), dim3(grid, n), dim3(256), 0, s, jobs, H, W, tau);
        CMAX_CHECK_LAUNCH();
    }
    return 0;
}

// Adjoint sweep: the forward-time chain runs from bin T-2 down to t0, the backward-time chain from bin 1 up to t0,
// each step adding J^T gV[neighbour] into gV[i] (atomics); step j of both chains shares a launch, the chains are
// aligned so that they reach bin t0 in the same (last) launch.
template <typename T>
int voxel_construct_adj(const T *V, int Tn, int t0, int H, int W, int scheme, T *gV, T *gF, hipStream_t s) {
    const int64_t sz = 2 * (int64_t)H * W;
    const T tau = (T)(1.0 / (double)Tn);
    const int grid = div_up((int64_t)H * W, 256);
    const int nb = t0, nf = Tn - 1 - t0, nstep = nb > nf ? nb : nf;
    for (int j = nstep; j >= 1; --j) {  // j = distance of the step's INPUT bin from t0, outermost first
        StepJobs<T> jobs = {};
        int n = 0;
        if (j <= nf) {  // step bin t0+j-1 -> t0+j (dt > 0): gV[t0+j-1] += J^T gV[t0+j]
            const int i = t0 + j - 1;
            jobs.src[n] = V + (int64_t)i * sz;
            jobs.gout[n] = gV + (int64_t)(i + 1) * sz;
            jobs.dst[n] = gV + (int64_t)i * sz;
            jobs.s[n++] = (T)1;
        }
        if (j <= nb) {  // step bin t0-j+1 -> t0-j (dt < 0): gV[t0-j+1] += J^T gV[t0-j]
            const int i = t0 - j + 1;
            jobs.src[n] = V + (int64_t)i * sz;
            jobs.gout[n] = gV + (int64_t)(i - 1) * sz;
            jobs.dst[n] = gV + (int64_t)i * sz;
            jobs.s[n++] = (T)-1;
        }
        if (scheme == CMAX_SCHEME_BURGERS)
            hipLaunchKernelGGL((k_flow_step_adj_jobs<T, CMAX_SCHEME_BURGERS>), dim3(grid, n), dim3(256), 0, s, jobs, H, W, tau);
        else
            hipLaunchKernelGGL((k_flow_step_adj_jobs<T, CMAX_SCHEME_UPWIND>), dim3(grid, n), dim3(256), 0, s, jobs, H, W, tau);
        CMAX_CHECK_LAUNCH();
    }
    CMAX_CHECK_HIP(hipMemcpyAsync(gF, gV + (int64_t)t0 * sz, sz * sizeof(T), hipMemcpyDeviceToDevice, s));
    return 0;
}

// Voxel and its tangent along dF in one sweep (dual numbers, cmax_flow_dual.h); same launch structure as
// voxel_construct.
template <typename T>
int voxel_construct_tan(const T *F, const T *dF, int Tn, int t0, int H, int W, int scheme, T *V, T *dV, hipStream_t s) {
    const int64_t sz = 2 * (int64_t)H * W;
    const T tau = (T)(1.0 / (double)Tn);
    const int grid = div_up((int64_t)H * W, 256);
    const int nb = t0, nf = Tn - 1 - t0, nstep = nb > nf ? nb : nf;
    CMAX_CHECK_HIP(hipMemcpyAsync(V + (int64_t)t0 * sz, F, sz * sizeof(T), hipMemcpyDeviceToDevice, s));
    CMAX_CHECK_HIP(hipMemcpyAsync(dV + (int64_t)t0 * sz, dF, sz * sizeof(T), hipMemcpyDeviceToDevice, s));
    for (int j = 1; j <= nstep; ++j) {
        DualJobs<T> jobs = {};
        int n = 0;
        if (j <= nb) {
            const int64_t a = (int64_t)(t0 - j + 1) * sz, b = (int64_t)(t0 - j) * sz;
            jobs.src[n] = V + a; jobs.dsrc[n] = dV + a; jobs.dst[n] = V + b; jobs.ddst[n] = dV + b;
            jobs.s[n++] = (T)-1;
        }
        if (j <= nf) {
            const int64_t a = (int64_t)(t0 + j - 1) * sz, b = (int64_t)(t0 + j) * sz;
            jobs.src[n] = V + a; jobs.dsrc[n] = dV + a; jobs.dst[n] = V + b; jobs.ddst[n] = dV + b;
            jobs.s[n++] = (T)1;
        }
        if (scheme == CMAX_SCHEME_BURGERS)
            hipLaunchKernelGGL((k_flow_step_dual<T, CMAX_SCHEME_BURGERS>), dim3(grid, n), dim3(256), 0, s, jobs, H, W, tau);
        else
            hipLaunchKernelGGL((k_flow_step_dual<T, CMAX_SCHEME_UPWIND>), dim3(grid, n), dim3(256), 0, s, jobs, H, W, tau);
        CMAX_CHECK_LAUNCH();
    }
    return 0;
}

// Adjoint sweep on dual numbers: gV / dgV hold (dL/dV, its tangent) on entry and are clobbered; gF = dL/dF (the
// first-order gradient, as voxel_construct_adj gives it) and dgF = its directional derivative along dF.
template <typename T>
int voxel_construct_adj_tan(const T *V, const T *dV, int Tn, int t0, int H, int W, int scheme, T *gV, T *dgV, T *gF, T *dgF, hipStream_t s) {
    const int64_t sz = 2 * (int64_t)H * W;
    const T tau = (T)(1.0 / (double)Tn);
    const int grid = div_up((int64_t)H * W, 256);
    const int nb = t0, nf = Tn - 1 - t0, nstep = nb > nf ? nb : nf;
    for (int j = nstep; j >= 1; --j) {
        DualJobs<T> jobs = {};
        int n = 0;
        if (j <= nf) {
            const int64_t a = (int64_t)(t0 + j - 1) * sz, b = (int64_t)(t0 + j) * sz;
            jobs.src[n] = V + a; jobs.dsrc[n] = dV + a; jobs.gout[n] = gV + b; jobs.dgout[n] = dgV + b; jobs.dst[n] = gV + a; jobs.ddst[n] = dgV + a;
            jobs.s[n++] = (T)1;
        }
        if (j <= nb) {
            const int64_t a = (int64_t)(t0 - j + 1) * sz, b = (int64_t)(t0 - j) * sz;
            jobs.src[n] = V + a; jobs.dsrc[n] = dV + a; jobs.gout[n] = gV + b; jobs.dgout[n] = dgV + b; jobs.dst[n] = gV + a; jobs.ddst[n] = dgV + a;
            jobs.s[n++] = (T)-1;
        }
        if (scheme == CMAX_SCHEME_BURGERS)
            hipLaunchKernelGGL((k_flow_step_adj_dual<T, CMAX_SCHEME_BURGERS>), dim3(grid, n), dim3(256), 0, s, jobs, H, W, tau);
        else
            hipLaunchKernelGGL((k_flow_step_adj_dual<T, CMAX_SCHEME_UPWIND>), dim3(grid, n), dim3(256), 0, s, jobs, H, W, tau);
        CMAX_CHECK_LAUNCH();
    }
    if (gF) CMAX_CHECK_HIP(hipMemcpyAsync(gF, gV + (int64_t)t0 * sz, sz * sizeof(T), hipMemcpyDeviceToDevice, s));
    CMAX_CHECK_HIP(hipMemcpyAsync(dgF, dgV + (int64_t)t0 * sz, sz * sizeof(T), hipMemcpyDeviceToDevice, s));
    return 0;
}

template int voxel_construct<float>(const float *, int, int, int, int, int, float *, hipStream_t);
template int voxel_construct_adj<float>(const float *, int, int, int, int, int, float *, float *, hipStream_t);

}  // namespace cmax

using namespace cmax;

extern "C" {

int cmax_flow_step(const void *F, int dtype, int H, int W, double dt, int scheme, void *out, cmax_stream_t stream) {
    CMAX_REQUIRE(F && out && F != out && H > 0 && W > 0, "flow_step");
    CMAX_REQUIRE(scheme == CMAX_SCHEME_BURGERS || scheme == CMAX_SCHEME_UPWIND, "flow_step: scheme");
    if (dtype == CMAX_F32) return flow_step<float>((const float *)F, H, W, dt, scheme, (float *)out, (hipStream_t)stream);
    if (dtype == CMAX_F64) return flow_step<double>((const double *)F, H, W, dt, scheme, (double *)out, (hipStream_t)stream);
    set_error("flow_step: dtype");
    return CMAX_EINVAL;
}

int cmax_flow_step_adj(const void *F, int dtype, int H, int W, double dt, int scheme, const void *gout, void *gF,
                       cmax_stream_t stream) {
    CMAX_REQUIRE(F && gout && gF && H > 0 && W > 0, "flow_step_adj");
    CMAX_REQUIRE(scheme == CMAX_SCHEME_BURGERS || scheme == CMAX_SCHEME_UPWIND, "flow_step_adj: scheme");
    if (dtype == CMAX_F32) return flow_step_adj<float>((const float *)F, H, W, dt, scheme, (const float *)gout, (float *)gF, (hipStream_t)stream);
    if (dtype == CMAX_F64) return flow_step_adj<double>((const double *)F, H, W, dt, scheme, (const double *)gout, (double *)gF, (hipStream_t)stream);
    set_error("flow_step_adj: dtype");
    return CMAX_EINVAL;
}

int cmax_voxel_construct(const void *F, int dtype, int Tn, int t0, int H, int W, int scheme, void *V, cmax_stream_t stream) {
    CMAX_REQUIRE(F && V && Tn > 0 && t0 >= 0 && t0 < Tn && H > 0 && W > 0, "voxel_construct");
    CMAX_REQUIRE(scheme == CMAX_SCHEME_BURGERS || scheme == CMAX_SCHEME_UPWIND, "voxel_construct: scheme");
    if (dtype == CMAX_F32) return voxel_construct<float>((const float *)F, Tn, t0, H, W, scheme, (float *)V, (hipStream_t)stream);
    if (dtype == CMAX_F64) return voxel_construct<double>((const double *)F, Tn, t0, H, W, scheme, (double *)V, (hipStream_t)stream);
    set_error("voxel_construct: dtype");
    return CMAX_EINVAL;
}

int cmax_voxel_construct_adj(const void *V, int dtype, int Tn, int t0, int H, int W, int scheme, void *gV, void *gF,
                             cmax_stream_t stream) {
    CMAX_REQUIRE(V && gV && gF && Tn > 0 && t0 >= 0 && t0 < Tn && H > 0 && W > 0, "voxel_construct_adj");
    CMAX_REQUIRE(scheme == CMAX_SCHEME_BURGERS || scheme == CMAX_SCHEME_UPWIND, "voxel_construct_adj: scheme");
    if (dtype == CMAX_F32) return voxel_construct_adj<float>((const float *)V, Tn, t0, H, W, scheme, (float *)gV, (float *)gF, (hipStream_t)stream);
    if (dtype == CMAX_F64) return voxel_construct_adj<double>((const double *)V, Tn, t0, H, W, scheme, (double *)gV, (double *)gF, (hipStream_t)stream);
    set_error("voxel_construct_adj: dtype");
    return CMAX_EINVAL;
}

int cmax_voxel_construct_tan(const void *F, const void *dF, int dtype, int Tn, int t0, int H, int W, int scheme, void *V, void *dV,
                             cmax_stream_t stream) {
    CMAX_REQUIRE(F && dF && V && dV && Tn > 0 && t0 >= 0 && t0 < Tn && H > 0 && W > 0, "voxel_construct_tan");
    CMAX_REQUIRE(scheme == CMAX_SCHEME_BURGERS || scheme == CMAX_SCHEME_UPWIND, "voxel_construct_tan: scheme");
    if (dtype == CMAX_F32) return voxel_construct_tan<float>((const float *)F, (const float *)dF, Tn, t0, H, W, scheme, (float *)V, (float *)dV, (hipStream_t)stream);
    if (dtype == CMAX_F64) return voxel_construct_tan<double>((const double *)F, (const double *)dF, Tn, t0, H, W, scheme, (double *)V, (double *)dV, (hipStream_t)stream);
    set_error("voxel_construct_tan: dtype");
    return CMAX_EINVAL;
}

int cmax_voxel_construct_adj_tan(const void *V, const void *dV, int dtype, int Tn, int t0, int H, int W, int scheme, void *gV, void *dgV,
                                 void *gF, void *dgF, cmax_stream_t stream) {
    CMAX_REQUIRE(V && dV && gV && dgV && dgF && Tn > 0 && t0 >= 0 && t0 < Tn && H > 0 && W > 0, "voxel_construct_adj_tan");
    CMAX_REQUIRE(scheme == CMAX_SCHEME_BURGERS || scheme == CMAX_SCHEME_UPWIND, "voxel_construct_adj_tan: scheme");
    if (dtype == CMAX_F32)
        return voxel_construct_adj_tan<float>((const float *)V, (const float *)dV, Tn, t0, H, W, scheme, (float *)gV, (float *)dgV, (float *)gF, (float *)dgF, (hipStream_t)stream);
    if (dtype == CMAX_F64)
        return voxel_construct_adj_tan<double>((const double *)V, (const double *)dV, Tn, t0, H, W, scheme, (double *)gV, (double *)dgV, (double *)gF, (double *)dgF, (hipStream_t)stream);
    set_error("voxel_construct_adj_tan: dtype");
    return CMAX_EINVAL;
}

int cmax_patch_to_dense(const void *motion, int dtype, int ph, int pw, int pad_h, int pad_w, int sw_h, int sw_w, int H, int W,
                        int adjoint, void *out, cmax_stream_t stream) {
    CMAX_REQUIRE(motion && out && ph > 0 && pw > 0 && pad_h >= 0 && pad_w >= 0 && sw_h > 0 && sw_w > 0 && H > 0 && W > 0, "patch_to_dense");
    CMAX_REQUIRE((ph + 2 * pad_h) * sw_h >= H && (pw + 2 * pad_w) * sw_w >= W, "patch_to_dense: up-sampled grid smaller than the image");
    hipStream_t s = (hipStream_t)stream;
    if (!adjoint) {
        const int grid = div_up(2 * (int64_t)H * W, 256);
        if (dtype == CMAX_F32) hipLaunchKernelGGL(k_patch_to_dense<float>, dim3(grid), dim3(256), 0, s, (const float *)motion, ph, pw, pad_h, pad_w, sw_h, sw_w, H, W, (float *)out);
        else if (dtype == CMAX_F64) hipLaunchKernelGGL(k_patch_to_dense<double>, dim3(grid), dim3(256), 0, s, (const double *)motion, ph, pw, pad_h, pad_w, sw_h, sw_w, H, W, (double *)out);
        else { set_error("patch_to_dense: dtype"); return CMAX_EINVAL; }
    } else {
        const int grid = 2 * ph * pw;
        if (dtype == CMAX_F32) hipLaunchKernelGGL(k_patch_to_dense_adj<float>, dim3(grid), dim3(256), 0, s, (const float *)motion, ph, pw, pad_h, pad_w, sw_h, sw_w, H, W, (float *)out);
        else if (dtype == CMAX_F64) hipLaunchKernelGGL(k_patch_to_dense_adj<double>, dim3(grid), dim3(256), 0, s, (const double *)motion, ph, pw, pad_h, pad_w, sw_h, sw_w, H, W, (double *)out);
        else { set_error("patch_to_dense: dtype"); return CMAX_EINVAL; }
    }
    CMAX_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
