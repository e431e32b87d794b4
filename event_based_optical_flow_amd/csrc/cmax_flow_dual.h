// Second-order machinery of the time-aware flow: the propagation step and its adjoint evaluated on DUAL numbers
// (value, directional derivative).  Needed for the exact Hessian-vector product of an objective that sees the
// patch flow through the Burgers / upwind voxel (what torch.autograd.functional.vhp differentiates in the
// reference, src/solver/scipy_autograd/torch_wrapper.py:51-73 over src/utils/flow_utils.py:99-161, 439-493, 567-639):
//   forward   (V_i, dV_i) -> (V_i+-1, dV_i+-1)            : voxel and its tangent along dF in one sweep
//   adjoint   lambda_i += J_i^T lambda_next,  dlambda_i += J_i^T dlambda_next + (dJ_i[dV_i])^T lambda_next
// Both come out of the SAME arithmetic as the first-order kernels of cmax_flow.hip, run on Dual<T>; sign(), the
// selectors of maximum / minimum and their tie rule are piecewise constant, i.e. carry no derivative -- exactly
// torch's convention.
#pragma once
#include "cmax_common.h"

namespace cmax {

template <typename T>
struct Dual {
    T v, d;
    __device__ __forceinline__ Dual() : v((T)0), d((T)0) {}
    __device__ __forceinline__ Dual(T v_, T d_) : v(v_), d(d_) {}
};
template <typename T>
__device__ __forceinline__ Dual<T> operator+(Dual<T> a, Dual<T> b) { return Dual<T>(a.v + b.v, a.d + b.d); }
template <typename T>
__device__ __forceinline__ Dual<T> operator-(Dual<T> a, Dual<T> b) { return Dual<T>(a.v - b.v, a.d - b.d); }
template <typename T>
__device__ __forceinline__ Dual<T> operator-(Dual<T> a) { return Dual<T>(-a.v, -a.d); }
template <typename T>
__device__ __forceinline__ Dual<T> operator*(Dual<T> a, Dual<T> b) { return Dual<T>(a.v * b.v, a.v * b.d + a.d * b.v); }
template <typename T>
__device__ __forceinline__ Dual<T> operator*(Dual<T> a, T b) { return Dual<T>(a.v * b, a.d * b); }
template <typename T>
__device__ __forceinline__ Dual<T> operator*(T a, Dual<T> b) { return Dual<T>(a * b.v, a * b.d); }

// piecewise-constant helpers on the value
template <typename T>
__device__ __forceinline__ T d_sgn(Dual<T> a) { return (T)((a.v > (T)0) - (a.v < (T)0)); }
template <typename T>
__device__ __forceinline__ T d_pos(Dual<T> a) { return a.v > (T)0 ? (T)1 : (T)0; }  // max(sign(a), 0)
template <typename T>
__device__ __forceinline__ T d_neg(Dual<T> a) { return a.v < (T)0 ? (T)-1 : (T)0; }  // min(sign(a), 0)
template <typename T>
__device__ __forceinline__ T d_dmax0(Dual<T> a) { return a.v > (T)0 ? (T)1 : (a.v == (T)0 ? (T)0.5 : (T)0); }
template <typename T>
__device__ __forceinline__ T d_dmin0(Dual<T> a) { return a.v < (T)0 ? (T)1 : (a.v == (T)0 ? (T)0.5 : (T)0); }
// maximum / minimum with 0 and |.|: derivative = selector * tangent
template <typename T>
__device__ __forceinline__ Dual<T> d_max0(Dual<T> a) { return Dual<T>(a.v > (T)0 ? a.v : (T)0, d_dmax0(a) * a.d); }
template <typename T>
__device__ __forceinline__ Dual<T> d_min0(Dual<T> a) { return Dual<T>(a.v < (T)0 ? a.v : (T)0, d_dmin0(a) * a.d); }
template <typename T>
__device__ __forceinline__ Dual<T> d_abs(Dual<T> a) { return Dual<T>(a.v < (T)0 ? -a.v : a.v, d_sgn(a) * a.d); }

template <typename T>
struct DualJobs {  // up to 3 jobs per launch (blockIdx.y): the two time directions (+ the copy of bin t0 in the first forward launch)
    const T *src[3], *dsrc[3];    // V_i, dV_i
    T *dst[3], *ddst[3];          // forward: V_next, dV_next; adjoint: lambda_i, dlambda_i (accumulated)
    const T *gout[3], *dgout[3];  // adjoint: lambda_next, dlambda_next
    T s[3];                       // +-1: time direction of the step; 0 (forward only): plain copy
};

// forward step on dual numbers: same formulas as flow_step_pixel (flow_utils.py:582-639 / 459-492)
template <typename T, int SCHEME>
__global__ void __launch_bounds__(256) k_flow_step_dual(DualJobs<T> jobs, int H, int W, T tau) {
    using N = Dual<T>;
    const int y = blockIdx.y;
    const T *F = jobs.src[y], *dF = jobs.dsrc[y];
    const T s = jobs.s[y];
    const int64_t hw = (int64_t)H * W;
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hw) return;
    if (s == (T)0) {  // copy job: bin t0 holds (F, dF) themselves
        jobs.dst[y][p] = F[p];
        jobs.dst[y][hw + p] = F[hw + p];
        jobs.ddst[y][p] = dF[p];
        jobs.ddst[y][hw + p] = dF[hw + p];
        return;
    }
    const int i = (int)(p / W), j = (int)(p % W);
    auto U = [&](int r, int c) { return N(s * F[(int64_t)r * W + c], s * dF[(int64_t)r * W + c]); };
    auto V = [&](int r, int c) { return N(s * F[hw + (int64_t)r * W + c], s * dF[hw + (int64_t)r * W + c]); };
    const N u = U(i, j), v = V(i, j), zero;
    N nu, nv;
    if (SCHEME == CMAX_SCHEME_BURGERS) {
        const int ip = i + 1 < H ? i + 1 : H - 1, im = i > 0 ? i - 1 : 0;
        const int jp = j + 1 < W ? j + 1 : W - 1, jm = j > 0 ? j - 1 : 0;
        const N uf = U(ip, j), ub = U(im, j), vf = V(i, jp), vb = V(i, jm);
        const N bu = ((u * u) * d_sgn(u) - (ub * ub) * d_pos(ub) - (uf * uf) * d_neg(uf)) * (T)0.5;
        const N bv = ((v * v) * d_sgn(v) - (vb * vb) * d_pos(vb) - (vf * vf) * d_neg(vf)) * (T)0.5;
        const N u_dy_back = j > 0 ? u - U(i, j - 1) : zero, u_dy_forw = j + 1 < W ? U(i, j + 1) - u : zero;
        const N v_dx_back = i > 0 ? v - V(i - 1, j) : zero, v_dx_forw = i + 1 < H ? V(i + 1, j) - v : zero;
        nu = u - (d_max0(v) * u_dy_back + d_min0(v) * u_dy_forw + bu) * tau;
        nv = v - (d_max0(u) * v_dx_back + d_min0(u) * v_dx_forw + bv) * tau;
    } else {
        const N u_dx_back = i > 0 ? u - U(i - 1, j) : zero, u_dx_forw = i + 1 < H ? U(i + 1, j) - u : zero;
        const N u_dy_back = j > 0 ? u - U(i, j - 1) : zero, u_dy_forw = j + 1 < W ? U(i, j + 1) - u : zero;
        const N v_dx_back = i > 0 ? v - V(i - 1, j) : zero, v_dx_forw = i + 1 < H ? V(i + 1, j) - v : zero;
        const N v_dy_back = j > 0 ? v - V(i, j - 1) : zero, v_dy_forw = j + 1 < W ? V(i, j + 1) - v : zero;
        nu = u - (d_max0(u) * u_dx_back + d_min0(u) * u_dx_forw + d_max0(v) * u_dy_back + d_min0(v) * u_dy_forw) * tau;
        nv = v - (d_max0(u) * v_dx_back + d_min0(u) * v_dx_forw + d_max0(v) * v_dy_back + d_min0(v) * v_dy_forw) * tau;
    }
    jobs.dst[y][p] = nu.v * s;
    jobs.dst[y][hw + p] = nv.v * s;
    jobs.ddst[y][p] = nu.d * s;
    jobs.ddst[y][hw + p] = nv.d * s;
}

// Tile geometry of the atomic-free adjoint steps (k_flow_step_adj_tiled in cmax_flow.hip and the dual version below):
// one workgroup owns a 16 x 32 tile of the destination, evaluates the scatter of the tile's pixels and of the ring
// around it, keeps what lands inside the tile in LDS and adds it to the destination with plain read-modify-writes.
constexpr int kAdjTileH = 16, kAdjTileW = 32, kAdjThreads = 512;
constexpr int kAdjDualThreads = 512;  // dual numbers: twice the registers; two workgroups per CU run the two time directions side by side

// adjoint step on dual numbers, scatter form like flow_step_adj_core: F -> (V_i, dV_i), upstream (lambda, dlambda) = (gnu, gnv).
// U / V: accessors of the signed dual field; GU / GV(row, col, value): sinks of the contributions of source pixel (i, j).
template <typename T, int SCHEME, typename AU, typename AV, typename SU, typename SV>
__device__ __forceinline__ void flow_step_adj_dual_core(AU U, AV V, SU GU, SV GV, int i, int j, int H, int W, T tau, Dual<T> gnu, Dual<T> gnv) {
    using N = Dual<T>;
    const N u = U(i, j), v = V(i, j), zero;
    const T mt = -tau;
    if (SCHEME == CMAX_SCHEME_BURGERS) {
        const int ip = i + 1 < H ? i + 1 : H - 1, im = i > 0 ? i - 1 : 0;
        const int jp = j + 1 < W ? j + 1 : W - 1, jm = j > 0 ? j - 1 : 0;
        const N uf = U(ip, j), ub = U(im, j), vf = V(i, jp), vb = V(i, jm);
        const N u_dy_back = j > 0 ? u - U(i, j - 1) : zero, u_dy_forw = j + 1 < W ? U(i, j + 1) - u : zero;
        const N v_dx_back = i > 0 ? v - V(i - 1, j) : zero, v_dx_forw = i + 1 < H ? V(i + 1, j) - v : zero;
        const N one((T)1, (T)0);
        // channel u
        N self_u = gnu * (one - d_abs(u) * tau);  // d(u|u|/2)/du = |u|
        GU(im, j, gnu * (-d_max0(ub)) * mt);
        GU(ip, j, gnu * d_min0(uf) * mt);
        N self_v = gnu * (u_dy_back * d_dmax0(v) + u_dy_forw * d_dmin0(v)) * mt;
        const N mvp = d_max0(v), mvn = d_min0(v);
        if (j > 0) {
            self_u = self_u + gnu * mvp * mt;
            GU(i, j - 1, -(gnu * mvp * mt));
        }
        if (j + 1 < W) {
            GU(i, j + 1, gnu * mvn * mt);
            self_u = self_u - gnu * mvn * mt;
        }
        // channel v
        self_v = self_v + gnv * (one - d_abs(v) * tau);
        GV(i, jm, gnv * (-d_max0(vb)) * mt);
        GV(i, jp, gnv * d_min0(vf) * mt);
        self_u = self_u + gnv * (v_dx_back * d_dmax0(u) + v_dx_forw * d_dmin0(u)) * mt;
        const N mup = d_max0(u), mun = d_min0(u);
        if (i > 0) {
            self_v = self_v + gnv * mup * mt;
            GV(i - 1, j, -(gnv * mup * mt));
        }
        if (i + 1 < H) {
            GV(i + 1, j, gnv * mun * mt);
            self_v = self_v - gnv * mun * mt;
        }
        GU(i, j, self_u);
        GV(i, j, self_v);
    } else {
        const N mup = d_max0(u), mun = d_min0(u), mvp = d_max0(v), mvn = d_min0(v);
        N self[2];
        for (int c = 0; c < 2; ++c) {
            const N g = c == 0 ? gnu : gnv;
            auto Cc = [&](int r, int q) { return c == 0 ? U(r, q) : V(r, q); };
            auto GC = [&](int r, int q, N val) {
                if (c == 0) GU(r, q, val);
                else GV(r, q, val);
            };
            const N f = c == 0 ? u : v;
            const N dx_back = i > 0 ? f - Cc(i - 1, j) : zero, dx_forw = i + 1 < H ? Cc(i + 1, j) - f : zero;
            const N dy_back = j > 0 ? f - Cc(i, j - 1) : zero, dy_forw = j + 1 < W ? Cc(i, j + 1) - f : zero;
            self[c] = self[c] + g;
            self[0] = self[0] + g * (dx_back * d_dmax0(u) + dx_forw * d_dmin0(u)) * mt;
            self[1] = self[1] + g * (dy_back * d_dmax0(v) + dy_forw * d_dmin0(v)) * mt;
            if (i > 0) {
                self[c] = self[c] + g * mup * mt;
                GC(i - 1, j, -(g * mup * mt));
            }
            if (i + 1 < H) {
                GC(i + 1, j, g * mun * mt);
                self[c] = self[c] - g * mun * mt;
            }
            if (j > 0) {
                self[c] = self[c] + g * mvp * mt;
                GC(i, j - 1, -(g * mvp * mt));
            }
            if (j + 1 < W) {
                GC(i, j + 1, g * mvn * mt);
                self[c] = self[c] - g * mvn * mt;
            }
        }
        GU(i, j, self[0]);
        GV(i, j, self[1]);
    }
}

// The jobs of a launch (both time directions) run one after the other in the workgroup.
// DET (deterministic handles, cmax_set_deterministic, through the patch plan): no LDS atomics -- every destination pixel of the
// tile evaluates the scatter of its (at most five) source pixels itself and keeps what lands on it, in a fixed order: ~4x the
// arithmetic of a kernel that is launch-bound, bit-identical results from run to run.
template <typename T, int SCHEME, bool DET = false>
__global__ void __launch_bounds__(kAdjDualThreads) k_flow_step_adj_dual(DualJobs<T> jobs, int n_jobs, int H, int W, T tau) {
    using N = Dual<T>;
    __shared__ T acc[4][DET ? 1 : kAdjTileH * kAdjTileW];  // lambda u, lambda v, dlambda u, dlambda v
    const int64_t hw = (int64_t)H * W;
    const int tiles_w = (W + kAdjTileW - 1) / kAdjTileW;
    const int tr = blockIdx.x / tiles_w, tc = blockIdx.x - tr * tiles_w;
    const int R0 = tr * kAdjTileH, C0 = tc * kAdjTileW;
    constexpr int RH = kAdjTileH + 2, RW = kAdjTileW + 2;
    // gridDim.y > 1: one job per workgroup (different destinations); else the jobs one after the other
    const int y_first = gridDim.y > 1 ? (int)blockIdx.y : 0, y_last = gridDim.y > 1 ? (int)blockIdx.y + 1 : n_jobs;
    for (int y = y_first; y < y_last; ++y) {
        const T *F = jobs.src[y], *dF = jobs.dsrc[y], *gout = jobs.gout[y], *dgout = jobs.dgout[y];
        T *gF = jobs.dst[y], *dgF = jobs.ddst[y];
        const T s = jobs.s[y];
        auto U = [&](int r, int c) { return N(s * F[(int64_t)r * W + c], s * dF[(int64_t)r * W + c]); };
        auto V = [&](int r, int c) { return N(s * F[hw + (int64_t)r * W + c], s * dF[hw + (int64_t)r * W + c]); };
        if (DET) {
            for (int q = threadIdx.x; q < kAdjTileH * kAdjTileW; q += kAdjDualThreads) {
                const int a = q / kAdjTileW, b = q - a * kAdjTileW, di = R0 + a, dj = C0 + b;
                if (di >= H || dj >= W) continue;
                N au, av;
                auto GU = [&](int r, int c, N val) { if (r == di && c == dj) au = au + val; };
                auto GV = [&](int r, int c, N val) { if (r == di && c == dj) av = av + val; };
#pragma unroll
                for (int k = 0; k < 5; ++k) {  // sources: above, left, the pixel itself, right, below
                    const int i = di + (k == 0 ? -1 : (k == 4 ? 1 : 0)), j = dj + (k == 1 ? -1 : (k == 3 ? 1 : 0));
                    if ((unsigned)i >= (unsigned)H || (unsigned)j >= (unsigned)W) continue;
                    const int64_t p = (int64_t)i * W + j;
                    flow_step_adj_dual_core<T, SCHEME>(U, V, GU, GV, i, j, H, W, tau, N(gout[p], dgout[p]), N(gout[hw + p], dgout[hw + p]));
                }
                const int64_t p = (int64_t)di * W + dj;
                gF[p] += au.v;
                gF[hw + p] += av.v;
                dgF[p] += au.d;
                dgF[hw + p] += av.d;
            }
            continue;
        }
        for (int q = threadIdx.x; q < kAdjTileH * kAdjTileW; q += kAdjDualThreads) acc[0][q] = acc[1][q] = acc[2][q] = acc[3][q] = (T)0;
        __syncthreads();
        for (int q = threadIdx.x; q < RH * RW; q += kAdjDualThreads) {
            const int a = q / RW, b = q - a * RW, i = R0 - 1 + a, j = C0 - 1 + b;
            if ((unsigned)i >= (unsigned)H || (unsigned)j >= (unsigned)W) continue;
            const int64_t p = (int64_t)i * W + j;
            auto GU = [&](int r, int c, N val) {
                const int lr = r - R0, lc = c - C0;
                if ((unsigned)lr < (unsigned)kAdjTileH && (unsigned)lc < (unsigned)kAdjTileW) {
                    atomic_add(&acc[0][lr * kAdjTileW + lc], val.v);
                    atomic_add(&acc[2][lr * kAdjTileW + lc], val.d);
                }
            };
            auto GV = [&](int r, int c, N val) {
                const int lr = r - R0, lc = c - C0;
                if ((unsigned)lr < (unsigned)kAdjTileH && (unsigned)lc < (unsigned)kAdjTileW) {
                    atomic_add(&acc[1][lr * kAdjTileW + lc], val.v);
                    atomic_add(&acc[3][lr * kAdjTileW + lc], val.d);
                }
            };
            flow_step_adj_dual_core<T, SCHEME>(U, V, GU, GV, i, j, H, W, tau, N(gout[p], dgout[p]), N(gout[hw + p], dgout[hw + p]));
        }
        __syncthreads();
        for (int q = threadIdx.x; q < kAdjTileH * kAdjTileW; q += kAdjDualThreads) {
            const int a = q / kAdjTileW, b = q - a * kAdjTileW, i = R0 + a, j = C0 + b;
            if (i < H && j < W) {
                const int64_t p = (int64_t)i * W + j;
                gF[p] += acc[0][q];
                gF[hw + p] += acc[1][q];
                dgF[p] += acc[2][q];
                dgF[hw + p] += acc[3][q];
            }
        }
        __syncthreads();
    }
}

}  // namespace cmax
