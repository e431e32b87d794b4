// Time-aware flow: explicit Burgers / upwind propagation of a dense flow field into a voxel of
// time bins, and the adjoints.  Restates src/utils/flow_utils.py:99-161 (voxel), 567-639 (Burgers),
// 439-493 (upwind) as one gather kernel per step: each thread owns one pixel and reads its
// 4-neighbourhood (the reference materialises ~30 intermediate ATen tensors per step).
#include <atomic>

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

// One pixel of one propagation step.  U / V: accessors of the SIGNED input field s * F at (row, col), only asked for
// in-image neighbours of (i, j); returns the signed new values (the caller multiplies by s again).
template <typename T, int SCHEME, typename AU, typename AV>
__device__ __forceinline__ void flow_step_core(AU U, AV V, int i, int j, int H, int W, T tau, T &nu, T &nv) {
    const T u = U(i, j), v = V(i, j);
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
}

template <typename T, int SCHEME>
__device__ __forceinline__ void flow_step_pixel(const T *__restrict__ F, int H, int W, T s, T tau, T *__restrict__ out) {
    const int64_t hw = (int64_t)H * W;
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hw) return;
    const int i = (int)(p / W), j = (int)(p % W);
    auto U = [&](int r, int c) { return s * F[(int64_t)r * W + c]; };
    auto V = [&](int r, int c) { return s * F[hw + (int64_t)r * W + c]; };
    T nu, nv;
    flow_step_core<T, SCHEME>(U, V, i, j, H, W, tau, nu, nv);
    out[p] = nu * s;
    out[hw + p] = nv * s;
}

// The whole voxel V[T,2,H,W] from F = V[t0] in ONE launch.  The step is a radius-1 stencil, so a workgroup can run a
// chain of S steps on a 16 x 32 output tile from a (16 + 2S) x (32 + 2S) input patch held in LDS, the valid region shrinking by
// one ring per step (1.5x redundant arithmetic at S = 5 -- nothing next to the 4-9 dependent launches it replaces:
// every step of the per-step form is a ~4.5 us dependent launch on a 260 x 346 field).  Both time directions run in the
// same workgroup, one after the other, from the same patch.  Dynamic LDS: 2 buffers x 2 channels x patch.
constexpr int kVoxTileH = 16, kVoxTileW = 32;  // 260 x 346: 17 x 11 = 187 workgroups, one round on 256 CUs
constexpr int kVoxThreads = 1024;                // about one patch cell per thread and step: the chain is a sequence of barriers

template <typename T, int SCHEME>
__global__ void __launch_bounds__(kVoxThreads) k_voxel_chain_tiled(const T *__restrict__ F, int Tn, int t0, int H, int W, T tau, int S, T *__restrict__ Vox,
                                                                    float *__restrict__ Vox32) {
    extern __shared__ unsigned char s_raw[];
    T *buf = reinterpret_cast<T *>(s_raw);
    const int ph = kVoxTileH + 2 * S, pw = kVoxTileW + 2 * S, cells = ph * pw;  // the patch
    const int64_t hw = (int64_t)H * W, sz = 2 * hw;
    const int tiles_w = (W + kVoxTileW - 1) / kVoxTileW;
    const int tr = blockIdx.x / tiles_w, tc = blockIdx.x - tr * tiles_w;
    const int R0 = tr * kVoxTileH - S, C0 = tc * kVoxTileW - S;  // origin of the patch
    const int nb = t0, nf = Tn - 1 - t0;
    auto in_tile = [&](int a, int b) { return a >= S && a < S + kVoxTileH && b >= S && b < S + kVoxTileW; };
    for (int dir = 0; dir < 2; ++dir) {
        const int nstep = dir == 0 ? nb : nf;
        if (dir == 1 && nstep == 0) break;  // (dir 0 still loads the patch: it writes bin t0)
        const T s = dir == 0 ? (T)-1 : (T)1;
        T *cur = buf, *nxt = buf + 2 * cells;
        __syncthreads();  // the previous direction is done with the buffers
        for (int q = threadIdx.x; q < cells; q += kVoxThreads) {
            const int a = q / pw, b = q - a * pw, r = R0 + a, c = C0 + b;
            const bool in = (unsigned)r < (unsigned)H && (unsigned)c < (unsigned)W;
            cur[q] = in ? F[(int64_t)r * W + c] : (T)0;
            cur[cells + q] = in ? F[hw + (int64_t)r * W + c] : (T)0;
            // bin t0 is a copy of F: written once, by the tile's own pixels
            if (dir == 0 && in && in_tile(a, b)) {
                Vox[(int64_t)t0 * sz + (int64_t)r * W + c] = cur[q];
                Vox[(int64_t)t0 * sz + hw + (int64_t)r * W + c] = cur[cells + q];
                if (Vox32) {  // the fp32 motion of the fused objective, written on the way
                    Vox32[(int64_t)t0 * sz + (int64_t)r * W + c] = (float)cur[q];
                    Vox32[(int64_t)t0 * sz + hw + (int64_t)r * W + c] = (float)cur[cells + q];
                }
            }
        }
        __syncthreads();
        for (int k = 1; k <= nstep; ++k) {
            const int bin = dir == 0 ? t0 - k : t0 + k;
            const int sh = ph - 2 * k, sw = pw - 2 * k;  // valid after step k: local rows [k, ph - k), columns [k, pw - k)
            for (int q = threadIdx.x; q < sh * sw; q += kVoxThreads) {
                const int a = k + q / sw, b = k + (q - (q / sw) * sw), r = R0 + a, c = C0 + b;
                if ((unsigned)r >= (unsigned)H || (unsigned)c >= (unsigned)W) continue;
                auto U = [&](int rr, int cc) { return s * cur[(rr - R0) * pw + (cc - C0)]; };
                auto V = [&](int rr, int cc) { return s * cur[cells + (rr - R0) * pw + (cc - C0)]; };
                T nu, nv;
                flow_step_core<T, SCHEME>(U, V, r, c, H, W, tau, nu, nv);
                nu *= s;
                nv *= s;
                nxt[a * pw + b] = nu;
                nxt[cells + a * pw + b] = nv;
                if (in_tile(a, b)) {
                    Vox[(int64_t)bin * sz + (int64_t)r * W + c] = nu;
                    Vox[(int64_t)bin * sz + hw + (int64_t)r * W + c] = nv;
                    if (Vox32) {
                        Vox32[(int64_t)bin * sz + (int64_t)r * W + c] = (float)nu;
                        Vox32[(int64_t)bin * sz + hw + (int64_t)r * W + c] = (float)nv;
                    }
                }
            }
            __syncthreads();
            T *t = cur;
            cur = nxt;
            nxt = t;
        }
    }
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
// U / V: accessors of the signed field s * F (in-image neighbours of (i, j) only); GU / GV(row, col, value): sinks of the
// gradient contributions; gnu / gnv: the incoming gradient of output pixel (i, j).
template <typename T, int SCHEME, typename AU, typename AV, typename SU, typename SV>
__device__ __forceinline__ void flow_step_adj_core(AU U, AV V, SU GU, SV GV, int i, int j, int H, int W, T tau, T gnu, T gnv) {
    const T u = U(i, j), v = V(i, j);
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
    flow_step_adj_core<T, SCHEME>(U, V, GU, GV, i, j, H, W, tau, gout[p], gout[hw + p]);
}

// The adjoint step without global atomics.  One workgroup owns a 16 x 32 tile of the destination: it evaluates the
// scatter of every source pixel of the tile and of the one-pixel ring around it (the stencil has radius 1), keeps the
// contributions that land inside its tile in an LDS accumulator (ds_add_f32 / ds_add_f64) and adds the tile to the
// destination with plain read-modify-writes.  The scatter form with global fp64 atomics cost 14.4 us per step on a
// 260 x 346 field (1.8M atomics); the jobs of a launch (both time directions) run one after the other in the workgroup,
// so two jobs may share their destination (the last step: both chains arrive at bin t0).

// DET (deterministic handles, through the patch plan): no LDS atomics -- every destination pixel evaluates the scatter of its
// (at most five) source pixels itself and keeps what lands on it, in a fixed order (see k_flow_step_adj_dual).
template <typename T, int SCHEME, bool DET = false>
__global__ void __launch_bounds__(kAdjThreads) k_flow_step_adj_tiled(StepJobs<T> jobs, int n_jobs, int H, int W, T tau) {
    __shared__ T acc[2][DET ? 1 : kAdjTileH * kAdjTileW];
    const int64_t hw = (int64_t)H * W;
    const int tiles_w = (W + kAdjTileW - 1) / kAdjTileW;
    const int tr = blockIdx.x / tiles_w, tc = blockIdx.x - tr * tiles_w;
    const int R0 = tr * kAdjTileH, C0 = tc * kAdjTileW;
    constexpr int RH = kAdjTileH + 2, RW = kAdjTileW + 2;
    // gridDim.y > 1: one job per workgroup (their destinations differ); else the jobs one after the other
    const int y_first = gridDim.y > 1 ? (int)blockIdx.y : 0, y_last = gridDim.y > 1 ? (int)blockIdx.y + 1 : n_jobs;
    for (int y = y_first; y < y_last; ++y) {
        const T *__restrict__ F = jobs.src[y];
        const T *__restrict__ gout = jobs.gout[y];
        T *__restrict__ dst = jobs.dst[y];
        const T s = jobs.s[y];
        auto U = [&](int r, int c) { return s * F[(int64_t)r * W + c]; };
        auto V = [&](int r, int c) { return s * F[hw + (int64_t)r * W + c]; };
        if (DET) {
            for (int q = threadIdx.x; q < kAdjTileH * kAdjTileW; q += kAdjThreads) {
                const int a = q / kAdjTileW, b = q - a * kAdjTileW, di = R0 + a, dj = C0 + b;
                if (di >= H || dj >= W) continue;
                T au = (T)0, av = (T)0;
                auto GU = [&](int r, int c, T val) { if (r == di && c == dj) au += val; };
                auto GV = [&](int r, int c, T val) { if (r == di && c == dj) av += val; };
#pragma unroll
                for (int k = 0; k < 5; ++k) {  // sources: above, left, the pixel itself, right, below
                    const int i = di + (k == 0 ? -1 : (k == 4 ? 1 : 0)), j = dj + (k == 1 ? -1 : (k == 3 ? 1 : 0));
                    if ((unsigned)i >= (unsigned)H || (unsigned)j >= (unsigned)W) continue;
                    const int64_t p = (int64_t)i * W + j;
                    flow_step_adj_core<T, SCHEME>(U, V, GU, GV, i, j, H, W, tau, gout[p], gout[hw + p]);
                }
                const int64_t p = (int64_t)di * W + dj;
                dst[p] += au;
                dst[hw + p] += av;
            }
            continue;
        }
        for (int q = threadIdx.x; q < kAdjTileH * kAdjTileW; q += kAdjThreads) {
            acc[0][q] = (T)0;
            acc[1][q] = (T)0;
        }
        __syncthreads();
        for (int q = threadIdx.x; q < RH * RW; q += kAdjThreads) {
            const int a = q / RW, b = q - a * RW, i = R0 - 1 + a, j = C0 - 1 + b;
            if ((unsigned)i >= (unsigned)H || (unsigned)j >= (unsigned)W) continue;
            auto GU = [&](int r, int c, T val) {
                const int lr = r - R0, lc = c - C0;
                if ((unsigned)lr < (unsigned)kAdjTileH && (unsigned)lc < (unsigned)kAdjTileW) atomic_add(&acc[0][lr * kAdjTileW + lc], val);
            };
            auto GV = [&](int r, int c, T val) {
                const int lr = r - R0, lc = c - C0;
                if ((unsigned)lr < (unsigned)kAdjTileH && (unsigned)lc < (unsigned)kAdjTileW) atomic_add(&acc[1][lr * kAdjTileW + lc], val);
            };
            const int64_t p = (int64_t)i * W + j;
            flow_step_adj_core<T, SCHEME>(U, V, GU, GV, i, j, H, W, tau, gout[p], gout[hw + p]);
        }
        __syncthreads();
        for (int q = threadIdx.x; q < kAdjTileH * kAdjTileW; q += kAdjThreads) {
            const int a = q / kAdjTileW, b = q - a * kAdjTileW, i = R0 + a, j = C0 + b;
            if (i < H && j < W) {
                const int64_t p = (int64_t)i * W + j;
                dst[p] += acc[0][q];
                dst[hw + p] += acc[1][q];
            }
        }
        __syncthreads();
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

// cmax_set_leaf_deterministic: the stand-alone adjoint entries take the order-free step kernels (no handle to read a mode from)
static std::atomic<int> g_leaf_det{0};

template <typename T>
static int flow_step_adj(const T *F, int H, int W, double dt, int scheme, const T *gout, T *gF, hipStream_t s) {
    const int64_t hw = (int64_t)H * W;
    if (dt == 0.0) {
        hipLaunchKernelGGL(k_axpy1<T>, dim3(div_up(2 * hw, 256)), dim3(256), 0, s, 2 * hw, gout, gF);
        CMAX_CHECK_LAUNCH();
        return 0;
    }
    const T sg = dt > 0 ? (T)1 : (T)-1, tau = (T)fabs(dt);
    if (g_leaf_det.load(std::memory_order_relaxed)) {  // every destination pixel evaluates the scatter of its five sources itself: no atomics
        StepJobs<T> jobs = {};
        jobs.src[0] = F;
        jobs.gout[0] = gout;
        jobs.dst[0] = gF;
        jobs.s[0] = sg;
        const dim3 agrid(div_up(H, kAdjTileH) * div_up(W, kAdjTileW), 1);
        if (scheme == CMAX_SCHEME_BURGERS)
            hipLaunchKernelGGL((k_flow_step_adj_tiled<T, CMAX_SCHEME_BURGERS, true>), agrid, dim3(kAdjThreads), 0, s, jobs, 1, H, W, tau);
        else
            hipLaunchKernelGGL((k_flow_step_adj_tiled<T, CMAX_SCHEME_UPWIND, true>), agrid, dim3(kAdjThreads), 0, s, jobs, 1, H, W, tau);
        CMAX_CHECK_LAUNCH();
        return 0;
    }
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
int voxel_construct(const T *F, int Tn, int t0, int H, int W, int scheme, T *V, hipStream_t s, float *V32 = nullptr, bool *wrote_v32 = nullptr) {
    if (wrote_v32) *wrote_v32 = false;
    const int64_t sz = 2 * (int64_t)H * W;
    const T tau = (T)(1.0 / (double)Tn);
    const int grid = div_up((int64_t)H * W, 256);
    const int nb = t0, nf = Tn - 1 - t0, nstep = nb > nf ? nb : nf;
    if (nstep == 0) {
        CMAX_CHECK_HIP(hipMemcpyAsync(V + (int64_t)t0 * sz, F, sz * sizeof(T), hipMemcpyDeviceToDevice, s));
        return 0;
    }
    {  // the whole chain in one launch while the (16 + 2 nstep)^2 patch fits LDS
        const size_t lds = (size_t)4 * (kVoxTileH + 2 * nstep) * (kVoxTileW + 2 * nstep) * sizeof(T);
        if (lds <= 60 * 1024) {
            const int tiles = div_up(H, kVoxTileH) * div_up(W, kVoxTileW);
            if (scheme == CMAX_SCHEME_BURGERS)
                hipLaunchKernelGGL((k_voxel_chain_tiled<T, CMAX_SCHEME_BURGERS>), dim3(tiles), dim3(kVoxThreads), lds, s, F, Tn, t0, H, W, tau, nstep, V, V32);
            else
                hipLaunchKernelGGL((k_voxel_chain_tiled<T, CMAX_SCHEME_UPWIND>), dim3(tiles), dim3(kVoxThreads), lds, s, F, Tn, t0, H, W, tau, nstep, V, V32);
            CMAX_CHECK_LAUNCH();
            if (wrote_v32) *wrote_v32 = V32 != nullptr;
            return 0;
        }
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
int voxel_construct_adj(const T *V, int Tn, int t0, int H, int W, int scheme, T *gV, T *gF, hipStream_t s, bool det = false) {
    const int64_t sz = 2 * (int64_t)H * W;
    const T tau = (T)(1.0 / (double)Tn);
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
        const int tiles = div_up(H, kAdjTileH) * div_up(W, kAdjTileW);
        // the two chains write different bins until the last step, where both arrive at bin t0 (then one workgroup
        // runs both jobs of its tile one after the other)
        const dim3 agrid(tiles, (n == 2 && jobs.dst[0] != jobs.dst[1]) ? 2 : 1);
        if (scheme == CMAX_SCHEME_BURGERS && det)
            hipLaunchKernelGGL((k_flow_step_adj_tiled<T, CMAX_SCHEME_BURGERS, true>), agrid, dim3(kAdjThreads), 0, s, jobs, n, H, W, tau);
        else if (scheme == CMAX_SCHEME_BURGERS)
            hipLaunchKernelGGL((k_flow_step_adj_tiled<T, CMAX_SCHEME_BURGERS>), agrid, dim3(kAdjThreads), 0, s, jobs, n, H, W, tau);
        else if (det)
            hipLaunchKernelGGL((k_flow_step_adj_tiled<T, CMAX_SCHEME_UPWIND, true>), agrid, dim3(kAdjThreads), 0, s, jobs, n, H, W, tau);
        else
            hipLaunchKernelGGL((k_flow_step_adj_tiled<T, CMAX_SCHEME_UPWIND>), agrid, dim3(kAdjThreads), 0, s, jobs, n, H, W, tau);
        CMAX_CHECK_LAUNCH();
    }
    if (gF) CMAX_CHECK_HIP(hipMemcpyAsync(gF, gV + (int64_t)t0 * sz, sz * sizeof(T), hipMemcpyDeviceToDevice, s));
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
    if (nstep == 0) {
        CMAX_CHECK_HIP(hipMemcpyAsync(V + (int64_t)t0 * sz, F, sz * sizeof(T), hipMemcpyDeviceToDevice, s));
        CMAX_CHECK_HIP(hipMemcpyAsync(dV + (int64_t)t0 * sz, dF, sz * sizeof(T), hipMemcpyDeviceToDevice, s));
    }
    for (int j = 1; j <= nstep; ++j) {
        DualJobs<T> jobs = {};
        int n = 0;
        // the first steps read (F, dF) directly; bin t0 is filled by a copy job of the same launch
        if (j <= nb) {
            const int64_t a = (int64_t)(t0 - j + 1) * sz, b = (int64_t)(t0 - j) * sz;
            jobs.src[n] = j == 1 ? F : V + a; jobs.dsrc[n] = j == 1 ? dF : dV + a; jobs.dst[n] = V + b; jobs.ddst[n] = dV + b;
            jobs.s[n++] = (T)-1;
        }
        if (j <= nf) {
            const int64_t a = (int64_t)(t0 + j - 1) * sz, b = (int64_t)(t0 + j) * sz;
            jobs.src[n] = j == 1 ? F : V + a; jobs.dsrc[n] = j == 1 ? dF : dV + a; jobs.dst[n] = V + b; jobs.ddst[n] = dV + b;
            jobs.s[n++] = (T)1;
        }
        if (j == 1) {
            jobs.src[n] = F; jobs.dsrc[n] = dF; jobs.dst[n] = V + (int64_t)t0 * sz; jobs.ddst[n] = dV + (int64_t)t0 * sz;
            jobs.s[n++] = (T)0;
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
int voxel_construct_adj_tan(const T *V, const T *dV, int Tn, int t0, int H, int W, int scheme, T *gV, T *dgV, T *gF, T *dgF, hipStream_t s,
                            bool det = false) {
    const int64_t sz = 2 * (int64_t)H * W;
    const T tau = (T)(1.0 / (double)Tn);
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
        // the two chains write different bins until the last step, where both arrive at bin t0
        const dim3 dgrid(div_up(H, kAdjTileH) * div_up(W, kAdjTileW), (n == 2 && jobs.dst[0] != jobs.dst[1]) ? 2 : 1);
        if (scheme == CMAX_SCHEME_BURGERS && det)
            hipLaunchKernelGGL((k_flow_step_adj_dual<T, CMAX_SCHEME_BURGERS, true>), dgrid, dim3(kAdjDualThreads), 0, s, jobs, n, H, W, tau);
        else if (scheme == CMAX_SCHEME_BURGERS)
            hipLaunchKernelGGL((k_flow_step_adj_dual<T, CMAX_SCHEME_BURGERS>), dgrid, dim3(kAdjDualThreads), 0, s, jobs, n, H, W, tau);
        else if (det)
            hipLaunchKernelGGL((k_flow_step_adj_dual<T, CMAX_SCHEME_UPWIND, true>), dgrid, dim3(kAdjDualThreads), 0, s, jobs, n, H, W, tau);
        else
            hipLaunchKernelGGL((k_flow_step_adj_dual<T, CMAX_SCHEME_UPWIND>), dgrid, dim3(kAdjDualThreads), 0, s, jobs, n, H, W, tau);
        CMAX_CHECK_LAUNCH();
    }
    if (gF) CMAX_CHECK_HIP(hipMemcpyAsync(gF, gV + (int64_t)t0 * sz, sz * sizeof(T), hipMemcpyDeviceToDevice, s));
    if (dgF) CMAX_CHECK_HIP(hipMemcpyAsync(dgF, dgV + (int64_t)t0 * sz, sz * sizeof(T), hipMemcpyDeviceToDevice, s));
    return 0;
}

template int voxel_construct<float>(const float *, int, int, int, int, int, float *, hipStream_t, float *, bool *);
template int voxel_construct_adj<float>(const float *, int, int, int, int, int, float *, float *, hipStream_t, bool);

}  // namespace cmax

using namespace cmax;

namespace cmax {
// plan-internal: fp64 voxel and, when the tiled kernel ran, its fp32 copy in the same launch (*wrote_v32)
int voxel_construct_f64_f32(const double *F, int Tn, int t0, int H, int W, int scheme, double *V, float *V32, bool *wrote_v32, hipStream_t s) {
    return voxel_construct<double>(F, Tn, t0, H, W, scheme, V, s, V32, wrote_v32);
}
// plan-internal: the adjoint sweeps with the order-free step kernels (deterministic handles)
int voxel_construct_adj_f64(const double *V, int Tn, int t0, int H, int W, int scheme, double *gV, hipStream_t s, bool det) {
    return voxel_construct_adj<double>(V, Tn, t0, H, W, scheme, gV, nullptr, s, det);
}
int voxel_construct_adj_tan_f64(const double *V, const double *dV, int Tn, int t0, int H, int W, int scheme, double *gV, double *dgV, hipStream_t s, bool det) {
    return voxel_construct_adj_tan<double>(V, dV, Tn, t0, H, W, scheme, gV, dgV, nullptr, nullptr, s, det);
}
}  // namespace cmax

extern "C" {

int cmax_flow_step(const void *F, int dtype, int H, int W, double dt, int scheme, void *out, cmax_stream_t stream) {
    CMAX_REQUIRE(F && out && F != out && H > 0 && W > 0, "flow_step");
    CMAX_REQUIRE(scheme == CMAX_SCHEME_BURGERS || scheme == CMAX_SCHEME_UPWIND, "flow_step: scheme");
    if (dtype == CMAX_F32) return flow_step<float>((const float *)F, H, W, dt, scheme, (float *)out, (hipStream_t)stream);
    if (dtype == CMAX_F64) return flow_step<double>((const double *)F, H, W, dt, scheme, (double *)out, (hipStream_t)stream);
    set_error("flow_step: dtype");
    return CMAX_EINVAL;
}

int cmax_set_leaf_deterministic(int enable) { return g_leaf_det.exchange(enable != 0 ? 1 : 0); }

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
    CMAX_REQUIRE(V && gV && Tn > 0 && t0 >= 0 && t0 < Tn && H > 0 && W > 0, "voxel_construct_adj");
    CMAX_REQUIRE(scheme == CMAX_SCHEME_BURGERS || scheme == CMAX_SCHEME_UPWIND, "voxel_construct_adj: scheme");
    if (dtype == CMAX_F32) return voxel_construct_adj<float>((const float *)V, Tn, t0, H, W, scheme, (float *)gV, (float *)gF, (hipStream_t)stream, g_leaf_det.load() != 0);
    if (dtype == CMAX_F64) return voxel_construct_adj<double>((const double *)V, Tn, t0, H, W, scheme, (double *)gV, (double *)gF, (hipStream_t)stream, g_leaf_det.load() != 0);
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
    CMAX_REQUIRE(V && dV && gV && dgV && Tn > 0 && t0 >= 0 && t0 < Tn && H > 0 && W > 0, "voxel_construct_adj_tan");
    CMAX_REQUIRE(scheme == CMAX_SCHEME_BURGERS || scheme == CMAX_SCHEME_UPWIND, "voxel_construct_adj_tan: scheme");
    if (dtype == CMAX_F32)
        return voxel_construct_adj_tan<float>((const float *)V, (const float *)dV, Tn, t0, H, W, scheme, (float *)gV, (float *)dgV, (float *)gF, (float *)dgF, (hipStream_t)stream, g_leaf_det.load() != 0);
    if (dtype == CMAX_F64)
        return voxel_construct_adj_tan<double>((const double *)V, (const double *)dV, Tn, t0, H, W, scheme, (double *)gV, (double *)dgV, (double *)gF, (double *)dgF, (hipStream_t)stream, g_leaf_det.load() != 0);
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
