// Fused contrast-maximization objective for MI355X (gfx950).
//
// One evaluation = what PatchContrastMaximization.get_arg_for_cost + cost.calculate +
// torch.autograd.grad compute in the reference (src/solver/patch_contrast_base.py:273-352,
// src/solver/scipy_autograd/torch_wrapper.py:30-49), without ever materialising warped events:
//
//   set_events (once per batch)  pack to 8 B/event (12-bit row | 12-bit col | 8-bit time bin, fp32
//                                normalised time) and counting-sort by source-pixel tile (binned handles:
//                                by (tile, time bin)); the host cuts the sorted stream into segments
//   K1  k_vote      one workgroup per SEGMENT (<= 2040 consecutive sorted events = a few neighbouring
//                   source tiles): warp, bounding box of the targets, votes accumulated in an LDS
//                   window as 12.20 fixed point with ds_add_u32 (fp32 LDS atomics run at 1/12 of the
//                   integer rate on gfx950, profiles/r01_microbench.txt), one coalesced global
//                   atomicAdd per touched window pixel; publishes its windows for K3
//   K2  image kernels: k_stats (fp64 sums of the contrast function into `nsub` sub-accumulators per
//                   image: same-address fp64 atomics serialise at ~12 ns), or fused with what follows --
//                   k_blur_stats_adj_var (variance with blur and a gradient: ONE kernel, the mean comes from K1's vote sums;
//                   value-only / multi-GPU: k_blur_stats_var + k_gimage_blur_adj_var),
//                   k_stats_gimage_gm / k_blur_stats_gimage_gm (gradient magnitude: statistics + the
//                   G = dL/dIWE, blurs included).  They also clear the OTHER vote buffer for
//                   the next evaluation (double buffering: no memset in the steady state) and the
//                   flow-gradient buffer, with write-through stores.  Plain variance needs no G image:
//                   K3 gathers the raw image (the mean cancels in the gather's differences except at the
//                   edge of the region); on owned groups its K2 even runs INSIDE the K3 launch (kFoldStatsInside)
//   K3  k_grad      per event: re-warp, gather G at the 4 corners -> dL/d(x',y') -> motion gradient
//                   (2-DoF: per-segment partials, for the plain variance together with the image
//                   statistics so that K2 is not launched at all; dense: runs of equal source pixel
//                   reduced in registers + one segmented scan, one atomic per run; voxel: LDS
//                   accumulators in block-scaled fixed point)
//   k_finish / k_finish_deferred   sum the per-segment partials (2-DoF), write loss + gradient
// Every launch covers all reference times of the objective (blockIdx.y).  The event kernels live in
// cmax_event_kernels.inc, compiled for 256-, 512- and 1024-thread workgroups (namespaces t256 / t512 /
// t1024); the host picks 512 above 1024 segments (1024 for the voxel K3).
//
// fp32 per event with the integer source pixel split from the fp32 displacement (keeps the
// bilinear fractions accurate to ulp(displacement) instead of ulp(coordinate)); fp64 reductions.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "cmax_comm.h"
#include "cmax_common.h"
#include "cmax_image_kernels.h"
#include "cmax_search_kernels.h"
#include "cmax_sort_kernels.h"
#include "cmax_radix_sort.h"

// Environment knobs (tuning experiments and A/B tests only; none changes results beyond rounding):
//   CMAX_NO_OWNED=1      never build the group-aligned "owned groups" work list (cmax_set_events)
//   CMAX_NO_RUN_SORT=1   leave the events of a source pixel in the order the tile sort produced (no ordering by time)
//   CMAX_VOTE_NS / CMAX_GRAD_NS = 256 | 512 | 1024   force the workgroup size of K1 / K3
//   CMAX_SEG_CAP=n        free-cut work lists: events per segment (<= the layout's cap; profiles/r04_ablation.txt 18)
//   CMAX_NO_STATS_INSIDE=1   plain variance on owned groups: k_stats as a launch of its own instead of inside K3's
//   CMAX_STAT_SWEEPS=n       statistics inside K3's launch: 4-pixel sweeps per statistics workgroup (fewer, longer workgroups)
//   CMAX_NO_FUSED_BLURVAR=1  blurred variance: k_blur_stats_var + k_gimage_blur_adj_var instead of k_blur_stats_adj_var
//   CMAX_NSUB=n          statistics sub-accumulators (cache lines) per image
//   CMAX_TAN2=0 | 1      2-DoF tangent-image path: never / also without a communicator
//   CMAX_PLAN_GRAPHS=1   replay the patch plan from captured hipGraphs (cmax_solver.hip)
//   CMAX_COMPACT=0       big segments read the plain 8-byte events instead of the 4.5-byte compact copy (fp32 time instead of 24-bit fixed point)
//   CMAX_DEBUG_COMPACT=1 cmax_set_events synchronises behind k_pack_compact and fails if an event did not fit its region's coordinates
//   CMAX_SORT=radix | bucket   per-batch order by the stable radix sort (cmax_radix_sort.h) / the two-level counting sort, whatever the size
namespace cmax {

constexpr int kSparseSegment = 512;  // voxel K3: segments below this many events add straight to memory (no LDS accumulators)
constexpr int kTile = 16;  // source-pixel tile edge of the counting sort
constexpr int kSegMax = 2040;                 // events per segment (+1 for the even-aligned start still fits 2048);
                                              // |sum of votes| <= 2040 * 2^20 < 2^31
constexpr int kWinCap = 8192;                 // LDS window capacity in 32-bit words (32 KiB)
constexpr int kAccCells = 3072;               // flow-gradient accumulator cells per channel in LDS (voxel K3): 12 (tile, bin) groups
constexpr int kAccCellsDense = 768;           // the same for the dense K3 with owned tiles: 3 source tiles
constexpr int kWinMaxW = 128;                 // widest window when the bounding box has to be clipped
constexpr int64_t kRadixMinEvents = 8000000;  // batches from here on are ordered by the stable radix sort (cmax_radix_sort.h): 8M events 550 vs 556 us,
                                              // 12M 765 vs 875, 20M 1177 vs 1419, 64M 3483 vs 4754 (profiles/r05_set_events.txt); below, the counting sort's fewer launches win
constexpr int kShiftWordsMax = 512;           // words of cell offsets per (reference time, segment): 4 bits per slot of a big segment
// (votes are accumulated as signed fixed point in the LDS windows: 12.20, big segments 13.19 -- kFixNS of cmax_event_kernels.inc)

struct EvView {
    const uint2 *ev;      // .x = row | col << 12 | bin << 24 ; .y = bits of fp32 tau = (t - tmin) / (tmax - tmin)
    const float *rx;      // fractional residual of the source coordinate (nullptr if integral)
    const float *ry;
    const float2 *rl;     // ... its low part (rxl, ryl): residual = (double)rx + (double)rxl -- read only by warp_exact (events on a cell border)
    const double *tau64;  // BINNED handles (the packed word's top byte holds the voxel bin): normalised time in fp64, same order as `ev` --
                          // read only for the few events whose cell is decided in fp64 (warp_exact); null on un-binned handles, whose
                          // packed word carries the time's residual beyond fp32 (tau_refined)
};

struct WarpParams {
    int H, W, Hp, Wp, ph, pw;  // un-padded sensor, padded image, padding
    int T;                     // voxel bins
    int ntc;                   // source tiles per tile row
    float d;                   // reference time as a fraction of the batch period
    float d_lo;                // ... what fp32 could not hold of it (d64 = (double)d + (double)d_lo): read only by warp_exact
    int normalize;             // normalize_t
    int motion_f64;            // 2-DoF only: `motion` points to double[2] (cmax_objective_t::motion_dtype == CMAX_F64)
    const double *tmm;         // device (tmin, tmax)
    const float *motion;       // theta[2] | flow[2,H,W] | voxel[T,2,H,W]
};

// One launch of an event kernel covers every reference time of the objective (blockIdx.y): a multi-focal cost
// warps the same events to 3 reference times, and on the solver's small batches three dependent launches per
// kernel class were three times the launch latency.
struct RefArgs {
    float d[4];       // reference time as a fraction of the batch period
    float d_lo[4];    // ... its fp32 remainder (K1's fp64 cell decisions: a reference time need not be a dyadic fraction -- "random" / float directions)
    float *img[4];    // K1: vote image to fill; K3: image to gather from (dL/dIWE or the IWE itself)
    double *stat[4];  // K1: statistics accumulators to reset (or null)
    double *raw_zero; // K1: [n_ref][kRawStride] gradient-sum lines of the 2-DoF K3 of the same evaluation to reset (or null)
    float *zero[4];   // K3, deferred statistics: vote image of the NEXT evaluation to clear (or null)
    int k0;           // index of the first reference time of this launch (statistics slot, partial-sum offset)
    int4 *win;        // [n_ref][nseg] LDS windows: written by K1, read by K3 of the same evaluation (or null)
    unsigned *shifts; // [n_ref][nseg][kSlots / 8] cell offsets of the events K1 decided in fp64 (phase_warp), 8 slots per word; valid where Window::bmask says so
    int windows_only; // K1: publish the windows (and offsets) and return -- for a K3 whose vote this was not (cmax_objective_finish)
    // cmax_objective_batch: blockIdx.z = candidate motion.  Element strides between consecutive candidates (0 for every other launch):
    int64_t z_img;    // floats between their vote images (img[k] / zero[k] are candidate 0's)
    int64_t z_raw;    // doubles between their raw-sum lines (stat[k] of K1, gpart of K3)
    int z_motion;     // floats between their motions
    // K1, blurred variance with a gradient: sum_p I[p] B[p] (B = blur^T 1_Omega = b(r) b(c)) = the sum of the blurred image over
    // Omega, accumulated while the votes are flushed -- the image kernel then knows the mean before it has blurred anything
    double *musum[4];        // kMuLines accumulators (one 128-byte line each) per reference time, or null
    // K3, kFoldStatsInside: musum[k] is READ (the mean), and
    int stat_blocks;         // leading workgroups of the grid that run the statistics (a multiple of 8: the XCD map of the others holds)
    int *ticket;             // their arrival counters (kTicketLines lines, zero between launches): the last one writes the loss
    double *musum_next;      // the other buffer of the vote sums: cleared for the next evaluation
    float band_b0, band_b1;  // b(0) = b(n - 1) and b(1) = b(n - 2) of border_weight (b = 1 elsewhere)
    // deterministic mode (cmax_set_deterministic): order-free integer accumulation
    long long *img64[4];       // K1: 2^-20 fixed-point vote image, 64-bit integer atomics instead of fp32 ones (or null)
    const unsigned *imax;      // K3: bits of max |image k| per statistics slot (the bound the fixed-point scale is derived from)
    long long *g64;            // K3: 2-DoF per-segment partial sums [n_ref][nseg][2] / flow-gradient accumulators, fixed point
    double *det_inv_scale;     // K3: [4] 1 / scale used by reference time k (2-DoF) or by all of them ([0], flow gradient)
    long long n_events;        // K3: events behind one accumulator at most (fixed-point headroom)
    // K1: the flow-gradient buffer the K3 of this evaluation ADDS into (work lists that are not group-aligned), cleared slice by slice
    // behind the event loads -- so that the statistics can run inside the K3 launch there too (kFoldStatsInside) instead of in a
    // launch of their own whose second job this was
    float4 *grad_zero;
    int64_t n_grad_zero4;
};

// the image-space kernels of an evaluation cover all reference times in one launch as well (blockIdx.y)
struct ImgArgs {
    const float *in[4];  // raw votes or blurred image, per reference time
    float *blurred[4];   // blurred image to write (kernels that blur)
    float *zero[4];      // vote image of the next evaluation to clear (or null)
    float *G[4];         // dL/dIWE to write
    float chain[4];      // fused statistics + G kernels: chain factor to fold into G (a constant of a cost that is not normalised;
                         // 1 when it depends on the finished statistics and K3 applies it: kFoldScale)
};

}  // namespace cmax

// Phase timeline of the event kernels (builds with -DCMAX_TIMELINE only, tools/timeline.py): thread 0 of the first 4096 workgroups of
// reference time 0 stamps the 100 MHz wall clock at the phase boundaries of K1 (kernel 0) and K3 (kernel 1) into
// g_timeline[kernel][workgroup][8] (round 6: kernel 2 = the fused image kernel between them); a stamp that is to follow the arrival of
// loaded data is handed a value computed from it.
#ifdef CMAX_TIMELINE
namespace cmax {
__device__ unsigned long long *g_timeline = nullptr;
__device__ __forceinline__ void timeline_stamp(int kernel, int idx) {
    if (g_timeline && threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 4096u) g_timeline[((size_t)kernel * 4096 + blockIdx.x) * 8 + idx] = wall_clock64();
}
template <typename T>
__device__ __forceinline__ void timeline_stamp_after(int kernel, int idx, T dep) {
    asm volatile("" ::"v"(dep));
    timeline_stamp(kernel, idx);
}
}  // namespace cmax
#define CMAX_STAMP(K, I) cmax::timeline_stamp(K, I)
#define CMAX_STAMP_AFTER(K, I, DEP) cmax::timeline_stamp_after(K, I, DEP)
#else
#define CMAX_STAMP(K, I) do { } while (0)
#define CMAX_STAMP_AFTER(K, I, DEP) do { } while (0)
#endif

struct cmax_handle_s {
    int H = 0, W = 0, ph = 0, pw = 0, Hp = 0, Wp = 0;
    int device = 0;
    int64_t n = 0, cap = 0;
    int64_t n_dropped = 0;  // events of the last batch whose source pixel was off the sensor (or NaN): not packed
    bool keep_outside = true;  // cmax_set_keep_outside (default since round 5): finite events off the sensor are packed (nearest sensor pixel + residual)
    int64_t n_outside = 0;      // ... how many of the last batch (2-DoF objectives only: a flow has no value there)
    bool has_frac = false;
    int n_time_bin = 0;
    bool slab_major = false;  // the (tile, bin) groups are ordered (tile row, time slab, tile column): cmax_set_time_slabs (large motions)
    // packed, sorted events
    uint2 *evp = nullptr;  // packed events, 8 B each, 16-byte aligned base (+2 elements of padding)
    float *rx = nullptr, *ry = nullptr;
    float2 *rl = nullptr;  // low parts of (rx, ry): written and read only for batches with fractional sources
    double *tau64 = nullptr;
    // staging SoA of the two-level sort (tile buckets before the per-tile ordering)
    uint2 *evp_alt = nullptr;
    float *rx_alt = nullptr, *ry_alt = nullptr;
    float2 *rl_alt = nullptr;
    double *tau64_alt = nullptr;
    int64_t cap_alt = 0;
    // sort scratch
    int *counts = nullptr;  // [nkeys + 1] events per tile -> tile offsets after the scan
    int *cursor = nullptr;  // [nkeys] per-tile cursor of the bucket pass, then active source pixels per tile
    int *scan_tmp = nullptr;  // [ceil(nkeys / 2048)] chunk sums of the scan
    int *rs_hist = nullptr;   // radix sort (large batches): [digits][workgroups] + 1 histogram / offsets of one pass
    int *rs_tmp = nullptr;    // ... chunk sums of its scan
    int64_t rs_hist_cap = 0;
    int nkeys = 0, ntr = 0, ntc = 0;
    int *d_flags = nullptr;  // [0] any fractional source coordinate, [1] dropped events, [2] events kept from off the sensor (cmax_set_keep_outside)
    bool long_runs = false;  // >= 8 events per active source pixel on average: the dense K3 reduces runs serially per thread
    bool mid = false;        // MID segments: up to 3064 events, four source tiles (five (tile, bin) groups) wide, event kernels of the m512 namespace -- when the
                             // standard cut would need more workgroups than the chip holds at once and this one does not (build_segments)
    bool big = false;        // BIG segments: up to 4088 events each, event kernels of the b512 / b1024 namespaces (batches of >= 8M events)
    int seg_max = 2040;      // events per segment of the current work list (kSegMax, or 4088 for big segments)
    bool small_acc = false;  // binned handle, owned groups of <= 3 groups per segment: the voxel K3 with kAccCellsDense accumulator cells (kGradOwnedSmall)
    bool owned = false;      // the work list gives every group (empty ones included) to exactly one segment, <= kAccCells / 256 groups each
    int *d_tile_start = nullptr;  // [ngroups + 1] first sorted event of every group (source tile, or (tile, time bin))
    // What the host reads back once per batch lives in ONE allocation, in this order: d_tmm (2 doubles) | d_flags (4 ints) |
    // d_active (ntiles ints) | d_tile_start (...): one device-to-host copy instead of four (each ~4 us of copy latency)
    int *d_batch = nullptr;   // base of that allocation (as ints)
    int *d_active = nullptr;  // [ntiles] source pixels that hold events, per tile (un-binned order; written by k_tile_sort)
    int4 *d_segs = nullptr;       // [nseg] (begin, count, first source tile, tiles spanned): work items of the event kernels
    int4 *d_win = nullptr;        // [4][nseg] LDS windows of the last objective vote (K1 -> K3 of the same evaluation)
    // compact copy of the sorted events, one region per segment of the work list (EventLoads in cmax_event_kernels.inc): big segments of
    // un-binned handles; valid for the current work list only (build_segments)
    char *cev = nullptr;
    int64_t cev_cap = 0;          // regions allocated
    bool compact = false;
    // cmax_objective_batch: K candidate motions per launch (allocated on first use, sized for batch_cap candidates)
    float *bimg = nullptr;        // [2 buffers][batch_cap][n_ref <= 4][npix] vote images
    double *braw = nullptr;       // [batch_cap][4][kRawStride] raw sums
    int4 *bwin = nullptr;         // [batch_cap][4][nseg] windows
    unsigned *bshifts = nullptr;  // [batch_cap][4][nseg][kShiftWordsMax]
    int batch_cap = 0, batch_seg_cap = 0, batch_cur = 0;
    int batch_zero[2] = {0, 0};  // the first batch_zero[b] images of buffer b are zero
    unsigned *d_shifts = nullptr; // [4][nseg][kShiftWordsMax] cell offsets of the events that vote decided in fp64 (RefArgs::shifts)
    // what those windows were computed for: K3 reuses them only for the same motion / model / reference times
    const float *win_motion = nullptr;
    int win_model = -2, win_nref = 0, win_T = 0, win_normalize = 0;
    float win_d[4] = {0.f, 0.f, 0.f, 0.f};
    uint64_t win_generation = ~(uint64_t)0;
    int nseg = 0, seg_cap = 0;
    // images
    float *imgs = nullptr;                                  // [2 buffers][5, Hp, Wp] raw votes: one per reference time + un-warped
    float *iweb[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // blurred copies
    float *G = nullptr, *Gt = nullptr;
    const float *last_iwe[4] = {nullptr, nullptr, nullptr, nullptr};
    // device scalars
    double *d_tmm = nullptr;  // [2]
    double *d_stat = nullptr;   // [5 slots][32 sub-accumulators][2] contrast statistics (slot 4 = un-warped image)
    // the handle's own vote images are double-buffered: k_stats of evaluation e zeroes the images of
    // evaluation e+1, so the steady state needs no memset node.  zero_mask[b] bit k: image k of buffer b is zero
    int cur_buf = 0;
    unsigned zero_mask[2] = {0u, 0u};
    double *d_gpart = nullptr;  // [4 reference times][nseg][2] per-segment 2-DoF partials
    double *d_raw = nullptr;    // [4 reference times][kRawStride] raw sums of the deferred 2-DoF K3 (cmax_objective's own copy)
    // cmax_objective_host: device outputs + pinned staging of one evaluation whose results go straight to the host
    double *d_host_result = nullptr;  // [8]
    void *d_host_grad = nullptr;
    int64_t host_grad_bytes = 0;
    double *hp_out = nullptr;  // pinned
    int64_t hp_out_cap = 0;
    unsigned long long host_seq = 0;  // run counter of the finishing kernel that writes into hp_out (polled by cmax_objective_host)
    int *d_ticket = nullptr;    // arrival counters of the statistics workgroups inside K3 (kFoldStatsInside): zero between launches
    double *d_musum = nullptr;  // [2 buffers][4 reference times][kMuStride] K1's sums for the blurred variance (RefArgs::musum)
    float4 *vote_clear4 = nullptr;      // objective_eval -> vote_images: gradient buffer K1 is to clear (RefArgs::grad_zero), consumed by the launch
    int64_t vote_nclear4 = 0;
    const void *grad_cleared_by_vote = nullptr;  // ... and which buffer the K1 launch of the current evaluation did clear
    int mu_buf = 0;             // buffer the next evaluation adds into (the other one is being cleared / is clear)
    bool mu_valid = false;      // the K1 launch of the current cmax_objective call filled d_musum[mu_buf]
    // orig-IWE cache key
    bool orig_valid = false;
    double orig_sigma = -1;
    int orig_cost = -1, orig_omit = -1;
    double tmin_host = 0.0, tmax_host = 0.0;  // batch extremes (copied once per batch)
    float *hvp_img = nullptr;                 // [6 kinds][4 reference times][Hp, Wp] scratch of cmax_objective_hvp (allocated on first use)
    double *d_stat_tan = nullptr;             // [4][kStatStride] tangent statistics
    // pinned host staging of the per-batch read-back (group starts, active pixels per tile, flags, time extremes) and
    // of the work list: copies to / from pageable memory are staged by the runtime and cost a synchronisation each
    int *hp_read = nullptr;
    int64_t hp_read_cap = 0;
    int4 *hp_segs = nullptr;
    int64_t hp_segs_cap = 0;
    hipEvent_t segs_copied = nullptr;  // the last upload of hp_segs has left the host buffer
    hipEvent_t read_done = nullptr;    // the per-batch read-back has arrived (the stream may still be busy behind it)
    float2 *search_range = nullptr;           // [search_cap] (tau_min, tau_max) per patch of cmax_patch_search
    int search_cap = 0;
    int64_t bytes = 0;
    uint64_t generation = 0;  // bumped by set_events / set_time_bins (device pointers and the work list change)
    uint64_t generation_counted = ~(uint64_t)0;
    // optional per-kernel-class timing with HIP events (cmax_set_profiling)
    bool profiling = false;
    int prof_repeat = 1;  // > 1: every hot launch is issued this many times inside its event bracket (timing only)
    std::vector<hipEvent_t> prof_ev[CMAX_PROF_CLASSES];  // class -> [start0, stop0, start1, stop1, ...]
    // 2-DoF tangent images (k_vote_tan2): [2 buffers][4 reference times][I: Hp Wp | E0: (Hp + 1) Wp | F1: Hp (Wp + 1)]
    float *tan = nullptr;
    int tan_cur = 0;
    unsigned tan_zero_mask[2] = {0u, 0u};  // bit k: planes of reference time k of buffer b are zero
    double *d_tanpart = nullptr;           // [4][kTanBlocks][6] partial sums of k_tan_stats_var
    // time-sliced multi-GPU evaluation: this rank's RCCL communicator (cmax_comm_init), or null
    cmax::Comm *comm = nullptr;
    // C2 in row bands behind K3 (cmax_comm_set_c2_bands): the owned-group K3 of a dense objective is launched band by band (whole
    // tile rows) and every band's rows of the gradient are all-reduced on a second stream while the next band's K3 runs
    int c2_bands = 1;
    std::vector<int> row_seg_start;  // [ntr + 1] first segment of every source tile row (owned work lists only)
    hipStream_t comm_stream = nullptr;
    std::vector<hipEvent_t> band_ev;  // [bands + 1]
    // deterministic mode (cmax_set_deterministic): every accumulation that depends on the order of events, workgroups or
    // atomics is done in integers (exact, associative) -- bit-identical IWE, loss and gradient from run to run
    bool deterministic = false;
    long long *img64 = nullptr;   // [5][npix] fixed-point vote images, all zero between evaluations
    unsigned *d_imax = nullptr;   // [kStatSlots] bits of max |image| per statistics slot
    long long *g64 = nullptr;     // fixed-point gradient accumulators (zero between evaluations)
    int64_t g64_cap = 0;
    double *d_det_inv_scale = nullptr;  // [4]
    float *Gt_det = nullptr;      // [4][npix] dL/d(blurred image) of the unfused image path
};

namespace cmax {

// kernel classes for cmax_set_profiling / cmax_read_profile
enum { kProfVote = 0, kProfStats = 1, kProfGimage = 2, kProfGrad = 3, kProfFinish = 4, kProfComm = 5 };
static_assert(kProfComm + 1 == CMAX_PROF_CLASSES, "profile classes");
constexpr size_t kProfMaxPairs = 16384;

// RAII: records a HIP event on the launch stream before and after the enclosed launch
struct ProfScope {
    cmax_handle_s *h;
    int cls;
    hipStream_t s;
    hipEvent_t stop = nullptr;
    ProfScope(cmax_handle_s *h_, int cls_, hipStream_t s_) : h(h_), cls(cls_), s(s_) {
        if (!h->profiling || h->prof_ev[cls].size() >= 2 * kProfMaxPairs) return;
        hipEvent_t start = nullptr;
        if (hipEventCreate(&start) != hipSuccess || hipEventCreate(&stop) != hipSuccess) {
            stop = nullptr;
            return;
        }
        (void)hipEventRecord(start, s);
        h->prof_ev[cls].push_back(start);
        h->prof_ev[cls].push_back(stop);
    }
    ~ProfScope() {
        if (stop) (void)hipEventRecord(stop, s);
    }
};

// ---------------------------------------------------------------------------------------------
// memory helpers
// ---------------------------------------------------------------------------------------------
template <typename T>
static int dev_alloc(cmax_handle_s *h, T **p, int64_t count) {
    if (count <= 0) count = 1;
    hipError_t e = hipMalloc((void **)p, (size_t)count * sizeof(T));
    if (e != hipSuccess) {
        set_error("hipMalloc(%lld bytes) failed: %s", (long long)(count * sizeof(T)), hipGetErrorString(e));
        return CMAX_ENOMEM;
    }
    h->bytes += count * (int64_t)sizeof(T);
    return 0;
}
template <typename T>
static void dev_free(T **p) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
}

// ---------------------------------------------------------------------------------------------
// set_events: pack + two-level sort (cmax_sort_kernels.h); the scan of the tile counts lives here
// ---------------------------------------------------------------------------------------------
__global__ void k_tmm_set(double *tmm, double lo, double hi) {
    tmm[0] = lo;
    tmm[1] = hi;
}

// Exclusive scan of counts[0..m) in place, counts[m] = total, in three small launches: sums of 2048-element
// chunks, scan of the chunk sums by one workgroup, scan inside every chunk (a single-workgroup scan of the
// 921k pixel keys of a 1280x720 sensor took 1.4 ms).
constexpr int kScanChunk = 2048;  // elements per workgroup (256 threads x 8)

// nonzero (optional): += number of keys with at least one event (one atomic per workgroup)
__global__ void __launch_bounds__(256) k_scan_sums(const int *__restrict__ counts, int m, int *__restrict__ chunk_sum, int *__restrict__ nonzero) {
    __shared__ int s_w[4], s_n[4];
    const int base = blockIdx.x * kScanChunk + threadIdx.x * 8;
    int s = 0, nz = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int c = base + u < m ? counts[base + u] : 0;
        s += c;
        nz += c != 0;
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, kWave);
        nz += __shfl_xor(nz, o, kWave);
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
        s_w[threadIdx.x / kWave] = s;
        s_n[threadIdx.x / kWave] = nz;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        chunk_sum[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        if (nonzero) atomicAdd(nonzero, s_n[0] + s_n[1] + s_n[2] + s_n[3]);
    }
}

// one workgroup: exclusive scan of the chunk sums in place (nchunk <= 1024 * per-thread loop), total -> *total_out
__global__ void __launch_bounds__(1024) k_scan_chunks(int *__restrict__ chunk_sum, int nchunk, int *__restrict__ total_out) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (nchunk + 1023) / 1024;
    const int b = t * per, e = min(b + per, nchunk);
    int s = 0;
    for (int i = b; i < e; ++i) s += chunk_sum[i];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan of the per-thread totals
        int v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;
    for (int i = b; i < e; ++i) {
        const int c = chunk_sum[i];
        chunk_sum[i] = run;
        run += c;
    }
    if (t == 1023) *total_out = part[1023];
}

__global__ void __launch_bounds__(256) k_scan_apply(int *__restrict__ counts, int m, const int *__restrict__ chunk_off) {
    __shared__ int s_w[4];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int base = blockIdx.x * kScanChunk + threadIdx.x * 8;
    int c[8], s = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        c[u] = base + u < m ? counts[base + u] : 0;
        s += c[u];
    }
    int incl = s;  // inclusive scan of the per-thread sums over the wave
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const int v = __shfl_up(incl, o, kWave);
        if (lane >= o) incl += v;
    }
    if (lane == kWave - 1) s_w[wave] = incl;
    __syncthreads();
    int run = chunk_off[blockIdx.x] + incl - s;
    for (int w = 0; w < wave; ++w) run += s_w[w];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        if (base + u < m) counts[base + u] = run;
        run += c[u];
    }
}

// m <= 4096: the whole scan in one workgroup (thread q owns counts[4q .. 4q+3]) instead of three launches
__global__ void __launch_bounds__(1024) k_scan_small(int *__restrict__ counts, int m) {
    __shared__ int s_w[1024 / kWave];
    const int t = threadIdx.x, lane = t & (kWave - 1), wave = t / kWave;
    int c[4], sum = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        c[j] = 4 * t + j < m ? counts[4 * t + j] : 0;
        sum += c[j];
    }
    int incl = sum;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const int v = __shfl_up(incl, o, kWave);
        if (lane >= o) incl += v;
    }
    if (lane == kWave - 1) s_w[wave] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int w = 0; w < wave; ++w) run += s_w[w];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (4 * t + j < m) counts[4 * t + j] = run;
        run += c[j];
    }
    if (t == 1023) counts[m] = run;
}

// ---------------------------------------------------------------------------------------------
// per-event warp shared by K1 and K3
// ---------------------------------------------------------------------------------------------
// v_cvt_flr_i32_f32: (int)floor(x) in one instruction, saturating
__device__ __forceinline__ int cvt_flr(float x) {
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

struct Warped {
    int row, col;  // top-left corner in the padded image
    float a, b;    // row / column fractions
    float dt;
    int src;       // source pixel linear index (un-padded), + bin * 2HW for voxel
    float f0, f1;  // dense / voxel: the flow at the source pixel (phase_warp bounds the rounding of dt * f with it)
};

// One event as the warp sees it, whichever way it was stored (plain 8-byte event: absolute coordinates, zero bases; compact event of a
// big segment: coordinates relative to the segment's tile strip, bases from the region's header -- cmax_event_kernels.inc).
struct EvDec {
    unsigned ixr, iyr;  // source row / column minus EvBase::row / col
    unsigned top;       // top byte of the plain packed word (voxel: the time bin)
    unsigned key;       // (row | col << 12 [| bin << 24]) absolute: the run key of the flow-gradient reduction (K3)
    float tmd;          // tau - d: calculate_dt before the time scale, rounded once
};
struct EvBase {  // workgroup-uniform
    unsigned row, col;  // what ixr / iyr are relative to (0 for plain events)
    unsigned src;       // row * W + col
};

// MODEL: -1 none (orig_iwe), 0 2-DoF, 1 dense, 2 voxel
template <int MODEL, bool FRAC>
__device__ __forceinline__ Warped warp_one(const EvView &ev, const EvDec &e, const EvBase &eb, int64_t i, bool valid, const WarpParams &wp, float tscale, float th0,
                                           float th1) {
    Warped w;
    const int ix = (int)e.ixr, iy = (int)e.iyr;
    w.dt = e.tmd * tscale;  // calculate_dt, src/warp.py:254-259
    float dx = 0.f, dy = 0.f;
    if (FRAC && valid) {  // `valid` false: an empty slot of the caller (zero event), i may be outside the arrays
        dx = ev.rx[i];
        dy = ev.ry[i];
    }
    w.src = (int)(__umul24((unsigned)ix, (unsigned)wp.W & 0xFFFFFFu) + (unsigned)iy + eb.src);  // 12-bit x 13-bit: v_mul_u32_u24 (full rate; v_mul_lo_u32 runs at a quarter)
    if (MODEL == CMAX_MODEL_2DOF) {
        dx = fmaf(w.dt, th0, dx);  // x' = x + dt*theta0, src/warp.py:506-515
        dy = fmaf(w.dt, th1, dy);
    } else if (MODEL == CMAX_MODEL_DENSE || MODEL == CMAX_MODEL_VOXEL) {
        const int hw = wp.H * wp.W;
        if (MODEL == CMAX_MODEL_VOXEL) w.src += (int)e.top * 2 * hw;  // (2 HW can exceed 24 bits: a full 32-bit multiply)
        // uniform base + unsigned 32-bit BYTE offset: one address instruction per event and a `global_load_dword v, voff, s[base]`
        // per channel, instead of 64-bit per-lane pointer arithmetic (the field is < 4 GiB: checked by the host)
        const unsigned off = (unsigned)w.src * 4u;
        const char *m0 = reinterpret_cast<const char *>(wp.motion), *m1 = reinterpret_cast<const char *>(wp.motion + hw);
#ifdef CMAX_AB_NOFLOW  // (A/B builds only, profiles/r05_ablation.txt 8: no flow reads at all -- what the dense kernels pay for them)
        w.f0 = 0.25f + 1e-9f * (float)off;
        w.f1 = -0.125f;
        (void)m0;
        (void)m1;
#else
        w.f0 = *reinterpret_cast<const float *>(m0 + off);
        w.f1 = *reinterpret_cast<const float *>(m1 + off);
#endif
        dx = fmaf(-w.dt, w.f0, dx);  // x' = x - dt*F[0,ix,iy], src/warp.py:305-306
        dy = fmaf(-w.dt, w.f1, dy);
    }
    // floor(x' + 1e-6) = ix + floor(dx + 1e-6) exactly because ix is an integer
    // (bilinear_vote_tensor, src/event_image_converter.py:340-345)
    // Round 6 (VALU diet): v_floor_f32 + v_cvt_i32_f32 per coordinate; the v_med3_f32 between them (a clamp to +-8192) is gone:
    // a displacement beyond +-8192 px (a diverged or non-finite motion) now lands on whatever cell the saturated conversion names --
    // (v_cvt_i32_f32 saturates) -- its packed (row, col) word may wrap, but every consumer derives box, window and indices from the SAME word, so such an event
    // is either outside every window of the fast path's 8192 words (-> the clipped path, whose every corner is tested against window
    // and image) or votes in bounds; it was never defined where.
    const float fx = floorf(dx + 1e-6f), fy = floorf(dy + 1e-6f);
    const int fxi = (int)fx, fyi = (int)fy;  // (v_cvt_i32_f32 saturates; through inline asm v_cvt_flr_i32_f32 + v_cvt_f32_i32 are two instructions as well, but
    w.a = dx - fx;                           // the asm operands cost the 512 x 8 K3 eleven VGPRs, i.e. a wave per SIMD)
    w.b = dy - fy;
    w.row = ix + fxi + (int)(eb.row + (unsigned)wp.ph);
    w.col = iy + fyi + (int)(eb.col + (unsigned)wp.pw);
    return w;
}

// EXACT CELLS (round 3: 2-DoF; round 4: every model).  The image is continuous across a cell border, the gradient is not (it takes
// its differences of dL/dIWE from the event's own cell): an event whose displacement lies within fp32 rounding of a border falls
// on the other side in any fp32 evaluation, and its term of the gradient then comes from the wrong side of the kink -- ~70 of a
// million events moved the 2-DoF gradient (ONE sum over all events) by 6e-4 relative against the reference's fp64 value (the
// reference's own fp32 path: 1.3e-3, tests/golden/cfg2_fp32_reference.npz), and a flow gradient by more than 1e-4 in the pixels
// such events come from (2-5 of 0.6-1.8M entries at the BASELINE sizes).  The fp32 displacement of warp_one is off by at most
// 7 * 2^-24 |motion| max(period, 1) max(1, |d|, |1 - d|) (rounding of tau, of tau - d, of the time scale, of the product, of the
// + 1e-6); an event whose fraction a or b comes closer than exact_margin to 0 or 1 is warped again here BY K1, with the arithmetic of
// src/warp.py:304-307 / 506-515 + src/event_image_converter.py:340 in fp64: the fp64 normalised time (EvView::tau64, a load only
// these events make), the caller's fp64 theta (2-DoF), the flow as the device holds it (fp32 values, fp64 arithmetic) -- 1e-5 ..
// 1e-4 of the events -- and K1 publishes the cell's offset from the fp32 one for K3 (phase_warp).
template <int MODEL, bool FRAC>
__device__ __forceinline__ Warped warp_exact(const EvView &ev, unsigned ex, int64_t i, double tau, double period, const WarpParams &wp, double m0, double m1) {
    Warped w;
    const int ix = (int)(ex & 0xFFFu), iy = (int)((ex >> 12) & 0xFFFu);
    const double dtd = (tau - ((double)wp.d + (double)wp.d_lo)) * period;
    const double sgn = MODEL == CMAX_MODEL_2DOF ? 1.0 : -1.0;  // x' = x + dt theta (warp.py:514)  |  x' = x - dt F (warp.py:305)
    double srx = 0.0, sry = 0.0;  // the source coordinate's residual as the reference's fp64 arithmetic sees it
    if (FRAC) {
        const float2 lo = ev.rl[i];
        srx = (double)ev.rx[i] + (double)lo.x;
        sry = (double)ev.ry[i] + (double)lo.y;
    }
    const double ddx = fma(sgn * dtd, m0, srx), ddy = fma(sgn * dtd, m1, sry);
    const double fxd = fmin(fmax(floor(ddx + 1e-6), -8192.0), 8192.0), fyd = fmin(fmax(floor(ddy + 1e-6), -8192.0), 8192.0);
    w.a = (float)(ddx - fxd);
    w.b = (float)(ddy - fyd);
    w.row = ix + (int)fxd + wp.ph;
    w.col = iy + (int)fyd + wp.pw;
    w.dt = 0.f;
    w.src = 0;
    w.f0 = 0.f;
    w.f1 = 0.f;
    return w;
}
// The margin m of an event: it is a candidate for warp_exact when a + 1e-6 or b + 1e-6 lies in [0, m) or (1 - m, 1).  With eps = 2^-24
// the fp32 displacement dx = fl(rx -/+ dt32 f), dt32 = fl(fl(tau32 - d32) period32), is off by at most
//   eps (3.5 |dt f| + kappa |f|),   kappa = 0 for the reference time "first" (d = 0: the rounding of tau is then relative to dt itself),
//                                   else (1 + |d| / 2) period (roundings of tau and d: absolute in dt, so they scale with |f| alone)
// -- 3.5: the subtraction, the period, the product, the fma and the `+ 1e-6` at eps / 2 each, tau at eps (d = 0), theta at eps / 2
// (2-DoF) -- and |dt| <= period max(|d|, |1 - d|).  m = 1.5 eps (4 period max(|d|, |1 - d|) + kappa) fm with fm = max |motion component|
// of the event (a function of the EVENT alone, so that nothing depends on how threads and events are matched).  Fractional sources add
// the rounding of rx and the fma's on a sum of magnitude up to 1.  Everything else is RELATIVE to the motion: a zero motion -- the
// optimiser's usual starting point, where every event sits ON a border -- has no candidates.
__device__ __forceinline__ float exact_margin_coef(float d, float tscale) {
    const float kappa = d == 0.f ? 0.f : (1.f + 0.5f * fabsf(d)) * tscale;
    return 0x1.8p-24f * (4.f * tscale * fmaxf(fabsf(d), fabsf(1.f - d)) + kappa);
}
template <bool FRAC>
__device__ __forceinline__ float exact_margin(float coef, float fm) { return fmaf(coef, fm, FRAC ? 0x1p-21f : 0.f); }

__device__ __forceinline__ float time_scale(const WarpParams &wp) {
    return wp.normalize ? 1.0f : (float)(wp.tmm[1] - wp.tmm[0]);
}

// Workgroup -> segment, XCD-aware: the dispatcher places block b on XCD b % 8, so giving XCD x the
// contiguous range [x * per, (x+1) * per) keeps neighbouring tiles (shared halo rows of the IWE / G
// windows, neighbouring flow pixels) in one XCD's L2.  Placement only affects speed.
__device__ __forceinline__ int segment_of_block(int nseg, unsigned bx = blockIdx.x) {
#ifdef CMAX_NO_XCD_MAP
    return (int)bx;
#else
    const int per = (nseg + 7) >> 3;
    return (int)(bx & 7u) * per + (int)(bx >> 3);
#endif
}

// Row stride of an LDS window of width w.  With a power-of-two stride (the first version: a shift per index) the bank of a
// cell is its COLUMN modulo 32 whatever its row -- a 20-pixel-wide window used 20 of the 32 banks, and the two rows of a
// 2 x 2 footprint always collided (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.55-0.70 in K1 and K3, profiles/r02_sq_*).  An odd
// stride walks the banks row by row; the index costs one v_mad_u32_u24 instead of a shift + add.
__host__ __device__ inline int window_stride(int w) { return (w < 15 ? 15 : w) | 1; }

struct Window {
    int r0, c0, h, w;  // top-left corner in the padded image, extent (clipped to the image and to LDS)
    int stride;        // LDS row stride in words: odd (window_stride), so that the rows of a window start in different banks
    bool clipped;      // the bounding box did not fit: votes / reads outside the window but inside the image exist
    bool border;       // the segment holds an event within fp32 rounding of a cell border (K1's verdict, re-used by K3)
    unsigned long long bmask;  // ... and where: bit q = such an event among the segment's slots [q * kSlots / 64, (q + 1) * kSlots / 64)
};
// A window as K1 publishes it for K3 of the same evaluation (RefArgs::win): position | geometry + flags | the 64 bits of bmask
// (the LDS row stride follows from the width)
__device__ __forceinline__ int4 pack_window(const Window &w) {
    return make_int4((int)(((unsigned)(w.r0 + 16384) << 16) | (unsigned)(w.c0 + 16384)),
                     w.h | (w.w << 12) | (w.clipped ? (1 << 28) : 0) | (w.border ? (1 << 29) : 0),
                     (int)(unsigned)w.bmask, (int)(unsigned)(w.bmask >> 32));
}
__device__ __forceinline__ Window unpack_window(int4 kw) {
    Window w;
    w.r0 = (int)((unsigned)kw.x >> 16) - 16384;
    w.c0 = (int)((unsigned)kw.x & 0xFFFFu) - 16384;
    w.h = kw.y & 0xFFF;
    w.w = (kw.y >> 12) & 0xFF;
    w.stride = window_stride(w.w);
    w.clipped = ((kw.y >> 28) & 1) != 0;
    w.border = ((kw.y >> 29) & 1) != 0;
    w.bmask = (unsigned long long)(unsigned)kw.z | ((unsigned long long)(unsigned)kw.w << 32);
    return w;
}
// how K3 obtains dL/dIWE: from a materialised G image; folded G = c2 (IWE - mu) with the statistics K2 left in
// `stat`; or (2-DoF) deferred -- K3 gathers the raw image and the image statistics itself, the chain factors are
// applied by k_finish_deferred, and K2 is not launched at all
// Linear index of pixel (r, c) of an image that is W wide, r >= 0: one full-rate v_mad_u32_u24 (rows and widths are < 2^13;
// a 64-bit r * W + c is a quarter-rate v_mad_u64_u32, a 32-bit one a quarter-rate v_mul_lo_u32 -- and the event kernels are
// VALU-bound on the large configurations: SQ_ACTIVE_INST_VALU x 8 waves ~ 0.8 of the SIMD cycles in K1 / K3 of cfg3).
__device__ __forceinline__ int pix_index(int r, int c, int W) { return (int)__umul24((unsigned)r, (unsigned)W & 0xFFFFFFu) + c; }


constexpr int kFoldNone = 0, kFoldStats = 1, kFoldDeferred = 2, kFoldScale = 3;  // kFoldScale: G image stored without its chain factor
// kFoldStatsInside: kFoldStats for a plain (not normalised) variance whose K2 runs INSIDE the K3 launch -- the first
// RefArgs::stat_blocks workgroups of the grid are k_stats (they also write the loss), the others gather; nothing in the
// gradient waits for the statistics: the chain factor is a constant and the mean comes from K1's vote sums (RefArgs::musum)
constexpr int kFoldStatsInside = 4;
constexpr int kGradRuns = 0, kGradStrided = 1, kGradOwned = 2, kGradDet = 3, kGradOwnedSmall = 4;  // k_grad's VARIANT (see cmax_event_kernels.inc)
[[maybe_unused]] constexpr int kDummy = kWinCap;     // masked path: 64 per-lane scratch words behind the window
constexpr int kScratch = 200;       // scratch words behind the window; the fast path sends the 2x2 footprint of an
                                    // empty slot to kWinCap + lane + {0, 1, stride, stride + 1}, stride <= 128
static_assert(kScratch >= 64 + kWinMaxW + 2, "scratch must hold a per-lane 2x2 footprint at the widest stride");

// Compiler barrier on a loaded value.  `x = cond ? p[i] : 0` -- and even an unconditional load whose only use is such a
// select -- is compiled into an exec-masked block that holds the load AND its s_waitcnt: a thread that wants four loads in flight
// gets four dependent round trips to memory (found in round 3 in the ISA of K1 / K3 / k_stats: profiles/r03_ablation.txt).  Load
// unconditionally (clamped index), pin() every value, select afterwards: the loads are then issued back to back.
__device__ __forceinline__ void pin(float &v) { asm volatile("" : "+v"(v)); }

// Stage the ROWS x COLS window of `img` whose top-left pixel is (r0, c0) into an LDS tile of row pitch LD floats, zero outside
// the image.  All of a thread's loads are issued before its first LDS store (see pin(): `in ? img[..] : 0` per iteration is one
// memory round trip per iteration).  256 threads.
template <int ROWS, int COLS, int LD>
__device__ __forceinline__ void stage_tile(float (*tile)[LD], const float *__restrict__ img, int r0, int c0, int H, int W) {
    constexpr int kN = ROWS * COLS, kIt = (kN + 255) / 256;
    float x[kIt];
#pragma unroll
    for (int u = 0; u < kIt; ++u) {
        const int q = threadIdx.x + 256 * u, a = q / COLS, b = q - a * COLS;
        const int r = min(max(r0 + a, 0), H - 1), c = min(max(c0 + b, 0), W - 1);
        x[u] = img[(int64_t)r * W + c];
    }
#pragma unroll
    for (int u = 0; u < kIt; ++u) pin(x[u]);
#pragma unroll
    for (int u = 0; u < kIt; ++u) {
        const int q = threadIdx.x + 256 * u, a = q / COLS, b = q - a * COLS;
        const int r = r0 + a, c = c0 + b;
        if (q < kN) tile[a][b] = ((unsigned)r < (unsigned)H && (unsigned)c < (unsigned)W) ? x[u] : 0.f;
    }
}

// 16 bytes of zeros, written through the L2 (sc1): see the deferred-statistics K3
typedef float float4_v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_zero4_sc1(float *p) {
    const float4_v z = {0.f, 0.f, 0.f, 0.f};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(z) : "memory");
}

// Clears n floats at `base`, the launch's threads striding over it (tid of nthreads), with write-through stores:
// buffers that the NEXT kernels fill with device-scope atomics (vote images, the flow gradient) must not stay
// parked in an XCD's L2.  16-byte stores when the buffer allows it.
__device__ __forceinline__ void zero_fill_sc1(float *base, int64_t n, int64_t tid, int64_t nthreads) {
    if (!base) return;
    if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(base) & 15u) == 0) {
        for (int64_t q = tid; q < (n >> 2); q += nthreads) store_zero4_sc1(base + 4 * q);
    } else {
        for (int64_t q = tid; q < n; q += nthreads) __hip_atomic_store(&base[q], 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// v_cvt_rpi_i32_f32: floor(x + 0.5) in one instruction (rndne + cvt are two)
__device__ __forceinline__ int cvt_rpi(float x) {
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
// v_mad_u32_u16 with the HIGH half of `packed` as first factor: (packed >> 16) * factor + addend in one instruction (`factor` wave-uniform,
// below 2^16), and v_add_u32 with the LOW half of `packed` through SDWA: the LDS word of a packed (row, col) in two instructions where
// shift, mask, multiply-add and add were four
__device__ __forceinline__ unsigned mad_hi16_u(unsigned packed, unsigned factor, unsigned addend) {
    unsigned r;
    asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(r) : "v"(packed), "s"(factor), "v"(addend));
    return r;
}
__device__ __forceinline__ unsigned add_lo16_u(unsigned a, unsigned packed) {
    unsigned r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(a), "v"(packed));
    return r;
}
// 24-bit multiplies with a UNIFORM second factor, spelled out: the compiler takes v_mul_lo_u32 / v_mul_hi_u32 whenever it cannot
// prove both factors below 2^24 (a run-time window stride: K1's vote index, one per event).  `b` must be wave-uniform.
__device__ __forceinline__ unsigned mul24_u(unsigned a, unsigned b) {
    unsigned r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "s"(b), "v"(a));
    return r;
}
__device__ __forceinline__ unsigned mad24_u(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}
typedef unsigned short ushort2_v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_min_u16(unsigned a, unsigned b) {  // v_pk_min_u16: both 16-bit halves at once
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(ushort2_v, a), __builtin_bit_cast(ushort2_v, b)));
}
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(ushort2_v, a), __builtin_bit_cast(ushort2_v, b)));
}

// ---- wave64 segmented inclusive scan without LDS traffic -------------------------------------------
// ds_bpermute-based __shfl_up costs ~16 cycles per wave instruction on gfx950 and the 6-step scan of
// (x, y, flag) was 60 % of the dense K3 (profiles/r01_ablation.txt).  DPP row shifts run on the VALU:
// 4 steps inside each 16-lane row, then three v_readlane carries across the rows.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {  // lanes whose source is outside the row / wave read 0
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int old, int v) {
    return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u0(unsigned v) {  // lanes without a source read 0 (bound_ctrl: no move of the identity first)
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
constexpr int kDppRowShr1 = 0x111, kDppRowShr2 = 0x112, kDppRowShr4 = 0x114, kDppRowShr8 = 0x118;
constexpr int kDppWaveShl1 = 0x130, kDppWaveShr1 = 0x138, kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143;

// head: 1 on the first lane of every run.  On return (vx, vy) of the LAST lane of a run hold the run's sums.
__device__ __forceinline__ void seg_scan64(int head, float &vx, float &vy, int lane) {
    int f = head;  // "a run starts between my row's first lane and me"
#define CMAX_SEG_STEP(CTRL)                                   \
    {                                                         \
        const float x2 = dpp_f<CTRL>(vx), y2 = dpp_f<CTRL>(vy); \
        const int f2 = dpp_i<CTRL>(0, f);                     \
        if (!f) {                                             \
            vx += x2;                                         \
            vy += y2;                                         \
            f |= f2;                                          \
        }                                                     \
    }
    CMAX_SEG_STEP(kDppRowShr1)
    CMAX_SEG_STEP(kDppRowShr2)
    CMAX_SEG_STEP(kDppRowShr4)
    CMAX_SEG_STEP(kDppRowShr8)
#undef CMAX_SEG_STEP
    // carry the run that crosses a row boundary: lanes of row r without a head before them continue
    // the run ending at the last lane of row r-1 (whose value already contains earlier carries)
#pragma unroll
    for (int r = 1; r < 4; ++r) {
        const float cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vx), 16 * r - 1));
        const float cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vy), 16 * r - 1));
        if ((lane >> 4) == r && !f) {
            vx += cx;
            vy += cy;
        }
    }
}  // 64 scratch words behind the window: target of masked lanes (branch-free phase B)

// ---------------------------------------------------------------------------------------------
// K2: contrast statistics of one image (+ zeroing of the NEXT evaluation's vote image).
//     Workgroup b adds its fp64 partials to sub-accumulator b % 32 of its slot:
//     stat[slot][sub][0] = sum x (variance) or sum gx^2+gy^2 (grad-mag), [1] = sum x^2.
//     Same-address fp64 atomics serialise at ~12 ns each (they cost 47 us in the first version), so
//     they are spread over 32 addresses and consumers add the 32 values.  The accumulators are zeroed
//     by workgroup 0 of the K1 launch that fills the image (stream order), so no memset node exists.
// ---------------------------------------------------------------------------------------------
constexpr int kStatBlocksMax = 512;
constexpr int kStatSlots = 5;  // reference times 0..3, slot 4 = un-warped image
constexpr int kStatSub = 32;   // sub-accumulators per slot
// Every sub-accumulator sits on its OWN 128-byte line: atomics to one cache line serialise in the L2 atomic unit whatever
// their address inside it (~7 ns each).  Packed 16 bytes apart, the up to 32 sub-accumulators of a slot shared 1-4 lines, and
// the 363 workgroups of a 260 x 346 image kernel spent 5 of their 9.8 us queueing on ONE line (profiles/r02_ablation.txt).
constexpr int kSubStride = 16;  // doubles between sub-accumulators
constexpr int kStatStride = kStatSub * kSubStride;
// raw sums of the deferred 2-DoF K3 (S1x, S1y, S2x, S2y, sum I, sum I^2): kRawLines accumulators per reference time, one line each
constexpr int kRawLines = CMAX_RAW_LINES, kRawStride = CMAX_RAW_DOUBLES;
static_assert(kRawLines == kStatSub && kRawStride == kStatStride && (kRawLines & (kRawLines - 1)) == 0, "K1 resets either with the same pattern");

struct ObjParams {
    int cost, normalized, minimize, negate, omit, n_ref;
    double mult[4];
    int H, W;  // padded image
    int nsub;  // sub-accumulators in use (<= kStatSub), grows with the number of k_stats workgroups
};

// raw contrast from the summed accumulators (variance: unbiased like torch.var, image_variance.py:55)
__device__ __forceinline__ double contrast_value(int cost, const double *acc, double npix, double *mu_out) {
    if (cost == CMAX_COST_VARIANCE) {
        const double mu = acc[0] / npix;
        if (mu_out) *mu_out = mu;
        return (acc[1] - acc[0] * mu) / (npix - 1.0);
    }
    return acc[0] / npix;
}

__device__ __forceinline__ double region_pixels(int H, int W, int omit) {
    const int i0 = omit ? 1 : 0;
    return (double)(H - 2 * i0) * (double)(W - 2 * i0);
}

// sum of the sub-accumulators of one slot.  WAVE: called by a whole converged wave -- lane u loads accumulator u,
// DPP reduction, broadcast from lane 63 (a serial walk over up to 32 addresses costs the event kernels 2-4 us of
// dependent scalar-load latency at the head of every workgroup).
template <bool WAVE = false>
__device__ __forceinline__ void stat_sum(const double *__restrict__ stat, int slot, int nsub, double (&acc)[2]) {
    if (WAVE) {
        const int lane = threadIdx.x & (kWave - 1);
        double a0 = 0.0, a1 = 0.0;
        if (lane < nsub) {
            a0 = stat[slot * kStatStride + kSubStride * lane];
            a1 = stat[slot * kStatStride + kSubStride * lane + 1];
        }
        a0 = wave_sum_lane63(a0);
        a1 = wave_sum_lane63(a1);
        acc[0] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a0), kWave - 1), __builtin_amdgcn_readlane(__double2loint(a0), kWave - 1));
        acc[1] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a1), kWave - 1), __builtin_amdgcn_readlane(__double2loint(a1), kWave - 1));
        return;
    }
    acc[0] = 0.0;
    acc[1] = 0.0;
    for (int u = 0; u < nsub; ++u) {
        acc[0] += stat[slot * kStatStride + kSubStride * u];
        acc[1] += stat[slot * kStatStride + kSubStride * u + 1];
    }
}

template <bool WAVE = false>
__device__ __forceinline__ double orig_value(const ObjParams &op, const double *stat) {
    // orig_iwe is NOT boundary-cropped for the variance (normalized_image_variance.py:40-41)
    const int omit_o = op.cost == CMAX_COST_VARIANCE ? 0 : op.omit;
    double acc[2];
    stat_sum<WAVE>(stat, 4, op.nsub, acc);
    return contrast_value(op.cost, acc, region_pixels(op.H, op.W, omit_o), nullptr);
}

// dL/dv_k: chain factor of reference time k, and the mean of its image (variance)
template <bool WAVE = false>
__device__ __forceinline__ double chain_coef(const ObjParams &op, const double *stat, int k, double *mu_out) {
    const double npix = region_pixels(op.H, op.W, op.omit);
    if (!op.normalized && op.cost != CMAX_COST_VARIANCE) {  // constant factor, no mean: the statistics are not needed
        const double c = op.mult[k] * (op.minimize ? -1.0 : 1.0);
        return op.negate ? -c : c;
    }
    double acc[2];
    stat_sum<WAVE>(stat, k, op.nsub, acc);
    const double v = contrast_value(op.cost, acc, npix, mu_out);
    double coef;
    if (!op.normalized) coef = op.mult[k] * (op.minimize ? -1.0 : 1.0);
    else {
        const double v_orig = orig_value<WAVE>(op, stat);
        coef = op.mult[k] * (op.minimize ? -v_orig / (v * v) : 1.0 / v_orig);
    }
    return op.negate ? -coef : coef;
}

// Arrival ticket of a group of workgroups inside one launch: called by ONE thread of workgroup `wg` of `total` after its
// device-scope atomics; true for exactly one caller, the last to arrive, and every other workgroup's atomics have been
// performed by then (the caller waits for its own acknowledgements first).  The counters are left at zero.  Same-line atomics
// serialise, hence kTicketSubs lines + one.  Four dependent fabric round trips (~5 us, profiles/r02_ablation.txt): only for work
// that is NOT at the end of its kernel.
constexpr int kTicketSubs = 16, kTicketLines = kTicketSubs + 1, kTicketLineInts = 32;
__device__ __forceinline__ bool arrive_last(int *ticket, int wg, int total) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int nsub = total < kTicketSubs ? total : kTicketSubs;
    const int sub = wg % kTicketSubs, per = (total - sub + kTicketSubs - 1) / kTicketSubs;
    int *line = ticket + sub * kTicketLineInts;
    if (__hip_atomic_fetch_add(line, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != per - 1) return false;
    __hip_atomic_store(line, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // nobody else arrives on this line any more
    int *master = ticket + kTicketSubs * kTicketLineInts;
    if (__hip_atomic_fetch_add(master, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != nsub - 1) return false;
    __hip_atomic_store(master, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

// loss of a plain variance objective with ONE reference time from accumulators that other workgroups of the same launch
// filled with device-scope atomics: device-scope loads (a plain load may be served from this XCD's L2)
__device__ __forceinline__ void write_result_variance_agent(const ObjParams &op, const double *stat, double *__restrict__ result) {
    double acc[2] = {0.0, 0.0};
    for (int u = 0; u < op.nsub; ++u) {
        acc[0] += __hip_atomic_load(&stat[kSubStride * u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc[1] += __hip_atomic_load(&stat[kSubStride * u + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const double v = contrast_value(CMAX_COST_VARIANCE, acc, region_pixels(op.H, op.W, op.omit), nullptr);
    const double loss = op.mult[0] * (op.minimize ? -v : v);
    result[0] = op.negate ? -loss : loss;
    result[1] = v;
    result[5] = 0.0;
}

// loss and per-reference-time contrasts -> result[0..5].  WAVE: called by a whole converged wave (the accumulators of a slot
// are gathered in ONE round trip instead of a serial walk over their cache lines); its first lane writes.
template <bool WAVE = false>
__device__ __forceinline__ void write_result(const ObjParams &op, const double *stat, double *__restrict__ result) {
    const double npix = region_pixels(op.H, op.W, op.omit);
    const double v_orig = op.normalized ? orig_value<WAVE>(op, stat) : 0.0;
    const bool writer = !WAVE || (threadIdx.x & (kWave - 1)) == 0;
    double loss = 0.0;
    for (int k = 0; k < op.n_ref; ++k) {
        double acc[2];
        stat_sum<WAVE>(stat, k, op.nsub, acc);
        const double v = contrast_value(op.cost, acc, npix, nullptr);
        if (writer) result[1 + k] = v;
        if (!op.normalized) loss += op.mult[k] * (op.minimize ? -v : v);
        else loss += op.mult[k] * (op.minimize ? v_orig / v : v / v_orig);
    }
    if (writer) {
        result[0] = op.negate ? -loss : loss;
        result[5] = v_orig;
    }
}

// THR: 256, or 1024 in deterministic mode (kStatSub workgroups, one per accumulator: they have to be large)
template <int COST, int THR = 256>
__global__ void __launch_bounds__(THR)
k_stats(const float *__restrict__ img, int H, int W, int omit, int nsub, double *__restrict__ stat_slot, float *__restrict__ zero_img,
        float4 *__restrict__ zero_extra, int64_t n_extra4, int64_t bs = 0) {
    __shared__ double smem[2 * (THR / 64)];
    img += blockIdx.y * bs;  // blockIdx.y: image of a batch (element stride bs)
    stat_slot += blockIdx.y * kStatStride;
    if (zero_img) zero_img += blockIdx.y * bs;
    // the flow-gradient buffer K3 accumulates into is cleared here (a hipMemsetAsync node costs 4-5 us)
    const int64_t gtid = (int64_t)blockIdx.x * THR + threadIdx.x, gthreads = (int64_t)gridDim.x * THR;
    const unsigned npix = (unsigned)H * (unsigned)W;
    const int i0 = omit ? 1 : 0;
    const unsigned stride = gridDim.x * (unsigned)THR;
    double v[2] = {0.0, 0.0};
    for (unsigned base = blockIdx.x * (unsigned)THR + threadIdx.x; base < npix; base += 4u * stride) {
        float x[4];
        bool in[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // 4 independent loads in flight per thread
            const unsigned p = base + (unsigned)u * stride;
            const unsigned r = p / (unsigned)W, c = p - r * (unsigned)W;
            in[u] = p < npix && (int)r >= i0 && (int)r < H - i0 && (int)c >= i0 && (int)c < W - i0;
            x[u] = 0.f;
            if (COST == CMAX_COST_VARIANCE) x[u] = img[p < npix ? p : npix - 1u];  // unconditional: see pin()
        }
        if (COST == CMAX_COST_VARIANCE) {
#pragma unroll
            for (int u = 0; u < 4; ++u) pin(x[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = in[u] ? x[u] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (COST == CMAX_COST_VARIANCE) {
                const double xd = (double)x[u];
                v[0] += xd;
                v[1] += xd * xd;
            } else if (in[u]) {
                const unsigned p = base + (unsigned)u * stride;
                const int r = (int)(p / (unsigned)W), c = (int)(p - (unsigned)r * (unsigned)W);
                float gx, gy;
                sobel8_f32(img, H, W, r, c, gx, gy);
                v[0] += (double)(gx * gx + gy * gy);
            }
        }
    }
    // The clearing stores go out AFTER the image loads have been consumed: vmcnt retires in issue order, so a
    // write-through store issued first sits on the critical path of every load behind it (k_stats 6.2 -> see profiles)
    if (blockIdx.y == 0) zero_fill_sc1((float *)zero_extra, 4 * n_extra4, gtid, gthreads);
    zero_fill_sc1(zero_img, (int64_t)H * W, gtid, gthreads);
    block_sum<2>(v, smem);
    if (threadIdx.x == 0) {
        double *a = stat_slot + kSubStride * (blockIdx.x % nsub);
        atomic_add(&a[0], v[0]);
        if (COST == CMAX_COST_VARIANCE) atomic_add(&a[1], v[1]);
    }
}

// K2 + K2b in one pass for the gradient-magnitude cost: statistics as k_stats<GRADMAG>, and the UNSCALED
// G' = (2 / n) / 8 * Sobel^T (gx, gy) 1_Omega  -- the chain factor of the objective (which needs the statistics of
// every image) is applied by K3 when it loads its window (kFoldScale), and commutes with the blur transpose.
// Blurred variance cost in two image kernels instead of four (every dependent launch costs ~4.5 us):
//   k_blur_stats_var      Ib = blur3(I) (written: K3 / cmax_copy_iwe read it), sum Ib, sum Ib^2 over Omega, zeroing
//   k_gimage_blur_adj_var G = blur3^T [ c (Ib - mu) 1_Omega ]  with c, mu from the finished statistics
__global__ void __launch_bounds__(256)
k_blur_stats_var(const float *__restrict__ in0, int64_t in_stride, int H, int W, ImgArgs ia, float k0, float k1, int omit, int nsub, double *__restrict__ stat_base,
                 float4 *__restrict__ zero_extra, int64_t n_extra4) {
    __shared__ double smem[2 * 4];
    const float *__restrict__ img = in0 + blockIdx.y * in_stride;  // (leading, preloaded arguments: the tile loads leave without waiting for the argument block)
    float *__restrict__ blurred = ia.blurred[blockIdx.y];
    float *__restrict__ zero_img = ia.zero[blockIdx.y];
    double *__restrict__ stat_slot = stat_base + blockIdx.y * kStatStride;
    const int64_t gtid = (int64_t)blockIdx.x * 256 + threadIdx.x, gthreads = (int64_t)gridDim.x * 256;
    const unsigned npix = (unsigned)H * (unsigned)W;
    const int i0 = omit ? 1 : 0;
    double v[2] = {0.0, 0.0};
    for (unsigned p = blockIdx.x * 256u + threadIdx.x; p < npix; p += gridDim.x * 256u) {
        const int i = (int)(p / (unsigned)W), j = (int)(p - (unsigned)i * (unsigned)W);
        const int im = refl101(i - 1, H), ip = refl101(i + 1, H), jm = refl101(j - 1, W), jp = refl101(j + 1, W);
        auto row = [&](int r) { return k1 * img[(int64_t)r * W + jm] + k0 * img[(int64_t)r * W + j] + k1 * img[(int64_t)r * W + jp]; };
        const float b = k1 * row(im) + k0 * row(i) + k1 * row(ip);  // same arithmetic as k_blur3
        blurred[p] = b;
        if (i >= i0 && i < H - i0 && j >= i0 && j < W - i0) {
            v[0] += (double)b;
            v[1] += (double)b * (double)b;
        }
    }
    if (blockIdx.y == 0) zero_fill_sc1((float *)zero_extra, 4 * n_extra4, gtid, gthreads);  // behind the loads (see k_stats)
    zero_fill_sc1(zero_img, (int64_t)H * W, gtid, gthreads);
    block_sum<2>(v, smem);
    if (threadIdx.x == 0) {
        double *a = stat_slot + kSubStride * (blockIdx.x % nsub);
        atomic_add(&a[0], v[0]);
        atomic_add(&a[1], v[1]);
    }
}

constexpr int kGmTileH = 8, kGmTileW = 32;  // pixels per workgroup of k_stats_gimage_gm (256 threads, one pixel each)
__global__ void __launch_bounds__(256)
k_stats_gimage_gm(const float *__restrict__ in0, int64_t in_stride, int H, int W, ImgArgs ia, int omit, int nsub, double *__restrict__ stat_base, float4 *__restrict__ zero_extra,
                  int64_t n_extra4) {
    __shared__ double smem[2 * 4];
    __shared__ float tile[kGmTileH + 4][kGmTileW + 4 + 1];  // image tile with a halo of 2 (zero outside the image)
    const float *__restrict__ img = in0 + blockIdx.y * in_stride;  // (leading, preloaded arguments: the tile loads leave without waiting for the argument block)
    float *__restrict__ zero_img = ia.zero[blockIdx.y];
    float *__restrict__ G = ia.G[blockIdx.y];
    double *__restrict__ stat_slot = stat_base + blockIdx.y * kStatStride;
    const int64_t gtid = (int64_t)blockIdx.x * 256 + threadIdx.x, gthreads = (int64_t)gridDim.x * 256;
    const int tiles_w = (W + kGmTileW - 1) / kGmTileW;
    const int tr = blockIdx.x / tiles_w, tc = blockIdx.x - tr * tiles_w;
    const int r0 = tr * kGmTileH - 2, c0 = tc * kGmTileW - 2;
    CMAX_STAMP(2, 0);
    stage_tile<kGmTileH + 4, kGmTileW + 4>(tile, img, r0, c0, H, W);
    CMAX_STAMP_AFTER(2, 1, tile[0][0]);
    __syncthreads();
    CMAX_STAMP(2, 2);
    if (blockIdx.y == 0) zero_fill_sc1((float *)zero_extra, 4 * n_extra4, gtid, gthreads);  // behind the loads (see k_stats)
    zero_fill_sc1(zero_img, (int64_t)H * W, gtid, gthreads);
    CMAX_STAMP(2, 3);
    const int i0 = omit ? 1 : 0;
    const float gscale = (float)((2.0 / region_pixels(H, W, omit)) / 8.0) * ia.chain[blockIdx.y];
    const int la = threadIdx.x / kGmTileW, lb = threadIdx.x - la * kGmTileW;  // pixel of this thread inside the tile
    const int i = r0 + 2 + la, j = c0 + 2 + lb;
    double v[2] = {0.0, 0.0};
    // A tile whose pixels all have their 3 x 3 neighbourhood inside Omega (every tile but those along the border): the
    // transpose of the Sobel pair applied to its own responses is ONE 5 x 5 stencil on the image,
    //   Sobel^T Sobel = A(dr) W(dc) + W(dr) A(dc),  A = autocorrelation of (-1, 0, 1) = (-1, 0, 2, 0, -1),  W = that of (1, 2, 1) = (1, 4, 6, 4, 1)
    // -- 25 LDS reads per pixel instead of the 108 of the 18 responses below (workgroup-uniform branch; k_stats_gimage_gm of
    // cfg3 5.7 -> see profiles/r02_ablation.txt).
    const int ri = (int)blockIdx.x / tiles_w * kGmTileH, ci = tc * kGmTileW;
    const bool interior = ri >= i0 + 1 && ri + kGmTileH - 1 <= H - i0 - 2 && ci >= i0 + 1 && ci + kGmTileW - 1 <= W - i0 - 2;
    if (interior) {
        float t5[5][5];
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = 0; b < 5; ++b) t5[a][b] = tile[la + a][lb + b];
        const float gxc = ((t5[3][1] + 2.f * t5[3][2] + t5[3][3]) - (t5[1][1] + 2.f * t5[1][2] + t5[1][3])) * 0.125f;
        const float gyc = ((t5[1][3] + 2.f * t5[2][3] + t5[3][3]) - (t5[1][1] + 2.f * t5[2][1] + t5[3][1])) * 0.125f;
        const float r_outer = -2.f * (t5[0][0] + t5[0][4] + t5[4][0] + t5[4][4]) - 4.f * (t5[0][1] + t5[0][2] + t5[0][3] + t5[4][1] + t5[4][2] + t5[4][3]);
        const float r_mid = -4.f * (t5[1][0] + t5[1][4] + t5[3][0] + t5[3][4]) + 8.f * (t5[1][2] + t5[3][2]);
        const float r_centre = -4.f * (t5[2][0] + t5[2][4]) + 8.f * (t5[2][1] + t5[2][3]) + 24.f * t5[2][2];
        G[(int64_t)i * W + j] = gscale * 0.125f * (r_outer + r_mid + r_centre);
        v[0] = (double)(gxc * gxc + gyc * gyc);
    } else {
        // Tiles along the border (workgroup-uniform; round 6).  They were the kernel's span: every thread computed the 18 Sobel responses of its
        // 3 x 3 neighbourhood (108 LDS reads) and added them under per-tap tests of Omega.  Now in two steps on LDS tiles: the responses of the
        // tile's pixels and a ring of one around them, ZERO outside Omega, once per pixel (12 reads); then the transpose as a plain 3 x 3
        // correlation (18 reads, no tests) -- the same products in the same order, the skipped ones now zeros.
        __shared__ float t_gx[kGmTileH + 2][kGmTileW + 2 + 1], t_gy[kGmTileH + 2][kGmTileW + 2 + 1];
        for (int q = threadIdx.x; q < (kGmTileH + 2) * (kGmTileW + 2); q += 256) {
            const int y = q / (kGmTileW + 2), x = q - y * (kGmTileW + 2);  // response of pixel (r0 + 1 + y, c0 + 1 + x): 3 x 3 window with top-left tile[y][x]
            const float(*t)[kGmTileW + 4 + 1] = tile;
            const int qi = r0 + 1 + y, qj = c0 + 1 + x;
            const bool in_om = qi >= i0 && qi < H - i0 && qj >= i0 && qj < W - i0;
            const float rx = ((t[y + 2][x] + 2.f * t[y + 2][x + 1] + t[y + 2][x + 2]) - (t[y][x] + 2.f * t[y][x + 1] + t[y][x + 2])) * 0.125f;
            const float ry = ((t[y][x + 2] + 2.f * t[y + 1][x + 2] + t[y + 2][x + 2]) - (t[y][x] + 2.f * t[y + 1][x] + t[y + 2][x])) * 0.125f;
            t_gx[y][x] = in_om ? rx : 0.f;
            t_gy[y][x] = in_om ? ry : 0.f;
        }
        __syncthreads();
        // transpose: output pixel q = (i - a, j - b) reads this pixel with tap (a, b); q outside Omega holds zeros
        float s = 0.f;
#pragma unroll
        for (int a = -1; a <= 1; ++a)
#pragma unroll
            for (int b = -1; b <= 1; ++b) {
                const float sx = (float)a * (b == 0 ? 2.f : 1.f);  // SX[a+1][b+1] = a * (2 - |b|)
                const float sy = (float)b * (a == 0 ? 2.f : 1.f);  // SY[a+1][b+1] = b * (2 - |a|)
                s += t_gx[la + 1 - a][lb + 1 - b] * sx + t_gy[la + 1 - a][lb + 1 - b] * sy;
            }
        if (i < H && j < W) G[(int64_t)i * W + j] = gscale * s;
        const float gxc = t_gx[la + 1][lb + 1], gyc = t_gy[la + 1][lb + 1];  // (zero outside Omega, the image included)
        v[0] = (double)(gxc * gxc + gyc * gyc);
    }
    CMAX_STAMP(2, 5);
    block_sum<2>(v, smem);
    CMAX_STAMP(2, 6);
    if (threadIdx.x == 0) atomic_add(&stat_slot[kSubStride * (blockIdx.x % nsub)], v[0]);
    CMAX_STAMP(2, 7);
}

// Blurred gradient-magnitude cost (the shipped YAML cost), whole image side of one reference time in ONE kernel:
//   Ib = blur3(I)  ->  statistics sum (gx^2 + gy^2) over Omega  ->  G' = (2/n)/8 Sobel^T(gx, gy) 1_Omega  ->  G = blur3^T G'
// (k_blur3 + k_stats_gimage_gm + k_blur3_adj = three dependent launches of ~4.5 us each on a 260 x 346 image).
// One 8 x 32 output tile per workgroup; the stages shrink a halo of 4 -> 3 -> 2 -> 1 -> 0 in LDS.  Arithmetic
// order as in the separate kernels.  G leaves without its chain factor (kFoldScale).
__global__ void __launch_bounds__(256)
k_blur_stats_gimage_gm(const float *__restrict__ in0, int64_t in_stride, int H, int W, ImgArgs ia, float k0, float k1, int omit, int nsub, double *__restrict__ stat_base,
                       float4 *__restrict__ zero_extra, int64_t n_extra4) {
    constexpr int TH = kGmTileH, TW = kGmTileW;
    const float *__restrict__ img = in0 + blockIdx.y * in_stride;  // (leading, preloaded arguments: the tile loads leave without waiting for the argument block)
    float *__restrict__ blurred = ia.blurred[blockIdx.y];
    float *__restrict__ zero_img = ia.zero[blockIdx.y];
    float *__restrict__ G = ia.G[blockIdx.y];
    double *__restrict__ stat_slot = stat_base + blockIdx.y * kStatStride;
    if (blockIdx.y > 0) n_extra4 = 0;  // the flow-gradient buffer is cleared once
    __shared__ double smem[2 * 4];
    __shared__ float t_i[TH + 8][TW + 8 + 1];   // raw image, halo 4 (only in-image cells are read)
    __shared__ float t_b[TH + 6][TW + 6 + 1];   // blurred image, halo 3, zero outside the image (Sobel zero padding)
    __shared__ float t_gx[TH + 4][TW + 4 + 1];  // Sobel/8 responses, halo 2
    __shared__ float t_gy[TH + 4][TW + 4 + 1];
    __shared__ float t_g[TH + 2][TW + 2 + 1];   // G', halo 1, zero outside the image
    const int64_t gtid = (int64_t)blockIdx.x * 256 + threadIdx.x, gthreads = (int64_t)gridDim.x * 256;
    const int tiles_w = (W + TW - 1) / TW;
    const int tr = blockIdx.x / tiles_w, tc = blockIdx.x - tr * tiles_w;
    const int R0 = tr * TH, C0 = tc * TW;  // top-left output pixel
    const int i0 = omit ? 1 : 0;
    auto in_img = [&](int r, int c) { return (unsigned)r < (unsigned)H && (unsigned)c < (unsigned)W; };
    stage_tile<TH + 8, TW + 8>(t_i, img, R0 - 4, C0 - 4, H, W);
    __syncthreads();
    zero_fill_sc1((float *)zero_extra, 4 * n_extra4, gtid, gthreads);  // behind the loads (see k_stats)
    zero_fill_sc1(zero_img, (int64_t)H * W, gtid, gthreads);
    for (int q = threadIdx.x; q < (TH + 6) * (TW + 6); q += 256) {
        const int a = q / (TW + 6), b = q - a * (TW + 6), r = R0 - 3 + a, c = C0 - 3 + b;
        float v = 0.f;
        if (in_img(r, c)) {  // reflect-101 neighbours are at most one pixel away: inside the halo-4 tile
            const int rm = refl101(r - 1, H) - (R0 - 4), rr = r - (R0 - 4), rp = refl101(r + 1, H) - (R0 - 4);
            const int cm = refl101(c - 1, W) - (C0 - 4), cc = c - (C0 - 4), cp = refl101(c + 1, W) - (C0 - 4);
            auto row = [&](int y) { return k1 * t_i[y][cm] + k0 * t_i[y][cc] + k1 * t_i[y][cp]; };
            v = k1 * row(rm) + k0 * row(rr) + k1 * row(rp);
        }
        t_b[a][b] = v;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < (TH + 4) * (TW + 4); q += 256) {
        const int a = q / (TW + 4), b = q - a * (TW + 4);  // response at (R0 - 2 + a, C0 - 2 + b): t_b window rows a..a+2, cols b..b+2
        t_gx[a][b] = ((t_b[a + 2][b] + 2.f * t_b[a + 2][b + 1] + t_b[a + 2][b + 2]) - (t_b[a][b] + 2.f * t_b[a][b + 1] + t_b[a][b + 2])) * 0.125f;
        t_gy[a][b] = ((t_b[a][b + 2] + 2.f * t_b[a + 1][b + 2] + t_b[a + 2][b + 2]) - (t_b[a][b] + 2.f * t_b[a + 1][b] + t_b[a + 2][b])) * 0.125f;
    }
    __syncthreads();
    const float gscale = (float)((2.0 / region_pixels(H, W, omit)) / 8.0) * ia.chain[blockIdx.y];
    for (int q = threadIdx.x; q < (TH + 2) * (TW + 2); q += 256) {
        const int a = q / (TW + 2), b = q - a * (TW + 2), r = R0 - 1 + a, c = C0 - 1 + b;
        float sum = 0.f;
        if (in_img(r, c)) {
#pragma unroll
            for (int da = -1; da <= 1; ++da)
#pragma unroll
                for (int db = -1; db <= 1; ++db) {
                    const int qi = r - da, qj = c - db;  // output pixel of the Sobel that reads (r, c) with tap (da, db)
                    if (qi < i0 || qi >= H - i0 || qj < i0 || qj >= W - i0) continue;
                    const float sx = (float)da * (db == 0 ? 2.f : 1.f), sy = (float)db * (da == 0 ? 2.f : 1.f);
                    sum += t_gx[a + 1 - da][b + 1 - db] * sx + t_gy[a + 1 - da][b + 1 - db] * sy;  // (qi, qj) in halo-2 coordinates
                }
        }
        t_g[a][b] = gscale * sum;
    }
    __syncthreads();
    const int la = threadIdx.x / TW, lb = threadIdx.x - la * TW;
    const int i = R0 + la, j = C0 + lb;
    double v[2] = {0.0, 0.0};
    if (i < H && j < W) {
        const int64_t p = (int64_t)i * W + j;
        blurred[p] = t_b[la + 3][lb + 3];
        G[p] = blur_adj_1d<float>(i, H, k0, k1, [&](int r) {
            return blur_adj_1d<float>(j, W, k0, k1, [&](int c) { return t_g[r - (R0 - 1)][c - (C0 - 1)]; });
        });
        if (i >= i0 && i < H - i0 && j >= i0 && j < W - i0) {
            const float gx = t_gx[la + 2][lb + 2], gy = t_gy[la + 2][lb + 2];
            v[0] = (double)(gx * gx + gy * gy);
        }
    }
    block_sum<2>(v, smem);
    if (threadIdx.x == 0) atomic_add(&stat_slot[kSubStride * (blockIdx.x % nsub)], v[0]);
}

// Blurred variance cost with a gradient, whole image side of one reference time in ONE kernel (k_blur_stats_var +
// k_gimage_blur_adj_var are two dependent launches of ~4.7 + 5.0 us on a 260 x 346 image):
//   G = c blur3^T [ (Ib - mu) 1_Omega ],  c = coef 2 / (n - 1)
// needs the mean of the blurred image before the transpose can start.  The blur is linear: sum_Omega Ib = sum_p I[p] B[p] with
// B = blur3^T 1_Omega = b(r) b(c), b = 1 except within two pixels of the border (border_weight) -- a sum over the VOTES: the number
// of events, plus what K1's workgroups along the border add up (RefArgs::musum).  The mean only matters where B or the mask
// 1_Omega vary (everywhere else the gather's differences cancel it), so the 2^-20 rounding of the fixed-point weights inside
// "four weights add up to 1" is immaterial.  So this kernel reads mu, blurs, sums (for the loss) and writes
// G' = 2 / (n - 1) blur3^T [ (Ib - mu) 1_Omega ]; K3 (kFoldScale) multiplies by coef once the statistics are complete.
// One 8 x 32 output tile per workgroup, halo 2 -> 1 -> 0 in LDS; arithmetic of the blur as in k_blur3.
constexpr int kMuLines = 32;                       // accumulators of K1's sum (same-line atomics serialise: one line each)
constexpr int kMuStride = kMuLines * kSubStride;   // doubles per reference time
__global__ void __launch_bounds__(256)
k_blur_stats_adj_var(const float *__restrict__ in0, int64_t in_stride, int H, int W, ImgArgs ia, float k0, float k1, int omit, int nsub, double *__restrict__ stat_base,
                     float4 *__restrict__ zero_extra, int64_t n_extra4, const double *__restrict__ musum, double *__restrict__ musum_next,
                     int64_t n_events) {
    constexpr int TH = 8, TW = 32;
    __shared__ float s_mu;
    if (threadIdx.x < kWave) {  // issued first: the latency hides behind the tile loads
        double m = threadIdx.x < kMuLines ? musum[blockIdx.y * kMuStride + threadIdx.x * kSubStride] : 0.0;
        m = wave_sum_lane63(m);
        if (threadIdx.x == kWave - 1) s_mu = (float)(((double)n_events + m) / region_pixels(H, W, omit));
        // the accumulators of the NEXT evaluation (the other buffer: nobody reads or adds to it during this one), written through
        if (blockIdx.x == 0 && threadIdx.x < kMuLines)
            __hip_atomic_store(&musum_next[blockIdx.y * kMuStride + threadIdx.x * kSubStride], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const float *__restrict__ img = in0 + blockIdx.y * in_stride;  // (leading, preloaded arguments: the tile loads leave without waiting for the argument block)
    float *__restrict__ blurred = ia.blurred[blockIdx.y];
    float *__restrict__ zero_img = ia.zero[blockIdx.y];
    float *__restrict__ G = ia.G[blockIdx.y];
    double *__restrict__ stat_slot = stat_base + blockIdx.y * kStatStride;
    if (blockIdx.y > 0) n_extra4 = 0;  // the flow-gradient buffer is cleared once
    __shared__ double smem[2 * 4];
    __shared__ float t_i[TH + 4][TW + 4 + 1];  // raw image, halo 2 (only in-image cells are read)
    __shared__ float t_b[TH + 2][TW + 2 + 1];  // blurred image, halo 1
    const int64_t gtid = (int64_t)blockIdx.x * 256 + threadIdx.x, gthreads = (int64_t)gridDim.x * 256;
    const int tiles_w = (W + TW - 1) / TW;
    const int tr = blockIdx.x / tiles_w, tc = blockIdx.x - tr * tiles_w;
    const int R0 = tr * TH, C0 = tc * TW;  // top-left output pixel
    const int i0 = omit ? 1 : 0;
    auto in_img = [&](int r, int c) { return (unsigned)r < (unsigned)H && (unsigned)c < (unsigned)W; };
    CMAX_STAMP(2, 0);
    // Round 6: BRANCH-FREE STENCILS.  tools/timeline.py had the two stencil phases of this kernel at 1.0 + 1.5 us of its workgroups' 4.8
    // (predicated taps: reflection, image and Omega tests per tap, each an LDS round trip behind a branch), and the tiles along the border
    // -- the slowest workgroups ARE the kernel's span -- at the same cost as everybody else.  Now the border is resolved where the tiles
    // are filled: the raw tile is staged WITH its reflect-101 halo (in-image reads only), the forward blur is a plain 3 x 3 stencil, the
    // masked, centred image d = (Ib - mu) 1_Omega gets a tile of its own, and blur^T is a 3 x 3 stencil on d whose per-axis tap weights
    // (k1, k0, k1; 0 beyond the image; 2 k1 for the reflected taps folded back onto rows / columns 1 and n - 2) are set up once per thread.
    {
        constexpr int kN = (TH + 4) * (TW + 4), kIt = (kN + 255) / 256;
        float x[kIt];
#pragma unroll
        for (int u = 0; u < kIt; ++u) {
            const int q = min((int)threadIdx.x + 256 * u, kN - 1), a = q / (TW + 4), b = q - a * (TW + 4);
            const int r = refl101(min(max(R0 - 2 + a, -2), H + 1), H), c = refl101(min(max(C0 - 2 + b, -2), W + 1), W);  // (H, W >= 4: the host's condition for this kernel)
            x[u] = img[(int64_t)r * W + c];
        }
#pragma unroll
        for (int u = 0; u < kIt; ++u) pin(x[u]);  // all loads in flight before the first LDS store (see pin())
#pragma unroll
        for (int u = 0; u < kIt; ++u) {
            const int q = threadIdx.x + 256 * u, a = q / (TW + 4), b = q - a * (TW + 4);
            if (q < kN) t_i[a][b] = x[u];
        }
    }
    CMAX_STAMP_AFTER(2, 1, t_i[0][0]);
    __syncthreads();
    CMAX_STAMP(2, 2);
    zero_fill_sc1((float *)zero_extra, 4 * n_extra4, gtid, gthreads);  // behind the loads (see k_stats)
    zero_fill_sc1(zero_img, (int64_t)H * W, gtid, gthreads);
    __shared__ float t_d[TH + 2][TW + 2 + 1];  // (Ib - mu) 1_Omega, halo 1
    {
        const float mu = s_mu;  // (published by the barrier above)
        for (int q = threadIdx.x; q < (TH + 2) * (TW + 2); q += 256) {
            const int a = q / (TW + 2), b = q - a * (TW + 2), r = R0 - 1 + a, c = C0 - 1 + b;  // t_b[a][b] = pixel (r, c) = t_i[a + 1][b + 1]
            auto row = [&](int y) { return k1 * t_i[y][b] + k0 * t_i[y][b + 1] + k1 * t_i[y][b + 2]; };
            const float vb = k1 * row(a) + k0 * row(a + 1) + k1 * row(a + 2);  // (the products and their order are k_blur3's)
            const bool in_om = r >= i0 && r < H - i0 && c >= i0 && c < W - i0;
            t_b[a][b] = in_img(r, c) ? vb : 0.f;
            t_d[a][b] = in_om ? vb - mu : 0.f;
        }
    }
    CMAX_STAMP(2, 3);
    __syncthreads();
    CMAX_STAMP(2, 4);
    const float gscale = (float)(2.0 / (region_pixels(H, W, omit) - 1.0)) * ia.chain[blockIdx.y];
    const int la = threadIdx.x / TW, lb = threadIdx.x - la * TW;
    const int i = R0 + la, j = C0 + lb;
    double v[2] = {0.0, 0.0};
    if (i < H && j < W) {
        const int64_t p = (int64_t)i * W + j;
        const float b = t_b[la + 1][lb + 1];
        blurred[p] = b;
        // blur^T along one axis of length n at index q: weight of d[q - 1] and of d[q + 1] (blur_adj_1d: a tap beyond the image does not exist;
        // rows / columns 1 and n - 2 also receive the reflected tap of their outer neighbour)
        const float rm = i >= 1 ? (i == 1 ? 2.f * k1 : k1) : 0.f, rp = i + 1 < H ? (i == H - 2 ? 2.f * k1 : k1) : 0.f;
        const float cm = j >= 1 ? (j == 1 ? 2.f * k1 : k1) : 0.f, cp = j + 1 < W ? (j == W - 2 ? 2.f * k1 : k1) : 0.f;
        auto rowadj = [&](int y) { return k0 * t_d[y][lb + 1] + cm * t_d[y][lb] + cp * t_d[y][lb + 2]; };
        G[p] = gscale * (k0 * rowadj(la + 1) + rm * rowadj(la) + rp * rowadj(la + 2));
        if (i >= i0 && i < H - i0 && j >= i0 && j < W - i0) {
            v[0] = (double)b;
            v[1] = (double)b * (double)b;
        }
    }
    CMAX_STAMP(2, 5);
    block_sum<2>(v, smem);
    CMAX_STAMP(2, 6);
    if (threadIdx.x == 0) {
        double *a = stat_slot + kSubStride * (blockIdx.x % nsub);
        atomic_add(&a[0], v[0]);
        atomic_add(&a[1], v[1]);
    }
    CMAX_STAMP(2, 7);
}

// b(p) of B = blur3^T 1_Omega along one axis of length n >= 4 (Omega = [i0, n - i0)): b(0) = b(n - 1), b(1) = b(n - 2), 1 elsewhere
static float border_weight(int p, int n, int i0, float k0, float k1) {
    auto one = [&](int r) { return (r >= i0 && r < n - i0) ? 1.f : 0.f; };
    float s = k0 * one(p);
    if (p - 1 >= 0) s += k1 * one(p - 1);
    if (p + 1 < n) s += k1 * one(p + 1);
    if (p == 1) s += k1 * one(0);      // the reflected taps of blur_adj_1d
    if (p == n - 2) s += k1 * one(n - 1);
    return s;
}

// 2-DoF, plain variance, tangent images (k_vote_tan2): everything the loss and the gradient need, in image space.
//   With m = 1_Omega (zero outside the image) and G = c (I - mu) m:
//     dL/dtheta0 = sum_q E0[q] (G~[q + (1,0)] - G~[q]) = c (S1x - mu S2x),
//       S1x = sum_q E0[q] (m I)[q + (1,0)] - (m I)[q],   S2x = sum_q E0[q] (m[q + (1,0)] - m[q]),   likewise S1y, S2y with F1
//   -- the six sums k_finish_deferred turns into loss, chain factors and gradient (the same quantities the deferred K3
//   gathers per event).  grid (kTanBlocks, n_ref); per-block partials [k][block][6]; clears the planes of the next evaluation.
constexpr int kTanBlocks = 128;
__global__ void __launch_bounds__(256)
k_tan_stats_var(const float *__restrict__ T, float *__restrict__ Tnext, int64_t tstride, int Hp, int Wp, int omit, int normalize,
                const double *__restrict__ tmm, double *__restrict__ part) {
    __shared__ double smem[6 * 4];
    const int k = blockIdx.y;
    const float *__restrict__ I = T + k * tstride;
    const float *__restrict__ E0 = I + (int64_t)Hp * Wp;        // row r of the image is stored row r + 1
    const float *__restrict__ F1 = E0 + (int64_t)(Hp + 1) * Wp;  // column c is stored column c + 1, row stride Wp + 1
    const float ts = normalize ? 1.f : (float)(tmm[1] - tmm[0]);  // the votes carry dt / period
    const int i0 = omit ? 1 : 0;
    const int npix = Hp * Wp;
    auto mask = [&](int r, int c) -> bool { return r >= i0 && r < Hp - i0 && c >= i0 && c < Wp - i0; };
    double v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int p = blockIdx.x * 256 + threadIdx.x; p < npix; p += gridDim.x * 256) {
        const int r = p / Wp, c = p - r * Wp;
        const bool m = mask(r, c), md = r + 1 < Hp && mask(r + 1, c), mr = c + 1 < Wp && mask(r, c + 1);
        const float x = m ? I[p] : 0.f, xd = md ? I[p + Wp] : 0.f, xr = mr ? I[p + 1] : 0.f;
        const float e = E0[(int64_t)(r + 1) * Wp + c] * ts, f = F1[(int64_t)r * (Wp + 1) + c + 1] * ts;
        v[0] += (double)(e * (xd - x));
        v[1] += (double)(f * (xr - x));
        v[2] += (double)(e * ((md ? 1.f : 0.f) - (m ? 1.f : 0.f)));
        v[3] += (double)(f * ((mr ? 1.f : 0.f) - (m ? 1.f : 0.f)));
        if (r == 0) {  // votes whose upper corner lies one row above the image
            const float eu = E0[c] * ts;
            v[0] += (double)(eu * x);
            v[2] += (double)(eu * (m ? 1.f : 0.f));
        }
        if (c == 0) {  // ... one column left of it
            const float fl = F1[(int64_t)r * (Wp + 1)] * ts;
            v[1] += (double)(fl * x);
            v[3] += (double)(fl * (m ? 1.f : 0.f));
        }
        v[4] += (double)x;
        v[5] += (double)x * (double)x;
    }
    if (Tnext) zero_fill_sc1(Tnext + k * tstride, tstride, (int64_t)blockIdx.x * 256 + threadIdx.x, (int64_t)gridDim.x * 256);
    // the four gradient sums go over the wave in fp32 (DPP adds; their terms carry fp32 rounding already) and in fp64 across
    // waves and workgroups, the two image sums stay fp64 throughout -- as in the deferred K3 (a block_sum of six doubles is
    // 36 dependent v_add_f64 behind 72 DPP moves)
    float *s_f = reinterpret_cast<float *>(smem + 2 * 4);  // [4][4] floats behind the doubles of block_sum<2>
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float w = (float)v[q];
        w += dpp_f<kDppRowShr1>(w);
        w += dpp_f<kDppRowShr2>(w);
        w += dpp_f<kDppRowShr4>(w);
        w += dpp_f<kDppRowShr8>(w);
        w += dpp_f<kDppRowBcast15>(w);
        w += dpp_f<kDppRowBcast31>(w);
        if (lane == kWave - 1) s_f[q * 4 + wave] = w;
    }
    double im[2] = {v[4], v[5]};
    block_sum<2>(im, smem);  // (its barriers also publish s_f)
    if (threadIdx.x == 0) {
        double *o = part + ((int64_t)k * gridDim.x + blockIdx.x) * 6;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = (double)s_f[q * 4] + (double)s_f[q * 4 + 1] + (double)s_f[q * 4 + 2] + (double)s_f[q * 4 + 3];
        o[4] = im[0];
        o[5] = im[1];
    }
}

__global__ void k_finalize(ObjParams op, const double *__restrict__ stat, double *__restrict__ result) {
    if (threadIdx.x == 0 && blockIdx.x == 0) write_result(op, stat, result);
}

// K2b: G[p] = dL/dv_k * dv_k/dI[p]   (needed when a blur transpose follows or for the grad-mag cost;
//      the plain-variance gradient is folded into K3 instead)
template <int COST>
__global__ void __launch_bounds__(256)
k_gimage(const float *__restrict__ img, ObjParams op, int k, const double *__restrict__ stat, float *__restrict__ G, int64_t bs = 0) {
    k += blockIdx.y;  // blockIdx.y: reference time of a batch (element stride bs)
    img += blockIdx.y * bs;
    G += blockIdx.y * bs;
    const int H = op.H, W = op.W;
    const int i0 = op.omit ? 1 : 0;
    const double npix = region_pixels(H, W, op.omit);
    double mu = 0.0;
    const double coef = chain_coef<true>(op, stat, k, &mu);  // every wave is converged here
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (int64_t)H * W) return;
    const int i = (int)(p / W), j = (int)(p % W);
    if (COST == CMAX_COST_VARIANCE) {
        const bool in = (i >= i0) && (i < H - i0) && (j >= i0) && (j < W - i0);
        G[p] = in ? (float)(coef * 2.0 * ((double)img[p] - mu) / (npix - 1.0)) : 0.f;
    } else {
        G[p] = (float)(coef * (2.0 / npix) / 8.0) * sobel8_adj_f32(img, H, W, i0, i, j);
    }
}

__global__ void __launch_bounds__(256)
k_gimage_blur_adj_var(ImgArgs ia, ObjParams op, const double *__restrict__ stat, float k0, float k1) {
    const int k = blockIdx.y;
    const float *__restrict__ blurred = ia.in[k];
    float *__restrict__ G = ia.G[k];
    const int H = op.H, W = op.W, i0 = op.omit ? 1 : 0;
    double mud = 0.0;
    const double coef = chain_coef<true>(op, stat, k, &mud);  // every wave is converged here
    const float c2 = (float)(coef * 2.0 / (region_pixels(H, W, op.omit) - 1.0)), mu = (float)mud;
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (int64_t)H * W) return;
    const int i = (int)(p / W), j = (int)(p % W);
    auto g = [&](int r, int c) -> float {  // dL/d(blurred image): what k_gimage<VARIANCE> writes
        const bool in = (r >= i0) && (r < H - i0) && (c >= i0) && (c < W - i0);
        return in ? c2 * (blurred[(int64_t)r * W + c] - mu) : 0.f;
    };
    G[p] = blur_adj_1d<float>(i, H, k0, k1, [&](int r) { return blur_adj_1d<float>(j, W, k0, k1, [&](int c) { return g(r, c); }); });
}

// =============================================================================================
// Exact Hessian-vector product (a18): what torch.autograd.functional.vhp returns for the
// objective (src/solver/scipy_autograd/torch_wrapper.py:51-73) -- the derivative of the analytic
// gradient along a tangent motion u with the pixel cells held fixed (floor has zero derivative):
//   H u = sum_k [ phi_k'' dv_k J^T G0_k + phi_k' ( (dJ^T/dtheta u) G0_k + J^T (dG0_k/dI) J u ) ]
//   T1 k_vote_tan   tangent image dI = J u        (votes with the derivatives of the 4 bilinear weights)
//   T2 k_stats_tan  <G0, dI> and mean(dI)  ;  k_gimage_tan  G' = phi' dG0 + phi'' dv G0
//   T3 k_grad_hvp   per event: mixed term M (da, db) with the current G + gather of G' -> J^T
// The tangent is scaled to unit max-norm by the caller so the derivative votes fit fixed point.
// =============================================================================================
struct TanParams {
    const float *u;      // tangent motion, same layout as the motion, scaled to |u|_inf = 1
    float fix, inv_fix;  // fixed-point scale of the derivative votes (set per workgroup from the arrays below)
    // one launch covers the reference times of the objective (blockIdx.y):
    float d[4], fixk[4];  // reference time as a fraction of the batch period, its fixed-point scale
    int64_t bs;           // element stride between the per-reference-time images
};

// deterministic mode of the second-order gather (k_grad_hvp): all null / zero in the default mode
struct HvpDet {
    long long *g64;        // 2-DoF: per-segment partial sums [n_ref][nseg][2]; flow models: fixed-point accumulators of the product
    const unsigned *imax;  // bits of max |G_k| at [k], of max |G'_k| at [4 + k]
    double *inv_scale;     // [4] 1 / scale of reference time k (2-DoF) or of the whole launch ([0])
    long long n_events;
    int n_ref;
};

// (da, db) of cached event slot: d(x', y')/d(motion) . u
template <int MODEL>
__device__ __forceinline__ void tangent_delta(const WarpParams &wp, const TanParams &tp, float dt, unsigned key, float u0, float u1,
                                              float &da, float &db) {
    if (MODEL == CMAX_MODEL_2DOF) {
        da = dt * u0;  // x' = x + dt * theta0
        db = dt * u1;
    } else {
        const int hw = wp.H * wp.W;
        const int ix = (int)(key & 0xFFFu), iy = (int)((key >> 12) & 0xFFFu), bin = (int)(key >> 24);
        const int src = bin * 2 * hw + ix * wp.W + iy;
        da = -dt * tp.u[src];  // x' = x - dt * F[0, ix, iy]
        db = -dt * tp.u[src + hw];
    }
}

// T2a: st[0] += sum_Omega dI (variance) or sum_Omega (gx dgx + gy dgy) (grad-mag); st[1] += sum_Omega I dI
template <int COST>
__global__ void __launch_bounds__(256)
k_stats_tan(const float *__restrict__ img, const float *__restrict__ dimg, int H, int W, int omit, int nsub, double *__restrict__ st,
            int64_t bs) {
    __shared__ double smem[2 * 4];
    img += blockIdx.y * bs;
    dimg += blockIdx.y * bs;
    st += blockIdx.y * kStatStride;
    const unsigned npix = (unsigned)H * (unsigned)W;
    const int i0 = omit ? 1 : 0;
    double v[2] = {0.0, 0.0};
    for (unsigned p = blockIdx.x * 256u + threadIdx.x; p < npix; p += gridDim.x * 256u) {
        const int r = (int)(p / (unsigned)W), c = (int)(p - (unsigned)r * (unsigned)W);
        if (r < i0 || r >= H - i0 || c < i0 || c >= W - i0) continue;
        if (COST == CMAX_COST_VARIANCE) {
            const double d = (double)dimg[p];
            v[0] += d;
            v[1] += d * (double)img[p];
        } else {
            float gx, gy, hx, hy;
            sobel8_f32(img, H, W, r, c, gx, gy);
            sobel8_f32(dimg, H, W, r, c, hx, hy);
            v[0] += (double)(gx * hx + gy * hy);
        }
    }
    block_sum<2>(v, smem);
    if (threadIdx.x == 0) {
        double *a = st + kSubStride * (blockIdx.x % nsub);
        atomic_add(&a[0], v[0]);
        if (COST == CMAX_COST_VARIANCE) atomic_add(&a[1], v[1]);
    }
}

// T2b: G'[p] = phi' * dG0[dI] + phi'' * dv * G0[I]   (stat: sums of I; st: tangent sums, slot-local pointer)
template <int COST>
__global__ void __launch_bounds__(256)
k_gimage_tan(const float *__restrict__ img, const float *__restrict__ dimg, ObjParams op, int k, const double *__restrict__ stat,
             const double *__restrict__ st, float *__restrict__ Gp, int64_t bs) {
    k += blockIdx.y;
    img += blockIdx.y * bs;
    dimg += blockIdx.y * bs;
    Gp += blockIdx.y * bs;
    st += blockIdx.y * kStatStride;
    const int H = op.H, W = op.W, i0 = op.omit ? 1 : 0;
    const double npix = region_pixels(H, W, op.omit);
    double acc[2], tacc[2] = {0.0, 0.0};
    stat_sum(stat, k, op.nsub, acc);
    for (int u = 0; u < op.nsub; ++u) {
        tacc[0] += st[kSubStride * u];
        tacc[1] += st[kSubStride * u + 1];
    }
    double mu = 0.0;
    const double v = contrast_value(COST, acc, npix, &mu);
    // phi(v): loss contribution of this reference time; phi' = chain factor, phi'' its derivative
    double p1, p2 = 0.0;
    if (!op.normalized) p1 = op.mult[k] * (op.minimize ? -1.0 : 1.0);
    else {
        const double v_orig = orig_value(op, stat);
        if (op.minimize) {
            p1 = -op.mult[k] * v_orig / (v * v);
            p2 = 2.0 * op.mult[k] * v_orig / (v * v * v);
        } else p1 = op.mult[k] / v_orig;
    }
    if (op.negate) {
        p1 = -p1;
        p2 = -p2;
    }
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (int64_t)H * W) return;
    const int i = (int)(p / W), j = (int)(p % W);
    if (COST == CMAX_COST_VARIANCE) {
        const double dmean = tacc[0] / npix;
        const double dv = 2.0 / (npix - 1.0) * (tacc[1] - mu * tacc[0]);  // <G0, dI>
        const bool in = (i >= i0) && (i < H - i0) && (j >= i0) && (j < W - i0);
        Gp[p] = in ? (float)((p1 * 2.0 * ((double)dimg[p] - dmean) + p2 * dv * 2.0 * ((double)img[p] - mu)) / (npix - 1.0)) : 0.f;
    } else {
        const double dv = 2.0 / npix * tacc[0];
        const double c = (2.0 / npix) / 8.0;
        Gp[p] = (float)(p1 * c) * sobel8_adj_f32(dimg, H, W, i0, i, j) + (float)(p2 * dv * c) * sobel8_adj_f32(img, H, W, i0, i, j);
    }
}

// ---------------------------------------------------------------------------------------------
// deterministic mode: fixed-point conversions
// ---------------------------------------------------------------------------------------------
// power-of-two scale s with  bound * n * s < 2^61: n terms of magnitude <= bound fit a 64-bit accumulator
// (n counted as at least 2048: a single term then stays below 2^50 and det_fixed's conversion is exact)
__device__ __forceinline__ double det_scale(double bound, long long n) {
    int e = 0;
    (void)frexp(bound * (double)(n > 2048 ? n : 2048), &e);  // bound n = f 2^e, f in [0.5, 1); 0 -> e = 0
    if (!(bound >= 0.0) || e > 1000) e = 1000;         // NaN / inf: any finite scale
    return ldexp(1.0, 60 - e);
}
// rint(x * s) as a 64-bit integer for |x s| < 2^51: x s + 1.5 2^52 carries the rounded integer in its low mantissa bits (one fma
// + a 64-bit subtraction; __double2ll_rn is a ~25-instruction sequence on gfx950, twice per event in the deterministic K3)
__device__ __forceinline__ long long det_fixed(double x, double s) {
    const double magic = 6755399441055744.0;  // 1.5 * 2^52
    return __double_as_longlong(fma(x, s, magic)) - __double_as_longlong(magic);
}
__device__ __forceinline__ void atomic_add_i64(long long *p, long long v) {
    atomicAdd(reinterpret_cast<unsigned long long *>(p), (unsigned long long)v);  // two's complement: wrap-around add
}

// fixed-point vote images -> fp32 images (blockIdx.y = image), the integer image is left zero for the next evaluation
struct FixedArgs {
    long long *src[5];
    float *dst[5];
    double inv_fix;  // 1 / the vote fixed point of the kernels that filled src: 2^-20, big segments 2^-19
    double inv_fixk[5];  // != 0: per-image scale instead (tangent votes: one fixed point per reference time)
};
__global__ void __launch_bounds__(256) k_fixed_to_image(FixedArgs fa, int64_t npix) {
    long long *__restrict__ src = fa.src[blockIdx.y];
    float *__restrict__ dst = fa.dst[blockIdx.y];
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (int64_t)gridDim.x * 256) {
        const long long v = src[p];
        dst[p] = (float)((double)v * (fa.inv_fixk[blockIdx.y] != 0.0 ? fa.inv_fixk[blockIdx.y] : fa.inv_fix));  // one rounding, of the exact sum
        if (v != 0) src[p] = 0;
    }
}

// max |image| of one statistics slot (order-free: integer max of the magnitude bits); imax[slot] zeroed by the caller
__global__ void __launch_bounds__(256) k_image_absmax(ImgArgs ia, int64_t npix, int slot0, unsigned *__restrict__ imax) {
    // one atomicMax per WORKGROUP on the image's word, and at most 256 workgroups per image: same-address atomics serialise at
    // ~12 ns each -- one per wave of a 1200-workgroup grid was 56 of the deterministic cfg3 evaluation's 225 us
    __shared__ unsigned s_m[4];
    const float *__restrict__ img = ia.in[blockIdx.y];
    unsigned m = 0u;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (int64_t)gridDim.x * 256)
        m = max(m, __float_as_uint(img[p]) & 0x7FFFFFFFu);
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, kWave));
    if ((threadIdx.x & (kWave - 1)) == 0) s_m[threadIdx.x / kWave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
        if (m) atomicMax(&imax[slot0 + blockIdx.y], m);
    }
}
__global__ void __launch_bounds__(256) k_fixed_to_grad(long long *__restrict__ g64, float *__restrict__ grad, int64_t n, const double *__restrict__ inv_scale) {
    const double is = inv_scale[0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const long long v = g64[i];
        grad[i] = (float)((double)v * is);
        if (v != 0) g64[i] = 0;
    }
}

// 2-DoF: gradient = sum over reference times and segments of the fixed-point partials (integer sums: any order)
__global__ void __launch_bounds__(256) k_finish_det(long long *__restrict__ gpart64, int nseg, int n_ref, const double *__restrict__ inv_scale,
                                                    double *__restrict__ gtheta) {
    __shared__ long long s_acc[2];
    double g0 = 0.0, g1 = 0.0;
    for (int k = 0; k < n_ref; ++k) {
        if (threadIdx.x < 2) s_acc[threadIdx.x] = 0;
        __syncthreads();
        long long a0 = 0, a1 = 0;
        long long *gp = gpart64 + (int64_t)k * nseg * 2;
        for (int i = threadIdx.x; i < nseg; i += blockDim.x) {
            a0 += gp[2 * i];
            a1 += gp[2 * i + 1];
        }
        atomic_add_i64(&s_acc[0], a0);
        atomic_add_i64(&s_acc[1], a1);
        __syncthreads();
        if (threadIdx.x == 0) {
            g0 += (double)s_acc[0] * inv_scale[k];  // reference times in fixed order
            g1 += (double)s_acc[1] * inv_scale[k];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        gtheta[0] = g0;
        gtheta[1] = g1;
    }
}

// 2-DoF objectives that keep statistics (everything but the deferred plain variance): K3 left sum dt g in doubles 0, 1 of kRawLines
// lines per reference time; gradient = their sum over lines and reference times.  One wave.  result_src -> result_dst / flag / seq:
// cmax_objective_host -- copy the result K3's first workgroup wrote to the device into pinned host memory beside the gradient and
// publish a run counter behind both.
__global__ void __launch_bounds__(64)
k_finish_lines(const double *__restrict__ raw, int n_ref, double *__restrict__ gtheta, const double *__restrict__ result_src,
               double *__restrict__ result_dst, volatile unsigned long long *flag, unsigned long long seq) {
    const int lane = threadIdx.x;
    double g0 = 0.0, g1 = 0.0;
    typedef double double2_v __attribute__((ext_vector_type(2)));
    const int line = lane < kRawLines ? lane : kRawLines - 1;  // unconditional 16-byte loads, masked afterwards (see k_finish_raw)
    double2_v a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        a[k] = double2_v{0.0, 0.0};
        if (k < n_ref) a[k] = *reinterpret_cast<const double2_v *>(raw + (int64_t)k * kRawStride + line * kSubStride);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        g0 += lane < kRawLines ? a[k].x : 0.0;
        g1 += lane < kRawLines ? a[k].y : 0.0;
    }
    g0 = wave_sum_lane63(g0);
    g1 = wave_sum_lane63(g1);
    if (result_dst && lane < 8) result_dst[lane] = result_src[lane];
    if (lane == kWave - 1) {
        gtheta[0] = g0;
        gtheta[1] = g1;
    }
    if (flag) {  // (ONE wave: every lane's stores above have left it when the fence returns, so lane 63 may publish for all of them)
        __threadfence_system();
        if (lane == kWave - 1) *flag = seq;
    }
}

}  // namespace cmax

// the event kernels, once per workgroup size
namespace cmax {
#define CMAX_SLOTS 2048
#define CMAX_WINCAP 8192
#define CMAX_ACC_TILES 3
#define CMAX_ACC_SMALL_GROUPS 3
#define CMAX_THREADS 256
#define CMAX_EVENT_NS t256
#include "cmax_event_kernels.inc"
#undef CMAX_THREADS
#undef CMAX_EVENT_NS
#define CMAX_THREADS 512
#define CMAX_EVENT_NS t512
#include "cmax_event_kernels.inc"
#undef CMAX_THREADS
#undef CMAX_EVENT_NS
#define CMAX_THREADS 1024
#define CMAX_EVENT_NS t1024
#include "cmax_event_kernels.inc"
#undef CMAX_THREADS
#undef CMAX_EVENT_NS
// mid segments (3064 events, four source tiles wide: cmax_handle_s::mid): 512 threads x 6 events
#undef CMAX_SLOTS
#undef CMAX_WINCAP
#undef CMAX_ACC_TILES
#undef CMAX_ACC_SMALL_GROUPS
#define CMAX_SLOTS 3072
#define CMAX_WINCAP 6144
#define CMAX_ACC_TILES 4
#define CMAX_ACC_SMALL_GROUPS 5
#define CMAX_THREADS 512
#define CMAX_EVENT_NS m512
#include "cmax_event_kernels.inc"
#undef CMAX_THREADS
#undef CMAX_EVENT_NS
// big segments (4088 events: cmax_handle_s::big): 512 threads x 8 events, 1024 x 4
#undef CMAX_SLOTS
#undef CMAX_WINCAP
#undef CMAX_ACC_TILES
#undef CMAX_ACC_SMALL_GROUPS
#define CMAX_SLOTS 4096
#define CMAX_WINCAP 8192
#define CMAX_ACC_TILES 6
#define CMAX_ACC_SMALL_GROUPS 3
#define CMAX_THREADS 512
#define CMAX_EVENT_NS b512
#include "cmax_event_kernels.inc"
#undef CMAX_THREADS
#undef CMAX_EVENT_NS
#define CMAX_THREADS 1024
#define CMAX_EVENT_NS b1024
#include "cmax_event_kernels.inc"
#undef CMAX_THREADS
#undef CMAX_EVENT_NS
#undef CMAX_SLOTS
#undef CMAX_WINCAP
#undef CMAX_ACC_TILES
#undef CMAX_ACC_SMALL_GROUPS

__global__ void k_empty(const int4 *) {}  // cmax_debug_launch_floor

// 2-DoF only: gradient = sum of the per-segment partials of every K3 launch (one workgroup).
__global__ void __launch_bounds__(256)
k_finish(const double *__restrict__ gpart, int n_gpart, double *__restrict__ gtheta) {
    __shared__ double s_red[2 * 4];
    double acc[2] = {0.0, 0.0};
    for (int i = threadIdx.x; i < n_gpart; i += blockDim.x) {
        acc[0] += gpart[2 * i];
        acc[1] += gpart[2 * i + 1];
    }
    block_sum<2>(acc, s_red);
    if (threadIdx.x == 0) {
        gtheta[0] = acc[0];
        gtheta[1] = acc[1];
    }
}

// 2-DoF, plain variance, deferred statistics: K3 left (S1x, S1y, S2x, S2y, sum I, sum I^2) per reference time with
//   S1 = sum_e dt * bilinear-difference(1_Omega I),  S2 = the same of 1_Omega.  With G = c (I - mu) 1_Omega,
//   c = 2 coef / (n - 1):  dL/dtheta = sum_k c_k (S1_k - mu_k S2_k);  loss and coef_k from the image sums.
// One function for the device (k_finish_deferred, k_finish_raw) and the host (cmax_finalize_raw_host).
__host__ __device__ inline void finalize_deferred(const ObjParams &op, const double (*S)[6], double v_orig, double *result, double *gtheta) {
    const int i0 = op.omit ? 1 : 0;
    const double npix = (double)(op.H - 2 * i0) * (double)(op.W - 2 * i0);
    double loss = 0.0, g0 = 0.0, g1 = 0.0;
    for (int k = 0; k < op.n_ref; ++k) {
        const double mu = S[k][4] / npix;
        const double v = (S[k][5] - S[k][4] * mu) / (npix - 1.0);  // unbiased like torch.var, image_variance.py:55
        result[1 + k] = v;
        double coef;
        if (!op.normalized) {
            loss += op.mult[k] * (op.minimize ? -v : v);
            coef = op.mult[k] * (op.minimize ? -1.0 : 1.0);
        } else {
            loss += op.mult[k] * (op.minimize ? v_orig / v : v / v_orig);
            coef = op.mult[k] * (op.minimize ? -v_orig / (v * v) : 1.0 / v_orig);
        }
        if (op.negate) coef = -coef;
        const double c = coef * 2.0 / (npix - 1.0);
        g0 += c * (S[k][0] - mu * S[k][2]);
        g1 += c * (S[k][1] - mu * S[k][3]);
    }
    result[0] = op.negate ? -loss : loss;
    result[5] = v_orig;
    if (gtheta) {
        gtheta[0] = g0;
        gtheta[1] = g1;
    }
}

// per-block partials [n_ref][nseg][6] -> loss + gradient (the tangent-image path: k_tan_stats_var leaves kTanBlocks partials)
__global__ void __launch_bounds__(256)
k_finish_deferred(ObjParams op, const double *__restrict__ stat, const double *__restrict__ gpart, int nseg,
                  double *__restrict__ result, double *__restrict__ gtheta) {
    __shared__ double s_red[6 * 4];
    __shared__ double s_sum[4][6];
    for (int k = 0; k < op.n_ref; ++k) {
        double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        const double *gp = gpart + (int64_t)k * nseg * 6;
        for (int i = threadIdx.x; i < nseg; i += blockDim.x) {
#pragma unroll
            for (int q = 0; q < 6; ++q) acc[q] += gp[6 * i + q];
        }
        block_sum<6>(acc, s_red);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int q = 0; q < 6; ++q) s_sum[k][q] = acc[q];
        }
    }
    if (threadIdx.x != 0) return;
    finalize_deferred(op, s_sum, op.normalized ? orig_value(op, stat) : 0.0, result, gtheta);
}

// The raw sums of the deferred K3 (kRawLines lines of six doubles per reference time) -> loss + gradient: ONE wave, lane l
// loads line l (all loads of all reference times in flight together), DPP sums, lane 63 finishes.  This launch exists only
// for callers that want the result ON THE DEVICE (cmax_objective); cmax_objective_raw / cmax_objective_host fold on the host.
// flag / seq (cmax_objective_host): result and gtheta are PINNED HOST memory; after they are visible system-wide the kernel writes
// the run counter `seq` behind them, which the host polls -- no copy engine, no hipStreamQuery.
__global__ void __launch_bounds__(64)
k_finish_raw(const double *__restrict__ raw, int n_ref, ObjParams op, const double *__restrict__ stat, double *__restrict__ result, double *__restrict__ gtheta,
             volatile unsigned long long *flag = nullptr, unsigned long long seq = 0) {
    // (raw and n_ref lead the parameter list: they arrive in SGPRs -- kernel-argument preload -- and the loads below leave at once, while
    // everything finalize_deferred needs is fetched from the argument block beside them: one round trip less in a one-wave kernel)
    // blockIdx.x = candidate motion of cmax_objective_batch (one workgroup otherwise)
    raw += (int64_t)blockIdx.x * n_ref * kRawStride;
    result += 8 * blockIdx.x;
    if (gtheta) gtheta += 2 * blockIdx.x;
    const int lane = threadIdx.x;
    double v[4][6];
    // every lane loads ITS line's six doubles as three 16-byte loads, all issued before the first use (lanes >= kRawLines re-read
    // the last line and are zeroed by a select: `lane < 32 ? raw[..] : 0` put each of the 6 x n_ref loads into an exec-masked
    // block with its own s_waitcnt -- six dependent round trips in a kernel that has only this one wave)
    typedef double double2_v __attribute__((ext_vector_type(2)));
    const int line = lane < kRawLines ? lane : kRawLines - 1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int q = 0; q < 6; ++q) v[k][q] = 0.0;
        if (k < n_ref) {  // uniform
            const double2_v *src = reinterpret_cast<const double2_v *>(raw + (int64_t)k * kRawStride + line * kSubStride);
            const double2_v a = src[0], b = src[1], c = src[2];
            v[k][0] = a.x;
            v[k][1] = a.y;
            v[k][2] = b.x;
            v[k][3] = b.y;
            v[k][4] = c.x;
            v[k][5] = c.y;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int q = 0; q < 6; ++q) v[k][q] = lane < kRawLines ? v[k][q] : 0.0;
    const double v_orig = op.normalized ? orig_value<true>(op, stat) : 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= op.n_ref) break;  // (uniform: 18 VALU instructions per sum, and there is only this one wave)
#pragma unroll
        for (int q = 0; q < 6; ++q) v[k][q] = wave_sum_lane63(v[k][q]);
    }
    if (lane == kWave - 1) {
        finalize_deferred(op, v, v_orig, result, gtheta);
        if (flag) {
            __threadfence_system();
            *flag = seq;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host-side orchestration
// ---------------------------------------------------------------------------------------------
// exclusive scan of h->counts[0..m) in place, counts[m] = total
// exclusive scan of counts[0 .. m) in place, counts[m] = total (tmp: ceil(m / 2048) + 1 ints)
static void launch_scan_buf(int *counts, int m, int *tmp, hipStream_t s, int *nonzero = nullptr, int *total_out = nullptr) {
    const int nchunk = div_up(m, kScanChunk);
    hipLaunchKernelGGL(k_scan_sums, dim3(nchunk), dim3(256), 0, s, counts, m, tmp, nonzero);
    hipLaunchKernelGGL(k_scan_chunks, dim3(1), dim3(1024), 0, s, tmp, nchunk, total_out ? total_out : counts + m);
    hipLaunchKernelGGL(k_scan_apply, dim3(nchunk), dim3(256), 0, s, counts, m, tmp);
}
static void launch_scan(cmax_handle_s *h, int m, hipStream_t s, int *nonzero = nullptr) {
    if (m <= 4096 && !nonzero) {
        hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, s, h->counts, m);
        return;
    }
    launch_scan_buf(h->counts, m, h->scan_tmp, s, nonzero);
}

static float ref_fraction(int ref_mode, double frac) {
    if (ref_mode == CMAX_REF_FIRST) return 0.f;
    if (ref_mode == CMAX_REF_LAST) return 1.f;
    return (float)frac;
}
// ... and what fp32 lost of it (zero for the named directions and every dyadic fraction)
static float ref_fraction_lo(int ref_mode, double frac) {
    if (ref_mode == CMAX_REF_FIRST || ref_mode == CMAX_REF_LAST) return 0.f;
    return (float)(frac - (double)(float)frac);
}

// 512-thread workgroups (4 events per thread) once the work list exceeds what the chip holds at once
// (the voxel K3 then even 1024 x 2)
static bool wide_groups(const cmax_handle_s *h) { return h->nseg > 1024; }
// tuning experiments only: CMAX_VOTE_NS / CMAX_GRAD_NS = 256 | 512 | 1024 force the workgroup size of K1 / K3
static int forced_ns(const char *name) {
    const char *e = getenv(name);
    return e ? atoi(e) : 0;
}

template <int MODEL>
static void launch_vote(cmax_handle_s *h, const EvView &ev, const WarpParams &wp, const RefArgs &ra, int n_ref, hipStream_t s, int nz = 1) {
    const dim3 grid(8 * ((h->nseg + 7) / 8), n_ref, nz);  // z: candidate motions of cmax_objective_batch
    ProfScope prof(h, kProfVote, s);
    const char *cev = (h->compact && h->big && MODEL != CMAX_MODEL_VOXEL) ? h->cev : nullptr;  // 6-byte events, one region per segment
#define CMAX_LAUNCH_VOTE(NS, FRAC)                                                                                           \
    do {                                                                                                                    \
        if (ra.musum[0]) hipLaunchKernelGGL((NS::k_vote<MODEL, FRAC, true>), grid, dim3(NS::kThr), 0, s, h->d_segs, h->nseg, ev.ev, cev, ev, wp, ra); \
        else hipLaunchKernelGGL((NS::k_vote<MODEL, FRAC>), grid, dim3(NS::kThr), 0, s, h->d_segs, h->nseg, ev.ev, cev, ev, wp, ra);     \
    } while (0)
    static const int force = forced_ns("CMAX_VOTE_NS");
    for (int rep = 0; rep < h->prof_repeat; ++rep) {
        if (h->big) {  // the work list holds segments of up to 4088 events: only the big-segment kernels can run it
            if (h->has_frac) CMAX_LAUNCH_VOTE(b512, true);
            else CMAX_LAUNCH_VOTE(b512, false);
        } else if (h->mid) {  // ... of up to 3064 events
            if (h->has_frac) CMAX_LAUNCH_VOTE(m512, true);
            else CMAX_LAUNCH_VOTE(m512, false);
        } else if (force ? force == 512 : h->nseg > 512) {  // as for K3: cfg2 (704 half-tile segments) K1 6.35 -> 5.97 us, evaluation 18.06 -> 17.38
            if (h->has_frac) CMAX_LAUNCH_VOTE(t512, true);
            else CMAX_LAUNCH_VOTE(t512, false);
        } else {
            if (h->has_frac) CMAX_LAUNCH_VOTE(t256, true);
            else CMAX_LAUNCH_VOTE(t256, false);
        }
    }
#undef CMAX_LAUNCH_VOTE
}

// workgroup size launch_grad picks (tuning knob CMAX_GRAD_NS aside)
static int grad_threads(const cmax_handle_s *h, int model) {
    static const int force = forced_ns("CMAX_GRAD_NS");
    if (h->big) return model == CMAX_MODEL_VOXEL ? 1024 : 512;
    if (h->mid) return 512;
    if (model == CMAX_MODEL_VOXEL && h->small_acc && h->owned && !force) return 512;
    if (force ? (force == 1024 && model == CMAX_MODEL_VOXEL) : (model == CMAX_MODEL_VOXEL && wide_groups(h))) return 1024;
    return (force ? force >= 512 : h->nseg > 512) ? 512 : 256;
}

// seg0 / seg_n: sub-range of the work list (a band of tile rows, see cmax_comm_set_c2_bands); seg_n < 0 = the whole list.  The
// kernels only see a shifted list (RefArgs::win is shifted by the caller).
template <int MODEL>
static void launch_grad(cmax_handle_s *h, const EvView &ev, const WarpParams &wp, const RefArgs &ra, int n_ref, int fold,
                        const ObjParams &op, double *gpart, float *gflow, double *result, bool owned, hipStream_t s, int seg0 = 0,
                        int seg_n = -1, int nz = 1) {
    const int4 *segs = h->d_segs + seg0;
    const int nseg = seg_n < 0 ? h->nseg : seg_n;
    const dim3 grid(8 * ((nseg + 7) / 8) + (fold == kFoldStatsInside ? ra.stat_blocks : 0), n_ref, nz);
    ProfScope prof(h, kProfGrad, s);
    const char *cev = (h->compact && h->big && MODEL != CMAX_MODEL_VOXEL) ? h->cev + (int64_t)seg0 * b512::kCompactStride : nullptr;
    if (h->deterministic) {  // one workgroup size, two ways of obtaining dL/dIWE (objective_finish runs the unfused image path)
#define CMAX_LAUNCH_DET(FRAC, FOLD)                                                                                                         \
    do {                                                                                                                                    \
        if (h->big) hipLaunchKernelGGL((b512::k_grad<MODEL, FRAC, FOLD, kGradDet>), grid, dim3(b512::kThr), 0, s, segs, nseg, ev.ev, cev, (const int4 *)ra.win, ra.stat_blocks, ev, wp, ra, op, h->d_stat, gpart, gflow, result); \
        else if (h->mid) hipLaunchKernelGGL((m512::k_grad<MODEL, FRAC, FOLD, kGradDet>), grid, dim3(m512::kThr), 0, s, segs, nseg, ev.ev, cev, (const int4 *)ra.win, ra.stat_blocks, ev, wp, ra, op, h->d_stat, gpart, gflow, result); \
        else hipLaunchKernelGGL((t256::k_grad<MODEL, FRAC, FOLD, kGradDet>), grid, dim3(t256::kThr), 0, s, segs, nseg, ev.ev, cev, (const int4 *)ra.win, ra.stat_blocks, ev, wp, ra, op, h->d_stat, gpart, gflow, result); \
    } while (0)
        if (h->has_frac) {
            if (fold == kFoldStats) CMAX_LAUNCH_DET(true, kFoldStats);
            else CMAX_LAUNCH_DET(true, kFoldNone);
        } else {
            if (fold == kFoldStats) CMAX_LAUNCH_DET(false, kFoldStats);
            else CMAX_LAUNCH_DET(false, kFoldNone);
        }
#undef CMAX_LAUNCH_DET
        return;
    }
    // dense model: runs of equal source pixel are reduced serially per thread when they are long (pixel-sorted
    // handle, >= 8 events per active pixel), else with a segmented scan per slot over lanes holding consecutive events;
    // owned groups (dense / voxel, one reference time, group-aligned work list): LDS accumulators + plain stores
    const bool strided = MODEL == CMAX_MODEL_DENSE && !(h->long_runs && h->n_time_bin == 0);
#define CMAX_LAUNCH_GRAD_L(NS, FRAC, FOLD, VARIANT) \
    hipLaunchKernelGGL((NS::k_grad<MODEL, FRAC, FOLD, VARIANT>), grid, dim3(NS::kThr), 0, s, segs, nseg, ev.ev, cev, (const int4 *)ra.win, ra.stat_blocks, ev, wp, ra, op, h->d_stat, gpart, gflow, result)
#define CMAX_LAUNCH_GRAD(NS, FRAC, FOLD)                                      \
    do {                                                                      \
        if constexpr (MODEL == CMAX_MODEL_DENSE) {                            \
            if (owned) {                                                      \
                CMAX_LAUNCH_GRAD_L(NS, FRAC, FOLD, kGradOwned);               \
            } else if (strided) {                                             \
                CMAX_LAUNCH_GRAD_L(NS, FRAC, FOLD, kGradStrided);             \
            } else {                                                          \
                CMAX_LAUNCH_GRAD_L(NS, FRAC, FOLD, kGradRuns);                \
            }                                                                 \
        } else if constexpr (MODEL == CMAX_MODEL_VOXEL) {                     \
            if (owned && small) {                                             \
                if constexpr (NS::kThr == 512 && NS::kSlots <= 3072) { CMAX_LAUNCH_GRAD_L(NS, FRAC, FOLD, kGradOwnedSmall); } \
            } else if (owned) {                                               \
                CMAX_LAUNCH_GRAD_L(NS, FRAC, FOLD, kGradOwned);               \
            } else {                                                          \
                CMAX_LAUNCH_GRAD_L(NS, FRAC, FOLD, kGradRuns);                \
            }                                                                 \
        } else {                                                              \
            CMAX_LAUNCH_GRAD_L(NS, FRAC, FOLD, kGradRuns);                    \
        }                                                                     \
    } while (0)
#define CMAX_LAUNCH_GRAD_FR(NS, FRAC)                                                        \
    if (fold == kFoldDeferred) {                                                             \
        if constexpr (MODEL == CMAX_MODEL_2DOF) { CMAX_LAUNCH_GRAD(NS, FRAC, kFoldDeferred); } \
    } else if (fold == kFoldStatsInside) {                                                   \
        if constexpr (MODEL == CMAX_MODEL_VOXEL && NS::kThr == 512 && NS::kSlots <= 3072) {   \
            if (!owned) { CMAX_LAUNCH_GRAD(NS, FRAC, kFoldStatsInside); }                    \
            else if (small) { CMAX_LAUNCH_GRAD_L(NS, FRAC, kFoldStatsInside, kGradOwnedSmall); } \
            else { CMAX_LAUNCH_GRAD_L(NS, FRAC, kFoldStatsInside, kGradOwned); }            \
        } else if constexpr (MODEL != CMAX_MODEL_2DOF) {                                     \
            if (owned) { CMAX_LAUNCH_GRAD_L(NS, FRAC, kFoldStatsInside, kGradOwned); }       \
            else { CMAX_LAUNCH_GRAD(NS, FRAC, kFoldStatsInside); }                           \
        }                                                                                    \
    } else if (fold == kFoldStats) {                                                         \
        CMAX_LAUNCH_GRAD(NS, FRAC, kFoldStats);                                              \
    } else if (fold == kFoldScale) {                                                         \
        CMAX_LAUNCH_GRAD(NS, FRAC, kFoldScale);                                              \
    } else {                                                                                 \
        CMAX_LAUNCH_GRAD(NS, FRAC, kFoldNone);                                               \
    }
#define CMAX_LAUNCH_GRAD_NS(NS)          \
    if (h->has_frac) {                   \
        CMAX_LAUNCH_GRAD_FR(NS, true)    \
    } else {                             \
        CMAX_LAUNCH_GRAD_FR(NS, false)   \
    }
    static const int force = forced_ns("CMAX_GRAD_NS");
    // voxel, owned groups of <= 3 groups per segment: 512 threads x 4 events with the small accumulator array (see build_segments)
    const bool small = MODEL == CMAX_MODEL_VOXEL && owned && h->small_acc && !h->big && !force;
    for (int rep = 0; rep < h->prof_repeat; ++rep) {
        if (h->big) {  // segments of up to 4088 events
            if constexpr (MODEL == CMAX_MODEL_VOXEL) {
                CMAX_LAUNCH_GRAD_NS(b1024)
            } else {
                CMAX_LAUNCH_GRAD_NS(b512)
            }
        } else if (h->mid) {  // ... of up to 3064 events
            CMAX_LAUNCH_GRAD_NS(m512)
        } else if (small) {
            CMAX_LAUNCH_GRAD_NS(t512)
        } else if (force ? (force == 1024 && MODEL == CMAX_MODEL_VOXEL) : (MODEL == CMAX_MODEL_VOXEL && wide_groups(h))) {  // measured: voxel K3 of cfg4 22.1 us (512 threads) -> 19.3 us
            if constexpr (MODEL == CMAX_MODEL_VOXEL) {
                CMAX_LAUNCH_GRAD_NS(t1024)
            }
        } else if (force ? force >= 512 : h->nseg > 512) {  // K3 hides its latencies better with 8 waves per workgroup (cfg2: 7.5 -> 7.1 us)
            CMAX_LAUNCH_GRAD_NS(t512)
        } else {
            CMAX_LAUNCH_GRAD_NS(t256)
        }
    }
#undef CMAX_LAUNCH_GRAD_NS
#undef CMAX_LAUNCH_GRAD_FR
#undef CMAX_LAUNCH_GRAD
#undef CMAX_LAUNCH_GRAD_L
}

static EvView ev_view(const cmax_handle_s *h) {
    EvView ev;
    ev.ev = h->evp;
    ev.rx = h->rx;
    ev.ry = h->ry;
    ev.rl = h->rl;
    ev.tau64 = h->n_time_bin > 0 ? h->tau64 : nullptr;
    return ev;
}

static WarpParams warp_params(const cmax_handle_s *h, const float *motion, int T, int ref_mode, double frac, int normalize, int motion_f64 = 0) {
    // wp.d is what the single-reference kernels (tangent / HVP) read; the batched kernels take it from RefArgs
    WarpParams wp;
    wp.H = h->H;
    wp.W = h->W;
    wp.Hp = h->Hp;
    wp.Wp = h->Wp;
    wp.ph = h->ph;
    wp.pw = h->pw;
    wp.T = T;
    wp.ntc = h->ntc;
    wp.d = ref_fraction(ref_mode, frac);
    wp.d_lo = ref_fraction_lo(ref_mode, frac);
    wp.normalize = normalize;
    wp.motion_f64 = motion_f64;
    wp.tmm = h->d_tmm;
    wp.motion = motion;
    return wp;
}

// raw votes of n_ref reference times (one launch) into imgs[k] (cleared here unless bit k of zero_mask says it is
// zero already); stat_slot0 >= 0: the statistics accumulators of slots stat_slot0 + k are reset by the launch
// mu_taps: non-null = also accumulate sum_p I[p] B[p] for the blurred variance (taps k0, k1 of the blur, omit_boundary in [2])
static int vote_images(cmax_handle_s *h, int model, const float *motion, int T, int n_ref, const int *ref_mode, const double *ref_frac,
                       int normalize, float *const *imgs, unsigned zero_mask, int stat_slot0, hipStream_t s, bool publish_windows = false,
                       const float *mu_taps = nullptr, int motion_f64 = 0, double *raw_reset = nullptr, double *raw_lines = nullptr) {
    const int64_t npix = (int64_t)h->Hp * h->Wp;
    RefArgs ra = {};
    ra.k0 = 0;
    ra.raw_zero = raw_lines;  // (the deferred objective resets its sums through ra.stat instead: raw_reset)
    ra.grad_zero = h->vote_clear4;
    ra.n_grad_zero4 = h->vote_nclear4;
    h->grad_cleared_by_vote = h->vote_clear4;
    h->vote_clear4 = nullptr;
    h->vote_nclear4 = 0;
    if (publish_windows && h->n > 0) {  // K3 of the same evaluation re-uses the LDS windows (see objective_finish)
        ra.win = h->d_win;
        ra.shifts = h->d_shifts;
        h->win_motion = motion;
        h->win_model = model;
        h->win_nref = n_ref;
        h->win_T = T;
        h->win_normalize = normalize;
        for (int k = 0; k < n_ref; ++k) h->win_d[k] = ref_fraction(ref_mode[k], ref_frac[k]);
        h->win_generation = h->generation;
    }
    const bool det = h->deterministic && h->n > 0;  // votes go to the integer images, imgs[k] are written by the conversion below
    for (int k = 0; k < n_ref; ++k) {
        if (!det && !((zero_mask >> k) & 1u)) CMAX_CHECK_HIP(hipMemsetAsync(imgs[k], 0, npix * sizeof(float), s));
        if (det) ra.img64[k] = h->img64 + (int64_t)k * npix;
        // accumulators K1's first workgroup resets: the statistics of slot stat_slot0 + k, or (deferred 2-DoF objective: no
        // statistics kernel runs) the raw sums K3 of the same evaluation adds into
        double *stat_zero = raw_reset ? raw_reset + (int64_t)k * kRawStride : (stat_slot0 >= 0 ? h->d_stat + (stat_slot0 + k) * kStatStride : nullptr);
        if (h->n == 0 && stat_zero)  // no K1 launch on this rank: reset the accumulators explicitly
            CMAX_CHECK_HIP(hipMemsetAsync(stat_zero, 0, kStatStride * sizeof(double), s));
        ra.d[k] = ref_fraction(ref_mode[k], ref_frac[k]);
        ra.d_lo[k] = ref_fraction_lo(ref_mode[k], ref_frac[k]);
        ra.img[k] = imgs[k];
        ra.stat[k] = stat_zero;
        if (mu_taps && !det && h->n > 0) ra.musum[k] = h->d_musum + ((int64_t)h->mu_buf * 4 + k) * kMuStride;
    }
    if (mu_taps) h->mu_valid = ra.musum[0] != nullptr;
    if (ra.musum[0]) {
        ra.band_b0 = border_weight(0, h->Hp, (int)mu_taps[2], mu_taps[0], mu_taps[1]);
        ra.band_b1 = border_weight(1, h->Hp, (int)mu_taps[2], mu_taps[0], mu_taps[1]);
    }
    if (h->n == 0) return 0;
    const EvView ev = ev_view(h);
    const WarpParams wp = warp_params(h, motion, T, ref_mode[0], ref_frac[0], normalize, motion_f64);
    switch (model) {
        case CMAX_MODEL_2DOF: launch_vote<CMAX_MODEL_2DOF>(h, ev, wp, ra, n_ref, s); break;
        case CMAX_MODEL_DENSE: launch_vote<CMAX_MODEL_DENSE>(h, ev, wp, ra, n_ref, s); break;
        case CMAX_MODEL_VOXEL: launch_vote<CMAX_MODEL_VOXEL>(h, ev, wp, ra, n_ref, s); break;
        default: launch_vote<-1>(h, ev, wp, ra, n_ref, s); break;
    }
    CMAX_CHECK_LAUNCH();
    if (det) {  // exact integer sums -> fp32 images, one rounding each; the integer images are zero again afterwards
        FixedArgs fa = {};
        fa.inv_fix = (h->big || h->mid) ? 1.0 / 524288.0 : 1.0 / 1048576.0;
        for (int k = 0; k < n_ref; ++k) {
            fa.src[k] = ra.img64[k];
            fa.dst[k] = imgs[k];
        }
        hipLaunchKernelGGL(k_fixed_to_image, dim3(stream_grid(npix, 256), n_ref), dim3(256), 0, s, fa, npix);
        CMAX_CHECK_LAUNCH();
    }
    return 0;
}

// one reference time
static int vote_image(cmax_handle_s *h, int model, const float *motion, int T, int ref_mode, double frac, int normalize,
                      float *raw, bool already_zero, int stat_slot, hipStream_t s) {
    float *imgs[1] = {raw};
    return vote_images(h, model, motion, T, 1, &ref_mode, &frac, normalize, imgs, already_zero ? 1u : 0u, stat_slot, s);
}

// the image the contrast is evaluated on: `raw`, or its blurred copy in `blur` when sigma > 0
static int blur_image(cmax_handle_s *h, double sigma, const float *raw, float *blur, const float **out, hipStream_t s) {
    *out = raw;
    if (sigma > 0) {
        const int64_t npix = (int64_t)h->Hp * h->Wp;
        double k0, k1;
        blur_taps(sigma, k0, k1);
        hipLaunchKernelGGL(k_blur3<float>, dim3(div_up(npix, 256)), dim3(256), 0, s, raw, h->Hp, h->Wp, (float)k0, (float)k1, blur);
        CMAX_CHECK_LAUNCH();
        *out = blur;
    }
    return 0;
}

static int stat_blocks(const cmax_handle_s *h) {
    // deterministic mode: one workgroup per sub-accumulator, so every accumulator receives exactly one addition
    if (h->deterministic) return kStatSub;
    // ~4 pixels per thread, at most kStatBlocksMax workgroups
    int64_t b = ((int64_t)h->Hp * h->Wp + 1023) / 1024;
    if (b < 1) b = 1;
    if (b > kStatBlocksMax) b = kStatBlocksMax;
    return (int)b;
}

// statistics of `img` -> stat[slot] (accumulators zeroed by the K1 launch); optionally zero `zero_img`
static int stat_subs(const cmax_handle_s *h) {
    if (h->deterministic) return kStatSub;
    // one 128-byte line per sub-accumulator (same-line atomics serialise).  Their readers are ONE wave per K3 workgroup (and
    // only for a normalised cost) and the wave that writes the loss, each with one load per lane -- when every wave of K3
    // gathered them at its head, more than 8 lines cost K3 more than they saved K2 (cfg3 K3 18.7 -> 21.1 us with 25 lines).
    // 16 lines: cfg3's image kernel (1200 workgroups) 6.0 -> 5.65 us, cfg4's (363) 6.5 -> 5.95.
    int n = stat_blocks(h) / 6;
    static const int forced = getenv("CMAX_NSUB") ? atoi(getenv("CMAX_NSUB")) : 0;  // tuning only
    if (forced > 0) return forced < kStatSub ? forced : kStatSub;
    return n < 4 ? 4 : (n > 16 ? 16 : n);
}

// zero_extra: optional buffer of n_extra floats (16-byte aligned, n_extra % 4 == 0) cleared by the same launch
static int launch_stats(cmax_handle_s *h, int cost, const float *img, int omit, int slot, float *zero_img, hipStream_t s,
                        float *zero_extra = nullptr, int64_t n_extra = 0) {
    const int grid = stat_blocks(h);
    const int nsub = stat_subs(h);
    double *stat_slot = h->d_stat + slot * kStatStride;
    ProfScope prof(h, kProfStats, s);
    for (int rep = 0; rep < h->prof_repeat; ++rep) {
        if (h->deterministic) {  // kStatSub workgroups of 1024 threads: every accumulator is written once, the summation order is fixed
            if (cost == CMAX_COST_VARIANCE)
                hipLaunchKernelGGL((k_stats<CMAX_COST_VARIANCE, 1024>), dim3(grid), dim3(1024), 0, s, img, h->Hp, h->Wp, omit, nsub, stat_slot, zero_img, (float4 *)zero_extra, n_extra / 4);
            else
                hipLaunchKernelGGL((k_stats<CMAX_COST_GRADMAG, 1024>), dim3(grid), dim3(1024), 0, s, img, h->Hp, h->Wp, omit, nsub, stat_slot, zero_img, (float4 *)zero_extra, n_extra / 4);
        } else if (cost == CMAX_COST_VARIANCE)
            hipLaunchKernelGGL(k_stats<CMAX_COST_VARIANCE>, dim3(grid), dim3(256), 0, s, img, h->Hp, h->Wp, omit, nsub, stat_slot, zero_img, (float4 *)zero_extra, n_extra / 4);
        else
            hipLaunchKernelGGL(k_stats<CMAX_COST_GRADMAG>, dim3(grid), dim3(256), 0, s, img, h->Hp, h->Wp, omit, nsub, stat_slot, zero_img, (float4 *)zero_extra, n_extra / 4);
    }
    CMAX_CHECK_LAUNCH();
    return 0;
}

// The event kernels read the sorted events in 16-byte pairs, so the pair that holds the last event of the last
// segment can reach one element past the batch.  That slot is not voted, but its pixel still indexes the flow
// field (dense / voxel warp): the two padding elements must decode to pixel (0, 0), bin 0 -- not to whatever the
// allocation held before.
static int pad_event_tail(cmax_handle_s *h, hipStream_t s) {
    CMAX_CHECK_HIP(hipMemsetAsync(h->evp + h->n, 0, 2 * sizeof(uint2), s));
    return 0;
}

// host copy of what one batch left on the device: [0] any fractional source coordinate, [1] dropped events,
// [2] events kept from off the sensor (cmax_set_keep_outside); batch time extremes
struct BatchReadback {
    int flags[4] = {0, 0, 0, 0};
    double tmm[2] = {0.0, 0.0};
    int64_t n_in = 0;
};

// Work list of the event kernels from the sorted events.  counts[] = exclusive scan of the sort-key histogram;
// a GROUP = `stride` consecutive keys = one source tile (pixel keys) or one (tile, time bin).
// Segments: <= kSegMax consecutive sorted events (fixed-point range) inside one tile row, spanning <= 12 groups
// (the flow-gradient accumulator of the voxel K3 holds kAccCells = 12 * 256 cells).  Two regimes, measured on MI355X:
//   * batches that fill the chip several times over (> 1024 full segments): cut every kSegMax events, group
//     boundaries ignored -- every workgroup full (dense K3 of cfg3 30 -> 25 us);
//   * smaller batches are latency-bound: small groups are merged while they fit, a group with more than
//     kSegMax events is split into EQUAL parts -- group-aligned windows are tighter and no workgroup is left
//     with a small remainder (cfg2: 2040 + 806 per tile -> 2 x 1423).
template <typename T>
static int pinned_reserve(T **p, int64_t *cap, int64_t count) {
    if (count <= *cap) return 0;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr;
    *cap = 0;
    const int64_t want = count + count / 2 + 64;
    // coherent + mapped: kernels write results and run counters straight into these buffers and the host polls them
    // (cmax_objective_host, the patch plan's tail) -- visibility must not hang on HIP_HOST_COHERENT; zeroed: a flag word is polled
    // before anything ever wrote it
    if (hipHostMalloc((void **)p, (size_t)want * sizeof(T), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {
        set_error("hipHostMalloc(%lld bytes) failed", (long long)(want * sizeof(T)));
        return CMAX_ENOMEM;
    }
    std::memset((void *)*p, 0, (size_t)want * sizeof(T));
    *cap = want;
    return 0;
}

// overlap: launched (by the caller's functor) BEHIND the read-back copies and an event, i.e. work the GPU does while the host
// waits for that event only, cuts the segments and uploads them -- the ordering of the pixel runs by time (k_run_time_sort),
// which moves events inside their groups and leaves the group starts alone (round 3: 0.16 -> see profiles, per 1M events)
template <typename Overlap>
static int build_segments(cmax_handle_s *h, int stride, hipStream_t s, BatchReadback *rb, Overlap overlap) {
    // slab-major handles (cmax_set_time_slabs): the groups are (tile row, slab, tile column) -- ntr * S rows of ntc groups, one group per
    // "tile" of that virtual grid; cut like an un-binned list, never owned (a source pixel belongs to S groups)
    const bool slab = stride == 1 && h->slab_major && h->n_time_bin > 0;
    const int T = (stride == 1 && !slab) ? h->n_time_bin : 1;  // groups per tile
    const int ngroups = h->ntr * h->ntc * (slab ? h->n_time_bin : T);
    const int ntiles = h->ntr * h->ntc;
    const bool want_active = rb && h->n_time_bin == 0;
    // the pinned read-back mirrors the device allocation: [2 doubles] extremes | [4] flags | [ntiles] active pixels | [ngroups + 1] group starts
    const int64_t off_tmm = 0, off_flags = 4, off_active = 8, off_start = 8 + ntiles;
    (void)want_active;
    int rc = pinned_reserve(&h->hp_read, &h->hp_read_cap, off_start + ngroups + 1);
    if (rc) return rc;
    const int *group_start = h->hp_read + off_start;
    CMAX_CHECK_HIP(hipMemcpyAsync(h->hp_read, h->d_batch, (size_t)(off_start + ngroups + 1) * sizeof(int), hipMemcpyDeviceToHost, s));
    if (!h->read_done) CMAX_CHECK_HIP(hipEventCreateWithFlags(&h->read_done, hipEventDisableTiming));
    CMAX_CHECK_HIP(hipEventRecord(h->read_done, s));
    rc = overlap();
    if (rc) return rc;
    {  // busy-wait: hipEventSynchronize sleeps and wakes up tens of microseconds late
        hipError_t e;
        while ((e = hipEventQuery(h->read_done)) == hipErrorNotReady) {
        }
        CMAX_CHECK_HIP(e);
    }
    if (rb) {
        for (int k = 0; k < 4; ++k) rb->flags[k] = h->hp_read[off_flags + k];
        std::memcpy(rb->tmm, h->hp_read + off_tmm, 2 * sizeof(double));
        h->has_frac = rb->flags[0] != 0;
        h->n = rb->n_in - rb->flags[1];
        if (h->generation_counted != h->generation) {  // a new batch (re-binning keeps the count of cmax_set_events)
            h->n_dropped = rb->flags[1];
            h->n_outside = rb->flags[2];
            h->generation_counted = h->generation;
        }
        int64_t active = 0;  // source pixels that hold events (un-binned order)
        if (want_active)
            for (int k = 0; k < ntiles; ++k) active += h->hp_read[off_active + k];
        h->long_runs = active > 0 && h->n >= 8 * active;
        h->tmin_host = rb->tmm[0];
        h->tmax_host = rb->tmm[1];
        rc = pad_event_tail(h, s);
        if (rc) return rc;
    }
    // un-binned handles: a segment spans <= 3 source tiles.  Sparse tiles used to be merged up to 12 wide; the LDS
    // window of such a segment (16 * span + displacement range wide) overflowed its 4096 words as soon as the motion
    // was a few pixels, and the clipped path (bounds tests, global atomics for the overflow) made those few workgroups
    // the tail of the launch: 1M events whose lower tile rows are sparse, K1 7.5 -> 6.9 us, K3 10.4 -> 7.3 us.
    // binned handles: 12 (tile, bin) groups, what the LDS accumulators of the voxel K3 hold -- unless every segment is
    // below kSparseSegment events anyway (then that kernel adds straight to memory): up to 3 tiles' worth of groups,
    // instead of ~100-event segments cut by the span (30k events, T = 10: 312 -> 125 segments)
    // BIG segments for batches that fill the chip many times over (CMAX_BIG_SEG=0 / 1 overrides for A/B runs): the per-workgroup
    // fixed costs of the event kernels are paid once per 4088 events instead of once per 2040.  Pays from ~2000 big segments on
    // (256 CUs x 3 resident workgroups, several rounds): measured (profiles/r03_ablation.txt 8) 20M events 104.3 -> 95.0 us per
    // evaluation, 64M events 323.6 -> 292.6; 5M events (1223 segments) 38.5 -> 38.2; below that fewer, longer workgroups LOSE
    // (2.5M events @720p 30.6 -> 33.1, 2M-event voxel batch 31.4 -> 33.5).
    // Between 522k and 8M events the choice follows the work list itself (tools/probe_rounds.py, profiles/r03_ablation.txt 17): both
    // group-aligned cuts are counted, and big segments are taken when they need <= 4/11 of the standard cut's workgroups (720p: 4M
    // events = 3600 one-tile segments of 1100 events vs 1200 three-tile ones, 41.5 -> 38.7 us; voxel T = 10: 4M events 47.1 -> 41.0)
    // -- at a ratio of 2 .. 2.5 the standard cut wins (cfg5's shard 30.6 vs 33.1, cfg4 31.4 vs 33.5) -- or, from 4M events on, when
    // the standard cut is not group-aligned at all (groups above 2040 events: 720p 7M 56.4 -> 47.9 us; cfg3 37.9 -> 36.9 us).
    static const int big_env = getenv("CMAX_BIG_SEG") ? atoi(getenv("CMAX_BIG_SEG")) : -1;
    auto owned_count = [&](int cut, int span_max) -> int {  // segments of the group-aligned cut; -1: a group exceeds `cut`
        for (int g = 0; g < ngroups; ++g)
            if (group_start[g + 1] - group_start[g] > cut) return -1;
        const int row_groups = h->ntc * T;
        int cnt = 0;
        for (int r0 = 0; r0 < ngroups; r0 += row_groups) {
            int g = r0;
            while (g < r0 + row_groups) {
                const int gs = g;
                int c = 0;
                while (g < r0 + row_groups && g - gs < span_max && c + (group_start[g + 1] - group_start[g]) <= cut) {
                    c += group_start[g + 1] - group_start[g];
                    ++g;
                }
                ++cnt;
            }
        }
        return cnt;
    };
    // MID segments (round 4; CMAX_MID_SEG=0 / 1 overrides for A/B runs): 3064 events, four source tiles or five well-filled (tile, bin)
    // groups wide.  A launch of more workgroups than the chip holds at once (256 CUs x 4 workgroups of 512 threads) runs in rounds, and
    // every round pays the ~2.5 us a workgroup waits for its first loads again (tools/timeline.py): the standard cut of cfg5's shard is
    // 1693 workgroups -- two rounds, the second two thirds full -- this one 900 in ONE round; cfg4: 1258 -> 748.  Taken when it fits one
    // round where the standard cut does not.
    static const int mid_env = getenv("CMAX_MID_SEG") ? atoi(getenv("CMAX_MID_SEG")) : -1;
    constexpr int kResidentGroups = 1024;  // workgroups of 512 threads the chip holds at once
    h->mid = false;
    if (slab) {
        h->big = false;
    } else if (big_env >= 0) {
        h->big = big_env != 0;
    } else if (h->n >= (int64_t)8000000) {
        h->big = true;
    } else if (h->n < (int64_t)256 * kSegMax) {
        h->big = false;
    } else {
        const bool small_std = T > 1 && h->n >= (int64_t)480 * ngroups;  // (see small_acc below)
        const int std_n = owned_count(kSegMax, T == 1 ? 3 : (small_std ? kAccCellsDense / 256 : kAccCells / 256));
        const int big_n = std_n >= 0 ? owned_count(4088, T == 1 ? 6 : kAccCells / 256) : -1;
        h->big = std_n >= 0 ? (int64_t)std_n * 4 >= (int64_t)big_n * 11 : h->n >= (int64_t)4000000;
        if (!h->big && mid_env != 0 && !getenv("CMAX_NO_OWNED")) {
            const int mid_n = owned_count(3064, T == 1 ? 4 : (small_std ? 5 : kAccCells / 256));
            h->mid = mid_env > 0 ? mid_n > 0 : (mid_n > 0 && mid_n <= kResidentGroups && std_n > kResidentGroups);
        }
    }
    h->seg_max = h->big ? 4088 : (h->mid ? 3064 : kSegMax);
    const int kSegCut = h->seg_max;
    int max_groups = T == 1 ? 3 : kAccCells / 256;
    static const int free_env = getenv("CMAX_FREE_CUT") ? atoi(getenv("CMAX_FREE_CUT")) : -1;  // tuning only
    const bool free_cut = free_env >= 0 ? free_env != 0 : h->n > (int64_t)1024 * kSegMax;
    // batches far below one full segment per CU (the solver's 30k-event slices): a workgroup walks its events
    // 8 (4) per thread, so 2040-event segments leave 15 workgroups with long serial work on a 256-CU chip.
    // Cap the segment at n / 512 (>= 256 events): cfg1-shaped K3 13 -> 5 us.
    int seg_cap = kSegCut;
    if (h->n < (int64_t)256 * kSegMax) seg_cap = (int)std::min<int64_t>(kSegCut, std::max<int64_t>(256, (h->n + 511) / 512));
    static const int cap_env = getenv("CMAX_SEG_CAP") ? atoi(getenv("CMAX_SEG_CAP")) : 0;  // tuning only
    if (cap_env > 0 && free_cut) seg_cap = std::min(kSegCut, cap_env & ~1);
    if (T > 1 && seg_cap < kSparseSegment) max_groups = std::max(max_groups, 3 * T);
    std::vector<int4> segs;
    // Owned groups (batches of at least one full segment per CU whose groups all fit a segment): every group -- empty
    // ones included -- belongs to exactly ONE segment made of whole consecutive groups of one tile row, at most 3 tiles
    // (LDS window) / kAccCells / 256 groups (LDS accumulators).  The flow-gradient kernel of a single-reference objective
    // then stores its groups' pixels instead of adding to them (k_grad, kGradOwned), and the buffer needs no clearing.
    h->owned = false;
    if (h->n >= (int64_t)256 * kSegMax && !getenv("CMAX_NO_OWNED") && !slab) {
        bool fits = true;
        for (int g = 0; g < ngroups && fits; ++g) fits = group_start[g + 1] - group_start[g] <= kSegCut;
        h->owned = fits;
    }
    h->row_seg_start.clear();
    // Binned handles whose groups are dense (>= 480 events per (tile, bin) group on average: four groups would not fit a segment
    // anyway): segments of <= 3 groups, so that the voxel K3 needs 768 accumulator cells per channel instead of 3072 -- 40 KB of LDS
    // per workgroup instead of 58, i.e. FOUR 512-thread workgroups per CU with 4 events per thread instead of two 1024-thread ones
    // with 2 (the per-thread fixed costs of K3 are spread over twice the events).  CMAX_SMALL_ACC=0 / 1 overrides (A/B runs).
    static const int small_env = getenv("CMAX_SMALL_ACC") ? atoi(getenv("CMAX_SMALL_ACC")) : -1;
    h->small_acc = T > 1 && h->owned && !h->big && (small_env >= 0 ? small_env != 0 : h->n >= (int64_t)480 * ngroups);
    if (h->mid && !h->owned) {  // (the mid layout exists for group-aligned work lists only: fall back to the standard cut)
        h->mid = false;
        h->seg_max = kSegMax;
        set_error("build_segments: mid layout chosen for a work list that is not group-aligned");
        return CMAX_ESTATE;
    }
    if (h->owned) {
        const int span_max = T == 1 ? (h->big ? 6 : (h->mid ? 4 : 3)) : (h->small_acc ? (h->mid ? 5 : kAccCellsDense / 256) : kAccCells / 256), row_groups = h->ntc * T;  // dense: kAccCellsDense (x 2 for big segments)
        for (int r0 = 0; r0 < ngroups; r0 += row_groups) {
            h->row_seg_start.push_back((int)segs.size());
            int g = r0;
            while (g < r0 + row_groups) {
                const int gs = g;
                int cnt = 0;
                while (g < r0 + row_groups && g - gs < span_max && cnt + (group_start[g + 1] - group_start[g]) <= kSegCut) {
                    cnt += group_start[g + 1] - group_start[g];
                    ++g;
                }
                segs.push_back(make_int4(group_start[gs], cnt, gs, g - gs));
            }
        }
        h->row_seg_start.push_back((int)segs.size());
    }
    int begin = 0, count = 0, row_of_begin = -1, g0 = 0, g_last = 0;
    auto close = [&]() {
        if (count > 0) segs.push_back(make_int4(begin, count, g0, g_last - g0 + 1));
        count = 0;
    };
    for (int g = 0; g < ngroups && !h->owned; ++g) {
        int b = group_start[g], c = group_start[g + 1] - group_start[g];
        const int trow = (g / T) / h->ntc;
        if (c == 0) continue;
        if (count > 0 && (trow != row_of_begin || g - g0 + 1 > max_groups || (!free_cut && count + c > seg_cap))) close();
        int limit = seg_cap;
        if (!free_cut && c > seg_cap) {
            const int parts = (c + seg_cap - 1) / seg_cap;
            limit = (c + parts - 1) / parts;
        }
        while (c > 0) {
            if (count == 0) {
                begin = b;
                row_of_begin = trow;
                g0 = g;
            }
            g_last = g;
            const int take = c < limit - count ? c : limit - count;
            count += take;
            b += take;
            c -= take;
            if (count >= limit) close();
        }
    }
    close();
    h->nseg = (int)segs.size();
    static const bool debug_segs = getenv("CMAX_DEBUG_SEGS") != nullptr;  // tools/probe_rounds.py
    if (debug_segs)
        fprintf(stderr, "[cmax] work list: n=%lld groups=%d segments=%d owned=%d big=%d small_acc=%d\n", (long long)h->n, ngroups, h->nseg, (int)h->owned,
                (int)h->big, (int)h->small_acc);
    if (h->nseg > h->seg_cap) {
        dev_free(&h->d_segs);
        dev_free(&h->d_gpart);
        dev_free(&h->d_win);
        dev_free(&h->d_shifts);
        // the workspace of cmax_objective_batch is sized by the work list too: allocated again on its next call
        dev_free(&h->bimg);
        dev_free(&h->braw);
        dev_free(&h->bwin);
        dev_free(&h->bshifts);
        h->batch_cap = h->batch_seg_cap = 0;
        h->seg_cap = 0;
        int rc = dev_alloc(h, &h->d_segs, h->nseg);
        if (!rc) rc = dev_alloc(h, &h->d_win, (int64_t)4 * h->nseg);
        if (!rc) rc = dev_alloc(h, &h->d_shifts, (int64_t)4 * h->nseg * kShiftWordsMax);
        if (!rc) rc = dev_alloc(h, &h->d_gpart, (int64_t)4 * h->nseg * 2);
        if (rc) return rc;
        // (K3 reads an offset word only where K1 flagged its block; cleared once so that no word ever holds allocator garbage)
        CMAX_CHECK_HIP(hipMemsetAsync(h->d_shifts, 0, (size_t)4 * h->nseg * kShiftWordsMax * sizeof(unsigned), s));
        h->seg_cap = h->nseg;
    }
    if (h->nseg > 0) {
        // upload from pinned memory, no wait: the buffer is only rewritten after the event below has passed
        if (h->segs_copied) CMAX_CHECK_HIP(hipEventSynchronize(h->segs_copied));
        else CMAX_CHECK_HIP(hipEventCreateWithFlags(&h->segs_copied, hipEventDisableTiming));
        rc = pinned_reserve(&h->hp_segs, &h->hp_segs_cap, h->nseg);
        if (rc) return rc;
        std::memcpy(h->hp_segs, segs.data(), segs.size() * sizeof(int4));
        CMAX_CHECK_HIP(hipMemcpyAsync(h->d_segs, h->hp_segs, segs.size() * sizeof(int4), hipMemcpyHostToDevice, s));
        CMAX_CHECK_HIP(hipEventRecord(h->segs_copied, s));
    }
    // Big segments of an un-binned handle: the compact copy the hot kernels read (6 B/event, one aligned region per segment).
    // CMAX_COMPACT=0 switches it off (A/B runs).
    static const int compact_env = getenv("CMAX_COMPACT") ? atoi(getenv("CMAX_COMPACT")) : 1;
    h->compact = false;
    // ... and long runs only: the dense K3 of a short-run batch that is not `owned` is kGradStrided, whose slot layout reads the 8-byte
    // events -- K1 must then warp with the same fp32 time (K3 follows K1's cells and windows without tests), so such a handle gets no compact
    // copy at all (big segments + short runs = a sparse batch on a very large sensor, or CMAX_BIG_SEG=1).
    if (h->big && T == 1 && !slab && h->n_time_bin == 0 && h->nseg > 0 && compact_env != 0 && h->long_runs) {
        if (h->nseg > h->cev_cap) {
            dev_free(&h->cev);
            h->cev_cap = 0;
            rc = dev_alloc(h, &h->cev, (int64_t)h->nseg * b512::kCompactStride);
            if (rc) return rc;
            h->cev_cap = h->nseg;
        }
        hipLaunchKernelGGL((b512::k_pack_compact<0>), dim3(h->nseg), dim3(b512::kThr), 0, s, (const int4 *)h->d_segs, h->nseg, (const uint2 *)h->evp, h->ntc,
                           h->cev, h->d_flags + 3);
        CMAX_CHECK_HIP(hipGetLastError());
        static const bool debug_compact = getenv("CMAX_DEBUG_COMPACT") != nullptr;
        if (debug_compact) {  // (k_pack_compact raises d_flags[3] when an event lies outside its segment's tile row / 8-tile span: never)
            int bad = 0;
            CMAX_CHECK_HIP(hipMemcpyAsync(&bad, h->d_flags + 3, sizeof(int), hipMemcpyDeviceToHost, s));
            CMAX_CHECK_HIP(hipStreamSynchronize(s));
            if (bad) {
                set_error("k_pack_compact: an event does not fit the compact coordinates of its segment");
                return CMAX_ESTATE;
            }
        }
        h->compact = true;
    }
    return 0;
}

// Order `n_in` events from `src` for h->n_time_bin (cmax_sort_kernels.h) into the handle's SoA, then build the work
// list.  first_flag: d_flags[first_flag .. 3] are cleared (0 for a new batch; 1 keeps "fractional sources" when re-binning).
template <typename SRC>
static int sort_events(cmax_handle_s *h, const SRC &src, int64_t n_in, bool reduce_time, int first_flag, hipStream_t s) {
    const int ntiles = h->ntr * h->ntc, T = h->n_time_bin;
    if (h->cap_alt < h->cap) {  // staging SoA of the bucket pass
        CMAX_CHECK_HIP(hipStreamSynchronize(s));
        dev_free(&h->evp_alt);
        dev_free(&h->rx_alt);
        dev_free(&h->ry_alt);
        dev_free(&h->rl_alt);
        dev_free(&h->tau64_alt);
        int rc = dev_alloc(h, &h->evp_alt, h->cap + 2);
        if (!rc) rc = dev_alloc(h, &h->rx_alt, h->cap);
        if (!rc) rc = dev_alloc(h, &h->ry_alt, h->cap);
        if (!rc) rc = dev_alloc(h, &h->rl_alt, h->cap);
        if (!rc) rc = dev_alloc(h, &h->tau64_alt, h->cap);
        if (rc) return rc;
        h->cap_alt = h->cap;
    }

    SortOut stage = {h->evp_alt, h->rx_alt, h->ry_alt, h->rl_alt, h->tau64_alt};
    SortOut fin = {h->evp, h->rx, h->ry, h->rl, h->tau64};
    unsigned long long *keys = reduce_time ? reinterpret_cast<unsigned long long *>(h->d_tmm) : nullptr;
    // Large batches: the STABLE radix sort (cmax_radix_sort.h) -- every pass streams, the events of a pixel come out by time without
    // k_run_time_sort, and the packed order is a function of the batch alone.  Small ones (a few launches fewer) keep the counting sort.
    static const char *sort_env = getenv("CMAX_SORT");
    const bool radix = sort_env ? strcmp(sort_env, "radix") == 0 : n_in >= kRadixMinEvents;
    if (radix) {
        const RsKey key = {h->ntc, T, (T == 0 || T * 256 <= kTileKeysMax) ? 1 : 0};
        const int ngroups_all = ntiles * (T > 0 ? T : 1);
        int gb = 1;
        while ((1 << gb) < ngroups_all) ++gb;
        // Digits of <= kRsMaxDigitBits bits: a pass writes one stream per (workgroup, digit), and its time follows the number of streams
        // (64M events: 1.41 ms per pass at 1024 digits, 0.53 at 32 -- cmax_radix_sort.h), so more, narrower passes win.
        // Un-binned handles: the whole key (8 pixel bits + tile bits) in equal digits -- every digit of it follows from the source pixel
        // alone.  Binned handles: the pixel byte first (the time bin, which needs the batch's extremes, sits above it), then the group bits.
        int nb[8], sh[8], P = 0, shift = 0;
        static const int digit_bits = getenv("CMAX_RS_BITS") ? std::min(kRsMaxDigitBits, std::max(2, atoi(getenv("CMAX_RS_BITS")))) : kRsMaxDigitBits;  // tuning only
        auto split = [&](int total_bits) {
            const int parts = div_up(total_bits, digit_bits);
            for (int q = 0, done = 0; q < parts; ++q) {
                const int bits = div_up(total_bits - done, parts - q);
                nb[P] = bits;
                sh[P++] = shift;
                shift += bits;
                done += bits;
            }
        };
        if (T == 0) {
            split(8 + gb);
        } else {
            if (key.fine) {
                nb[P] = 8;
                sh[P++] = 0;
                shift = 8;
            }
            split(gb);
        }
        int maxbits = 0;
        for (int q = 0; q < P; ++q) maxbits = std::max(maxbits, nb[q]);
        const int nwg = (int)std::min<int64_t>(kRsMaxGroups, std::max<int64_t>(1, div_up(n_in, (int64_t)kRsChunk)));
        const int64_t range = (int64_t)div_up(div_up(n_in, (int64_t)nwg), (int64_t)kRsThreads) * kRsThreads;
        const int64_t need = ((int64_t)1 << maxbits) * nwg + 1;
        if (need > h->rs_hist_cap) {
            CMAX_CHECK_HIP(hipStreamSynchronize(s));
            dev_free(&h->rs_hist);
            dev_free(&h->rs_tmp);
            h->rs_hist_cap = 0;
            int rc = dev_alloc(h, &h->rs_hist, need);
            if (!rc) rc = dev_alloc(h, &h->rs_tmp, div_up(need, (int64_t)kScanChunk) + 2);
            if (rc) return rc;
            h->rs_hist_cap = need;
        }
        hipLaunchKernelGGL(k_rs_clear, dim3(div_up(std::max(ntiles, 4), 256)), dim3(256), 0, s, h->d_active, ntiles, h->d_flags, first_flag, keys);
        if (!key.fine && keys)  // (the first digit needs the time bin: the batch's extremes first)
            hipLaunchKernelGGL((k_rs_time_extremes<SRC>), dim3(nwg), dim3(kRsThreads), 0, s, src, n_in, keys);
        // pass q writes buffer (P - 1 - q) % 2 (0: the handle's own arrays, 1: the staging copy), so that the last pass lands in the own arrays.
        // Re-binning reads the own arrays: when pass 0 would write them too, the two sets swap roles first (the source then IS the staging set).
        if (!std::is_same<SRC, RawSource<float>>::value && !std::is_same<SRC, RawSource<double>>::value && (P - 1) % 2 == 0) {
            std::swap(h->evp, h->evp_alt);
            std::swap(h->rx, h->rx_alt);
            std::swap(h->ry, h->ry_alt);
            std::swap(h->rl, h->rl_alt);
            std::swap(h->tau64, h->tau64_alt);
            std::swap(stage, fin);
        }
        // the number of packed events (every pass's scan writes it again): where the other pipeline keeps its total, outside the histogram
        // buffer the next pass overwrites
        int *total = h->counts + ntiles;
        for (int q = 0; q < P; ++q) {
            const int D = 1 << nb[q], m = D * nwg;
            const SortOut &out = (P - 1 - q) % 2 == 0 ? fin : stage;
            const SortOut &in = (P - 1 - q) % 2 == 0 ? stage : fin;
            if (q == 0) hipLaunchKernelGGL((k_rs_hist_src<SRC>), dim3(nwg), dim3(kRsThreads), (size_t)D * sizeof(int), s, src, n_in, range, key, nb[q], h->rs_hist, h->d_flags, keys);
            else hipLaunchKernelGGL(k_rs_hist, dim3(nwg), dim3(kRsThreads), (size_t)D * sizeof(int), s, (const uint2 *)in.evp, (const int *)total, range, key, sh[q], nb[q], h->rs_hist);
            launch_scan_buf(h->rs_hist, m, h->rs_tmp, s, nullptr, total);
            const size_t lds = (size_t)D * (sizeof(int) + sizeof(unsigned long long));
            if (q == 0) hipLaunchKernelGGL((k_rs_scatter_src<SRC>), dim3(nwg), dim3(kRsThreads), lds, s, src, n_in, range, key, nb[q], (const int *)h->rs_hist, (const int *)h->d_flags, out);
            else hipLaunchKernelGGL(k_rs_scatter, dim3(nwg), dim3(kRsThreads), lds, s, in, (const int *)total, range, key, sh[q], nb[q], (const int *)h->rs_hist, (const int *)h->d_flags, out);
            CMAX_CHECK_LAUNCH();
        }
        hipLaunchKernelGGL(k_rs_meta, dim3(div_up(n_in, 256)), dim3(256), 0, s, (const uint2 *)fin.evp, (const int *)total, key, ngroups_all, h->d_tile_start,
                           T == 0 ? h->d_active : (int *)nullptr, std::max(1, ntiles), keys);
        CMAX_CHECK_LAUNCH();
    } else {
    const int grid = (int)div_up(n_in, (int64_t)kSortChunk);
    hipLaunchKernelGGL(k_sort_clear, dim3(div_up(ntiles + 1, 256)), dim3(256), 0, s, h->counts, h->cursor, ntiles, h->d_flags, first_flag, keys);
    hipLaunchKernelGGL((k_bucket_hist<SRC>), dim3(grid), dim3(kSortThreads), 0, s, src, n_in, h->ntc, ntiles, h->counts, h->d_flags, keys);
    launch_scan(h, ntiles, s);
    hipLaunchKernelGGL((k_bucket_scatter<SRC>), dim3(grid), dim3(kSortThreads), 0, s, src, n_in, h->ntc, ntiles, T, h->counts, h->cursor, h->d_flags, stage);
    hipLaunchKernelGGL(k_tile_sort, dim3(ntiles), dim3(kTileSortThreads), 0, s, ntiles, T, h->counts, stage, fin, h->d_tile_start, h->d_active, h->d_flags, keys);
    CMAX_CHECK_LAUNCH();
    }
    if (T > 0 && h->slab_major) {  // slab-major group order (cmax_sort_kernels.h, S4b): final SoA -> staging SoA, the two swap roles
        const int ngroups = ntiles * T;
        hipLaunchKernelGGL(k_slab_offsets, dim3(1), dim3(1024), 0, s, h->d_tile_start, ngroups, h->ntc, T, h->cursor);
        hipLaunchKernelGGL(k_slab_regroup, dim3(ngroups), dim3(256), 0, s, h->d_tile_start, h->cursor, h->ntc, T, fin, stage, h->d_flags);
        CMAX_CHECK_LAUNCH();
        CMAX_CHECK_HIP(hipMemcpyAsync(h->d_tile_start, h->cursor, (size_t)(ngroups + 1) * sizeof(int), hipMemcpyDeviceToDevice, s));
        std::swap(h->evp, h->evp_alt);
        std::swap(h->rx, h->rx_alt);
        std::swap(h->ry, h->ry_alt);
        std::swap(h->rl, h->rl_alt);
        std::swap(h->tau64, h->tau64_alt);
    }
    static const bool run_sort = !getenv("CMAX_NO_RUN_SORT");
    BatchReadback rb;
    rb.n_in = n_in;
    return build_segments(h, T > 0 ? 1 : 256, s, &rb, [&]() -> int {
        if (T == 0 && run_sort && !radix) {
            // pixel runs ordered by time: final SoA -> staging SoA, then the two swap roles (same capacities).  Runs on the GPU
            // while the host cuts the segments: it needs nothing from the host and changes nothing the host reads back.
            hipLaunchKernelGGL(k_run_time_sort, dim3(div_up(n_in, kRunSortChunk)), dim3(256), 0, s, fin, stage, h->counts + ntiles, h->d_flags);
            CMAX_CHECK_LAUNCH();
            std::swap(h->evp, h->evp_alt);
            std::swap(h->rx, h->rx_alt);
            std::swap(h->ry, h->ry_alt);
            std::swap(h->rl, h->rl_alt);
            std::swap(h->tau64, h->tau64_alt);
        }
        return 0;
    });
}

bool handle_is_deterministic(cmax_handle_t h) { return h && h->deterministic; }

void handle_get_eval_state(cmax_handle_t h, HandleEvalState *out) {
    out->cur_buf = h->cur_buf;
    out->zero_mask[0] = h->zero_mask[0];
    out->zero_mask[1] = h->zero_mask[1];
    out->orig_valid = h->orig_valid ? 1 : 0;
    out->orig_cost = h->orig_cost;
    out->orig_omit = h->orig_omit;
    out->orig_sigma = h->orig_sigma;
    for (int k = 0; k < 4; ++k) out->last_iwe[k] = h->last_iwe[k];
    out->generation = h->generation;
    out->profiling = h->profiling ? 1 : 0;
    out->deterministic = h->deterministic ? 1 : 0;
    out->mu_buf = h->mu_buf;
}

uint64_t handle_state_key(const HandleEvalState &st, int kind) {
    uint64_t sb;
    std::memcpy(&sb, &st.orig_sigma, sizeof(sb));
    uint64_t k = (uint64_t)kind;
    k = k * 1000003u + (uint64_t)st.cur_buf;
    k = k * 1000003u + st.zero_mask[0];
    k = k * 1000003u + st.zero_mask[1];
    k = k * 1000003u + (uint64_t)st.orig_valid;
    k = k * 1000003u + (uint64_t)st.deterministic;
    k = k * 1000003u + (uint64_t)st.mu_buf;
    k = k * 1000003u + (uint64_t)(st.orig_cost + 7);
    k = k * 1000003u + (uint64_t)(st.orig_omit + 7);
    k = k * 1000003u + sb;
    return k;
}

void handle_set_eval_state(cmax_handle_t h, const HandleEvalState *in) {
    h->cur_buf = in->cur_buf;
    h->zero_mask[0] = in->zero_mask[0];
    h->zero_mask[1] = in->zero_mask[1];
    h->orig_valid = in->orig_valid != 0;
    h->orig_cost = in->orig_cost;
    h->orig_omit = in->orig_omit;
    h->orig_sigma = in->orig_sigma;
    for (int k = 0; k < 4; ++k) h->last_iwe[k] = in->last_iwe[k];
    h->mu_buf = in->mu_buf;
}

}  // namespace cmax

using namespace cmax;

extern "C" {

int cmax_create(int H, int W, int ph, int pw, cmax_handle_t *out) {
    CMAX_REQUIRE(out != nullptr, "create: out");
    CMAX_REQUIRE(H > 0 && W > 0 && H <= 4096 && W <= 4096 && ph >= 0 && pw >= 0, "create: image size must be in 1..4096");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device visible");
        return CMAX_ENODEV;
    }
    cmax_handle_s *h = new cmax_handle_s();
    h->H = H;
    h->W = W;
    h->ph = ph;
    h->pw = pw;
    h->Hp = H + 2 * ph;
    h->Wp = W + 2 * pw;
    CMAX_CHECK_HIP(hipGetDevice(&h->device));
    h->ntr = div_up(H, kTile);
    h->ntc = div_up(W, kTile);
    h->nkeys = h->ntr * h->ntc * kTile * kTile;
    const int64_t npix = (int64_t)h->Hp * h->Wp;
    int rc = dev_alloc(h, &h->imgs, 2 * 5 * npix);
    for (int k = 0; k < 5 && !rc; ++k) rc = dev_alloc(h, &h->iweb[k], npix);
    if (!rc) rc = dev_alloc(h, &h->G, 4 * npix);  // dL/dIWE of up to 4 reference times
    if (!rc) rc = dev_alloc(h, &h->Gt, npix);
    {
        const int ntiles = h->ntr * h->ntc;
        const int64_t nints = 4 /* tmm */ + 4 /* flags */ + ntiles + ((int64_t)ntiles * 256 + 1);  // tile starts: up to 255 time bins per tile
        if (!rc) rc = dev_alloc(h, &h->d_batch, nints);
        if (!rc) {
            h->d_tmm = reinterpret_cast<double *>(h->d_batch);
            h->d_flags = h->d_batch + 4;
            h->d_active = h->d_batch + 8;
            h->d_tile_start = h->d_batch + 8 + ntiles;
        }
    }
    if (!rc) rc = dev_alloc(h, &h->d_stat, kStatSlots * kStatStride);
    if (!rc) rc = dev_alloc(h, &h->counts, h->nkeys + 1);
    if (!rc) rc = dev_alloc(h, &h->cursor, h->nkeys);
    if (!rc) rc = dev_alloc(h, &h->scan_tmp, div_up(h->nkeys, kScanChunk) + 1);
    if (!rc) rc = dev_alloc(h, &h->d_raw, 4 * kRawStride);
    if (!rc) rc = dev_alloc(h, &h->d_host_result, 8);
    if (!rc) rc = dev_alloc(h, &h->d_musum, 2 * 4 * kMuStride);
    if (!rc) rc = dev_alloc(h, &h->d_ticket, kTicketLines * kTicketLineInts);
    if (!rc && hipMemset(h->d_ticket, 0, kTicketLines * kTicketLineInts * sizeof(int)) != hipSuccess) {
        set_error("cmax_create: clearing the arrival counters failed");
        rc = CMAX_ENOMEM;
    }
    if (!rc && hipMemset(h->d_musum, 0, 2 * 4 * kMuStride * sizeof(double)) != hipSuccess) {
        set_error("cmax_create: clearing the accumulators failed");
        rc = CMAX_ENOMEM;
    }
    if (rc) {
        cmax_destroy(h);
        return rc;
    }
    *out = h;
    return 0;
}

int cmax_destroy(cmax_handle_t h) {
    if (!h) return 0;
    comm_destroy(h->comm);
    h->comm = nullptr;
    if (h->comm_stream) (void)hipStreamDestroy(h->comm_stream);
    for (hipEvent_t e : h->band_ev) (void)hipEventDestroy(e);
    dev_free(&h->tan);
    dev_free(&h->d_tanpart);
    dev_free(&h->img64);
    dev_free(&h->d_imax);
    dev_free(&h->g64);
    dev_free(&h->d_det_inv_scale);
    dev_free(&h->Gt_det);
    dev_free(&h->imgs);
    for (int k = 0; k < 5; ++k) dev_free(&h->iweb[k]);
    dev_free(&h->G);
    dev_free(&h->Gt);
    dev_free(&h->d_batch);  // d_tmm, d_flags, d_active and d_tile_start live inside it
    h->d_tmm = nullptr;
    h->d_flags = nullptr;
    h->d_tile_start = nullptr;
    dev_free(&h->d_stat);
    dev_free(&h->d_musum);
    dev_free(&h->d_raw);
    dev_free(&h->d_host_result);
    if (h->d_host_grad) (void)hipFree(h->d_host_grad);
    if (h->hp_out) (void)hipHostFree(h->hp_out);
    dev_free(&h->d_ticket);
    dev_free(&h->hvp_img);
    dev_free(&h->d_stat_tan);
    dev_free(&h->search_range);
    if (h->hp_read) (void)hipHostFree(h->hp_read);
    if (h->hp_segs) (void)hipHostFree(h->hp_segs);
    if (h->segs_copied) (void)hipEventDestroy(h->segs_copied);
    if (h->read_done) (void)hipEventDestroy(h->read_done);
    dev_free(&h->d_gpart);
    dev_free(&h->counts);
    dev_free(&h->cursor);
    dev_free(&h->scan_tmp);
    dev_free(&h->rs_hist);
    dev_free(&h->rs_tmp);
    dev_free(&h->d_segs);
    dev_free(&h->d_win);
    dev_free(&h->d_shifts);
    dev_free(&h->bimg);
    dev_free(&h->braw);
    dev_free(&h->bwin);
    dev_free(&h->bshifts);
    dev_free(&h->evp);
    dev_free(&h->rx);
    dev_free(&h->ry);
    dev_free(&h->rl);
    dev_free(&h->tau64);
    dev_free(&h->cev);
    dev_free(&h->evp_alt);
    dev_free(&h->rx_alt);
    dev_free(&h->ry_alt);
    dev_free(&h->rl_alt);
    dev_free(&h->tau64_alt);
    for (int c = 0; c < CMAX_PROF_CLASSES; ++c)
        for (hipEvent_t e : h->prof_ev[c]) (void)hipEventDestroy(e);
    delete h;
    return 0;
}

int cmax_set_events(cmax_handle_t h, const void *events, int dtype, int64_t n, int have_tminmax, double tmin, double tmax,
                    int n_time_bin, cmax_stream_t stream) {
    CMAX_REQUIRE(h != nullptr, "set_events: handle");
    CMAX_REQUIRE(n >= 0 && n < (int64_t)2147483647 && (n == 0 || events), "set_events: n / events");
    CMAX_REQUIRE(dtype == CMAX_F32 || dtype == CMAX_F64, "set_events: dtype");
    CMAX_REQUIRE(n_time_bin >= 0 && n_time_bin <= 255, "set_events: n_time_bin must be in 0..255");
    hipStream_t s = (hipStream_t)stream;
    h->orig_valid = false;
    h->n = 0;
    h->n_dropped = 0;
    h->n_outside = 0;
    ++h->generation;
    h->generation_counted = ~(uint64_t)0;
    if (n > h->cap) {
        // the old buffers may still be in use by work queued on the stream
        CMAX_CHECK_HIP(hipStreamSynchronize(s));
        dev_free(&h->evp);
        dev_free(&h->rx);
        dev_free(&h->ry);
        dev_free(&h->rl);
        dev_free(&h->tau64);
        int rc = dev_alloc(h, &h->evp, n + 2);  // +2: the vector loads may touch one event past the end
        if (!rc) rc = dev_alloc(h, &h->rx, n);
        if (!rc) rc = dev_alloc(h, &h->ry, n);
        if (!rc) rc = dev_alloc(h, &h->rl, n);
        if (!rc) rc = dev_alloc(h, &h->tau64, n);
        if (rc) return rc;
        h->cap = n;
        dev_free(&h->evp_alt);  // the staging SoA follows (sort_events)
        dev_free(&h->rx_alt);
        dev_free(&h->ry_alt);
        dev_free(&h->rl_alt);
        dev_free(&h->tau64_alt);
        h->cap_alt = 0;
    }
    // global time extremes: given (a time slice of a larger batch), or reduced by the first sort kernel
    if (have_tminmax) {
        CMAX_REQUIRE(tmax >= tmin, "set_events: tmax < tmin");
        hipLaunchKernelGGL(k_tmm_set, dim3(1), dim3(1), 0, s, h->d_tmm, tmin, tmax);
        CMAX_CHECK_LAUNCH();
    } else if (n == 0) {
        hipLaunchKernelGGL(k_tmm_set, dim3(1), dim3(1), 0, s, h->d_tmm, (double)INFINITY, -(double)INFINITY);
        CMAX_CHECK_LAUNCH();
    }
    h->n_time_bin = n_time_bin;
    h->slab_major = false;  // the slab order is a property of the batch cmax_set_time_slabs re-ordered, not of the handle
    if (n == 0) {
        CMAX_CHECK_HIP(hipMemsetAsync(h->d_flags, 0, 4 * sizeof(int), s));
        h->has_frac = false;
        h->nseg = 0;
        return 0;
    }
    // pack + order (tile-major; by pixel or by time bin inside a tile) + work list; one host synchronisation
    const int keyed = have_tminmax ? 0 : 1;
    const int keep = h->keep_outside ? 1 : 0;
    if (dtype == CMAX_F32) return sort_events(h, RawSource<float>{(const float *)events, h->d_tmm, h->H, h->W, keyed, keep}, n, keyed != 0, 0, s);
    return sort_events(h, RawSource<double>{(const double *)events, h->d_tmm, h->H, h->W, keyed, keep}, n, keyed != 0, 0, s);
}

// Re-ordering the packed events (time bins, time slabs) makes a new work list but NOT a new batch: the events the batch dropped /
// kept from off the sensor were counted by cmax_set_events from the raw input; the re-sort reads the packed SoA with those flags
// cleared and must not overwrite the counts (the guards of the dense / voxel objectives read n_outside)
static void bump_generation_keep_counts(cmax_handle_s *h) {
    const bool counted = h->generation_counted == h->generation;
    ++h->generation;
    if (counted) h->generation_counted = h->generation;
}

int cmax_set_time_bins(cmax_handle_t h, int n_time_bin, cmax_stream_t stream) {
    CMAX_REQUIRE(h != nullptr, "set_time_bins: handle");
    CMAX_REQUIRE(n_time_bin >= 0 && n_time_bin <= 255, "set_time_bins: n_time_bin must be in 0..255");
    if (n_time_bin == h->n_time_bin && !h->slab_major) return 0;
    h->n_time_bin = n_time_bin;
    h->slab_major = false;
    bump_generation_keep_counts(h);
    if (h->n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    // re-order the packed events in place (through the staging SoA); [0] "fractional sources" stays what it was
    return sort_events(h, PackedSource{h->evp, h->rx, h->ry, h->rl, h->tau64, h->has_frac ? 1 : 0}, h->n, false, 1, s);
}

int cmax_set_time_slabs(cmax_handle_t h, int n_slab, cmax_stream_t stream) {
    CMAX_REQUIRE(h != nullptr, "set_time_slabs: handle");
    CMAX_REQUIRE(n_slab >= 0 && n_slab <= 64, "set_time_slabs: n_slab must be in 0..64");
    if (n_slab <= 1) n_slab = 0;  // one slab is the un-binned order
    if (n_slab == 0 && !h->slab_major) return h->n_time_bin == 0 ? 0 : cmax_set_time_bins(h, 0, stream);
    if (n_slab == h->n_time_bin && h->slab_major) return 0;
    h->n_time_bin = n_slab;
    h->slab_major = n_slab > 0;
    bump_generation_keep_counts(h);
    if (h->n == 0) return 0;
    return sort_events(h, PackedSource{h->evp, h->rx, h->ry, h->rl, h->tau64, h->has_frac ? 1 : 0}, h->n, false, 1, (hipStream_t)stream);
}

int cmax_iwe(cmax_handle_t h, int model, const float *motion, int T, int ref_mode, double ref_frac, int normalize_t,
             double sigma, float *iwe_out, cmax_stream_t stream) {
    CMAX_REQUIRE(h != nullptr && iwe_out != nullptr, "iwe: handle / output");
    CMAX_REQUIRE(model < 0 || motion != nullptr, "iwe: motion");
    CMAX_REQUIRE(model <= CMAX_MODEL_VOXEL, "iwe: model");
    CMAX_REQUIRE(model != CMAX_MODEL_VOXEL || (T > 0 && T == h->n_time_bin), "iwe: voxel T must match the handle's time bins");
    CMAX_REQUIRE(model != CMAX_MODEL_VOXEL || !h->slab_major, "iwe: a voxel motion needs time BINS (cmax_set_time_bins), this handle is in slab order");
    CMAX_REQUIRE(model <= CMAX_MODEL_2DOF || h->n_outside == 0, "iwe: the batch holds events off the sensor (cmax_set_keep_outside): 2-DoF / un-warped only");
    hipStream_t s = (hipStream_t)stream;
    float *raw = sigma > 0 ? h->G : iwe_out;  // G is scratch outside cmax_objective
    int rc = vote_image(h, model, motion, T, ref_mode, ref_frac, normalize_t, raw, false, -1, s);
    if (rc) return rc;
    const float *img = nullptr;
    return blur_image(h, sigma, raw, iwe_out, &img, s);
}

// the objective descriptor as the kernels see it
static ObjParams obj_params(const cmax_handle_s *h, const cmax_objective_t *d) {
    ObjParams op;
    op.cost = d->cost;
    op.normalized = d->normalized;
    op.minimize = d->minimize;
    op.negate = d->negate;
    op.omit = d->omit_boundary;
    op.n_ref = d->n_ref;
    for (int k = 0; k < 4; ++k) op.mult[k] = d->mult[k];
    op.H = h->Hp;
    op.W = h->Wp;
    op.nsub = stat_subs(h);
    return op;
}

static int check_objective_args(cmax_handle_t h, const cmax_objective_t *d, const float *motion) {
    CMAX_REQUIRE(h && d && motion, "objective: null pointer");
    CMAX_REQUIRE(d->model >= CMAX_MODEL_2DOF && d->model <= CMAX_MODEL_VOXEL, "objective: model");
    CMAX_REQUIRE(d->cost == CMAX_COST_VARIANCE || d->cost == CMAX_COST_GRADMAG, "objective: cost");
    CMAX_REQUIRE(d->n_ref >= 1 && d->n_ref <= 4, "objective: n_ref");
    CMAX_REQUIRE(d->motion_dtype == CMAX_F32 || (d->motion_dtype == CMAX_F64 && d->model == CMAX_MODEL_2DOF),
                 "objective: motion_dtype must be CMAX_F32, or CMAX_F64 for the 2-DoF model");
    CMAX_REQUIRE(d->model != CMAX_MODEL_VOXEL || (d->T > 0 && d->T == h->n_time_bin), "objective: voxel T must match the handle's time bins");
    CMAX_REQUIRE(d->model != CMAX_MODEL_VOXEL || !h->slab_major, "objective: a voxel motion needs time BINS (cmax_set_time_bins), this handle is in slab order");
    CMAX_REQUIRE(d->model == CMAX_MODEL_2DOF || h->n_outside == 0,
                 "objective: the batch holds events off the sensor: only the 2-DoF model is defined for them -- a flow has no value there (cmax_set_keep_outside(h, 0) before cmax_set_events drops them)");
    CMAX_REQUIRE(!d->omit_boundary || (h->Hp > 2 && h->Wp > 2), "objective: image too small for omit_boundary");
    CMAX_REQUIRE((int64_t)(d->model == CMAX_MODEL_VOXEL ? d->T : 1) * 2 * h->H * h->W * 4 < ((int64_t)1 << 32),
                 "objective: the motion field must be smaller than 4 GiB (32-bit byte offsets in the event kernels)");
    return 0;
}

static bool orig_cache_hit(const cmax_handle_s *h, const cmax_objective_t *d) {
    return h->orig_valid && h->orig_sigma == d->sigma && h->orig_cost == d->cost && h->orig_omit == d->omit_boundary;
}

// K3 stores every element of the flow gradient itself (one writer per pixel): needs the group-aligned work list, ONE reference
// time (several would add into the same pixels) and the sort order that matches the model (dense: tiles; voxel: (tile, bin) of
// the same T)
static bool owned_groups_apply(const cmax_handle_s *h, const cmax_objective_t *d, const void *grad) {
    return !h->deterministic && grad && h->owned && h->n > 0 && d->n_ref == 1 &&
           ((d->model == CMAX_MODEL_DENSE && h->n_time_bin == 0) || (d->model == CMAX_MODEL_VOXEL && h->n_time_bin == d->T));
}

// votes of every reference time (+ the un-warped image when needed) into images[0 .. n_images);
// zero_mask bit k: images[k] is already zero (the handle's double-buffered images)
// K1 without its votes: the LDS windows and the fp64 cell decisions (RefArgs::win / shifts) of this warp, for a K3 that follows
static int publish_windows(cmax_handle_s *h, const cmax_objective_t *d, const float *motion, hipStream_t s) {
    RefArgs ra = {};
    ra.win = h->d_win;
    ra.shifts = h->d_shifts;
    ra.windows_only = 1;
    for (int k = 0; k < d->n_ref; ++k) {
        ra.d[k] = ref_fraction(d->ref_mode[k], d->ref_frac[k]);
        ra.d_lo[k] = ref_fraction_lo(d->ref_mode[k], d->ref_frac[k]);
    }
    const EvView ev = ev_view(h);
    const WarpParams wp = warp_params(h, motion, d->T, d->ref_mode[0], d->ref_frac[0], d->normalize_t, d->motion_dtype == CMAX_F64);
    switch (d->model) {
        case CMAX_MODEL_2DOF: launch_vote<CMAX_MODEL_2DOF>(h, ev, wp, ra, d->n_ref, s); break;
        case CMAX_MODEL_DENSE: launch_vote<CMAX_MODEL_DENSE>(h, ev, wp, ra, d->n_ref, s); break;
        default: launch_vote<CMAX_MODEL_VOXEL>(h, ev, wp, ra, d->n_ref, s); break;
    }
    CMAX_CHECK_LAUNCH();
    h->win_motion = nullptr;  // (nobody may take these windows for those of a vote)
    return 0;
}

static int objective_vote(cmax_handle_t h, const cmax_objective_t *d, const float *motion, float *images, unsigned zero_mask,
                          int *n_images_out, hipStream_t s, bool want_mu = false, double *raw_reset = nullptr, double *raw_lines = nullptr) {
    const int64_t npix = (int64_t)h->Hp * h->Wp;
    h->mu_valid = false;
    {
        float *imgs[4];
        for (int k = 0; k < d->n_ref; ++k) imgs[k] = images + k * npix;
        float mu_taps[3] = {0.f, 0.f, d->omit_boundary ? 1.f : 0.f};
        if (want_mu) {
            double k0 = 1, k1 = 0;  // no blur: B = 1_Omega
            if (d->sigma > 0) blur_taps(d->sigma, k0, k1);
            mu_taps[0] = (float)k0;
            mu_taps[1] = (float)k1;
        }
        int rc = vote_images(h, d->model, motion, d->T, d->n_ref, d->ref_mode, d->ref_frac, d->normalize_t, imgs, zero_mask, 0, s, true,
                             want_mu ? mu_taps : nullptr, d->motion_dtype == CMAX_F64, raw_reset, raw_lines);
        if (rc) return rc;
    }
    int n_images = d->n_ref;
    if (d->normalized && !orig_cache_hit(h, d)) {  // un-warped image, once per batch (patch_contrast_base.py:295-301)
        int rc = vote_image(h, -1, nullptr, 0, CMAX_REF_FIRST, 0.0, 1, images + (int64_t)d->n_ref * npix,
                            (zero_mask >> d->n_ref) & 1u, 4, s);
        if (rc) return rc;
        ++n_images;
    }
    *n_images_out = n_images;
    return 0;
}

int cmax_objective_vote(cmax_handle_t h, const cmax_objective_t *d, const void *motion_v, float *images, int *n_images_host,
                        cmax_stream_t stream) {
    const float *motion = static_cast<const float *>(motion_v);  // double theta[2] when d->motion_dtype == CMAX_F64
    int rc = check_objective_args(h, d, motion);
    if (rc) return rc;
    CMAX_REQUIRE(images && n_images_host, "objective_vote: images / n_images_host");
    return objective_vote(h, d, motion, images, 0u, n_images_host, (hipStream_t)stream);
}

// zero_next: base of the image buffer of the NEXT evaluation (k_stats of image k zeroes zero_next[k]) or nullptr
// reuse_windows: the caller guarantees that `motion` still holds the values of the vote that published h->d_win
// (cmax_objective / cmax_objective_dist: same call).  The stand-alone cmax_objective_finish cannot know -- a caching
// allocator hands the same address to a different motion -- and lets K3 derive its windows again.
// raw: the accumulators of the deferred 2-DoF K3 (null: the handle's own); raw_is_reset: K1 of this evaluation cleared them;
// raw_only: stop there -- the caller folds the sums itself (cmax_objective_raw / cmax_objective_host), no finishing launch
static bool deferred_applies(const cmax_handle_s *h, const cmax_objective_t *d, const void *grad) {
    return !h->deterministic && grad && d->model == CMAX_MODEL_2DOF && d->cost == CMAX_COST_VARIANCE && !(d->sigma > 0) && h->n > 0 &&
           (int64_t)h->Hp * h->Wp <= (int64_t)h->nseg * 8192;
}

// every 2-DoF objective of the default mode leaves its gradient as sums in kRawLines lines (deferred: six sums, loss included)
static bool two_dof_lines(const cmax_handle_s *h, const cmax_objective_t *d, const void *grad) {
    return !h->deterministic && grad && d->model == CMAX_MODEL_2DOF && h->n > 0;
}

// C2 in row bands (cmax_comm_set_c2_bands).  WHETHER the gradient is exchanged in bands, and in which, must not depend on anything a
// rank knows about its own slice (a rank without events, or whose work list is not group-aligned, issued ONE all-reduce
// while its peers issued `bands` grouped ones -- different collective sequences, i.e. a hang): the bands follow from the sensor's tile
// rows and the setting alone, every rank issues the same `bands` grouped all-reduces on its second stream, and only HOW its K3 runs
// in front of them is local -- band by band where the work list allows it (overlap), in one launch or not at all otherwise.
static int c2_band_count(const cmax_handle_s *h, const cmax_objective_t *d, const void *grad, cmax::Comm *c2_comm) {
    return (c2_comm && grad && d->model == CMAX_MODEL_DENSE && h->c2_bands > 1) ? std::min(h->c2_bands, h->ntr) : 1;
}
static int c2_prepare_bands(cmax_handle_s *h, int bands) {
    if (!h->comm_stream) CMAX_CHECK_HIP(hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
    while ((int)h->band_ev.size() < bands + 1) {
        hipEvent_t e = nullptr;
        CMAX_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->band_ev.push_back(e);
    }
    return 0;
}
// RANK-CONSISTENT SCALARS (round 6).  After C1 every rank evaluates the contrast on the same reduced image, but with
// fp64 atomics in its own order: result[8] (loss and statistics) agrees across ranks to ~1e-16 relative, not bit for bit -- and every
// rank runs its own copy of the optimiser (src/solver/scipy_autograd/scipy_minimize.py:100-117), where ONE differing Wolfe / Armijo
// comparison means different call sequences, mismatched collectives, a hang.  The gradient is identical by construction (it is
// all-reduced); the scalars are made so by the same exchange: every rank but 0 zeroes its result[8] behind its last writer and the
// eight doubles ride in the gradient's (grouped) all-reduce -- x + 0 + ... + 0 is exact, so all ranks leave with rank 0's bits.  A
// value-only evaluation gets the 64-byte exchange on its own.  A 1-rank communicator runs the same sequence.
static int result_zero_unless_rank0(cmax::Comm *comm, double *result, hipStream_t s) {
    if (comm && result && comm_rank(comm) != 0) CMAX_CHECK_HIP(hipMemsetAsync(result, 0, 8 * sizeof(double), s));
    return 0;
}
static int result_exchange_alone(cmax::Comm *comm, double *result, hipStream_t s) {
    int rc = result_zero_unless_rank0(comm, result, s);
    return rc ? rc : comm_allreduce(comm, result, 8, kCommF64, kCommSum, s);
}
// the gradient's all-reduce with result[8] riding along (one grouped RCCL call)
static int grad_and_result_exchange(cmax::Comm *comm, void *grad, size_t gcount, CommType gtype, double *result, hipStream_t s) {
    int rc = result_zero_unless_rank0(comm, result, s);
    if (rc) return rc;
    void *bufs[2] = {grad, result};
    const size_t counts[2] = {gcount, 8};
    const CommType types[2] = {gtype, kCommF64};
    return comm_allreduce_group(comm, bufs, counts, types, result ? 2 : 1, kCommSum, s);
}

// band b's rows of both channels, all-reduced on the second stream once `s` has passed this point (band 0 carries result[8], see above)
static int c2_exchange_band(cmax_handle_s *h, cmax::Comm *c2_comm, float *grad, int b, int bands, hipStream_t s, double *result = nullptr) {
    const int tr0 = (int)((int64_t)h->ntr * b / bands), tr1 = (int)((int64_t)h->ntr * (b + 1) / bands);
    const int64_t hw = (int64_t)h->H * h->W;
    if (b == 0 && result) {
        int rc = result_zero_unless_rank0(c2_comm, result, s);
        if (rc) return rc;
    }
    CMAX_CHECK_HIP(hipEventRecord(h->band_ev[b], s));
    CMAX_CHECK_HIP(hipStreamWaitEvent(h->comm_stream, h->band_ev[b], 0));
    const int r0 = tr0 * kTile, r1 = std::min(tr1 * kTile, h->H);
    void *bufs[3] = {grad + (int64_t)r0 * h->W, grad + hw + (int64_t)r0 * h->W, result};
    const size_t counts[3] = {(size_t)(r1 - r0) * h->W, (size_t)(r1 - r0) * h->W, 8};
    const CommType types[3] = {kCommF32, kCommF32, kCommF64};
    return comm_allreduce_group(c2_comm, bufs, counts, types, (b == 0 && result) ? 3 : 2, kCommSum, h->comm_stream);
}
// the reduced gradient back to the caller's stream
static int c2_join_bands(cmax_handle_s *h, int bands, hipStream_t s) {
    CMAX_CHECK_HIP(hipEventRecord(h->band_ev[bands], h->comm_stream));
    CMAX_CHECK_HIP(hipStreamWaitEvent(s, h->band_ev[bands], 0));
    return 0;
}
// every band behind a gradient that is already complete on this rank
static int c2_exchange_all_bands(cmax_handle_s *h, cmax::Comm *c2_comm, float *grad, int bands, hipStream_t s, double *result) {
    int rc = c2_prepare_bands(h, bands);
    for (int b = 0; b < bands && !rc; ++b) rc = c2_exchange_band(h, c2_comm, grad, b, bands, s, result);
    return rc ? rc : c2_join_bands(h, bands, s);
}

static int objective_finish(cmax_handle_t h, const cmax_objective_t *d, const float *motion, const float *images, int n_images,
                            float *zero_next, double *result, void *grad, hipStream_t s, bool reuse_windows, double *raw = nullptr,
                            bool raw_is_reset = false, bool raw_only = false, cmax::Comm *c2_comm = nullptr, bool *c2_done = nullptr,
                            volatile unsigned long long *host_flag = nullptr, unsigned long long host_seq = 0) {
    int rc = 0;
    const int Hp = h->Hp, Wp = h->Wp;
    const int64_t npix = (int64_t)Hp * Wp;
    const int64_t gcount = d->model == CMAX_MODEL_2DOF ? 2 : (int64_t)(d->model == CMAX_MODEL_VOXEL ? d->T : 1) * 2 * h->H * h->W;
    const size_t gbytes = d->model == CMAX_MODEL_2DOF ? 2 * sizeof(double) : (size_t)gcount * sizeof(float);

    const ObjParams op = obj_params(h, d);

    // statistics of the un-warped image (slot 4) are cached per batch
    if (d->normalized) {
        if (n_images == d->n_ref + 1) {
            const float *img = nullptr;
            rc = blur_image(h, d->sigma, images + (int64_t)d->n_ref * npix, h->iweb[4], &img, s);
            if (rc) return rc;
            // orig_iwe is NOT boundary-cropped for the variance (normalized_image_variance.py:40-41)
            const int omit_o = d->cost == CMAX_COST_VARIANCE ? 0 : d->omit_boundary;
            rc = launch_stats(h, d->cost, img, omit_o, 4, zero_next ? zero_next + (int64_t)d->n_ref * npix : nullptr, s);
            if (rc) return rc;
            h->orig_valid = true;
            h->orig_sigma = d->sigma;
            h->orig_cost = d->cost;
            h->orig_omit = d->omit_boundary;
        } else if (!orig_cache_hit(h, d)) {
            set_error("objective_finish: normalised cost needs the un-warped image (n_images == n_ref + 1)");
            return CMAX_ESTATE;
        }
    }

    // 2-DoF + plain variance with a gradient: K3 gathers the image statistics itself and k_finish_deferred applies
    // the chain factors -- no K2 launch (unless a workgroup's slice of the image would get long)
    const bool two_dof = d->model == CMAX_MODEL_2DOF;
    const bool fold_var = d->cost == CMAX_COST_VARIANCE && !(d->sigma > 0);
    // deterministic mode: the unfused image path (k_blur3, k_stats with one workgroup per accumulator, k_gimage, k_blur3_adj:
    // every sum in a fixed order) and the integer-accumulating K3
    const bool det = h->deterministic;
    const bool deferred = deferred_applies(h, d, grad);
    const bool lines = two_dof_lines(h, d, grad);  // (includes the deferred case)
    if (raw_only && !lines) {
        set_error("objective_raw: this objective has no raw form (2-DoF, a non-empty batch, not deterministic)");
        return CMAX_EINVAL;
    }
    if (lines && !raw) raw = h->d_raw;
    if (lines && !raw_is_reset) CMAX_CHECK_HIP(hipMemsetAsync(raw, 0, (size_t)d->n_ref * kRawStride * sizeof(double), s));
    // gradient magnitude with a gradient: K2 and K2b (and the blurs) are one kernel: statistics + G image without its chain factor
    const bool fused_gm = !det && grad && d->cost == CMAX_COST_GRADMAG && h->n > 0;
    const bool blur_var = !det && d->cost == CMAX_COST_VARIANCE && d->sigma > 0;  // blur + statistics in one kernel
    // ... with a gradient: + the un-centred, unscaled A' = 2 / (n - 1) blur^T [Ib 1_Omega] (k_blur_stats_adj_var); K3 applies the
    // chain factor (kFoldScale) and, along the image border, the mean
    const bool fused_bv = blur_var && grad && h->n > 0 && reuse_windows && h->mu_valid;  // (decided by objective_eval: blurvar_from_votes)
    // owned groups: K3 stores every element of the flow gradient itself (one writer per pixel) -- nothing to clear.
    // Needs the group-aligned work list, ONE reference time (several would add into the same pixels) and the sort
    // order that matches the model (dense: tiles; voxel: (tile, bin) of the same T).
    const bool owned = owned_groups_apply(h, d, grad);
    // plain variance, not normalised, owned groups, K1 of this call summed its votes: the statistics run inside the K3 launch
    const bool cleared_by_vote = grad && reuse_windows && h->grad_cleared_by_vote == grad && !owned;  // K1 of this call cleared the gradient buffer
    h->grad_cleared_by_vote = nullptr;
    const bool stats_inside = (owned || cleared_by_vote) && fold_var && !d->normalized && reuse_windows && h->mu_valid && !det;
    const bool grad_cleared_by_stats = grad && !two_dof && !owned && !cleared_by_vote && !det && h->n > 0 && gcount % 4 == 0 && ((uintptr_t)grad & 15u) == 0;
    float4 *clear4 = grad_cleared_by_stats ? (float4 *)grad : nullptr;
    const int64_t nclear4 = grad_cleared_by_stats ? gcount / 4 : 0;
    double k0 = 0, k1 = 0;
    if (d->sigma > 0) blur_taps(d->sigma, k0, k1);
    const dim3 gm_grid(div_up(Hp, kGmTileH) * div_up(Wp, kGmTileW), d->n_ref);  // one 8 x 32 pixel tile per workgroup

    // per-reference-time pointers of the image kernels (every kernel class covers all reference times in one launch)
    ImgArgs ia = {};
    for (int k = 0; k < d->n_ref; ++k) {
        ia.in[k] = images + k * npix;
        ia.blurred[k] = h->iweb[k];
        ia.zero[k] = zero_next ? zero_next + k * npix : nullptr;
        ia.G[k] = h->G + k * npix;
        // not normalised: the chain factor of chain_coef is a constant -- the fused image kernels fold it into G and K3 reads a finished image
        ia.chain[k] = op.normalized ? 1.f : (float)((op.negate ? -1.0 : 1.0) * op.mult[k] * (op.minimize ? -1.0 : 1.0));
        h->last_iwe[k] = d->sigma > 0 ? h->iweb[k] : images + k * npix;  // the image the contrast is evaluated on
    }

    // ---- contrast statistics
    if (fused_bv) {
        ProfScope prof(h, kProfStats, s);
        for (int rep = 0; rep < h->prof_repeat; ++rep)
            hipLaunchKernelGGL(k_blur_stats_adj_var, gm_grid, dim3(256), 0, s, ia.in[0], npix, Hp, Wp, ia, (float)k0, (float)k1, d->omit_boundary, op.nsub,
                               h->d_stat, clear4, nclear4, h->d_musum + (int64_t)h->mu_buf * 4 * kMuStride,
                               h->d_musum + (int64_t)(h->mu_buf ^ 1) * 4 * kMuStride, h->n);
        CMAX_CHECK_LAUNCH();
        h->mu_buf ^= 1;  // the next evaluation adds into the buffer this launch cleared
        h->mu_valid = false;
    } else if (blur_var) {
        ProfScope prof(h, kProfStats, s);
        for (int rep = 0; rep < h->prof_repeat; ++rep)
            hipLaunchKernelGGL(k_blur_stats_var, dim3(stat_blocks(h), d->n_ref), dim3(256), 0, s, ia.in[0], npix, Hp, Wp, ia, (float)k0, (float)k1, d->omit_boundary,
                               op.nsub, h->d_stat, clear4, nclear4);
        CMAX_CHECK_LAUNCH();
    } else if (!(deferred || fused_gm || stats_inside)) {  // those get their statistics from K3 / from the fused image kernel below
        for (int k = 0; k < d->n_ref; ++k) {
            const float *img = nullptr;
            rc = blur_image(h, d->sigma, images + k * npix, h->iweb[k], &img, s);
            if (rc) return rc;
            rc = launch_stats(h, d->cost, img, d->omit_boundary, k, ia.zero[k], s, k == 0 ? (float *)clear4 : nullptr, k == 0 ? 4 * nclear4 : 0);
            if (rc) return rc;
        }
    }
    if (!grad || h->n == 0) {  // value only (or a rank without events): the loss needs its own tiny launch
        ProfScope prof(h, kProfFinish, s);
        hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, s, op, h->d_stat, result);
        CMAX_CHECK_LAUNCH();
    }
    if (!grad) return 0;
    const int bands = c2_band_count(h, d, grad, c2_comm);  // the same on every rank
    if (h->n == 0) {  // this rank holds no events of the batch
        CMAX_CHECK_HIP(hipMemsetAsync(grad, 0, gbytes, s));
        if (bands > 1) {  // ... and still takes part in every band's exchange
            rc = c2_exchange_all_bands(h, c2_comm, (float *)grad, bands, s, result);
            if (rc) return rc;
            if (c2_done) *c2_done = true;
        }
        return 0;
    }

    // ---- backward: dL/dIWE (folded into K3 for the plain variance; otherwise a G image per reference time)
    if (!two_dof && !grad_cleared_by_stats && !cleared_by_vote && !owned && !det) CMAX_CHECK_HIP(hipMemsetAsync(grad, 0, gbytes, s));
    const int fold = deferred ? kFoldDeferred
                              : (stats_inside ? kFoldStatsInside
                                              : (fold_var ? kFoldStats : (((fused_gm || fused_bv) && d->normalized) ? kFoldScale : kFoldNone)));
    if (det) {
        // bound of the per-event terms: max |image the contrast is evaluated on| per reference time (integer max: order-free)
        CMAX_CHECK_HIP(hipMemsetAsync(h->d_imax, 0, kStatSlots * sizeof(unsigned), s));
        ImgArgs im = {};
        for (int k = 0; k < d->n_ref; ++k) im.in[k] = h->last_iwe[k];
        hipLaunchKernelGGL(k_image_absmax, dim3(std::min(stream_grid(npix, 1024), 256), d->n_ref), dim3(256), 0, s, im, npix, 0, h->d_imax);
        CMAX_CHECK_LAUNCH();
        if (fold == kFoldNone) {  // dL/dIWE with its chain factor: k_gimage on the (blurred) image, then the blur transpose
            if (d->sigma > 0 && !h->Gt_det) {
                rc = dev_alloc(h, &h->Gt_det, 4 * npix);
                if (rc) return rc;
            }
            float *gdst = d->sigma > 0 ? h->Gt_det : h->G;
            const dim3 igrid(div_up(npix, 256), d->n_ref);
            // the n_ref images are not contiguous (last_iwe[k] = blurred copies or the caller's buffer): one launch each
            for (int k = 0; k < d->n_ref; ++k) {
                if (d->cost == CMAX_COST_VARIANCE)
                    hipLaunchKernelGGL(k_gimage<CMAX_COST_VARIANCE>, dim3(igrid.x), dim3(256), 0, s, h->last_iwe[k], op, k, h->d_stat, gdst + k * npix, (int64_t)0);
                else
                    hipLaunchKernelGGL(k_gimage<CMAX_COST_GRADMAG>, dim3(igrid.x), dim3(256), 0, s, h->last_iwe[k], op, k, h->d_stat, gdst + k * npix, (int64_t)0);
            }
            if (d->sigma > 0) hipLaunchKernelGGL(k_blur3_adj<float>, igrid, dim3(256), 0, s, h->Gt_det, Hp, Wp, (float)k0, (float)k1, h->G, npix);
            CMAX_CHECK_LAUNCH();
        }
        if (!two_dof && gcount > h->g64_cap) {
            CMAX_CHECK_HIP(hipStreamSynchronize(s));
            dev_free(&h->g64);
            h->g64_cap = 0;
            rc = dev_alloc(h, &h->g64, gcount);
            if (rc) return rc;
            h->g64_cap = gcount;
            CMAX_CHECK_HIP(hipMemsetAsync(h->g64, 0, (size_t)gcount * sizeof(long long), s));
        }
    }
    if (fused_bv) {
        // dL/dIWE was written (without mean and chain factor) by the statistics kernel
    } else if (blur_var) {
        ImgArgs ib = ia;
        for (int k = 0; k < d->n_ref; ++k) ib.in[k] = h->iweb[k];
        ProfScope prof(h, kProfGimage, s);
        for (int rep = 0; rep < h->prof_repeat; ++rep)
            hipLaunchKernelGGL(k_gimage_blur_adj_var, dim3(div_up(npix, 256), d->n_ref), dim3(256), 0, s, ib, op, h->d_stat, (float)k0, (float)k1);
        CMAX_CHECK_LAUNCH();
    } else if (fused_gm) {
        ProfScope prof(h, kProfStats, s);
        for (int rep = 0; rep < h->prof_repeat; ++rep) {
            if (d->sigma > 0)  // raw votes -> blurred image, statistics and the finished G (blur transpose included)
                hipLaunchKernelGGL(k_blur_stats_gimage_gm, gm_grid, dim3(256), 0, s, ia.in[0], npix, Hp, Wp, ia, (float)k0, (float)k1, d->omit_boundary, op.nsub,
                                   h->d_stat, clear4, nclear4);
            else
                hipLaunchKernelGGL(k_stats_gimage_gm, gm_grid, dim3(256), 0, s, ia.in[0], npix, Hp, Wp, ia, d->omit_boundary, op.nsub, h->d_stat, clear4, nclear4);
        }
        CMAX_CHECK_LAUNCH();
    }
    // ---- per-event gather, all reference times in one launch
    RefArgs ra = {};
    ra.k0 = 0;
    bool same_vote = reuse_windows && h->win_generation == h->generation && h->win_motion == motion && h->win_model == d->model &&
                     h->win_nref == d->n_ref && h->win_T == d->T && h->win_normalize == d->normalize_t;
    for (int k = 0; k < d->n_ref; ++k) {
        ra.d[k] = ref_fraction(d->ref_mode[k], d->ref_frac[k]);
        ra.d_lo[k] = ref_fraction_lo(d->ref_mode[k], d->ref_frac[k]);
        ra.img[k] = (fold == kFoldNone || fold == kFoldScale) ? h->G + k * npix : const_cast<float *>(h->last_iwe[k]);
        ra.zero[k] = (deferred || stats_inside) ? ia.zero[k] : nullptr;
        same_vote = same_vote && h->win_d[k] == ra.d[k];
    }
    // K3 follows the windows -- and the fp64 cell decisions -- of a K1 launch with exactly this warp: that of the same evaluation, or
    // (the stand-alone cmax_objective_finish, whose vote the handle cannot vouch for) a K1 launch that only publishes them
    if (!same_vote && grad && h->n > 0) {
        rc = publish_windows(h, d, motion, s);
        if (rc) return rc;
    }
    ra.win = h->d_win;
    ra.shifts = h->d_shifts;
    ra.n_events = h->n;
    if (stats_inside) {
        // Sweeps of 4 pixels per thread and statistics workgroup: two (round 3, standard segments: K3 of cfg5 with 1 / 2 / 4 / 8 sweeps 18.8 /
        // 17.9 / 18.3 / 18.3 us) -- unless the gathering workgroups fit the chip in ONE round and the statistics workgroups are what
        // spills into a second: then four, if that makes the launch fit (round 4, mid segments: cfg5's shard is 900 gathering workgroups;
        // with 232 statistics workgroups 1132 > 1024 resident, with 120 it fits -- K3 16.1 -> 14.8 us, the evaluation 28.8 -> 27.4 us;
        // 3 / 4 / 6 / 8 / 16 sweeps: 28.8 / 27.4 / 27.6 / 27.8 / 31.3 us, profiles/r04_ablation.txt 15).  CMAX_STAT_SWEEPS overrides.
        static const int sweeps_env = getenv("CMAX_STAT_SWEEPS") ? std::max(1, atoi(getenv("CMAX_STAT_SWEEPS"))) : 0;
        const int gthreads = grad_threads(h, d->model);
        auto blocks_for = [&](int sweeps) {
            const int64_t per_block = 4 * (int64_t)gthreads * sweeps;
            return (int)std::min<int64_t>(8 * div_up(div_up(npix, per_block), 8), 8 * (kStatBlocksMax / 8));
        };
        int sweeps = sweeps_env ? sweeps_env : 2;
        if (!sweeps_env) {
            const int resident = 256 * (2048 / gthreads), gather = 8 * div_up(h->nseg, 8) * d->n_ref;  // workgroups the chip holds at once
            if (gather <= resident && gather + blocks_for(2) * d->n_ref > resident && gather + blocks_for(4) * d->n_ref <= resident) sweeps = 4;
        }
        ra.stat_blocks = blocks_for(sweeps);
        ra.ticket = h->d_ticket;
        ra.musum[0] = h->d_musum + (int64_t)h->mu_buf * 4 * kMuStride;
        ra.musum_next = h->d_musum + (int64_t)(h->mu_buf ^ 1) * 4 * kMuStride;
        h->mu_buf ^= 1;  // the next evaluation adds into the buffer this launch clears
        h->mu_valid = false;
    }
    if (det) {
        ra.imax = h->d_imax;
        ra.g64 = two_dof ? reinterpret_cast<long long *>(h->d_gpart) : h->g64;  // (d_gpart: [4][nseg][6] doubles, 2 used per segment)
        ra.det_inv_scale = h->d_det_inv_scale;
        ra.n_events = h->n;
    }
    const EvView ev = ev_view(h);
    const WarpParams wp = warp_params(h, motion, d->T, d->ref_mode[0], d->ref_frac[0], d->normalize_t, d->motion_dtype == CMAX_F64);
    // the first workgroup writes the loss (deferred: the finishing step does).  Raw form of an objective with statistics: into doubles
    // 8..15 of the first line of the raw buffer (the sums use doubles 0, 1 of every line)
    double *res = deferred ? nullptr : ((raw_only && lines) ? raw + 8 : result);
    // C2 in row bands (cmax_comm_set_c2_bands): dense objective on an owned work list under a communicator.  The owned K3 STORES
    // every gradient element of its segments' tiles, so after the launch that covers tile rows [a, b) the pixel rows [16 a, 16 b)
    // of both channels are final on this rank: they are all-reduced on the handle's second stream while the caller's stream
    // already runs the next band's K3.  One event per band hands the rows over, one event hands the reduced gradient back.
    if (bands > 1 && owned && (int)h->row_seg_start.size() == h->ntr + 1) {  // K3 band by band, each band's rows exchanged behind it
        rc = c2_prepare_bands(h, bands);
        if (rc) return rc;
        for (int b = 0; b < bands; ++b) {
            const int tr0 = (int)((int64_t)h->ntr * b / bands), tr1 = (int)((int64_t)h->ntr * (b + 1) / bands);
            const int s0 = h->row_seg_start[tr0], s1 = h->row_seg_start[tr1];
            RefArgs rb = ra;
            rb.win += s0;  // (one reference time: the windows of segment i sit at win[i])
            rb.shifts += (int64_t)s0 * (h->big ? 512 : (h->mid ? 384 : 256));
            if (s1 > s0) launch_grad<CMAX_MODEL_DENSE>(h, ev, wp, rb, d->n_ref, fold, op, nullptr, (float *)grad, b == 0 ? res : nullptr, owned, s, s0, s1 - s0);
            CMAX_CHECK_LAUNCH();
            rc = c2_exchange_band(h, c2_comm, (float *)grad, b, bands, s, result);
            if (rc) return rc;
        }
        rc = c2_join_bands(h, bands, s);
        if (rc) return rc;
        if (c2_done) *c2_done = true;
        return 0;
    }
    switch (d->model) {
        case CMAX_MODEL_2DOF: launch_grad<CMAX_MODEL_2DOF>(h, ev, wp, ra, d->n_ref, fold, op, lines ? raw : h->d_gpart, nullptr, res, false, s); break;
        case CMAX_MODEL_DENSE: launch_grad<CMAX_MODEL_DENSE>(h, ev, wp, ra, d->n_ref, fold, op, nullptr, (float *)grad, res, owned, s); break;
        default: launch_grad<CMAX_MODEL_VOXEL>(h, ev, wp, ra, d->n_ref, fold, op, nullptr, (float *)grad, res, owned, s); break;
    }
    CMAX_CHECK_LAUNCH();
    if (det) {
        ProfScope prof(h, kProfFinish, s);
        if (two_dof)
            hipLaunchKernelGGL(k_finish_det, dim3(1), dim3(256), 0, s, reinterpret_cast<long long *>(h->d_gpart), h->nseg, d->n_ref, h->d_det_inv_scale, (double *)grad);
        else
            hipLaunchKernelGGL(k_fixed_to_grad, dim3(stream_grid(gcount, 256)), dim3(256), 0, s, h->g64, (float *)grad, gcount, h->d_det_inv_scale);
        CMAX_CHECK_LAUNCH();
    }
    if (bands > 1) {  // this rank's work list does not allow K3 in bands (or the mode is deterministic): the same exchanges behind one launch
        rc = c2_exchange_all_bands(h, c2_comm, (float *)grad, bands, s, result);
        if (rc) return rc;
        if (c2_done) *c2_done = true;
        return 0;
    }
    if (det) {
    } else if (deferred) {
        if (!raw_only) {
            ProfScope prof(h, kProfFinish, s);
            hipLaunchKernelGGL(k_finish_raw, dim3(1), dim3(64), 0, s, (const double *)raw, op.n_ref, op, h->d_stat, result, (double *)grad, host_flag, host_seq);
            CMAX_CHECK_LAUNCH();
        }
    } else if (two_dof) {
        if (!raw_only) {
            ProfScope prof(h, kProfFinish, s);
            // (host_flag: cmax_objective_host -- `result` and `grad` are pinned host memory; K3's first workgroup wrote the loss there)
            hipLaunchKernelGGL(k_finish_lines, dim3(1), dim3(64), 0, s, raw, d->n_ref, (double *)grad, (const double *)nullptr, (double *)nullptr,
                               host_flag, host_seq);
            CMAX_CHECK_LAUNCH();
        }
    }
    return 0;
}

int cmax_objective_finish(cmax_handle_t h, const cmax_objective_t *d, const void *motion_v, const float *images, int n_images,
                          double *result, void *grad, cmax_stream_t stream) {
    const float *motion = static_cast<const float *>(motion_v);  // double theta[2] when d->motion_dtype == CMAX_F64
    int rc = check_objective_args(h, d, motion);
    if (rc) return rc;
    CMAX_REQUIRE(images && result, "objective_finish: images / result");
    CMAX_REQUIRE(n_images == d->n_ref || n_images == d->n_ref + 1, "objective_finish: n_images");
    return objective_finish(h, d, motion, images, n_images, nullptr, result, grad, (hipStream_t)stream, false);
}

// 2-DoF, plain variance: K1T (image + two tangent images) -> [ONE all-reduce] -> k_tan_stats_var -> k_finish_deferred.
// No pass over the events for the gradient, and across GPUs no second exchange: after the all-reduce every rank holds
// the whole batch's three images and finishes loss and gradient on its own.
static bool tan2_applicable(const cmax_handle_s *h, const cmax_objective_t *d, const void *grad, bool dist) {
    static const int force = getenv("CMAX_TAN2") ? atoi(getenv("CMAX_TAN2")) : -1;  // tuning: 1 = also on one GPU, 0 = never
    if (force == 0 || h->deterministic || !grad) return false;
    if (d->model != CMAX_MODEL_2DOF || d->cost != CMAX_COST_VARIANCE || d->sigma > 0) return false;
    // Rank-consistent scalars: behind the single exchange every rank finishes loss AND gradient on its own -- from the reduced planes in a
    // fixed order (per-workgroup partials, one folding wave: bit-identical on every rank), but a NORMALISED variance also reads the
    // un-warped image's cached statistics, which each rank summed with atomics in its own order.  Across real ranks that objective
    // takes the two-exchange path, whose gradient is all-reduced and whose scalars ride along.
    if (d->normalized && h->comm && comm_nranks(h->comm) > 1) return false;
    if (d->normalized && !(h->orig_valid && h->orig_sigma == d->sigma && h->orig_cost == d->cost && h->orig_omit == d->omit_boundary))
        return false;  // the first evaluation of a batch builds the un-warped image's statistics on the standard path
    return dist || force == 1;
}

static int objective_eval_tan2(cmax_handle_t h, const cmax_objective_t *d, const float *motion, double *result, void *grad, hipStream_t s,
                               cmax::Comm *comm) {
    const int Hp = h->Hp, Wp = h->Wp, nr = d->n_ref;
    const int64_t tsize = (int64_t)3 * Hp * Wp + Hp + Wp;
    const int64_t tstride = (tsize + 3) & ~(int64_t)3;  // 16-byte aligned planes: the clearing stores are 16 bytes wide
    int rc = 0;
    if (!h->tan) {
        rc = dev_alloc(h, &h->tan, 2 * 4 * tstride);
        if (!rc) rc = dev_alloc(h, &h->d_tanpart, 4 * kTanBlocks * 6);
        if (rc) return rc;
        h->tan_zero_mask[0] = h->tan_zero_mask[1] = 0u;
    }
    float *cur = h->tan + (int64_t)h->tan_cur * 4 * tstride, *nxt = h->tan + (int64_t)(h->tan_cur ^ 1) * 4 * tstride;
    RefArgs ra = {};
    for (int k = 0; k < nr; ++k) {
        if (!((h->tan_zero_mask[h->tan_cur] >> k) & 1u)) CMAX_CHECK_HIP(hipMemsetAsync(cur + k * tstride, 0, (size_t)tstride * sizeof(float), s));
        ra.d[k] = ref_fraction(d->ref_mode[k], d->ref_frac[k]);
        ra.d_lo[k] = ref_fraction_lo(d->ref_mode[k], d->ref_frac[k]);
        ra.img[k] = cur + k * tstride;
    }
    const unsigned used = (1u << nr) - 1u;
    h->tan_zero_mask[h->tan_cur] &= ~used;
    const WarpParams wp = warp_params(h, motion, 0, d->ref_mode[0], d->ref_frac[0], d->normalize_t, d->motion_dtype == CMAX_F64);
    if (h->n > 0) {
        const EvView ev = ev_view(h);
        const dim3 grid(8 * ((h->nseg + 7) / 8), nr);
        ProfScope prof(h, kProfVote, s);
        for (int rep = 0; rep < h->prof_repeat; ++rep) {
            if (h->big) {
                if (h->has_frac) hipLaunchKernelGGL((b512::k_vote_tan2<true>), grid, dim3(b512::kThr), 0, s, h->d_segs, h->nseg, ev, wp, ra);
                else hipLaunchKernelGGL((b512::k_vote_tan2<false>), grid, dim3(b512::kThr), 0, s, h->d_segs, h->nseg, ev, wp, ra);
            } else if (h->mid) {
                if (h->has_frac) hipLaunchKernelGGL((m512::k_vote_tan2<true>), grid, dim3(m512::kThr), 0, s, h->d_segs, h->nseg, ev, wp, ra);
                else hipLaunchKernelGGL((m512::k_vote_tan2<false>), grid, dim3(m512::kThr), 0, s, h->d_segs, h->nseg, ev, wp, ra);
            } else if (h->has_frac) hipLaunchKernelGGL((t256::k_vote_tan2<true>), grid, dim3(t256::kThr), 0, s, h->d_segs, h->nseg, ev, wp, ra);
            else hipLaunchKernelGGL((t256::k_vote_tan2<false>), grid, dim3(t256::kThr), 0, s, h->d_segs, h->nseg, ev, wp, ra);
        }
        CMAX_CHECK_LAUNCH();
    }
    if (comm) {  // the ONE exchange of a 2-DoF evaluation: all planes of all reference times are contiguous
        ProfScope prof(h, kProfComm, s);
        rc = comm_allreduce(comm, cur, (size_t)nr * tstride, kCommF32, kCommSum, s);
        if (rc) return rc;
    }
    for (int k = 0; k < nr; ++k) h->last_iwe[k] = cur + k * tstride;  // the raw image is the first plane
    {
        ProfScope prof(h, kProfStats, s);
        for (int rep = 0; rep < h->prof_repeat; ++rep)
            hipLaunchKernelGGL(k_tan_stats_var, dim3(kTanBlocks, nr), dim3(256), 0, s, cur, nxt, tstride, Hp, Wp, d->omit_boundary, d->normalize_t,
                               h->d_tmm, h->d_tanpart);
        CMAX_CHECK_LAUNCH();
    }
    h->tan_zero_mask[h->tan_cur ^ 1] |= used;
    h->tan_cur ^= 1;
    const ObjParams op = obj_params(h, d);
    {
        ProfScope prof(h, kProfFinish, s);
        hipLaunchKernelGGL(k_finish_deferred, dim3(1), dim3(256), 0, s, op, h->d_stat, h->d_tanpart, kTanBlocks, result, (double *)grad);
        CMAX_CHECK_LAUNCH();
    }
    return 0;
}

// vote -> [all-reduce of the images] -> finish -> [all-reduce of the gradient], all on the handle's double-buffered images
// raw_out: non-null = stop at the raw sums of the deferred 2-DoF K3 (cmax_objective_raw; `grad` is then only a non-null marker)
static int objective_eval(cmax_handle_t h, const cmax_objective_t *d, const float *motion, double *result, void *grad, hipStream_t s,
                          cmax::Comm *comm, double *raw_out = nullptr, volatile unsigned long long *host_flag = nullptr,
                          unsigned long long host_seq = 0, bool skip_c2 = false) {
    const bool dist = comm != nullptr;  // also a 1-rank communicator: the same enqueue sequence, RCCL included
    // (never with a host flag: the tangent-image path ends in kernels that know nothing of it -- ADVICE r3)
    if (!raw_out && !host_flag && !skip_c2 && (h->n > 0 || dist) && tan2_applicable(h, d, grad, dist)) return objective_eval_tan2(h, d, motion, result, grad, s, comm);
    if (raw_out && !two_dof_lines(h, d, grad)) {
        set_error("objective_raw: this objective has no raw form (2-DoF, a non-empty batch, not deterministic)");
        return CMAX_EINVAL;
    }
    const int64_t gcount = d->model == CMAX_MODEL_2DOF ? 2 : (int64_t)(d->model == CMAX_MODEL_VOXEL ? d->T : 1) * 2 * h->H * h->W;
    const size_t gbytes = d->model == CMAX_MODEL_2DOF ? 2 * sizeof(double) : (size_t)gcount * sizeof(float);
    // empty batch: loss 0, zero gradient (patch_contrast_base.py:253-255).  A time slice may be empty while the batch is not.
    if (h->n == 0 && !dist) {
        CMAX_CHECK_HIP(hipMemsetAsync(result, 0, 8 * sizeof(double), s));
        if (grad) CMAX_CHECK_HIP(hipMemsetAsync(grad, 0, gbytes, s));
        return 0;
    }
    const int64_t npix = (int64_t)h->Hp * h->Wp;
    float *cur = h->imgs + (int64_t)h->cur_buf * 5 * npix;
    float *nxt = h->imgs + (int64_t)(h->cur_buf ^ 1) * 5 * npix;
    int n_images = 0;
    // blurred variance with a gradient on one GPU: K1 also sums the votes against blur^T 1_Omega, so that ONE image kernel can
    // blur, sum and write dL/dIWE (k_blur_stats_adj_var).  Across GPUs the sum would need its own exchange: two image kernels.
    static const bool no_fused_bv = getenv("CMAX_NO_FUSED_BLURVAR") != nullptr;  // tuning: k_blur_stats_var + k_gimage_blur_adj_var
    const bool blurvar_from_votes = !dist && grad && !h->deterministic && h->n > 0 && d->cost == CMAX_COST_VARIANCE && d->sigma > 0 &&
                                    h->Hp >= 4 && h->Wp >= 4 && !no_fused_bv;
    // plain variance (no blur, not normalised) on owned groups: nothing in the gradient needs the finished statistics once the
    // mean comes from the votes, so K2 runs inside the K3 launch (kFoldStatsInside)
    static const bool no_stats_inside = getenv("CMAX_NO_STATS_INSIDE") != nullptr;  // tuning: k_stats as a launch of its own
    // ... and on work lists that are not group-aligned (K3 adds its gradient with atomics) once K1 clears the gradient buffer -- the
    // other job of the separate statistics launch (round 4: 20M events 88.5 -> see profiles/r04_ablation.txt 16)
    const bool inside_ok = !dist && !no_stats_inside && d->cost == CMAX_COST_VARIANCE && !(d->sigma > 0) && !d->normalized && h->Hp >= 4 && h->Wp >= 4;
    const bool clear_by_vote = inside_ok && grad && !owned_groups_apply(h, d, grad) && d->model != CMAX_MODEL_2DOF && d->n_ref == 1 && !h->deterministic &&
                               h->n > 0 && gcount % 4 == 0 && ((uintptr_t)grad & 15u) == 0;
    if (clear_by_vote) {
        h->vote_clear4 = (float4 *)grad;
        h->vote_nclear4 = gcount / 4;
    }
    const bool var_from_votes = inside_ok && (owned_groups_apply(h, d, grad) || clear_by_vote);
    // 2-DoF objectives: K1 clears the raw sums its K3 adds into -- through its statistics slot when the objective keeps none
    // (deferred), through RefArgs::raw_zero otherwise
    double *raw = two_dof_lines(h, d, grad) ? (raw_out ? raw_out : h->d_raw) : nullptr;
    const bool raw_deferred = raw && deferred_applies(h, d, grad);
    int rc = objective_vote(h, d, motion, cur, h->zero_mask[h->cur_buf], &n_images, s, blurvar_from_votes || var_from_votes,
                            raw_deferred ? raw : nullptr, raw_deferred ? nullptr : raw);
    if (rc) return rc;
    const unsigned used = (1u << n_images) - 1u;
    h->zero_mask[h->cur_buf] &= ~used;  // now holds votes
    if (dist) {  // C1: every reference time (+ the un-warped image) in one call; the images are contiguous
        ProfScope prof(h, kProfComm, s);
        rc = comm_allreduce(comm, cur, (size_t)n_images * npix, kCommF32, kCommSum, s);
        if (rc) return rc;
    }
    bool c2_done = false;  // the gradient was all-reduced in row bands behind K3 already
    rc = objective_finish(h, d, motion, cur, n_images, nxt, result, grad, s, true, raw, raw != nullptr, raw_out != nullptr,
                          dist && grad && !raw_out && !skip_c2 ? comm : nullptr, &c2_done, host_flag, host_seq);
    if (h->mu_valid) {  // K1 summed its votes and nothing consumed (and cleared) them -- an error on the way, or a path that does
        // not use them after all: the next evaluation must find clean accumulators
        (void)hipMemsetAsync(h->d_musum + (int64_t)h->mu_buf * 4 * kMuStride, 0, (size_t)4 * kMuStride * sizeof(double), s);
        h->mu_valid = false;
    }
    if (rc) return rc;
    h->zero_mask[h->cur_buf ^ 1] |= used;  // zeroed by this evaluation's k_stats launches
    h->cur_buf ^= 1;
    if (dist && grad && !raw_out && !c2_done && !skip_c2) {  // C2, result[8] riding along (rank-consistent scalars)
        ProfScope prof(h, kProfComm, s);
        rc = grad_and_result_exchange(comm, grad, (size_t)gcount, d->model == CMAX_MODEL_2DOF ? kCommF64 : kCommF32, result, s);
        if (rc) return rc;
    } else if (dist && !grad && !raw_out && !skip_c2) {  // value only: the 64 bytes on their own
        ProfScope prof(h, kProfComm, s);
        rc = result_exchange_alone(comm, result, s);
        if (rc) return rc;
    }
    return 0;
}

int cmax_objective(cmax_handle_t h, const cmax_objective_t *d, const void *motion_v, double *result, void *grad,
                   cmax_stream_t stream) {
    const float *motion = static_cast<const float *>(motion_v);  // double theta[2] when d->motion_dtype == CMAX_F64
    int rc = check_objective_args(h, d, motion);
    if (rc) return rc;
    CMAX_REQUIRE(result, "objective: result");
    return objective_eval(h, d, motion, result, grad, (hipStream_t)stream, nullptr);
}

int cmax_objective_dist(cmax_handle_t h, const cmax_objective_t *d, const void *motion_v, double *result, void *grad,
                        cmax_stream_t stream) {
    const float *motion = static_cast<const float *>(motion_v);  // double theta[2] when d->motion_dtype == CMAX_F64
    int rc = check_objective_args(h, d, motion);
    if (rc) return rc;
    CMAX_REQUIRE(result, "objective_dist: result");
    return objective_eval(h, d, motion, result, grad, (hipStream_t)stream, h->comm);
}

// K candidate motions in ONE launch pair.  The reference's gradient-free paths evaluate the objective for batches of sampled motions
// (src/solver/base.py:738-758 run_optuna; src/solver/patch_contrast_pyramid.py:363-415), and a line search asks for several steps
// along one direction: each evaluation of such a batch alone is two dependent launches and a finishing one, and at cfg2's size those
// launches' floor is 40 % of it (roofline.launch_floor_us).  Here blockIdx.z of K1 / K3 is the candidate -- every candidate its own
// vote image, raw-sum lines and windows -- and the finishing kernel runs one wave per candidate.
// Fast path: 2-DoF, plain image variance (sigma 0, not normalised), default mode, one GPU -- the headline objective.  Everything
// else is evaluated candidate by candidate inside this call (same results, no speed-up).
int cmax_objective_batch(cmax_handle_t h, const cmax_objective_t *d, const void *motions_v, int K, double *results, void *grads,
                         cmax_stream_t stream) {
    const float *motions = static_cast<const float *>(motions_v);
    int rc = check_objective_args(h, d, motions);
    if (rc) return rc;
    CMAX_REQUIRE(results != nullptr && K >= 1 && K <= 64, "objective_batch: results / 1 <= K <= 64");
    hipStream_t s = (hipStream_t)stream;
    const bool two_dof = d->model == CMAX_MODEL_2DOF;
    const int64_t gcount = two_dof ? 2 : (int64_t)(d->model == CMAX_MODEL_VOXEL ? d->T : 1) * 2 * h->H * h->W;
    const int64_t mfloats = two_dof ? (d->motion_dtype == CMAX_F64 ? 4 : 2) : gcount;  // floats per candidate motion
    const size_t gbytes = two_dof ? 2 * sizeof(double) : (size_t)gcount * sizeof(float);
    const bool fast = K > 1 && grads && deferred_applies(h, d, grads) && !d->normalized && !h->comm && !h->profiling;
    if (!fast) {
        for (int z = 0; z < K; ++z) {
            rc = objective_eval(h, d, motions + z * mfloats, results + 8 * z, grads ? (char *)grads + (size_t)z * gbytes : nullptr, s, nullptr);
            if (rc) return rc;
        }
        return 0;
    }
    const int64_t npix = (int64_t)h->Hp * h->Wp;
    const int nr = d->n_ref;
    if (K > h->batch_cap || h->nseg > h->batch_seg_cap) {
        CMAX_CHECK_HIP(hipStreamSynchronize(s));
        dev_free(&h->bimg);
        dev_free(&h->braw);
        dev_free(&h->bwin);
        dev_free(&h->bshifts);
        const int cap = std::max(K, h->batch_cap), scap = std::max(h->nseg, h->batch_seg_cap);
        h->batch_cap = h->batch_seg_cap = 0;  // (a failed allocation below must not leave a later call believing the workspace is there)
        rc = dev_alloc(h, &h->bimg, (int64_t)2 * cap * 4 * npix);
        if (!rc) rc = dev_alloc(h, &h->braw, (int64_t)cap * 4 * kRawStride);
        if (!rc) rc = dev_alloc(h, &h->bwin, (int64_t)cap * 4 * scap);
        if (!rc) rc = dev_alloc(h, &h->bshifts, (int64_t)cap * 4 * scap * kShiftWordsMax);
        if (rc) return rc;
        CMAX_CHECK_HIP(hipMemsetAsync(h->bshifts, 0, (size_t)cap * 4 * scap * kShiftWordsMax * sizeof(unsigned), s));
        h->batch_cap = cap;
        h->batch_seg_cap = scap;
        h->batch_zero[0] = h->batch_zero[1] = 0;
    }
    // candidate z, reference time k: image at bimg[buffer][(z * nr + k) * npix]; the launch's images are contiguous
    float *cur = h->bimg + (int64_t)h->batch_cur * h->batch_cap * 4 * npix;
    float *nxt = h->bimg + (int64_t)(h->batch_cur ^ 1) * h->batch_cap * 4 * npix;
    if (h->batch_zero[h->batch_cur] < K * nr) CMAX_CHECK_HIP(hipMemsetAsync(cur, 0, (size_t)K * nr * npix * sizeof(float), s));
    const ObjParams op = obj_params(h, d);
    const EvView ev = ev_view(h);
    const WarpParams wp = warp_params(h, motions, 0, d->ref_mode[0], d->ref_frac[0], d->normalize_t, d->motion_dtype == CMAX_F64);
    RefArgs ra = {};
    ra.win = h->bwin;
    ra.shifts = h->bshifts;
    ra.z_img = (int64_t)nr * npix;
    ra.z_raw = (int64_t)nr * kRawStride;
    ra.z_motion = (int)mfloats;
    for (int k = 0; k < nr; ++k) {
        ra.d[k] = ref_fraction(d->ref_mode[k], d->ref_frac[k]);
        ra.d_lo[k] = ref_fraction_lo(d->ref_mode[k], d->ref_frac[k]);
        ra.img[k] = cur + k * npix;
        ra.stat[k] = h->braw + (int64_t)k * kRawStride;  // K1's first workgroup of every (candidate, reference time) resets its lines
        ra.zero[k] = nxt + k * npix;                      // K3 clears the other buffer's image of the same (candidate, reference time)
    }
    ra.n_events = h->n;
    launch_vote<CMAX_MODEL_2DOF>(h, ev, wp, ra, nr, s, K);
    CMAX_CHECK_LAUNCH();
    launch_grad<CMAX_MODEL_2DOF>(h, ev, wp, ra, nr, kFoldDeferred, op, h->braw, nullptr, nullptr, false, s, 0, -1, K);
    CMAX_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_finish_raw, dim3(K), dim3(64), 0, s, (const double *)h->braw, nr, op, h->d_stat, results, (double *)grads,
                       (volatile unsigned long long *)nullptr, 0ull);
    CMAX_CHECK_LAUNCH();
    h->batch_zero[h->batch_cur] = 0;           // holds votes now
    h->batch_zero[h->batch_cur ^ 1] = K * nr;  // K3 cleared the first K * nr images of the other buffer (a larger batch clears it itself)
    h->batch_cur ^= 1;
    return 0;
}

int cmax_objective_has_raw(cmax_handle_t h, const cmax_objective_t *d) {
    if (!h || !d) return 0;
    if (!two_dof_lines(h, d, (const void *)h)) return 0;
    // (a NORMALISED plain variance runs the deferred K3, whose fold needs the un-warped image's statistics from the device)
    return deferred_applies(h, d, (const void *)h) && d->normalized ? 0 : 1;
}

int cmax_objective_raw(cmax_handle_t h, const cmax_objective_t *d, const void *motion_v, double *raw, cmax_stream_t stream) {
    const float *motion = static_cast<const float *>(motion_v);
    int rc = check_objective_args(h, d, motion);
    if (rc) return rc;
    CMAX_REQUIRE(raw, "objective_raw: raw");
    CMAX_REQUIRE(cmax_objective_has_raw(h, d), "objective_raw: this objective has no raw form (see cmax_objective_has_raw)");
    return objective_eval(h, d, motion, nullptr, (void *)raw, (hipStream_t)stream, nullptr, raw);
}

int cmax_finalize_raw_host(cmax_handle_t h, const cmax_objective_t *d, const double *raw_host, double *result_host, double *grad_host) {
    CMAX_REQUIRE(h && d && raw_host && result_host, "finalize_raw_host: null pointer");
    CMAX_REQUIRE(d->n_ref >= 1 && d->n_ref <= 4 && d->model == CMAX_MODEL_2DOF, "finalize_raw_host: descriptor");
    for (int k = 0; k < 8; ++k) result_host[k] = 0.0;
    if (!deferred_applies(h, d, (const void *)h)) {
        // objectives that keep statistics: K3's first workgroup wrote result[8] into doubles 8..15 of the first line, every
        // workgroup added sum dt g into doubles 0, 1 of a line
        for (int k = 0; k < 8; ++k) result_host[k] = raw_host[8 + k];
        if (grad_host) {
            double g0 = 0.0, g1 = 0.0;
            for (int k = 0; k < d->n_ref; ++k)
                for (int l = 0; l < kRawLines; ++l) {
                    g0 += raw_host[(int64_t)k * kRawStride + l * kSubStride];
                    g1 += raw_host[(int64_t)k * kRawStride + l * kSubStride + 1];
                }
            grad_host[0] = g0;
            grad_host[1] = g1;
        }
        return 0;
    }
    CMAX_REQUIRE(!d->normalized, "finalize_raw_host: a normalised plain variance needs the un-warped image's statistics (use cmax_objective_host)");
    const ObjParams op = obj_params(h, d);
    double S[4][6];
    for (int k = 0; k < d->n_ref; ++k)
        for (int q = 0; q < 6; ++q) {
            double a = 0.0;
            for (int l = 0; l < kRawLines; ++l) a += raw_host[(int64_t)k * kRawStride + l * kSubStride + q];
            S[k][q] = a;
        }
    finalize_deferred(op, S, 0.0, result_host, grad_host);
    return 0;
}

// busy-wait for the stream: hipStreamSynchronize sleeps in the driver and wakes up ~10 us late (profiles/r02_solver_objective.txt)
static int spin_until_done(hipStream_t s) {
    hipError_t e;
    while ((e = hipStreamQuery(s)) == hipErrorNotReady) {
    }
    if (e != hipSuccess) {
        set_error("stream error while waiting for the evaluation: %s", hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    std::atomic_signal_fence(std::memory_order_seq_cst);
#endif
}

int cmax_objective_host(cmax_handle_t h, const cmax_objective_t *d, const void *motion_v, double *result_host, void *grad_host,
                        cmax_stream_t stream) {
    const float *motion = static_cast<const float *>(motion_v);
    int rc = check_objective_args(h, d, motion);
    if (rc) return rc;
    CMAX_REQUIRE(result_host, "objective_host: result_host");
    hipStream_t s = (hipStream_t)stream;
    if (grad_host && two_dof_lines(h, d, grad_host)) {
        // 2-DoF objectives: K1 [-> image kernel] -> K3 -> the one-wave finishing kernel, which writes (loss and) gradient STRAIGHT INTO PINNED HOST
        // MEMORY and then a run counter behind them; the host polls that word.  No copy engine and no driver call between the
        // last kernel and the caller (a 4 KB device-to-host copy of the raw sums + hipStreamQuery polling was 5-6 us slower per
        // evaluation, profiles/r03_ablation.txt 10).
        rc = pinned_reserve(&h->hp_out, &h->hp_out_cap, (int64_t)4 * kRawStride);
        if (rc) return rc;
        volatile unsigned long long *flag = reinterpret_cast<volatile unsigned long long *>(h->hp_out + 16);
        const unsigned long long seq = ++h->host_seq;
        for (int k = 0; k < 8; ++k) h->hp_out[k] = 0.0;  // (the kernel writes the entries the descriptor uses)
        rc = objective_eval(h, d, motion, h->hp_out, h->hp_out + 8, s, nullptr, nullptr, flag, seq);
        if (rc) return rc;
        bool seen = false;
        for (long spin = 0; spin < 20000000L; ++spin) {  // tens of ms at most, then ask the runtime
            if (*flag == seq) {
                seen = true;
                break;
            }
            cpu_relax();
        }
        if (!seen) {
            rc = spin_until_done(s);
            if (rc) return rc;
            if (*flag != seq) {
                set_error("objective_host: the finishing kernel did not report (stream finished without its run counter)");
                return CMAX_ESTATE;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        std::memcpy(result_host, h->hp_out, 8 * sizeof(double));
        std::memcpy(grad_host, h->hp_out + 8, 2 * sizeof(double));
        return 0;
    }
    const int64_t gcount = d->model == CMAX_MODEL_2DOF ? 2 : (int64_t)(d->model == CMAX_MODEL_VOXEL ? d->T : 1) * 2 * h->H * h->W;
    const int64_t gbytes = grad_host ? (d->model == CMAX_MODEL_2DOF ? 2 * (int64_t)sizeof(double) : gcount * (int64_t)sizeof(float)) : 0;
    if (gbytes > h->host_grad_bytes) {
        CMAX_CHECK_HIP(hipStreamSynchronize(s));
        if (h->d_host_grad) (void)hipFree(h->d_host_grad);
        h->d_host_grad = nullptr;
        h->host_grad_bytes = 0;
        CMAX_CHECK_HIP(hipMalloc(&h->d_host_grad, (size_t)gbytes));
        h->host_grad_bytes = gbytes;
        h->bytes += gbytes;
    }
    rc = pinned_reserve(&h->hp_out, &h->hp_out_cap, 8 + (gbytes + 7) / 8);
    if (rc) return rc;
    rc = objective_eval(h, d, motion, h->d_host_result, grad_host ? h->d_host_grad : nullptr, s, nullptr);
    if (rc) return rc;
    CMAX_CHECK_HIP(hipMemcpyAsync(h->hp_out, h->d_host_result, 8 * sizeof(double), hipMemcpyDeviceToHost, s));
    if (grad_host) CMAX_CHECK_HIP(hipMemcpyAsync(h->hp_out + 8, h->d_host_grad, (size_t)gbytes, hipMemcpyDeviceToHost, s));
    rc = spin_until_done(s);
    if (rc) return rc;
    std::memcpy(result_host, h->hp_out, 8 * sizeof(double));
    if (grad_host) std::memcpy(grad_host, h->hp_out + 8, (size_t)gbytes);
    return 0;
}

int cmax_set_deterministic(cmax_handle_t h, int enable) {
    CMAX_REQUIRE(h != nullptr, "set_deterministic: handle");
    if (enable) {  // each buffer on its own: a failed allocation must not leave a later call believing everything is there
        const int64_t npix = (int64_t)h->Hp * h->Wp;
        if (!h->img64) {
            int rc = dev_alloc(h, &h->img64, 5 * npix);
            if (rc) return rc;
            if (hipMemset(h->img64, 0, (size_t)5 * npix * sizeof(long long)) != hipSuccess) {
                dev_free(&h->img64);
                set_error("set_deterministic: clearing the integer images failed");
                return CMAX_ENOMEM;
            }
        }
        if (!h->d_imax) {
            int rc = dev_alloc(h, &h->d_imax, 8);  // [0..4]: images of the objective; HVP: [0..3] max |G_k|, [4..7] max |G'_k|
            if (rc) return rc;
        }
        if (!h->d_det_inv_scale) {
            int rc = dev_alloc(h, &h->d_det_inv_scale, 4);
            if (rc) return rc;
        }
    }
    h->deterministic = enable != 0;
    h->win_generation = ~(uint64_t)0;  // windows published by the other mode's K1 stay valid, but keep the state simple
    return 0;
}

int cmax_get_deterministic(cmax_handle_t h, int *enabled) {
    CMAX_REQUIRE(h != nullptr && enabled != nullptr, "get_deterministic");
    *enabled = h->deterministic ? 1 : 0;
    return 0;
}

int cmax_comm_set_c2_bands(cmax_handle_t h, int bands) {
    CMAX_REQUIRE(h != nullptr && bands >= 1 && bands <= 64, "comm_set_c2_bands: bands must be in 1..64");
    h->c2_bands = bands;
    return 0;
}

int cmax_comm_available(char *path_host, int path_capacity) { return comm_available(path_host, path_capacity); }

int cmax_comm_unique_id(void *id_host) {
    CMAX_REQUIRE(id_host != nullptr, "comm_unique_id: id_host");
    return comm_unique_id(id_host);
}

int cmax_comm_init(cmax_handle_t h, const void *id_host, int nranks, int rank) {
    CMAX_REQUIRE(h != nullptr, "comm_init: handle");
    CMAX_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "comm_init: need 0 <= rank < nranks");
    if (h->comm) {
        comm_destroy(h->comm);
        h->comm = nullptr;
    }
    if (nranks == 1 && !id_host) return 0;  // a single slice is the whole batch: no communicator, RCCL is never loaded
    CMAX_REQUIRE(id_host != nullptr, "comm_init: id_host");
    int dev = -1;
    CMAX_CHECK_HIP(hipGetDevice(&dev));
    if (dev != h->device) {
        set_error("comm_init: the current HIP device is %d but the handle lives on device %d", dev, h->device);
        return CMAX_ESTATE;
    }
    return comm_create(id_host, nranks, rank, &h->comm);
}

int cmax_comm_destroy(cmax_handle_t h) {
    CMAX_REQUIRE(h != nullptr, "comm_destroy: handle");
    comm_destroy(h->comm);
    h->comm = nullptr;
    return 0;
}

int cmax_comm_info(cmax_handle_t h, int *nranks, int *rank, int *rccl_version) {
    CMAX_REQUIRE(h != nullptr, "comm_info: handle");
    if (nranks) *nranks = comm_nranks(h->comm);
    if (rank) *rank = comm_rank(h->comm);
    if (rccl_version) *rccl_version = h->comm ? comm_version() : 0;
    return 0;
}

int cmax_comm_allreduce(cmax_handle_t h, void *buf, int64_t count, int dtype, int op, cmax_stream_t stream) {
    CMAX_REQUIRE(h != nullptr && (buf != nullptr || count == 0) && count >= 0, "comm_allreduce: handle / buffer");
    CMAX_REQUIRE(dtype == CMAX_F32 || dtype == CMAX_F64, "comm_allreduce: dtype");
    CMAX_REQUIRE(op >= 0 && op <= 2, "comm_allreduce: op must be 0 (sum), 1 (min) or 2 (max)");
    ProfScope prof(h, kProfComm, (hipStream_t)stream);
    return comm_allreduce(h->comm, buf, (size_t)count, dtype == CMAX_F64 ? kCommF64 : kCommF32,
                          op == 1 ? kCommMin : (op == 2 ? kCommMax : kCommSum), (hipStream_t)stream);
}

}  // extern "C"

namespace cmax {

// tangent votes of every reference time (blockIdx.y) into draw + k * tp.bs (zeroed by the caller)
template <int MODEL>
static void launch_vote_tan(cmax_handle_s *h, const EvView &ev, const WarpParams &wp, const TanParams &tp, int n_ref, float *draw, hipStream_t s,
                            long long *draw64 = nullptr) {
    const dim3 grid(8 * ((h->nseg + 7) / 8), n_ref);
    if (h->big) {
        if (h->has_frac) hipLaunchKernelGGL((b512::k_vote_tan<MODEL, true>), grid, dim3(b512::kThr), 0, s, ev, wp, tp, h->d_segs, h->nseg, draw, h->d_stat_tan, draw64);
        else hipLaunchKernelGGL((b512::k_vote_tan<MODEL, false>), grid, dim3(b512::kThr), 0, s, ev, wp, tp, h->d_segs, h->nseg, draw, h->d_stat_tan, draw64);
    } else if (h->mid) {
        if (h->has_frac) hipLaunchKernelGGL((m512::k_vote_tan<MODEL, true>), grid, dim3(m512::kThr), 0, s, ev, wp, tp, h->d_segs, h->nseg, draw, h->d_stat_tan, draw64);
        else hipLaunchKernelGGL((m512::k_vote_tan<MODEL, false>), grid, dim3(m512::kThr), 0, s, ev, wp, tp, h->d_segs, h->nseg, draw, h->d_stat_tan, draw64);
    } else if (h->has_frac) hipLaunchKernelGGL((t256::k_vote_tan<MODEL, true>), grid, dim3(t256::kThr), 0, s, ev, wp, tp, h->d_segs, h->nseg, draw, h->d_stat_tan, draw64);
    else hipLaunchKernelGGL((t256::k_vote_tan<MODEL, false>), grid, dim3(t256::kThr), 0, s, ev, wp, tp, h->d_segs, h->nseg, draw, h->d_stat_tan, draw64);
}

template <int MODEL>
static void launch_grad_hvp(cmax_handle_s *h, const EvView &ev, const WarpParams &wp, const TanParams &tp, int n_ref, const float *G,
                            const float *Gp, double *gpart, float *hflow, hipStream_t s, const HvpDet &det) {
    const dim3 grid(8 * ((h->nseg + 7) / 8), n_ref);
    if (h->big) {
        if (h->has_frac) hipLaunchKernelGGL((b512::k_grad_hvp<MODEL, true>), grid, dim3(b512::kThr), 0, s, ev, wp, tp, h->d_segs, h->nseg, G, Gp, gpart, hflow, det);
        else hipLaunchKernelGGL((b512::k_grad_hvp<MODEL, false>), grid, dim3(b512::kThr), 0, s, ev, wp, tp, h->d_segs, h->nseg, G, Gp, gpart, hflow, det);
    } else if (h->mid) {
        if (h->has_frac) hipLaunchKernelGGL((m512::k_grad_hvp<MODEL, true>), grid, dim3(m512::kThr), 0, s, ev, wp, tp, h->d_segs, h->nseg, G, Gp, gpart, hflow, det);
        else hipLaunchKernelGGL((m512::k_grad_hvp<MODEL, false>), grid, dim3(m512::kThr), 0, s, ev, wp, tp, h->d_segs, h->nseg, G, Gp, gpart, hflow, det);
    } else if (h->has_frac) hipLaunchKernelGGL((t256::k_grad_hvp<MODEL, true>), grid, dim3(t256::kThr), 0, s, ev, wp, tp, h->d_segs, h->nseg, G, Gp, gpart, hflow, det);
    else hipLaunchKernelGGL((t256::k_grad_hvp<MODEL, false>), grid, dim3(t256::kThr), 0, s, ev, wp, tp, h->d_segs, h->nseg, G, Gp, gpart, hflow, det);
}

}  // namespace cmax

extern "C" {

// comm: the handle holds a time slice of the batch -- the images and the tangent images are all-reduced (ONE grouped call: both are
// sums over events), everything in image space runs redundantly on every rank, the second-order gather covers this rank's events;
// reduce_hv: ... and the product is all-reduced too (cmax_objective_hvp_dist); the patch plan reduces its 2 n_patch numbers instead
static int objective_hvp_impl(cmax_handle_t h, const cmax_objective_t *d, const float *motion, const float *tangent, void *hv, hipStream_t s,
                              cmax::Comm *comm, bool reduce_hv) {
    int rc = 0;
    const bool dist = comm != nullptr;
    const int Hp = h->Hp, Wp = h->Wp;
    const int64_t npix = (int64_t)Hp * Wp;
    const bool two_dof = d->model == CMAX_MODEL_2DOF;
    const int64_t gcount = two_dof ? 2 : (int64_t)(d->model == CMAX_MODEL_VOXEL ? d->T : 1) * 2 * h->H * h->W;
    const size_t gbytes = two_dof ? 2 * sizeof(double) : (size_t)gcount * sizeof(float);
    CMAX_CHECK_HIP(hipMemsetAsync(hv, 0, gbytes, s));
    if (h->n == 0 && !dist) return 0;
    if (!h->hvp_img) {
        rc = dev_alloc(h, &h->hvp_img, 6 * 4 * npix);
        if (!rc) rc = dev_alloc(h, &h->d_stat_tan, 4 * kStatStride);
        if (!rc) {
            dev_free(&h->Gt);
            rc = dev_alloc(h, &h->Gt, 4 * npix);
        }
        if (rc) return rc;
    }
    // [6 kinds][4 reference times][npix]: every kernel of the chain covers all reference times in one launch
    // (blockIdx.y, element stride npix)
    const int64_t bs = npix;
    float *I = h->hvp_img, *Ib = I + 4 * npix, *dI = I + 8 * npix, *dIb = I + 12 * npix, *Gp = I + 16 * npix, *Gpt = I + 20 * npix;
    const int nr = d->n_ref;

    const ObjParams op = obj_params(h, d);
    const int nsub = op.nsub;

    // statistics of the un-warped image (slot 4), cached per batch like in cmax_objective
    if (d->normalized && !orig_cache_hit(h, d)) {
        rc = vote_image(h, -1, nullptr, 0, CMAX_REF_FIRST, 0.0, 1, I, false, 4, s);
        if (rc) return rc;
        if (dist) {
            ProfScope prof(h, kProfComm, s);
            rc = comm_allreduce(comm, I, (size_t)npix, kCommF32, kCommSum, s);
            if (rc) return rc;
        }
        const float *img = nullptr;
        rc = blur_image(h, d->sigma, I, Ib, &img, s);
        if (rc) return rc;
        const int omit_o = d->cost == CMAX_COST_VARIANCE ? 0 : d->omit_boundary;
        rc = launch_stats(h, d->cost, img, omit_o, 4, nullptr, s);
        if (rc) return rc;
        h->orig_valid = true;
        h->orig_sigma = d->sigma;
        h->orig_cost = d->cost;
        h->orig_omit = d->omit_boundary;
    }

    // derivative votes are bounded by 2 |dt|_max (the tangent has unit max-norm): fixed-point scale per reference time
    const double period = d->normalize_t ? 1.0 : (h->tmax_host - h->tmin_host);
    const EvView ev = ev_view(h);
    double k0 = 0, k1 = 0;
    if (d->sigma > 0) blur_taps(d->sigma, k0, k1);
    const dim3 igrid(div_up(npix, 256), nr), sgrid(stat_blocks(h), nr);
    TanParams tp = {};
    tp.u = tangent;
    tp.bs = bs;
    for (int k = 0; k < nr; ++k) {
        const double dref = ref_fraction(d->ref_mode[k], d->ref_frac[k]);
        double dtmax = fabs(dref) > fabs(1.0 - dref) ? fabs(dref) : fabs(1.0 - dref);
        dtmax *= period > 0 ? period : 1.0;
        if (dtmax < 1e-30) dtmax = 1e-30;
        tp.d[k] = (float)dref;
        tp.fixk[k] = (float)(1073741824.0 / ((double)h->seg_max * 2.0 * dtmax));  // 2^30 / (events * max |derivative vote|)
    }
    tp.fix = tp.fixk[0];
    tp.inv_fix = 1.f / tp.fix;
    const WarpParams wp = warp_params(h, motion, d->T, d->ref_mode[0], d->ref_frac[0], d->normalize_t, d->motion_dtype == CMAX_F64);
    // images, their blur and statistics (slots 0 .. nr-1)
    CMAX_CHECK_HIP(hipMemsetAsync(I, 0, (size_t)nr * npix * sizeof(float), s));
    {
        float *imgs[4];
        for (int k = 0; k < nr; ++k) imgs[k] = I + k * npix;
        rc = vote_images(h, d->model, motion, d->T, nr, d->ref_mode, d->ref_frac, d->normalize_t, imgs, 0xFu, 0, s);
        if (rc) return rc;
    }
    // T1: tangent images (their blur follows below)
    CMAX_CHECK_HIP(hipMemsetAsync(dI, 0, (size_t)nr * npix * sizeof(float), s));
    // deterministic mode (round 3): the tangent votes go to the integer images (all zero between uses) and are rounded to fp32
    // once; the second-order gather below accumulates integers -- the product is then bit-identical from run to run like the
    // loss and the gradient
    const bool det = h->deterministic;
    long long *dI64 = det ? h->img64 : nullptr;
    if (h->n > 0) {
        switch (d->model) {
            case CMAX_MODEL_2DOF: launch_vote_tan<CMAX_MODEL_2DOF>(h, ev, wp, tp, nr, dI, s, dI64); break;
            case CMAX_MODEL_DENSE: launch_vote_tan<CMAX_MODEL_DENSE>(h, ev, wp, tp, nr, dI, s, dI64); break;
            default: launch_vote_tan<CMAX_MODEL_VOXEL>(h, ev, wp, tp, nr, dI, s, dI64); break;
        }
    }
    CMAX_CHECK_LAUNCH();
    if (det && h->n > 0) {
        FixedArgs fa = {};
        for (int k = 0; k < nr; ++k) {
            fa.src[k] = h->img64 + (int64_t)k * npix;
            fa.dst[k] = dI + k * npix;
            fa.inv_fixk[k] = 1.0 / (double)tp.fixk[k];
        }
        hipLaunchKernelGGL(k_fixed_to_image, dim3(stream_grid(npix, 256), nr), dim3(256), 0, s, fa, npix);
        CMAX_CHECK_LAUNCH();
    }
    if (dist) {  // C1 of the product: images and tangent images of all reference times, one grouped call
        ProfScope prof(h, kProfComm, s);
        void *bufs[2] = {I, dI};
        const size_t counts[2] = {(size_t)nr * npix, (size_t)nr * npix};
        const CommType types[2] = {kCommF32, kCommF32};
        rc = comm_allreduce_group(comm, bufs, counts, types, 2, kCommSum, s);
        if (rc) return rc;
    }
    if (h->n == 0) {  // this rank holds no events of the batch: nothing to gather, but it takes part in every exchange
        if (dist && reduce_hv) {
            ProfScope prof(h, kProfComm, s);
            rc = comm_allreduce(comm, hv, (size_t)gcount, two_dof ? kCommF64 : kCommF32, kCommSum, s);
        }
        return rc;
    }
    const float *img = I, *dimg = dI;
    if (d->sigma > 0) {
        hipLaunchKernelGGL(k_blur3<float>, igrid, dim3(256), 0, s, I, Hp, Wp, (float)k0, (float)k1, Ib, bs);
        img = Ib;
    }
    if (d->cost == CMAX_COST_VARIANCE)
        hipLaunchKernelGGL(k_stats<CMAX_COST_VARIANCE>, sgrid, dim3(256), 0, s, img, Hp, Wp, d->omit_boundary, nsub, h->d_stat, (float *)nullptr, (float4 *)nullptr, (int64_t)0, bs);
    else
        hipLaunchKernelGGL(k_stats<CMAX_COST_GRADMAG>, sgrid, dim3(256), 0, s, img, Hp, Wp, d->omit_boundary, nsub, h->d_stat, (float *)nullptr, (float4 *)nullptr, (int64_t)0, bs);
    CMAX_CHECK_LAUNCH();
    // (the tangent images were voted above, next to the images)
    if (d->sigma > 0) {
        hipLaunchKernelGGL(k_blur3<float>, igrid, dim3(256), 0, s, dI, Hp, Wp, (float)k0, (float)k1, dIb, bs);
        dimg = dIb;
    }
    // T2: tangent statistics, G (current) and G' (tangent), blur transposes
    float *Gk = d->sigma > 0 ? h->Gt : h->G, *Gpk = d->sigma > 0 ? Gpt : Gp;
    if (d->cost == CMAX_COST_VARIANCE) {
        hipLaunchKernelGGL(k_stats_tan<CMAX_COST_VARIANCE>, sgrid, dim3(256), 0, s, img, dimg, Hp, Wp, d->omit_boundary, nsub, h->d_stat_tan, bs);
        hipLaunchKernelGGL(k_gimage<CMAX_COST_VARIANCE>, igrid, dim3(256), 0, s, img, op, 0, h->d_stat, Gk, bs);
        hipLaunchKernelGGL(k_gimage_tan<CMAX_COST_VARIANCE>, igrid, dim3(256), 0, s, img, dimg, op, 0, h->d_stat, h->d_stat_tan, Gpk, bs);
    } else {
        hipLaunchKernelGGL(k_stats_tan<CMAX_COST_GRADMAG>, sgrid, dim3(256), 0, s, img, dimg, Hp, Wp, d->omit_boundary, nsub, h->d_stat_tan, bs);
        hipLaunchKernelGGL(k_gimage<CMAX_COST_GRADMAG>, igrid, dim3(256), 0, s, img, op, 0, h->d_stat, Gk, bs);
        hipLaunchKernelGGL(k_gimage_tan<CMAX_COST_GRADMAG>, igrid, dim3(256), 0, s, img, dimg, op, 0, h->d_stat, h->d_stat_tan, Gpk, bs);
    }
    if (d->sigma > 0) {
        hipLaunchKernelGGL(k_blur3_adj<float>, igrid, dim3(256), 0, s, h->Gt, Hp, Wp, (float)k0, (float)k1, h->G, bs);
        hipLaunchKernelGGL(k_blur3_adj<float>, igrid, dim3(256), 0, s, Gpt, Hp, Wp, (float)k0, (float)k1, Gp, bs);
    }
    CMAX_CHECK_LAUNCH();
    // T3
    HvpDet hd = {};
    if (det) {
        // the bounds the fixed-point scale is derived from: max |G_k| and max |G'_k| (integer maxima: order-free)
        CMAX_CHECK_HIP(hipMemsetAsync(h->d_imax, 0, 8 * sizeof(unsigned), s));
        ImgArgs ig = {}, ip = {};
        for (int k = 0; k < nr; ++k) {
            ig.in[k] = h->G + k * npix;
            ip.in[k] = Gp + k * npix;
        }
        const dim3 mgrid(std::min(stream_grid(npix, 1024), 256), nr);
        hipLaunchKernelGGL(k_image_absmax, mgrid, dim3(256), 0, s, ig, npix, 0, h->d_imax);
        hipLaunchKernelGGL(k_image_absmax, mgrid, dim3(256), 0, s, ip, npix, 4, h->d_imax);
        CMAX_CHECK_LAUNCH();
        if (!two_dof && gcount > h->g64_cap) {
            CMAX_CHECK_HIP(hipStreamSynchronize(s));
            dev_free(&h->g64);
            h->g64_cap = 0;
            rc = dev_alloc(h, &h->g64, gcount);
            if (rc) return rc;
            h->g64_cap = gcount;
            CMAX_CHECK_HIP(hipMemsetAsync(h->g64, 0, (size_t)gcount * sizeof(long long), s));
        }
        hd.g64 = two_dof ? reinterpret_cast<long long *>(h->d_gpart) : h->g64;  // (d_gpart: [4][nseg][2] doubles = as many 64-bit integers)
        hd.imax = h->d_imax;
        hd.inv_scale = h->d_det_inv_scale;
        hd.n_events = h->n;
        hd.n_ref = nr;
    }
    switch (d->model) {
        case CMAX_MODEL_2DOF: launch_grad_hvp<CMAX_MODEL_2DOF>(h, ev, wp, tp, nr, h->G, Gp, h->d_gpart, nullptr, s, hd); break;
        case CMAX_MODEL_DENSE: launch_grad_hvp<CMAX_MODEL_DENSE>(h, ev, wp, tp, nr, h->G, Gp, nullptr, (float *)hv, s, hd); break;
        default: launch_grad_hvp<CMAX_MODEL_VOXEL>(h, ev, wp, tp, nr, h->G, Gp, nullptr, (float *)hv, s, hd); break;
    }
    CMAX_CHECK_LAUNCH();
    if (det) {
        if (two_dof)
            hipLaunchKernelGGL(k_finish_det, dim3(1), dim3(256), 0, s, reinterpret_cast<long long *>(h->d_gpart), h->nseg, nr, h->d_det_inv_scale, (double *)hv);
        else
            hipLaunchKernelGGL(k_fixed_to_grad, dim3(stream_grid(gcount, 256)), dim3(256), 0, s, h->g64, (float *)hv, gcount, h->d_det_inv_scale);
        CMAX_CHECK_LAUNCH();
    } else if (two_dof) {
        hipLaunchKernelGGL(k_finish, dim3(1), dim3(256), 0, s, h->d_gpart, d->n_ref * h->nseg, (double *)hv);
        CMAX_CHECK_LAUNCH();
    }
    if (dist && reduce_hv) {
        ProfScope prof(h, kProfComm, s);
        rc = comm_allreduce(comm, hv, (size_t)gcount, two_dof ? kCommF64 : kCommF32, kCommSum, s);
        if (rc) return rc;
    }
    return 0;
}

int cmax_objective_hvp(cmax_handle_t h, const cmax_objective_t *d, const void *motion_v, const float *tangent, void *hv,
                       cmax_stream_t stream) {
    const float *motion = static_cast<const float *>(motion_v);  // double theta[2] when d->motion_dtype == CMAX_F64
    int rc = check_objective_args(h, d, motion);
    if (rc) return rc;
    CMAX_REQUIRE(tangent && hv, "objective_hvp: tangent / hv");
    return objective_hvp_impl(h, d, motion, tangent, hv, (hipStream_t)stream, nullptr, false);
}

int cmax_objective_hvp_dist(cmax_handle_t h, const cmax_objective_t *d, const void *motion_v, const float *tangent, void *hv,
                            cmax_stream_t stream) {
    const float *motion = static_cast<const float *>(motion_v);
    int rc = check_objective_args(h, d, motion);
    if (rc) return rc;
    CMAX_REQUIRE(tangent && hv, "objective_hvp_dist: tangent / hv");
    return objective_hvp_impl(h, d, motion, tangent, hv, (hipStream_t)stream, h->comm, true);
}

}  // extern "C"

// library-internal (cmax_solver.hip): the evaluation / the product of a time-sliced batch with the motion gradient LEFT AS THIS
// RANK'S SHARE -- the patch plan carries it through the (linear) adjoints of the voxel chain and of the patch interpolation and
// all-reduces 2 n_patch numbers instead of 2 H W [T]
namespace cmax {
bool handle_has_comm(cmax_handle_t h) { return h && h->comm != nullptr; }
int objective_dist_local_grad(cmax_handle_t h, const cmax_objective_t *d, const float *motion, double *result, void *grad, hipStream_t s) {
    int rc = check_objective_args(h, d, motion);
    if (rc) return rc;
    return objective_eval(h, d, motion, result, grad, s, h->comm, nullptr, nullptr, 0, true);
}
int objective_hvp_dist_local(cmax_handle_t h, const cmax_objective_t *d, const float *motion, const float *tangent, void *hv, hipStream_t s) {
    int rc = check_objective_args(h, d, motion);
    if (rc) return rc;
    return objective_hvp_impl(h, d, motion, tangent, hv, s, h->comm, false);
}
int handle_allreduce_sum(cmax_handle_t h, void *buf, size_t count, bool f64, hipStream_t s) {
    if (!h || !h->comm) return 0;
    ProfScope prof(h, kProfComm, s);
    return comm_allreduce(h->comm, buf, count, f64 ? kCommF64 : kCommF32, kCommSum, s);
}
// ... with `n_scalars` doubles that every rank computed redundantly (the per-term result[8] of the patch plan) made RANK 0'S on every
// rank by the same grouped call (rank-consistent scalars, see result_zero_unless_rank0); buf may be null (value-only evaluation)
int handle_allreduce_sum_with_scalars(cmax_handle_t h, void *buf, size_t count, bool f64, double *scalars, int n_scalars, hipStream_t s) {
    if (!h || !h->comm) return 0;
    ProfScope prof(h, kProfComm, s);
    if (comm_rank(h->comm) != 0) CMAX_CHECK_HIP(hipMemsetAsync(scalars, 0, (size_t)n_scalars * sizeof(double), s));
    void *bufs[2] = {scalars, buf};
    const size_t counts[2] = {(size_t)n_scalars, count};
    const CommType types[2] = {kCommF64, f64 ? kCommF64 : kCommF32};
    return comm_allreduce_group(h->comm, bufs, counts, types, (buf && count) ? 2 : 1, kCommSum, s);
}
}  // namespace cmax

extern "C" {


int cmax_sizeof_objective(void) { return (int)sizeof(cmax_objective_t); }

static void prof_clear(cmax_handle_s *h) {
    for (int c = 0; c < CMAX_PROF_CLASSES; ++c) {
        for (hipEvent_t e : h->prof_ev[c]) (void)hipEventDestroy(e);
        h->prof_ev[c].clear();
    }
}

int cmax_set_profiling(cmax_handle_t h, int enable) {
    CMAX_REQUIRE(h != nullptr, "set_profiling");
    CMAX_REQUIRE(enable >= 0 && enable <= 64, "set_profiling: 0 (off), 1 (bracket every launch) or 2..64 (repeat count)");
    prof_clear(h);
    h->profiling = enable != 0;
    h->prof_repeat = enable > 1 ? enable : 1;
    return 0;
}

int cmax_read_profile_all(cmax_handle_t h, double *total_ms_host, int64_t *count_host) {
    CMAX_REQUIRE(h && total_ms_host && count_host, "read_profile");
    for (int c = 0; c < CMAX_PROF_CLASSES; ++c) {
        double tot = 0.0;
        const size_t np = h->prof_ev[c].size() / 2;
        for (size_t i = 0; i < np; ++i) {
            CMAX_CHECK_HIP(hipEventSynchronize(h->prof_ev[c][2 * i + 1]));
            float ms = 0.f;
            CMAX_CHECK_HIP(hipEventElapsedTime(&ms, h->prof_ev[c][2 * i], h->prof_ev[c][2 * i + 1]));
            tot += (double)ms;
        }
        total_ms_host[c] = tot;
        count_host[c] = (int64_t)np * (c <= kProfGrad ? h->prof_repeat : 1);  // only the four hot classes are repeated
    }
    prof_clear(h);
    return 0;
}

int cmax_read_profile(cmax_handle_t h, double *total_ms_host, int64_t *count_host) {
    CMAX_REQUIRE(h && total_ms_host && count_host, "read_profile");
    double ms[CMAX_PROF_CLASSES];
    int64_t cnt[CMAX_PROF_CLASSES];
    const int rc = cmax_read_profile_all(h, ms, cnt);
    if (rc) return rc;
    for (int c = 0; c < 4; ++c) {
        total_ms_host[c] = ms[c];
        count_host[c] = cnt[c];
    }
    return 0;
}

int cmax_copy_iwe(cmax_handle_t h, int k, float *iwe_out, cmax_stream_t stream) {
    CMAX_REQUIRE(h && iwe_out && k >= 0 && k < 4, "copy_iwe");
    if (!h->last_iwe[k]) {
        set_error("copy_iwe: no objective evaluated yet for slot %d", k);
        return CMAX_ESTATE;
    }
    CMAX_CHECK_HIP(hipMemcpyAsync(iwe_out, h->last_iwe[k], (size_t)h->Hp * h->Wp * sizeof(float), hipMemcpyDeviceToDevice,
                                  (hipStream_t)stream));
    return 0;
}

int cmax_patch_search(cmax_handle_t h, int n_patch, const int *boxes, int img_h, int img_w, int n_cand, const float *cand,
                      double sigma, float *gm_out, int *count_out, cmax_stream_t stream) {
    CMAX_REQUIRE(h && boxes && gm_out && count_out, "patch_search: null argument");
    CMAX_REQUIRE(n_patch > 0 && n_cand >= 0 && (n_cand == 0 || cand), "patch_search: n_patch > 0, n_cand >= 0");
    CMAX_REQUIRE(img_h > 0 && img_w > 0 && sigma >= 0.0, "patch_search: patch image size / sigma");
    CMAX_REQUIRE(n_cand < 65535, "patch_search: at most 65534 candidates per patch");
    const size_t lds = (size_t)2 * img_h * img_w * sizeof(float);
    if (lds > 64 * 1024 - 256) {
        set_error("patch_search: a %d x %d patch image does not fit the 64 KB workgroup LDS budget", img_h, img_w);
        return CMAX_EINVAL;
    }
    if (h->n == 0) {
        set_error("patch_search: no events set");
        return CMAX_ESTATE;
    }
    hipStream_t s = (hipStream_t)stream;
    if (n_patch > h->search_cap) {
        dev_free(&h->search_range);
        h->search_cap = 0;
        int rc = dev_alloc(h, &h->search_range, (size_t)n_patch);
        if (rc) return rc;
        h->search_cap = n_patch;
    }
    CMAX_REQUIRE(h->n_outside == 0, "patch_search: the batch holds events off the sensor (cmax_set_keep_outside)");
    SearchArgs a;
    a.evp = h->evp;
    a.rx = h->rx;
    a.ry = h->ry;
    a.tile_start = h->d_tile_start;
    a.groups_per_tile = h->n_time_bin > 0 ? h->n_time_bin : 1;
    a.ntr = h->ntr;
    a.ntc = h->ntc;
    a.has_frac = h->has_frac ? 1 : 0;
    a.slab_major = h->slab_major && h->n_time_bin > 0 ? 1 : 0;
    a.boxes = (const int4 *)boxes;
    a.img_h = img_h;
    a.img_w = img_w;
    hipLaunchKernelGGL(k_search_range, dim3(n_patch), dim3(kSearchThreads), 0, s, a, h->search_range, count_out);
    CMAX_CHECK_LAUNCH();
    const int radius = sigma > 0.0 ? (int)(4.0 * sigma + 0.5) : 0;  // scipy.ndimage.gaussian_filter, truncate = 4
    hipLaunchKernelGGL(k_patch_search, dim3(n_patch, n_cand + 1), dim3(kSearchThreads), lds, s, a, h->search_range, n_cand,
                       (const float2 *)cand, (float)(h->tmax_host - h->tmin_host), (float)sigma, radius, gm_out);
    CMAX_CHECK_LAUNCH();
    return 0;
}

int cmax_batch_info(cmax_handle_t h, int64_t *n_packed, int64_t *n_dropped, int *has_fractional, int *owned_groups) {
    CMAX_REQUIRE(h != nullptr, "batch_info");
    if (n_packed) *n_packed = h->n;
    if (n_dropped) *n_dropped = h->n_dropped;
    if (has_fractional) *has_fractional = h->has_frac ? 1 : 0;
    if (owned_groups) *owned_groups = h->owned ? 1 : 0;
    return 0;
}

int cmax_set_keep_outside(cmax_handle_t h, int on) {
    CMAX_REQUIRE(h != nullptr, "set_keep_outside");
    h->keep_outside = on != 0;  // takes effect with the next cmax_set_events
    return 0;
}

int cmax_batch_outside(cmax_handle_t h, int64_t *n_outside) {
    CMAX_REQUIRE(h != nullptr && n_outside != nullptr, "batch_outside");
    *n_outside = h->n_outside;
    return 0;
}

int cmax_work_list_info(cmax_handle_t h, int *n_segments, int *segment_events, int *small_accumulators) {
    CMAX_REQUIRE(h != nullptr, "work_list_info");
    if (n_segments) *n_segments = h->nseg;
    if (segment_events) *segment_events = h->seg_max;
    if (small_accumulators) *small_accumulators = h->small_acc ? 1 : 0;
    return 0;
}

int cmax_handle_info(cmax_handle_t h, int64_t *n_events, int64_t *workspace_bytes) {
    CMAX_REQUIRE(h != nullptr, "handle_info");
    if (n_events) *n_events = h->n;
    if (workspace_bytes) *workspace_bytes = h->bytes;
    return 0;
}

// tools/timeline.py: where the event kernels record their phase stamps (device buffer of 2 x 4096 x 8 u64; NULL: off).  Only
// libraries built with -DCMAX_TIMELINE record anything; the others return CMAX_ESTATE.
int cmax_debug_timeline(void *device_buffer) {
#ifdef CMAX_TIMELINE
    unsigned long long *p = static_cast<unsigned long long *>(device_buffer);
    CMAX_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(cmax::g_timeline), &p, sizeof(p)));
    return 0;
#else
    (void)device_buffer;
    set_error("debug_timeline: this library was built without -DCMAX_TIMELINE");
    return CMAX_ESTATE;
#endif
}

int cmax_debug_packed_events(cmax_handle_t h, void *events_out, int *group_start_out, int *n_groups_host, cmax_stream_t stream) {
    CMAX_REQUIRE(h != nullptr && events_out != nullptr, "debug_packed_events");
    const int ngroups = h->ntr * h->ntc * (h->n_time_bin > 0 ? h->n_time_bin : 1);
    if (n_groups_host) *n_groups_host = ngroups;
    if (h->n > 0) CMAX_CHECK_HIP(hipMemcpyAsync(events_out, h->evp, (size_t)h->n * sizeof(uint2), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    if (group_start_out)
        CMAX_CHECK_HIP(hipMemcpyAsync(group_start_out, h->d_tile_start, (size_t)(ngroups + 1) * sizeof(int), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

// Launch floor of the current work list: `pairs` times two dependent EMPTY launches with the grids of K1 and K3 of a single-reference
// objective on this batch (the launch structure of the headline evaluation).  bench.py brackets it with events: what the evaluation
// would cost if its kernels did nothing (profiles/r03_launch_floor.txt measured the same with tools/microbench_launch.hip).
int cmax_debug_launch_floor(cmax_handle_t h, int pairs, cmax_stream_t stream) {
    CMAX_REQUIRE(h != nullptr && pairs > 0, "debug_launch_floor");
    CMAX_REQUIRE(h->n > 0 && h->nseg > 0, "debug_launch_floor: no events set");
    const dim3 grid(8 * ((h->nseg + 7) / 8));
    const int k1 = (h->big || h->mid) ? 512 : (h->nseg > 512 ? 512 : 256), k3 = grad_threads(h, CMAX_MODEL_2DOF);
    for (int i = 0; i < pairs; ++i) {
        hipLaunchKernelGGL(k_empty, grid, dim3(k1), 0, (hipStream_t)stream, h->d_segs);
        hipLaunchKernelGGL(k_empty, grid, dim3(k3), 0, (hipStream_t)stream, h->d_segs);
    }
    CMAX_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
