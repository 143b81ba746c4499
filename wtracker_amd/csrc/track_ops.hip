// Batched per-cycle predictors over a device-resident bbox track: the arithmetic of the reference's
// OptimalController (median head position of the next imaging phase), PolyfitController (weighted least-squares polynomial
// through sampled head centres, extrapolated) and NumpyDataset.create_from_config (ResMLP training pairs), for ALL cycles /
// rows of a track in one launch each.  These are a few dozen flops per cycle: latency-bound gathers, one thread per sample,
// fp64 like the reference's numpy code.  No MFMA, no LDS tiling — there is nothing to tile.
#include "wtk_kernels.h"

#include <cmath>

namespace wtk {
namespace {

template <typename T> __device__ __forceinline__ bool load_center(const T *track, int n_frames, int f, double &cx, double &cy) {
    if (f < 0 || f >= n_frames) return false;
    const double x = (double)track[4 * (long long)f + 0], y = (double)track[4 * (long long)f + 1];
    const double w = (double)track[4 * (long long)f + 2], h = (double)track[4 * (long long)f + 3];
    cx = x + w / 2; // BoxUtils.center / optimal_controller.py:12-14
    cy = y + h / 2;
    return isfinite(cx) && isfinite(cy);
}

__device__ __forceinline__ double median_sorted_insert(double *v, int n) {
    for (int i = 1; i < n; ++i) { // insertion sort: n <= kTrackMaxWindow
        const double key = v[i];
        int j = i - 1;
        while (j >= 0 && v[j] > key) {
            v[j + 1] = v[j];
            --j;
        }
        v[j + 1] = key;
    }
    return (n & 1) ? v[n / 2] : (v[n / 2 - 1] + v[n / 2]) / 2.0; // numpy.median: mean of the two middle values
}

// OptimalController.provide_movement_vector (optimal_controller.py:16-32) up to the camera offset: median centre of the
// finite rows among frames [(cycle+1)*cyc, +imaging)
template <typename T> __global__ __launch_bounds__(64) void track_median_kernel(const TrackMedianArgs a) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= a.n_samples) return;
    const T *track = reinterpret_cast<const T *>(a.track);
    const long long lo = (long long)(a.cycles[i] + 1) * a.cycle_frame_num;
    double xs[kTrackMaxWindow], ys[kTrackMaxWindow];
    int n = 0;
    for (int k = 0; k < a.imaging_frame_num; ++k) {
        const long long f = lo + k;
        double cx, cy;
        if (f < a.n_frames && load_center(track, a.n_frames, (int)f, cx, cy)) xs[n] = cx, ys[n] = cy, ++n;
    }
    if (n == 0) {
        a.pred[2 * i] = a.pred[2 * i + 1] = 0.0;
        a.valid[i] = 0;
        return;
    }
    a.pred[2 * i] = median_sorted_insert(xs, n);
    a.pred[2 * i + 1] = median_sorted_insert(ys, n);
    a.valid[i] = 1;
}

// Singular value decomposition of the n x K matrix A (n <= kTrackMaxTimes, K <= kTrackMaxCoef) by one-sided Jacobi (Hestenes)
// rotations of its COLUMNS, fp64: on return the columns of A are mutually orthogonal (A = U diag(s) as columns, s_p = |A[:,p]|)
// and V holds the accumulated rotations, so A_in = U diag(s) V^T.  Working on the matrix itself (not on the Gram matrix) keeps
// singular values down to ~eps * s_max resolved, which is what numpy's rcond = len(t) * eps needs.
// Two thresholds on the cosine |a_p . a_q| / (|a_p| |a_q|) of a column pair:
//   * a pair is ROTATED while its cosine exceeds 1e-16, i.e. down to the rounding noise of the dot product itself: at <= 16 x 8 a sweep is a few
//     hundred flops, so all 30 sweeps are the normal case, and on the degree-7 / 16-sample problems (scaled-Vandermonde condition ~1e9) the last
//     sub-eps rotations decide on which side of numpy's rcond cut-off the smallest singular value falls — stopping at a few eps (tried in round 4)
//     changed the reference's integer moves of tests/golden/polyfit_highdeg.json;
//   * CONVERGENCE is judged separately ("converged" must be distinguishable from "gave up") and in ABSOLUTE terms: the largest
//     |a_p . a_q| met in the last sweep run must be <= kJacobiConverged = 64 eps (16 samples x 4 eps) times the largest squared column norm,
//     i.e. the singular values are settled to ~eps * s_max — all a least-squares solve that drops everything below rcond * s_max can use.
//     (A relative test on the cosine cannot be met on these problems: a column of norm s_min carries the eps * s_max rounding noise of the
//     rotations that produced it, so its cosines stall near eps * s_max / s_min ~ 1e-7; tried first in round 4, it flagged the degree-7
//     goldens.)  The function returns false otherwise and the caller flags the sample instead of using the decomposition.
constexpr double kJacobiRotate = 1e-16;
constexpr double kJacobiConverged = 64.0 * 2.220446049250313e-16;
__device__ bool jacobi_svd_columns(double (&A)[kTrackMaxTimes][kTrackMaxCoef], double (&V)[kTrackMaxCoef][kTrackMaxCoef], int n, int K) {
    for (int i = 0; i < K; ++i)
        for (int j = 0; j < K; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    double worst = 0.0, top = 0.0; // largest |a_p . a_q| and largest squared column norm met in the most recent sweep
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
        worst = 0.0, top = 0.0;
        for (int p = 0; p < K; ++p)
            for (int q = p + 1; q < K; ++q) {
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
                for (int j = 0; j < n; ++j) alpha += A[j][p] * A[j][p], beta += A[j][q] * A[j][q], gamma += A[j][p] * A[j][q];
                top = fmax(top, fmax(alpha, beta));
                if (!(fabs(gamma) > kJacobiRotate * sqrt(alpha * beta)) || gamma == 0.0) continue; // orthogonal to the last bit (or a zero column)
                worst = fmax(worst, fabs(gamma));
                rotated = true;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int j = 0; j < n; ++j) {
                    const double ap = A[j][p], aq = A[j][q];
                    A[j][p] = c * ap - sn * aq;
                    A[j][q] = sn * ap + c * aq;
                }
                for (int k = 0; k < K; ++k) {
                    const double vp = V[k][p], vq = V[k][q];
                    V[k][p] = c * vp - sn * vq;
                    V[k][q] = sn * vp + c * vq;
                }
            }
        if (!rotated) return true;
    }
    return worst <= kJacobiConverged * top;
}

// PolyfitController.provide_movement_vector (polyfit_controller.py:54-84) up to the camera offsets (the fit commutes with the
// translation by the camera corner): numpy.polynomial.polynomial.polyfit(t, centres, deg, w=weights) restated — weighted
// Vandermonde with columns scaled to unit norm, minimum-norm least squares (numpy.linalg.lstsq = LAPACK gelsd: singular values
// <= rcond * s_max are treated as zero, rcond = len(t) * eps with len(t) = the finite samples) — then polyval at `t_eval`.
// The (<= 16 x 8) problem is solved through a one-sided Jacobi SVD of the scaled matrix itself, so the cut-off is numpy's at
// every admitted degree (an eigen-decomposition of the Gram matrix, which round 2 used, only resolves singular-value ratios
// down to ~3e-7 and truncated directions numpy keeps at degree >= 4-5 over tens of frames).
template <typename T> __global__ __launch_bounds__(64) void track_polyfit_kernel(const TrackPolyfitArgs a) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= a.n_samples) return;
    const T *track = reinterpret_cast<const T *>(a.track);
    const int K = a.degree + 1;
    const long long base = (long long)a.cycles[i] * a.cycle_frame_num;
    double tt[kTrackMaxTimes], ww[kTrackMaxTimes], px[kTrackMaxTimes], py[kTrackMaxTimes];
    int n = 0;
    for (int j = 0; j < a.n_times; ++j) {
        const long long f = base + a.times[j];
        double cx, cy;
        if (f >= 0 && f < a.n_frames && load_center(track, a.n_frames, (int)f, cx, cy)) tt[n] = (double)a.times[j], ww[n] = a.weights[j], px[n] = cx, py[n] = cy, ++n;
    }
    if (n == 0) {
        a.pred[2 * i] = a.pred[2 * i + 1] = 0.0;
        a.valid[i] = 0;
        return;
    }
    // scaled weighted Vandermonde: L[j][p] = w_j t_j^p / scl_p
    double scl[kTrackMaxCoef];
    for (int p = 0; p < K; ++p) scl[p] = 0.0;
    for (int j = 0; j < n; ++j) {
        double tp = 1.0; // t^p by repeated multiplication, as numpy's vander
        for (int p = 0; p < K; ++p) {
            const double v = ww[j] * tp;
            scl[p] += v * v;
            tp *= tt[j];
        }
    }
    for (int p = 0; p < K; ++p) scl[p] = scl[p] > 0.0 ? sqrt(scl[p]) : 1.0;
    double L[kTrackMaxTimes][kTrackMaxCoef], V[kTrackMaxCoef][kTrackMaxCoef];
    for (int j = 0; j < n; ++j) {
        double tp = 1.0;
        for (int p = 0; p < K; ++p) L[j][p] = ww[j] * tp / scl[p], tp *= tt[j];
    }
    if (!jacobi_svd_columns(L, V, n, K)) { // columns still visibly non-orthogonal after 30 sweeps (never seen): "no prediction" rather than a silently wrong fit
        a.pred[2 * i] = a.pred[2 * i + 1] = 0.0;
        a.valid[i] = 0;
        return;
    }
    double s2[kTrackMaxCoef], s2max = 0.0;
    for (int e = 0; e < K; ++e) {
        double q = 0.0;
        for (int j = 0; j < n; ++j) q += L[j][e] * L[j][e];
        s2[e] = q;
        s2max = fmax(s2max, q);
    }
    const double rcond = (double)n * 2.220446049250313e-16; // numpy: len(x) * finfo(float64).eps
    double cx[kTrackMaxCoef], cy[kTrackMaxCoef];
    for (int p = 0; p < K; ++p) cx[p] = cy[p] = 0.0;
    for (int e = 0; e < K; ++e) {
        if (!(s2[e] > rcond * rcond * s2max)) continue; // s_e <= rcond * s_max: null-space direction, the minimum-norm solution leaves it at zero
        double dx = 0.0, dy = 0.0; // (u_e . rhs) / s_e = (L_rot[:,e] . rhs) / s_e^2
        for (int j = 0; j < n; ++j) dx += L[j][e] * (ww[j] * px[j]), dy += L[j][e] * (ww[j] * py[j]);
        dx /= s2[e], dy /= s2[e];
        for (int p = 0; p < K; ++p) cx[p] += V[p][e] * dx, cy[p] += V[p][e] * dy;
    }
    // polyval (Horner, highest power first) of c / scl at t_eval
    double x = cx[K - 1] / scl[K - 1], y = cy[K - 1] / scl[K - 1];
    for (int p = K - 2; p >= 0; --p) x = cx[p] / scl[p] + x * a.t_eval, y = cy[p] / scl[p] + y * a.t_eval;
    a.pred[2 * i] = x;
    a.pred[2 * i + 1] = y;
    a.valid[i] = 1;
}

// NumpyDataset.create_from_config (neural/dataset.py:42-96): row r holds the boxes at r + input_frames and the box centres at
// r + pred_frames; float64 -> float32 FIRST, then relative to the (float32) corner of the first input box; rows with any NaN
// are flagged in `keep` (the caller compacts).  One thread per candidate row.
template <typename T> __global__ __launch_bounds__(64) void track_pairs_kernel(const TrackPairsArgs a) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= a.n_rows) return;
    const T *track = reinterpret_cast<const T *>(a.track);
    const long long r = (long long)a.row0 + i;
    float *X = a.X + (long long)i * 4 * a.n_in;
    float *Y = a.Y + (long long)i * 2 * a.n_out;
    bool ok = true;
    for (int j = 0; j < a.n_in; ++j) {
        const long long f = r + a.in_frames[j];
        for (int c = 0; c < 4; ++c) {
            const double v = (f >= 0 && f < a.n_frames) ? (double)track[4 * f + c] : nan("");
            ok = ok && !isnan(v);
            X[4 * j + c] = (float)v;
        }
    }
    for (int j = 0; j < a.n_out; ++j) {
        const long long f = r + a.out_frames[j];
        double cx = nan(""), cy = nan("");
        if (f >= 0 && f < a.n_frames) {
            cx = (double)track[4 * f + 0] + (double)track[4 * f + 2] / 2;
            cy = (double)track[4 * f + 1] + (double)track[4 * f + 3] / 2;
        }
        ok = ok && !isnan(cx) && !isnan(cy);
        Y[2 * j] = (float)cx;
        Y[2 * j + 1] = (float)cy;
    }
    const float x0 = X[0], y0 = X[1];
    for (int j = 0; j < a.n_out; ++j) Y[2 * j] -= x0, Y[2 * j + 1] -= y0;
    for (int j = 0; j < a.n_in; ++j) X[4 * j] -= x0, X[4 * j + 1] -= y0;
    a.keep[i] = ok ? 1 : 0;
}

// rank of every margin among the batch's (ties broken by row): the K smallest go to slots[rank]; NaN counts as +inf.
// The weak rows (margin < thr) are counted by all 256 threads (wave ballots + one LDS add per wave); what the ceiling K cuts
// off is ADDED to *n_overflow, so that a caller with K < B can see that rows kept their fast result without a second look.
__global__ __launch_bounds__(256) void recheck_select_kernel(const RecheckArgs a) {
    __shared__ float m[1024];
    __shared__ int weak;
    if (threadIdx.x == 0) weak = 0;
    for (int i = threadIdx.x; i < a.B; i += 256) {
        const float v = a.margins[i];
        m[i] = v == v ? v : 3.4e38f;
    }
    __syncthreads();
    int mine = 0;
    for (int i = threadIdx.x; i < a.B; i += 256) {
        const float mi = m[i];
        int rank = 0;
        for (int j = 0; j < a.B; ++j) rank += (m[j] < mi || (m[j] == mi && j < i)) ? 1 : 0;
        if (rank < a.K) a.slots[rank] = i;
        mine += mi < a.thr ? 1 : 0;
    }
    if (a.n_weak || a.n_overflow) {
        for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off, 64);
        if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&weak, mine);
        __syncthreads();
        if (threadIdx.x == 0) {
            const int n = weak;
            if (a.n_weak) *a.n_weak = n < a.K ? n : a.K;
            if (a.n_overflow && n > a.K) *a.n_overflow += n - a.K;  // one block per launch, launches of a stream are ordered: a plain add
        }
    }
}

__global__ __launch_bounds__(64) void recheck_merge_kernel(const RecheckArgs a) {
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= a.K) return;
    const int row = a.slots[k];
    if ((unsigned)row >= (unsigned)a.B || !(a.margins[row] < a.thr)) return;
    reinterpret_cast<float4 *>(a.dst_xywh)[row] = reinterpret_cast<const float4 *>(a.src_xywh)[k];
    if (a.dst_conf && a.src_conf) a.dst_conf[row] = a.src_conf[k];
    if (a.dst_anchor && a.src_anchor) a.dst_anchor[row] = a.src_anchor[k];
    if (a.n_replaced) atomicAdd(a.n_replaced, 1);
}

// ---- deferred second look: queue of weak rows over several fast passes
// one block: queue position of every weak row of the batch (row order), the addresses its re-detected row will be written to, the new length
__global__ __launch_bounds__(256) void recheck_enqueue_plan_kernel(const RecheckQueueArgs a) {
    __shared__ int weak[1024];
    __shared__ int total;
    const int len0 = *a.q_len;
    for (int i = threadIdx.x; i < a.B; i += 256) {
        const float v = a.margins[i];
        weak[i] = (v == v && v < a.thr) ? 1 : 0; // NaN margin = no decision to revisit
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.B; i += 256) {
        int before = 0;
        for (int j = 0; j < i; ++j) before += weak[j];
        int p = -1;
        if (weak[i] && len0 + before < a.q_cap) {
            p = len0 + before;
            a.q_xywh[p] = a.dst_xywh + 4 * (long long)i;
            a.q_conf[p] = a.dst_conf ? a.dst_conf + i : nullptr;
            a.q_anchor[p] = a.dst_anchor ? a.dst_anchor + i : nullptr;
        }
        a.pos[i] = p;
        if (i == a.B - 1) total = before + weak[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int fit = min(total, a.q_cap - len0);
        *a.q_len = len0 + fit;
        if (a.n_overflow && total > fit) *a.n_overflow += total - fit;
    }
}

// blockIdx.y = batch row; its frame goes to its queue slot, 16 bytes per thread and step (rows that are not queued exit at once)
__global__ __launch_bounds__(256) void recheck_enqueue_copy_kernel(const RecheckQueueArgs a) {
    const int b = blockIdx.y;
    const int p = a.pos[b];
    if (p < 0) return;
    const uint4 *src = reinterpret_cast<const uint4 *>(a.frames + (long long)b * a.frame_bytes);
    uint4 *dst = reinterpret_cast<uint4 *>(a.q_frames + (long long)p * a.frame_bytes);
    const long long n16 = a.frame_bytes / 16;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) dst[i] = src[i];
}

__global__ __launch_bounds__(64) void recheck_scatter_kernel(const RecheckQueueArgs a) {
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= *a.q_len || k >= a.q_cap) return;
    *reinterpret_cast<float4 *>(a.q_xywh[k]) = reinterpret_cast<const float4 *>(a.src_xywh)[k];
    if (a.q_conf[k] && a.src_conf) *a.q_conf[k] = a.src_conf[k];
    if (a.q_anchor[k] && a.src_anchor) *a.q_anchor[k] = a.src_anchor[k];
    if (a.n_replaced) atomicAdd(a.n_replaced, 1);
}

// the weak rows' (frame, view centre) for the second look of the views entry point: row i of the second look = batch row slots[i]
__global__ __launch_bounds__(64) void recheck_gather_views_kernel(const int *slots, int K, const int *frame_index, const int *pos_xy, int *idx_out, int *pos_out) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= K) return;
    const int b = slots[i];
    idx_out[i] = frame_index ? frame_index[b] : b;
    pos_out[2 * i] = pos_xy[2 * b], pos_out[2 * i + 1] = pos_xy[2 * b + 1];
}

} // namespace

hipError_t launch_recheck_gather_views(const int *slots, int K, const int *frame_index, const int *pos_xy, int *idx_out, int *pos_out, hipStream_t stream) {
    if (K <= 0 || !slots || !pos_xy || !idx_out || !pos_out) return hipErrorInvalidValue;
    hipLaunchKernelGGL(recheck_gather_views_kernel, dim3((unsigned)((K + 63) / 64)), dim3(64), 0, stream, slots, K, frame_index, pos_xy, idx_out, pos_out);
    return hipGetLastError();
}

hipError_t launch_recheck_enqueue(const RecheckQueueArgs &a, hipStream_t stream) {
    if (a.B <= 0 || a.B > 1024 || a.q_cap <= 0 || a.frame_bytes <= 0 || a.frame_bytes % 16) return hipErrorInvalidValue;
    hipLaunchKernelGGL(recheck_enqueue_plan_kernel, dim3(1), dim3(256), 0, stream, a);
    const long long n16 = a.frame_bytes / 16;
    const unsigned bx = (unsigned)((n16 + 256 * 8 - 1) / (256 * 8) > 64 ? 64 : (n16 + 256 * 8 - 1) / (256 * 8)); // <= 64 blocks per frame, >= 8 steps each
    hipLaunchKernelGGL(recheck_enqueue_copy_kernel, dim3(bx ? bx : 1, (unsigned)a.B), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_recheck_scatter(const RecheckQueueArgs &a, hipStream_t stream) {
    if (a.q_cap <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(recheck_scatter_kernel, dim3((unsigned)((a.q_cap + 63) / 64)), dim3(64), 0, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return hipMemsetAsync(a.q_len, 0, sizeof(int), stream); // the queue is empty again (stream order: after the scatter)
}

hipError_t launch_recheck_select(const RecheckArgs &a, hipStream_t stream) {
    if (a.B <= 0 || a.B > 1024 || a.K <= 0 || a.K > a.B) return hipErrorInvalidValue;
    hipLaunchKernelGGL(recheck_select_kernel, dim3(1), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_recheck_merge(const RecheckArgs &a, hipStream_t stream) {
    if (a.B <= 0 || a.K <= 0 || a.K > a.B) return hipErrorInvalidValue;
    hipLaunchKernelGGL(recheck_merge_kernel, dim3((unsigned)((a.K + 63) / 64)), dim3(64), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_track_median(const TrackMedianArgs &a, int track_f64, hipStream_t stream) {
    if (a.n_samples <= 0 || a.imaging_frame_num <= 0 || a.imaging_frame_num > kTrackMaxWindow || a.cycle_frame_num <= 0) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((a.n_samples + 63) / 64));
    if (track_f64)
        hipLaunchKernelGGL(track_median_kernel<double>, grid, dim3(64), 0, stream, a);
    else
        hipLaunchKernelGGL(track_median_kernel<float>, grid, dim3(64), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_track_polyfit(const TrackPolyfitArgs &a, int track_f64, hipStream_t stream) {
    if (a.n_samples <= 0 || a.n_times <= 0 || a.n_times > kTrackMaxTimes || a.degree < 0 || a.degree + 1 > kTrackMaxCoef || a.cycle_frame_num <= 0)
        return hipErrorInvalidValue;
    const dim3 grid((unsigned)((a.n_samples + 63) / 64));
    if (track_f64)
        hipLaunchKernelGGL(track_polyfit_kernel<double>, grid, dim3(64), 0, stream, a);
    else
        hipLaunchKernelGGL(track_polyfit_kernel<float>, grid, dim3(64), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_track_pairs(const TrackPairsArgs &a, int track_f64, hipStream_t stream) {
    if (a.n_rows <= 0 || a.n_in <= 0 || a.n_in > kTrackMaxTimes || a.n_out <= 0 || a.n_out > kTrackMaxTimes) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((a.n_rows + 63) / 64));
    if (track_f64)
        hipLaunchKernelGGL(track_pairs_kernel<double>, grid, dim3(64), 0, stream, a);
    else
        hipLaunchKernelGGL(track_pairs_kernel<float>, grid, dim3(64), 0, stream, a);
    return hipGetLastError();
}

} // namespace wtk
