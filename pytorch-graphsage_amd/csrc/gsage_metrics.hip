// gsage_metrics.hip -- the metrics of the train / eval log on the device (gfx950).
//
// Replaces ProblemMetrics (reference problem.py:44-64: sklearn f1_score(average = micro | macro) on
// argmax / thresholded predictions, mean absolute error), which train.py:150 calls on EVERY batch after
// copying the predictions and targets to the host.  Here one launch counts true positives / false
// positives / false negatives per class with integer atomics (exact, order-independent), a second,
// single-workgroup launch turns the counts into the two F1 numbers: 8 bytes go to the host instead of
// the [B, C] predictions, and nothing but the print needs them.
//
//   micro = 2 TP / (2 TP + FP + FN) over all classes (sums);  macro = mean of the per-class F1.
//   classification: classes = the labels that occur in y_true or y_pred (sklearn's default label set:
//                   a class nobody has or predicts does not enter the macro mean);
//   multilabel:     every label enters the macro mean, F1 = 0 when it has no positive at all
//                   (sklearn's zero_division default).
#include "gsage_common.h"

namespace gsage {

// counts: int32 [3][C] = tp | fp | fn, then one more: targets outside [0, C) (sklearn would add such a label to
// the label set; the caller gets the count back and falls back to the host metric instead of dropping it)
__global__ void __launch_bounds__(256)
k_metric_counts_cls(const float *__restrict__ logits, int64_t ld, const int64_t *__restrict__ y, int64_t B,
                    int32_t C, int32_t *__restrict__ counts)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < B; i += stride) {
        const float *row = logits + i * ld;
        int best = 0;
        float bv = row[0];
        for (int c = 1; c < C; ++c) {           // first maximum wins, like np.argmax -- and so does the first NaN
            const float v = row[c];
            if (v > bv || (v != v && bv == bv)) { bv = v; best = c; }
        }
        const int64_t t = y[i];
        if (t < 0 || t >= C) {
            atomicAdd(counts + 3 * C, 1);
        } else if (t == best) {
            atomicAdd(counts + best, 1);
        } else {
            atomicAdd(counts + C + best, 1);
            atomicAdd(counts + 2 * C + (int)t, 1);
        }
    }
}

template <typename TY>
__global__ void __launch_bounds__(256)
k_metric_counts_ml(const float *__restrict__ logits, int64_t ld, const TY *__restrict__ y, int64_t ldy, int64_t B,
                   int32_t C, int32_t *__restrict__ counts)
{
    const int64_t total = B * C;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
        const int64_t i = t / C;
        const int c = (int)(t - i * C);
        const bool pred = logits[i * ld + c] > 0.f;
        const bool truth = y[i * ldy + c] != (TY)0;
        if (pred && truth) atomicAdd(counts + c, 1);
        else if (pred) atomicAdd(counts + C + c, 1);
        else if (truth) atomicAdd(counts + 2 * C + c, 1);
    }
}

// out[0] = micro, out[1] = macro, out[2] = targets outside [0, C); one workgroup.  present_only: macro over
// classes with any count.  Doubles: the reference rounds float64 results to 5 decimals (ujson double_precision).
__global__ void __launch_bounds__(256)
k_metric_f1(const int32_t *__restrict__ counts, int32_t C, int present_only, double *__restrict__ out)
{
    __shared__ double s_tp[256], s_fp[256], s_fn[256], s_f1[256], s_n[256];
    double tp = 0, fp = 0, fn = 0, f1 = 0, n = 0;
    for (int c = threadIdx.x; c < C; c += 256) {
        const double a = counts[c], b = counts[C + c], d = counts[2 * C + c];
        tp += a; fp += b; fn += d;
        const double den = 2.0 * a + b + d;
        if (den > 0) f1 += 2.0 * a / den;
        if (!present_only || den > 0) n += 1.0;
    }
    s_tp[threadIdx.x] = tp; s_fp[threadIdx.x] = fp; s_fn[threadIdx.x] = fn; s_f1[threadIdx.x] = f1;
    s_n[threadIdx.x] = n;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            s_tp[threadIdx.x] += s_tp[threadIdx.x + o]; s_fp[threadIdx.x] += s_fp[threadIdx.x + o];
            s_fn[threadIdx.x] += s_fn[threadIdx.x + o]; s_f1[threadIdx.x] += s_f1[threadIdx.x + o];
            s_n[threadIdx.x] += s_n[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double den = 2.0 * s_tp[0] + s_fp[0] + s_fn[0];
        out[0] = den > 0 ? 2.0 * s_tp[0] / den : 0.0;
        out[1] = s_n[0] > 0 ? s_f1[0] / s_n[0] : 0.0;
        out[2] = (double)counts[3 * C];
    }
}

// A TRAINING BATCH in one launch (B x C small: 512 x 41 at the headline configuration): one workgroup counts in LDS
// and finalises -- the per-batch log line of train.py:150-158 costs the step's stream one ~4 us launch instead of three
// (zero-fill, counts, finalisation: ~13 us of a 92 us step).  MODE 0: classification (int64 class ids), 1 / 2:
// multilabel with float / int64 indicator targets.  Same integer counts, same doubles as the three-launch route.
constexpr int METRIC_SMALL_C = 1024;
constexpr int METRIC_TILE = 10496;        // floats of the one-launch classification route's LDS tile (41 KiB: 256 rows x 41)
template <int MODE>
__global__ void __launch_bounds__(256)
k_metric_f1_small(const float *__restrict__ logits, int64_t ld, const void *__restrict__ yv, int64_t ldy, int32_t B,
                  int32_t C, double *__restrict__ out)
{
    __shared__ int32_t cnt[3 * METRIC_SMALL_C + 1];
    __shared__ double s_tp[256], s_fp[256], s_fn[256], s_f1[256], s_n[256];
    __shared__ float tile[MODE == 0 ? METRIC_TILE : 1];
    for (int i = threadIdx.x; i < 3 * C + 1; i += 256) cnt[i] = 0;
    __syncthreads();
    if (MODE == 0) {
        // argmax per row, a thread per row -- out of an LDS tile that the workgroup fills with COALESCED loads (a thread
        // reading its own row from HBM strides C floats: 64 lines per wave-wide load; that version took 20 us at
        // 512 x 41, most of what the per-batch log cost the CLI; competing for the row with LDS atomics took 31: forty
        // lanes of a wave hit the same word).  Rows sit Cp = C | 1 floats apart: an odd stride has no bank conflicts.
        const int Cp = C | 1;
        const int rows_per_tile = min(256, METRIC_TILE / Cp);
        const int64_t *y = (const int64_t *)yv;
        for (int row0 = 0; row0 < B; row0 += rows_per_tile) {
            const int rows = min(rows_per_tile, B - row0);
            // (EVERY load of the tile in flight at once -- 41 per thread: one workgroup cannot hide a memory round trip
            //  (~1.5 us) behind anything, so the kernel costs as many of them as it makes dependent rounds: a loop of
            //  single load -> LDS store pairs made 82 (20 us at 512 x 41), batches of eight 12)
            const int total = rows * C;
            constexpr int PER = METRIC_TILE / 256;
            // (r, c) of element threadIdx.x + 256 u without a division per element: ONE workgroup runs one wave per
            // SIMD, so every instruction of a dependent chain costs its full latency -- 164 runtime divisions per
            // thread were ~10 us of this kernel
            const int dr = 256 / C, dc = 256 - dr * C;
            int r = (int)threadIdx.x / C, c = (int)threadIdx.x - r * C;
            float v[PER];
            int off[PER];
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const bool ok = (int)threadIdx.x + u * 256 < total;
                off[u] = ok ? r * Cp + c : -1;
                v[u] = logits[ok ? (int64_t)(row0 + r) * ld + c : (int64_t)row0 * ld];
                r += dr; c += dc;
                if (c >= C) { c -= C; r += 1; }
            }
#pragma unroll
            for (int u = 0; u < PER; ++u)
                if (off[u] >= 0) tile[off[u]] = v[u];
            __syncthreads();
            if ((int)threadIdx.x < rows) {
                const float *row = tile + threadIdx.x * Cp;
                int best = 0;
                float bv = row[0];
                for (int c = 1; c < C; ++c) {   // first maximum wins, like np.argmax -- and so does the first NaN
                    const float v = row[c];
                    if (v > bv || (v != v && bv == bv)) { bv = v; best = c; }
                }
                const int i = row0 + threadIdx.x;
                const int64_t t = y[i];
                if (t < 0 || t >= C) {
                    atomicAdd(cnt + 3 * C, 1);
                } else if (t == best) {
                    atomicAdd(cnt + best, 1);
                } else {
                    atomicAdd(cnt + C + best, 1);
                    atomicAdd(cnt + 2 * C + (int)t, 1);
                }
            }
            __syncthreads();
        }
    } else {
        const int total = B * C;
        // (row, column) of element threadIdx.x + 256 k stepped, not divided (see MODE 0); eight element pairs in flight
        const int dr = 256 / C, dc = 256 - dr * C;
        int i = (int)threadIdx.x / C, c = (int)threadIdx.x - i * C;
        for (int base = threadIdx.x; base < total; base += 256 * 8) {
            bool pred[8], truth[8];
            int col[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool ok = base + u * 256 < total;
                col[u] = ok ? c : -1;
                const int64_t lo = ok ? (int64_t)i * ld + c : 0, yo = ok ? (int64_t)i * ldy + c : 0;
                pred[u] = logits[lo] > 0.f;
                truth[u] = MODE == 1 ? ((const float *)yv)[yo] != 0.f : ((const int64_t *)yv)[yo] != 0;
                i += dr; c += dc;
                if (c >= C) { c -= C; i += 1; }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (col[u] < 0) continue;
                if (pred[u] && truth[u]) atomicAdd(cnt + col[u], 1);
                else if (pred[u]) atomicAdd(cnt + C + col[u], 1);
                else if (truth[u]) atomicAdd(cnt + 2 * C + col[u], 1);
            }
        }
    }
    __syncthreads();
    const int present_only = MODE == 0;
    double tp = 0, fp = 0, fn = 0, f1 = 0, n = 0;
    for (int c = threadIdx.x; c < C; c += 256) {
        const double a = cnt[c], b = cnt[C + c], d = cnt[2 * C + c];
        tp += a; fp += b; fn += d;
        const double den = 2.0 * a + b + d;
        if (den > 0) f1 += 2.0 * a / den;
        if (!present_only || den > 0) n += 1.0;
    }
    s_tp[threadIdx.x] = tp; s_fp[threadIdx.x] = fp; s_fn[threadIdx.x] = fn; s_f1[threadIdx.x] = f1;
    s_n[threadIdx.x] = n;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            s_tp[threadIdx.x] += s_tp[threadIdx.x + o]; s_fp[threadIdx.x] += s_fp[threadIdx.x + o];
            s_fn[threadIdx.x] += s_fn[threadIdx.x + o]; s_f1[threadIdx.x] += s_f1[threadIdx.x + o];
            s_n[threadIdx.x] += s_n[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double den = 2.0 * s_tp[0] + s_fp[0] + s_fn[0];
        out[0] = den > 0 ? 2.0 * s_tp[0] / den : 0.0;
        out[1] = s_n[0] > 0 ? s_f1[0] / s_n[0] : 0.0;
        out[2] = (double)cnt[3 * C];
    }
}

// out[0] = mean |a - b| over n elements (one workgroup: the log line of a batch / a fold)
__global__ void __launch_bounds__(256)
k_metric_mae(const float *__restrict__ a, const float *__restrict__ b, int64_t n, double *__restrict__ out)
{
    __shared__ double s[256];
    double acc = 0;
    for (int64_t i = threadIdx.x; i < n; i += 256) acc += fabs((double)a[i] - (double)b[i]);
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = n > 0 ? s[0] / (double)n : 0.0;
}

__global__ void k_zero_i32(int32_t *p, int32_t n)
{
    for (int i = threadIdx.x; i < n; i += 256) p[i] = 0;
}

static inline int blocks_for(int64_t items)
{
    int64_t b = ceil_div(items, 256);
    return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}

}  // namespace gsage

using namespace gsage;

extern "C" {

int gsage_metric_f1(const float *logits, int64_t ld, const void *targets, int multilabel, int targets_f32,
                    int64_t ldy, int64_t B, int32_t C, int32_t *counts, double *out, void *stream)
{
    GSAGE_REQUIRE(logits && targets && counts && out, "metric_f1: null pointer");
    GSAGE_REQUIRE(B >= 0 && C >= 1 && ld >= C && (!multilabel || ldy >= C), "metric_f1: bad sizes");
    GSAGE_REQUIRE(multilabel || !targets_f32, "metric_f1: classification targets are int64 class ids");
    hipStream_t s = (hipStream_t)stream;
    if (B > 0 && C <= METRIC_SMALL_C && B * (int64_t)C <= 64 * 1024) {       // a training batch: one launch
        if (!multilabel)
            launch(k_metric_f1_small<0>, dim3(1), dim3(256), 0, s, logits, ld, targets, ldy, (int32_t)B, C, out);
        else if (targets_f32)
            launch(k_metric_f1_small<1>, dim3(1), dim3(256), 0, s, logits, ld, targets, ldy, (int32_t)B, C, out);
        else
            launch(k_metric_f1_small<2>, dim3(1), dim3(256), 0, s, logits, ld, targets, ldy, (int32_t)B, C, out);
        return check_launch("metric_f1");
    }
    launch(k_zero_i32, dim3(1), dim3(256), 0, s, counts, 3 * C + 1);
    int rc = check_launch("metric_zero");
    if (rc != GSAGE_OK) return rc;
    if (B > 0) {
        if (!multilabel)
            launch(k_metric_counts_cls, dim3(blocks_for(B)), dim3(256), 0, s, logits, ld, (const int64_t *)targets, B,
                   C, counts);
        else if (targets_f32)
            launch(k_metric_counts_ml<float>, dim3(blocks_for(B * C)), dim3(256), 0, s, logits, ld,
                   (const float *)targets, ldy, B, C, counts);
        else
            launch(k_metric_counts_ml<int64_t>, dim3(blocks_for(B * C)), dim3(256), 0, s, logits, ld,
                   (const int64_t *)targets, ldy, B, C, counts);
        rc = check_launch("metric_counts");
        if (rc != GSAGE_OK) return rc;
    }
    launch(k_metric_f1, dim3(1), dim3(256), 0, s, (const int32_t *)counts, C, multilabel ? 0 : 1, out);
    return check_launch("metric_f1");
}

int gsage_metric_mae(const float *y_true, const float *y_pred, int64_t n, double *out, void *stream)
{
    GSAGE_REQUIRE(y_true && y_pred && out && n >= 0, "metric_mae: bad arguments");
    launch(k_metric_mae, dim3(1), dim3(256), 0, (hipStream_t)stream, y_true, y_pred, n, out);
    return check_launch("metric_mae");
}

}  // extern "C"
