// gsage_head.hip -- the classification head of GSSupervised, forward AND backward, in two launches.
//
// Replaces (reference models.py:90-91, problem.py:34 and their autograd under models.py:100):
//     out   = F.normalize(emb, p=2, dim=1, eps=1e-12)
//     preds = fc(out)                               nn.Linear(2h, n_classes)
//     loss  = F.cross_entropy(preds, targets)       (mean over the batch)
//     loss.backward() -> d emb, d fc.weight, d fc.bias
// which stock PyTorch runs as ~35 sub-5-microsecond kernels (normalise, addmm, log_softmax, nll,
// their backwards, a 41x512x256 GEMM through hipBLASLt, reductions, fills).  The work is tiny
// (B x 2h x C = 512 x 256 x 41), so the cost is launches, not flops: here a workgroup keeps
// fc.weight in LDS and walks its rows, a second tiny kernel sums the per-workgroup partial
// weight gradients deterministically.
#include "gsage_common.h"

namespace gsage {

constexpr int HEAD_CMAX = 64;      // classes handled per lane pass
constexpr int HEAD_DMAX = 1024;    // max embedding width (2h)

struct HeadParams {
    const float *E;          // [B, lde] fp32 embedding (last SAGE layer output)
    const float *W;          // [C, D] fc.weight
    const float *bias;       // [C]
    const int64_t *targets;  // [B] class ids (or [n_batches, B] with batch_idx)
    const int64_t *batch_idx;
    int64_t n_batches;
    const int32_t *n_valid;  // optional: live rows of the batch (one word, or [n_batches] with batch_idx)
    float *preds;            // [B, C] logits
    void *dE;                // [B, ldd] gradient w.r.t. E (bf16 or fp32)
    float *partial;          // [grid, C*D + C + 1] per-workgroup dW | db | loss
    int64_t lde, ldd;
    int32_t B, C, D, rows_per_wg, dE_dtype;
};

__device__ __forceinline__ float wave_sum64(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max64(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum over 256 threads (4 waves) through a 4-float LDS scratch
__device__ __forceinline__ float block_sum(float v, float *red)
{
    v = wave_sum64(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// R rows are processed TOGETHER by the workgroup: every phase (normalise, logits, softmax,
// d z, dW accumulation) handles all R rows between two barriers, so a workgroup pays ~6 barriers
// for R rows instead of ~9 per row, fc.weight is read from LDS once per R rows, and the R
// independent dot products give the LDS latency some ILP.  Thread t owns columns t + 256 j.
template <int KPT, int R>
__global__ void __launch_bounds__(256)
k_head_ce(const HeadParams p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int C = p.C, D = p.D;
    const int ldw = D + 1;                        // +1 float: conflict-free column walks
    float *Ws = lds;                              // [C][ldw]
    float *zs = Ws + C * ldw;                     // [R][D] normalised rows
    float *part = zs + R * D;                     // [4 waves][R][HEAD_CMAX] partial logits
    float *dls = part + 4 * R * HEAD_CMAX;        // [R][HEAD_CMAX] d logits
    float *red = dls + R * HEAD_CMAX;             // [4][R]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // every global input is requested up front (one memory round trip on the critical path, not
    // one per phase): this wave's row target, the bias, the R embedding rows, then fc.weight
    const int row0 = blockIdx.x * R;
    const int64_t bq = p.batch_idx ? (int64_t)((uint64_t)*p.batch_idx % (uint64_t)p.n_batches) : 0;
    const int64_t *tgt = p.targets + bq * p.B;
    // rows past Bv are padding (the reference's chunks are not all of one size): no loss, no gradient
    const int Bv = p.n_valid ? min(max(p.n_valid[bq], 1), p.B) : p.B;
    int64_t my_target[(R + 3) / 4];
#pragma unroll
    for (int q = 0; q < (R + 3) / 4; ++q) {
        const int i = row0 + wave + 4 * q;
        my_target[q] = (wave + 4 * q < R && i < p.B) ? tgt[i] : -1;
    }
    const float my_bias = (lane < C) ? p.bias[lane] : 0.f;
    float z[R][KPT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const int k = tid + 256 * j;
            z[r][j] = (k < D && row0 + r < p.B) ? p.E[(int64_t)(row0 + r) * p.lde + k] : 0.f;
        }
    // fc.weight -> LDS, 8 independent loads in flight per thread before the stores
    for (int k = tid; k < D; k += 256) {
        int c = 0;
        for (; c + 8 <= C; c += 8) {
            float w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = p.W[(int64_t)(c + u) * D + k];
#pragma unroll
            for (int u = 0; u < 8; ++u) Ws[(c + u) * ldw + k] = w[u];
        }
        for (; c < C; ++c) Ws[c * ldw + k] = p.W[(int64_t)c * D + k];
    }

    auto block_sum_rows = [&](float (&v)[R]) {     // sums each v[r] over the 256 threads
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = wave_sum64(v[r]);
        __syncthreads();
        if (lane == 0)
#pragma unroll
            for (int r = 0; r < R; ++r) red[wave * R + r] = v[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = (red[r] + red[R + r]) + (red[2 * R + r] + red[3 * R + r]);
    };

    const float invB = 1.f / (float)Bv;
    // 1. L2 normalise (F.normalize: x / max(||x||, 1e-12)); rows past B behave as zero rows
    float ss[R], nrm[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        ss[r] = 0.f;
#pragma unroll
        for (int j = 0; j < KPT; ++j) ss[r] += z[r][j] * z[r][j];
    }
    block_sum_rows(ss);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        nrm[r] = fmaxf(sqrtf(ss[r]), 1e-12f);
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const int k = tid + 256 * j;
            z[r][j] = z[r][j] / nrm[r];
            if (k < D) zs[r * D + k] = z[r][j];
        }
    }
    __syncthreads();
    // 2. logits: lane c of wave w sums its quarter of the columns, for all R rows at once
    {
        float s[R];
#pragma unroll
        for (int r = 0; r < R; ++r) s[r] = 0.f;
        if (lane < C) {
            const int k0 = wave * ((D + 3) / 4), k1 = min(D, k0 + (D + 3) / 4);
            const float *wr = Ws + lane * ldw;
            for (int k = k0; k < k1; ++k) {
                const float w = wr[k];
#pragma unroll
                for (int r = 0; r < R; ++r) s[r] += zs[r * D + k] * w;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) part[(wave * R + r) * HEAD_CMAX + lane] = s[r];
    }
    __syncthreads();
    // 3. softmax / loss / d logits: wave r handles row r (extra rows loop when R > 4)
    float acc_db = 0.f, acc_loss = 0.f;
    for (int r = wave; r < R; r += 4) {
        const int i = row0 + r;
        const bool ok = lane < C && i < Bv;
        float logit = -INFINITY;
        if (lane < C)
            logit = part[(0 * R + r) * HEAD_CMAX + lane] + part[(1 * R + r) * HEAD_CMAX + lane] +
                    part[(2 * R + r) * HEAD_CMAX + lane] + part[(3 * R + r) * HEAD_CMAX + lane] + my_bias;
        const float mx = wave_max64(logit);
        const float ex = (lane < C) ? expf(logit - mx) : 0.f;
        const float den = wave_sum64(ex);
        const int64_t t = my_target[(r - wave) / 4];
        const float dl = ok ? (ex / den - ((int64_t)lane == t ? 1.f : 0.f)) * invB : 0.f;
        dls[r * HEAD_CMAX + lane] = dl;
        if (lane < C && i < p.B) p.preds[(int64_t)i * C + lane] = logit;
        if (i < Bv && (int64_t)lane == t) acc_loss += -(logit - mx - logf(den));
        acc_db += dl;                                   // lane c accumulates db[c] over its rows
    }
    __syncthreads();
    // 4. d z, d emb, dW accumulation (thread <-> columns)
    float dz[R][KPT], zdz[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        zdz[r] = 0.f;
#pragma unroll
        for (int j = 0; j < KPT; ++j) dz[r][j] = 0.f;
    }
    float accW[HEAD_CMAX][KPT];
#pragma unroll
    for (int c = 0; c < HEAD_CMAX; ++c) {
        if (c < C) {                                    // block-uniform
            float dl[R];
#pragma unroll
            for (int r = 0; r < R; ++r) dl[r] = dls[r * HEAD_CMAX + c];
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                const int k = tid + 256 * j;
                const float w = (k < D) ? Ws[c * ldw + k] : 0.f;
                float a = 0.f;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    dz[r][j] += dl[r] * w;
                    a += dl[r] * z[r][j];
                }
                accW[c][j] = a;
            }
        } else {
#pragma unroll
            for (int j = 0; j < KPT; ++j) accW[c][j] = 0.f;
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < KPT; ++j) zdz[r] += z[r][j] * dz[r][j];
    block_sum_rows(zdz);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = row0 + r;
        if (i >= p.B) continue;
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const int k = tid + 256 * j;
            if (k < D) {
                const float g = (dz[r][j] - z[r][j] * zdz[r]) / nrm[r];
                if (p.dE_dtype == GSAGE_BF16)
                    ((uint16_t *)p.dE)[(int64_t)i * p.ldd + k] = f32_to_bf16(g);
                else
                    ((float *)p.dE)[(int64_t)i * p.ldd + k] = g;
            }
        }
    }

    float *out = p.partial + (int64_t)blockIdx.x * ((int64_t)C * D + C + 1);
#pragma unroll
    for (int c = 0; c < HEAD_CMAX; ++c)
        if (c < C)
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                const int k = tid + 256 * j;
                if (k < D) out[c * D + k] = accW[c][j];
            }
    // db[c]: lane c of every wave holds its rows' share; loss likewise
    __syncthreads();
    part[wave * HEAD_CMAX + lane] = acc_db;
    const float l = wave_sum64(acc_loss);
    if (lane == 0) red[wave] = l;
    __syncthreads();
    if (wave == 0) {
        if (lane < C)
            out[C * D + lane] = (part[lane] + part[HEAD_CMAX + lane]) + (part[2 * HEAD_CMAX + lane] + part[3 * HEAD_CMAX + lane]);
        if (lane == 0) out[C * D + C] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

// dW | db | loss = sum over workgroups of the partials (deterministic order).  64 outputs per
// block; the 4 waves each sum a quarter of the workgroup range with 8 loads in flight.
__global__ void __launch_bounds__(256)
k_head_reduce(const float *__restrict__ partial, int32_t n_wg, int64_t width, int64_t cd, int32_t C,
              float *__restrict__ dW, float *__restrict__ db, float *__restrict__ loss, float inv_b)
{
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t t = (int64_t)blockIdx.x * 64 + lane;
    const int per = (n_wg + 3) / 4;
    const int g0 = q * per, g1 = min(n_wg, g0 + per);
    float s = 0.f;
    if (t < width) {
        int g = g0;
        for (; g + 8 <= g1; g += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(g + u) * width + t];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; g < g1; ++g) s += partial[(int64_t)g * width + t];
    }
    red[q][lane] = s;
    __syncthreads();
    if (q == 0 && t < width) {
        s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        if (t < cd) dW[t] = s;
        else if (t < cd + C) db[t - cd] = s;
        else if (loss) *loss = s * inv_b;
    }
}


// ---- L1 head of the reference's regression problems (Pokec: problem.py:39-42 behind models.py:100) -------------------
//     z = E / max(||E||_2, 1e-12);  preds[i] = <z_i, w> + b;  loss = F.l1_loss(preds [B,1], targets [B])
// The reference calls the loss with targets.squeeze(): [B,1] against [B] BROADCASTS to [B,B], i.e.
//     loss = mean_{i,j} |p_i - t_j|,   d loss / d p_i = (1/B^2) sum_j sign(p_i - t_j)        (sign(0) = 0, as torch's)
// kept as is (it is what the recorded Pokec result was trained with).  Two launches (every d p_i needs every
// prediction): (a) a wave per row: norm, prediction; (b) 16 rows per workgroup: d p, d E, and one partial row
// [d W | d b | loss] per workgroup for gsage_finalize_grads.  As stock torch ops the head was ~15 launches per step.
constexpr int L1_BMAX = 2048;
constexpr int L1_TMAX = 8192;      // targets of a GLOBAL batch (gsage_head_l1_sharded: every rank's)
constexpr int L1_ROWS = 16;
__device__ __forceinline__ void store_out(uint16_t *p, float v) { *p = f32_to_bf16(v); }
__device__ __forceinline__ void store_out(float *p, float v) { *p = v; }

__global__ void __launch_bounds__(256)
k_head_l1_pred(const float *__restrict__ E, int64_t lde, const float *__restrict__ W, const float *__restrict__ bias,
               int32_t B, int32_t D, float *__restrict__ preds, float *__restrict__ inv)
{
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= B) return;
    float ss = 0.f, dot = 0.f;
    for (int c = lane; c < D; c += 64) {
        const float e = E[(int64_t)i * lde + c];
        ss += e * e;
        dot += e * W[c];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ss += __shfl_xor(ss, o, 64); dot += __shfl_xor(dot, o, 64); }
    if (lane == 0) {
        const float nrm = fmaxf(sqrtf(ss), 1e-12f);          // F.normalize's eps
        inv[i] = 1.f / nrm;
        preds[i] = dot / nrm + bias[0];
    }
}

template <typename TD>
__global__ void __launch_bounds__(256)
k_head_l1_bwd(const float *__restrict__ E, int64_t lde, const float *__restrict__ W, const float *__restrict__ targets,
              const float *__restrict__ preds, const float *__restrict__ inv, int32_t Ball, int32_t D,
              TD *__restrict__ dE, int64_t ldd, float *__restrict__ partial, const int32_t *__restrict__ n_valid,
              int32_t T_all)
{
    __shared__ float ts[L1_TMAX], dps[L1_ROWS], ivs[L1_ROWS], lss[L1_ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * L1_ROWS;
    // rows past B are padding (n_valid): they enter neither the B x B pairs nor the gradient
    const int32_t B = n_valid ? min(max(n_valid[0], 1), Ball) : Ball;
    // T_all > 0: one shard of a data-parallel batch -- the pairs are (this rank's rows) x (the GLOBAL batch's targets)
    const int32_t T = T_all > 0 ? T_all : B;
    for (int j = tid; j < T; j += 256) ts[j] = targets[j];
    __syncthreads();
    const float scale = 1.f / ((float)B * (float)T);
    // d p of the workgroup's rows: a wave per row, lanes over the targets
    for (int r = wave; r < L1_ROWS; r += 4) {
        const int i = r0 + r;
        float cnt = 0.f, l = 0.f;
        if (i < B) {
            const float pv = preds[i];
            for (int j = lane; j < T; j += 64) {
                const float d = pv - ts[j];
                cnt += (float)((d > 0.f) - (d < 0.f));
                l += fabsf(d);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { cnt += __shfl_xor(cnt, o, 64); l += __shfl_xor(l, o, 64); }
        if (lane == 0) { dps[r] = scale * cnt; lss[r] = scale * l; ivs[r] = i < B ? inv[i] : 0.f; }
    }
    __syncthreads();
    // d E_i = (dz_i - z_i <z_i, dz_i>) / ||E_i||, dz_i = dp_i w
    for (int r = wave; r < L1_ROWS; r += 4) {
        const int i = r0 + r;
        if (i >= B) {
            if (i < Ball)                    // padding rows: an explicit zero gradient
                for (int c = lane; c < D; c += 64) store_out(dE + (int64_t)i * ldd + c, 0.f);
            continue;
        }
        const float iv = ivs[r], dp = dps[r];
        float zdz = 0.f;
        for (int c = lane; c < D; c += 64) zdz += (E[(int64_t)i * lde + c] * iv) * (dp * W[c]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) zdz += __shfl_xor(zdz, o, 64);
        for (int c = lane; c < D; c += 64) {
            const float z = E[(int64_t)i * lde + c] * iv;
            store_out(dE + (int64_t)i * ldd + c, (dp * W[c] - z * zdz) * iv);
        }
    }
    // the workgroup's partial row: dW[c] = sum_i dp_i z_i[c] | db = sum_i dp_i | loss
    float *row = partial + (int64_t)blockIdx.x * (D + 2);
    for (int c = tid; c < D; c += 256) {
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < L1_ROWS; ++r)
            if (r0 + r < B) acc += dps[r] * (E[(int64_t)(r0 + r) * lde + c] * ivs[r]);
        row[c] = acc;
    }
    if (tid == 0) {
        float dbv = 0.f, lv = 0.f;
        for (int r = 0; r < L1_ROWS; ++r) { dbv += dps[r]; lv += lss[r]; }      // (rows past B hold zeros)
        row[D] = dbv;
        row[D + 1] = lv;
    }
}

}  // namespace gsage

using namespace gsage;

extern "C" {

int64_t gsage_head_ce_scratch(int32_t B, int32_t C, int32_t D)
{
    const int rows_per_wg = D <= 256 ? 4 : 1;
    const int64_t n_wg = (B + rows_per_wg - 1) / rows_per_wg;
    return n_wg * ((int64_t)C * D + C + 1);
}

int gsage_head_ce(const float *E, int64_t lde, const float *W, const float *bias,
                  const int64_t *targets, int32_t B, int32_t C, int32_t D, float *preds, void *dE,
                  int dE_dtype, int64_t ldd, float *dW, float *db, float *loss, float *scratch,
                  const int64_t *batch_idx, int64_t n_batches, void *stream)
{
    const int32_t *n_valid = take_head_n_valid();     // (consumed before any return path: never left for a later launch)
    GSAGE_REQUIRE(!batch_idx || n_batches > 0, "head_ce: bad target queue");
    GSAGE_REQUIRE(E && W && bias && targets && preds && dE && scratch, "head_ce: null pointer");
    GSAGE_REQUIRE(B > 0 && C > 0 && C <= HEAD_CMAX && D > 0 && D <= HEAD_DMAX,
                  "head_ce: needs 1 <= n_classes <= %d and 1 <= width <= %d", HEAD_CMAX, HEAD_DMAX);
    GSAGE_REQUIRE(dE_dtype == GSAGE_BF16 || dE_dtype == GSAGE_F32, "head_ce: bad dE dtype");
    HeadParams p;
    p.E = E; p.W = W; p.bias = bias; p.targets = targets; p.batch_idx = batch_idx; p.n_batches = n_batches; p.preds = preds; p.dE = dE;
    p.n_valid = n_valid;
    p.partial = scratch; p.lde = lde; p.ldd = ldd; p.B = B; p.C = C; p.D = D; p.rows_per_wg = D <= 256 ? 4 : 1;
    p.dE_dtype = dE_dtype;
    const int n_wg = (B + p.rows_per_wg - 1) / p.rows_per_wg;
    const int R = p.rows_per_wg;
    const size_t lds = sizeof(float) * ((size_t)C * (D + 1) + (size_t)R * D + 4 * R * HEAD_CMAX + R * HEAD_CMAX + 4 * R + 16);
    GSAGE_REQUIRE(lds <= 160 * 1024, "head_ce: fc.weight does not fit in LDS");
    if (D <= 256)
        launch(k_head_ce<1, 4>, dim3(n_wg), dim3(256), lds, (hipStream_t)stream, p);
    else if (D <= 512)
        launch(k_head_ce<2, 1>, dim3(n_wg), dim3(256), lds, (hipStream_t)stream, p);
    else
        launch(k_head_ce<4, 1>, dim3(n_wg), dim3(256), lds, (hipStream_t)stream, p);
    int rc = check_launch("head_ce");
    if (rc != GSAGE_OK || dW == nullptr || db == nullptr) return rc;   // caller reduces the partials
    const int64_t width = (int64_t)C * D + C + 1;
    launch(k_head_reduce, dim3((unsigned)ceil_div(width, 64)), dim3(256), 0,
                       (hipStream_t)stream, (const float *)scratch, n_wg, width, (int64_t)C * D, C, dW,
                       db, loss, 1.f / (float)B);
    return check_launch("head_reduce");
}

int gsage_head_l1_scratch(int64_t B, int64_t D)
{
    return (int)(ceil_div(B, (int64_t)L1_ROWS) * (D + 2) + B);
}

static int head_l1_launch(const float *E, int64_t lde, const float *W, const float *bias, const float *targets, int64_t T,
                          int64_t B, int64_t D, float *preds, void *dE, int dE_dtype, int64_t ldd, float *scratch,
                          void *stream)
{
    const int32_t *nv = take_head_n_valid();          // (consumed before any return path: never left for a later launch)
    GSAGE_REQUIRE(E && W && bias && targets && preds && dE && scratch, "head_l1: null pointer");
    GSAGE_REQUIRE(B > 0 && B <= L1_BMAX && D > 0 && lde >= D && ldd >= D, "head_l1: needs 1 <= B <= %d", L1_BMAX);
    GSAGE_REQUIRE(T >= 0 && T <= L1_TMAX, "head_l1_sharded: needs <= %d targets", L1_TMAX);
    GSAGE_REQUIRE(dE_dtype == GSAGE_BF16 || dE_dtype == GSAGE_F32, "head_l1: bad dE dtype");
    const int n_wg = (int)ceil_div(B, (int64_t)L1_ROWS);
    float *inv = scratch + (int64_t)n_wg * (D + 2);
    hipStream_t s = (hipStream_t)stream;
    launch(k_head_l1_pred, dim3((unsigned)ceil_div(B, (int64_t)4)), dim3(256), 0, s, E, lde, W, bias, (int32_t)B, (int32_t)D,
           preds, inv);
    int rc = check_launch("head_l1_pred");
    if (rc != GSAGE_OK) return rc;
    if (dE_dtype == GSAGE_BF16)
        launch(k_head_l1_bwd<uint16_t>, dim3(n_wg), dim3(256), 0, s, E, lde, W, targets, (const float *)preds,
               (const float *)inv, (int32_t)B, (int32_t)D, (uint16_t *)dE, ldd, scratch, nv, (int32_t)T);
    else
        launch(k_head_l1_bwd<float>, dim3(n_wg), dim3(256), 0, s, E, lde, W, targets, (const float *)preds,
               (const float *)inv, (int32_t)B, (int32_t)D, (float *)dE, ldd, scratch, nv, (int32_t)T);
    return check_launch("head_l1_bwd");
}

int gsage_head_l1(const float *E, int64_t lde, const float *W, const float *bias, const float *targets, int64_t B,
                  int64_t D, float *preds, void *dE, int dE_dtype, int64_t ldd, float *scratch, void *stream)
{
    return head_l1_launch(E, lde, W, bias, targets, 0, B, D, preds, dE, dE_dtype, ldd, scratch, stream);
}

int gsage_head_l1_sharded(const float *E, int64_t lde, const float *W, const float *bias, const float *targets,
                          int64_t T, int64_t B, int64_t D, float *preds, void *dE, int dE_dtype, int64_t ldd,
                          float *scratch, void *stream)
{
    GSAGE_REQUIRE(T > 0, "head_l1_sharded: needs the global batch's targets");
    return head_l1_launch(E, lde, W, bias, targets, T, B, D, preds, dE, dE_dtype, ldd, scratch, stream);
}

}  // extern "C"
