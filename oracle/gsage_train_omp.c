/*
 * oracle/gsage_train_omp.c -- OpenMP CPU restatement of one GSSupervised.train_step for the
 * north-star configuration (sparse sampler, identity prep, mean aggregators, classification),
 * generic in depth.  TEST INFRASTRUCTURE / CPU BASELINE, not product: only tests/ and bench.py's
 * cpu_baseline leg load it ("the build's own C++ CPU restatement of the
 * same train_step path (OpenMP, all cores)").
 *
 * Follows the reference line by line in meaning, not in shape (it never materialises the sampled
 * neighbour rows of the last hop, the reference's models.py:80 does):
 *   frontier            models.py:73-81     ids -> sampler per hop (nn_modules.py:80-101: out[i*n+j] =
 *                                           data[indptr[id_i] + sel[i,j] % deg_i], 0 for an empty row)
 *   layer stacking      models.py:85-86     level l turns hops 0..L-l into hops 0..L-l-1
 *   MeanAggregator      nn_modules.py:196-204   act(cat[x Wx^T, mean_j(neib_j) Wn^T]), no bias
 *   head                models.py:90-91     F.normalize(p=2, dim=1, eps=1e-12) -> fc
 *   loss                problem.py:34       F.cross_entropy, mean over the batch
 *   train_step          models.py:97-104    backward, clip_grad_norm(5) (coef = 5/(norm+1e-6) when < 1),
 *                                           Adam(betas .9/.999, eps 1e-8, L2 weight decay added to grad)
 * Parity status: PINNED -- tests/test_oracle_omp.py checks predictions, loss, gradient norm, clipped
 * gradients and post-step weights against tests/golden/{model,engine}_kat.npz (outputs of the
 * reference itself) for the 2- and 3-layer mean cases.
 *
 * Parameter vector layout (= the reference's state_dict order): for each layer l: fc_x.weight
 * [h_l, din_l], fc_neib.weight [h_l, din_l] (din_0 = D, din_l = 2 h_{l-1}); then fc.weight
 * [C, 2 h_{L-1}], fc.bias [C].
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

#define MAXL 5

static int64_t pick(const int64_t *indptr, const int64_t *data, int64_t id, int64_t s)
{
    const int64_t beg = indptr[id], deg = indptr[id + 1] - beg;
    return deg > 0 ? data[beg + s % deg] : 0;      /* numpy: x % 0 == 0 -> the dummy node */
}

/* C[m, n] = sum_k A[m, k] * W[n, k]      (A: [M, lda], W: [N, ldw], C: [M, ldc], columns c0.. of C) */
static void gemm_nt(int64_t M, int N, int K, const float *A, int64_t lda, const float *W, int64_t ldw,
                    float *C, int64_t ldc, int c0)
{
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
        const float *a = A + m * lda;
        for (int n = 0; n < N; ++n) {
            const float *w = W + (int64_t)n * ldw;
            float s = 0.f;
#pragma omp simd reduction(+ : s)
            for (int k = 0; k < K; ++k) s += a[k] * w[k];
            C[m * ldc + c0 + n] = s;
        }
    }
}

/* out[m, k] = sum_n G[m, g0 + n] * W[n, k]   (input gradient of a projection) */
static void gemm_nn(int64_t M, int N, int K, const float *G, int64_t ldg, int g0, const float *W, int64_t ldw,
                    float *out, int64_t ldo)
{
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
        float *o = out + m * ldo;
        for (int k = 0; k < K; ++k) o[k] = 0.f;
        for (int n = 0; n < N; ++n) {
            const float g = G[m * ldg + g0 + n];
            if (g == 0.f) continue;
            const float *w = W + (int64_t)n * ldw;
#pragma omp simd
            for (int k = 0; k < K; ++k) o[k] += g * w[k];
        }
    }
}

/* dW[n, k] += sum_m G[m, g0 + n] * A[m, k]: M split over threads, partial sums reduced in chunk order */
static void wgrad_tn(int64_t M, int N, int K, const float *G, int64_t ldg, int g0, const float *A, int64_t lda,
                     float *dW)
{
    int T = omp_get_max_threads();
    if (T > M / 32) T = (int)(M / 32);
    if (T < 1) T = 1;
    float *part = (float *)calloc((size_t)T * N * K, sizeof(float));
#pragma omp parallel for schedule(static) num_threads(T)
    for (int t = 0; t < T; ++t) {
        float *p = part + (size_t)t * N * K;
        const int64_t m0 = M * t / T, m1 = M * (t + 1) / T;
        for (int64_t m = m0; m < m1; ++m) {
            const float *a = A + m * lda;
            for (int n = 0; n < N; ++n) {
                const float g = G[m * ldg + g0 + n];
                if (g == 0.f) continue;
                float *row = p + (size_t)n * K;
#pragma omp simd
                for (int k = 0; k < K; ++k) row[k] += g * a[k];
            }
        }
    }
    const int64_t NK = (int64_t)N * K;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < NK; ++i) {
        float s = 0.f;
        for (int t = 0; t < T; ++t) s += part[(size_t)t * NK + i];
        dW[i] += s;
    }
    free(part);
}

/* out[i, :] = mean_j src[(i*n + j), :]  (ids == NULL) or mean_j table[ids[i*n+j], :] */
static void seg_mean(const float *src, int64_t ld, const int64_t *ids, int64_t M, int n, int D, float *out)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < M; ++i) {
        float *o = out + i * D;
        for (int c = 0; c < D; ++c) o[c] = 0.f;
        for (int j = 0; j < n; ++j) {
            const int64_t r = ids ? ids[i * n + j] : i * n + j;
            const float *s = src + r * ld;
#pragma omp simd
            for (int c = 0; c < D; ++c) o[c] += s[c];
        }
        const float fn = (float)n;
        for (int c = 0; c < D; ++c) o[c] /= fn;
    }
}

int gso_train_step_mean(const int64_t *indptr, const int64_t *data, int64_t n_rows, const float *feats,
                        int64_t ld, int32_t D, int32_t L, const int32_t *fan, const int32_t *h, int32_t C,
                        float *params, float *grads, float *adam_m, float *adam_v, int64_t n_params,
                        int64_t adam_t, float lr, float wd, const int64_t *ids, const int64_t *targets,
                        int32_t B, const int64_t *const *sel, float *preds, float *loss_out,
                        float *gradnorm_out)
{
    if (L < 1 || L > MAXL - 1 || B < 1) return -1;
    int64_t size[MAXL + 1];
    int64_t *idk[MAXL + 1];
    size[0] = B;
    idk[0] = (int64_t *)ids;
    for (int k = 1; k <= L; ++k) {
        size[k] = size[k - 1] * fan[k - 1];
        idk[k] = (int64_t *)malloc(sizeof(int64_t) * size[k]);
    }
    for (int64_t i = 0; i < B; ++i)
        if (ids[i] < 0 || ids[i] >= n_rows) return -2;
    for (int k = 1; k <= L; ++k) {
        const int n = fan[k - 1];
        const int64_t *par = idk[k - 1], *s = sel[k - 1];
        int64_t *out = idk[k];
#pragma omp parallel for schedule(static)
        for (int64_t t = 0; t < size[k]; ++t) out[t] = pick(indptr, data, par[t / n], s[t]);
    }

    /* parameter offsets */
    int din[MAXL];
    int64_t oWx[MAXL], oWn[MAXL], oWfc, obfc, off = 0;
    for (int l = 0; l < L; ++l) {
        din[l] = l == 0 ? D : 2 * h[l - 1];
        oWx[l] = off; off += (int64_t)h[l] * din[l];
        oWn[l] = off; off += (int64_t)h[l] * din[l];
    }
    const int E = 2 * h[L - 1];
    oWfc = off; off += (int64_t)C * E;
    obfc = off; off += C;
    if (off != n_params) return -3;

    /* X[l][k]: input rows of level l at hop k (k <= L-l-1 as "x", k+1 as neighbours);  A[l][k]: means.
       Level 0 never materialises the neighbours of its last hop: A[0][k] is gathered straight from the table. */
    float *X[MAXL + 1][MAXL + 1] = {{0}}, *A[MAXL][MAXL] = {{0}};
    for (int k = 0; k < L; ++k) {
        X[0][k] = (float *)malloc(sizeof(float) * size[k] * D);
        A[0][k] = (float *)malloc(sizeof(float) * size[k] * D);
        seg_mean(feats, ld, idk[k], size[k], 1, D, X[0][k]);
        seg_mean(feats, ld, idk[k + 1], size[k], fan[k], D, A[0][k]);
    }
    for (int l = 0; l < L; ++l) {
        const int hl = h[l], dw = din[l];
        for (int k = 0; k < L - l; ++k) {
            if (l > 0) {
                A[l][k] = (float *)malloc(sizeof(float) * size[k] * dw);
                seg_mean(X[l][k + 1], dw, NULL, size[k], fan[k], dw, A[l][k]);
            }
            float *O = (float *)malloc(sizeof(float) * size[k] * 2 * hl);
            gemm_nt(size[k], hl, dw, X[l][k], dw, params + oWx[l], dw, O, 2 * hl, 0);
            gemm_nt(size[k], hl, dw, A[l][k], dw, params + oWn[l], dw, O, 2 * hl, hl);
            if (l < L - 1) {
                const int64_t tot = size[k] * 2 * hl;
#pragma omp parallel for schedule(static)
                for (int64_t i = 0; i < tot; ++i) O[i] = O[i] > 0.f ? O[i] : 0.f;
            }
            X[l + 1][k] = O;
        }
    }

    /* head + loss + d emb */
    const float *emb = X[L][0];
    const float *Wfc = params + oWfc, *bfc = params + obfc;
    float *dE = (float *)calloc((size_t)B * E, sizeof(float));
    float *z = (float *)malloc(sizeof(float) * B * E);
    float *dl = (float *)malloc(sizeof(float) * B * C);
    memset(grads, 0, sizeof(float) * n_params);
    double loss = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : loss)
    for (int64_t i = 0; i < B; ++i) {
        const float *e = emb + i * E;
        double ss = 0.0;
        for (int c = 0; c < E; ++c) ss += (double)e[c] * e[c];
        float nrm = (float)sqrt(ss);
        if (nrm < 1e-12f) nrm = 1e-12f;
        float *zi = z + i * E;
        for (int c = 0; c < E; ++c) zi[c] = e[c] / nrm;
        float *lg = preds + i * C;
        float mx = -INFINITY;
        for (int c = 0; c < C; ++c) {
            const float *w = Wfc + (int64_t)c * E;
            float s = 0.f;
#pragma omp simd reduction(+ : s)
            for (int q = 0; q < E; ++q) s += zi[q] * w[q];
            lg[c] = s + bfc[c];
            if (lg[c] > mx) mx = lg[c];
        }
        double den = 0.0;
        for (int c = 0; c < C; ++c) den += exp((double)lg[c] - mx);
        const int64_t t = targets[i];
        loss += -((double)lg[t] - mx - log(den));
        float *d = dl + i * C;
        for (int c = 0; c < C; ++c) d[c] = (float)((exp((double)lg[c] - mx) / den - (c == t ? 1.0 : 0.0)) / B);
        /* dz = dl Wfc;  de = (dz - z <z, dz>) / nrm */
        float *de = dE + i * E;
        for (int c = 0; c < C; ++c) {
            const float *w = Wfc + (int64_t)c * E;
            const float g = d[c];
            for (int q = 0; q < E; ++q) de[q] += g * w[q];
        }
        double dot = 0.0;
        for (int q = 0; q < E; ++q) dot += (double)zi[q] * de[q];
        for (int q = 0; q < E; ++q) de[q] = (de[q] - zi[q] * (float)dot) / nrm;
    }
    loss /= B;
    wgrad_tn(B, C, E, dl, C, 0, z, E, grads + oWfc);
    for (int c = 0; c < C; ++c) {
        double s = 0.0;
        for (int64_t i = 0; i < B; ++i) s += dl[i * C + c];
        grads[obfc + c] = (float)s;
    }

    /* backward through the levels: dX[l][k] = gradient w.r.t. X[l][k] (post-activation of level l-1) */
    float *dX[MAXL + 1][MAXL + 1] = {{0}};
    dX[L][0] = dE;
    for (int l = L - 1; l >= 0; --l) {
        const int hl = h[l], dlv = din[l];
        for (int k = 0; k < L - l; ++k) {
            float *G = dX[l + 1][k];                  /* [size[k], 2 hl] */
            if (l < L - 1) {                          /* ReLU of this level's output */
                const float *O = X[l + 1][k];
                const int64_t tot = size[k] * 2 * hl;
#pragma omp parallel for schedule(static)
                for (int64_t i = 0; i < tot; ++i) G[i] = O[i] > 0.f ? G[i] : 0.f;
            }
            wgrad_tn(size[k], hl, dlv, G, 2 * hl, 0, X[l][k], dlv, grads + oWx[l]);
            wgrad_tn(size[k], hl, dlv, G, 2 * hl, hl, A[l][k], dlv, grads + oWn[l]);
            if (l > 0) {
                float *gx = (float *)malloc(sizeof(float) * size[k] * dlv);
                float *ga = (float *)malloc(sizeof(float) * size[k] * dlv);
                gemm_nn(size[k], hl, dlv, G, 2 * hl, 0, params + oWx[l], dlv, gx, dlv);
                gemm_nn(size[k], hl, dlv, G, 2 * hl, hl, params + oWn[l], dlv, ga, dlv);
                if (!dX[l][k]) dX[l][k] = (float *)calloc((size_t)size[k] * dlv, sizeof(float));
                if (!dX[l][k + 1]) dX[l][k + 1] = (float *)calloc((size_t)size[k + 1] * dlv, sizeof(float));
                float *dx = dX[l][k], *dn = dX[l][k + 1];
                const int n = fan[k];
                const float inv = 1.f / (float)n;
#pragma omp parallel for schedule(static)
                for (int64_t i = 0; i < size[k]; ++i) {
                    for (int c = 0; c < dlv; ++c) dx[i * dlv + c] += gx[i * dlv + c];
                    for (int j = 0; j < n; ++j)
                        for (int c = 0; c < dlv; ++c) dn[(i * n + j) * dlv + c] += ga[i * dlv + c] * inv;
                }
                free(gx);
                free(ga);
            }
        }
    }

    /* clip_grad_norm(5) + Adam */
    double sq = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : sq)
    for (int64_t i = 0; i < n_params; ++i) sq += (double)grads[i] * grads[i];
    const float total = (float)sqrt(sq);
    const float coef = 5.0f / (total + 1e-6f);
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    const double bc1 = 1.0 - pow((double)b1, (double)adam_t), bc2 = 1.0 - pow((double)b2, (double)adam_t);
    const float step_size = (float)(lr / bc1), rs = (float)(1.0 / sqrt(bc2));
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n_params; ++i) {
        float g = grads[i];
        if (coef < 1.f) { g *= coef; grads[i] = g; }
        if (wd != 0.f) g += wd * params[i];
        const float m = b1 * adam_m[i] + (1.f - b1) * g;
        const float v = b2 * adam_v[i] + (1.f - b2) * g * g;
        adam_m[i] = m;
        adam_v[i] = v;
        params[i] -= step_size * (m / (sqrtf(v) * rs + eps));
    }
    if (loss_out) *loss_out = (float)loss;
    if (gradnorm_out) *gradnorm_out = total;

    for (int k = 1; k <= L; ++k) free(idk[k]);
    for (int l = 0; l <= L; ++l)
        for (int k = 0; k <= L; ++k) {
            free(X[l][k]);
            if (l < L || k > 0) free(dX[l][k]);
        }
    for (int l = 0; l < L; ++l)
        for (int k = 0; k < L; ++k) free(A[l][k]);
    free(dE);
    free(z);
    free(dl);
    return 0;
}

int gso_omp_threads(void) { return omp_get_max_threads(); }
void gso_omp_set_threads(int n) { omp_set_num_threads(n); }
