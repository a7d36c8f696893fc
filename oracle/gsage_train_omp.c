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
#include <immintrin.h>
#include <omp.h>

#define MAXL 5

/* wall time per phase since load: 0 gathers / means, 1 forward products, 3 weight gradients, 4 input gradients */
static double g_prof[8];
#define PROF(i, call) do { const double t_ = omp_get_wtime(); call; g_prof[i] += omp_get_wtime() - t_; } while (0)

static int64_t pick(const int64_t *indptr, const int64_t *data, int64_t id, int64_t s)
{
    const int64_t beg = indptr[id], deg = indptr[id + 1] - beg;
    return deg > 0 ? data[beg + s % deg] : 0;      /* numpy: x % 0 == 0 -> the dummy node */
}

static inline float hsum8(__m256 v)
{
    __m128 lo = _mm256_castps256_ps128(v), hi = _mm256_extractf128_ps(v, 1);
    lo = _mm_add_ps(lo, hi);
    lo = _mm_hadd_ps(lo, lo);
    lo = _mm_hadd_ps(lo, lo);
    return _mm_cvtss_f32(lo);
}

/* C[m, n] = sum_k A[m, k] * W[n, k]      (A: [M, lda], W: [N, ldw], C: [M, ldc], columns c0.. of C)
 * Register block: two rows of A against four rows of W, eight accumulators along k (AVX2 + FMA), the rows of A stay
 * in L1 while the block walks W (L2). */
static void gemm_nt(int64_t M, int N, int K, const float *A, int64_t lda, const float *W, int64_t ldw,
                    float *C, int64_t ldc, int c0)
{
    const int64_t M2 = (M + 1) / 2;
    const int K8 = K & ~7;
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < M2; ++q) {
        const int64_t m = 2 * q;
        const int two = m + 1 < M;
        const float *a0 = A + m * lda, *a1 = two ? a0 + lda : a0;
        float *cr0 = C + m * ldc + c0, *cr1 = two ? cr0 + ldc : cr0;
        int n = 0;
        for (; n + 4 <= N; n += 4) {
            const float *w0 = W + (int64_t)n * ldw, *w1 = w0 + ldw, *w2 = w1 + ldw, *w3 = w2 + ldw;
            __m256 s00 = _mm256_setzero_ps(), s01 = s00, s02 = s00, s03 = s00, s10 = s00, s11 = s00, s12 = s00, s13 = s00;
            for (int k = 0; k < K8; k += 8) {
                const __m256 x0 = _mm256_loadu_ps(a0 + k), x1 = _mm256_loadu_ps(a1 + k);
                __m256 w = _mm256_loadu_ps(w0 + k);
                s00 = _mm256_fmadd_ps(x0, w, s00); s10 = _mm256_fmadd_ps(x1, w, s10);
                w = _mm256_loadu_ps(w1 + k);
                s01 = _mm256_fmadd_ps(x0, w, s01); s11 = _mm256_fmadd_ps(x1, w, s11);
                w = _mm256_loadu_ps(w2 + k);
                s02 = _mm256_fmadd_ps(x0, w, s02); s12 = _mm256_fmadd_ps(x1, w, s12);
                w = _mm256_loadu_ps(w3 + k);
                s03 = _mm256_fmadd_ps(x0, w, s03); s13 = _mm256_fmadd_ps(x1, w, s13);
            }
            float r0[4] = {hsum8(s00), hsum8(s01), hsum8(s02), hsum8(s03)};
            float r1[4] = {hsum8(s10), hsum8(s11), hsum8(s12), hsum8(s13)};
            for (int k = K8; k < K; ++k) {
                const float x0 = a0[k], x1 = a1[k];
                r0[0] += x0 * w0[k]; r0[1] += x0 * w1[k]; r0[2] += x0 * w2[k]; r0[3] += x0 * w3[k];
                r1[0] += x1 * w0[k]; r1[1] += x1 * w1[k]; r1[2] += x1 * w2[k]; r1[3] += x1 * w3[k];
            }
            for (int j = 0; j < 4; ++j) cr0[n + j] = r0[j];
            if (two)
                for (int j = 0; j < 4; ++j) cr1[n + j] = r1[j];
        }
        for (; n < N; ++n) {
            const float *w = W + (int64_t)n * ldw;
            float t0 = 0.f, t1 = 0.f;
#pragma omp simd reduction(+ : t0, t1)
            for (int k = 0; k < K; ++k) { t0 += a0[k] * w[k]; t1 += a1[k] * w[k]; }
            cr0[n] = t0;
            if (two) cr1[n] = t1;
        }
    }
}

/* out[m, k] = sum_n G[m, g0 + n] * W[n, k]   (input gradient of a projection); four rows of W per pass over out[m] */
static void gemm_nn(int64_t M, int N, int K, const float *G, int64_t ldg, int g0, const float *W, int64_t ldw,
                    float *out, int64_t ldo)
{
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
        float *o = out + m * ldo;
        const float *g = G + m * ldg + g0;
        for (int k = 0; k < K; ++k) o[k] = 0.f;
        int n = 0;
        for (; n + 4 <= N; n += 4) {
            const float g0v = g[n], g1v = g[n + 1], g2v = g[n + 2], g3v = g[n + 3];
            if (g0v == 0.f && g1v == 0.f && g2v == 0.f && g3v == 0.f) continue;      /* ReLU'd rows are half zeros */
            const float *w0 = W + (int64_t)n * ldw, *w1 = w0 + ldw, *w2 = w1 + ldw, *w3 = w2 + ldw;
#pragma omp simd
            for (int k = 0; k < K; ++k) o[k] += g0v * w0[k] + g1v * w1[k] + g2v * w2[k] + g3v * w3[k];
        }
        for (; n < N; ++n) {
            const float gv = g[n];
            const float *w = W + (int64_t)n * ldw;
#pragma omp simd
            for (int k = 0; k < K; ++k) o[k] += gv * w[k];
        }
    }
}

/* dW[n, k] += sum_m G[m, g0 + n] * A[m, k].  Tasks = (M chunk) x (block of output rows n): at most WG_MCH partial
 * copies of dW whatever the thread count (one copy per THREAD was 158 MB of zero-fill and reduction per call on 128
 * threads), and the threads that share an M chunk read the same rows of A out of the shared cache.  Partials are summed
 * in chunk order. */
#define WG_MCH 8
#define WG_NB 8
static void wgrad_tn(int64_t M, int N, int K, const float *G, int64_t ldg, int g0, const float *A, int64_t lda,
                     float *dW)
{
    int C = WG_MCH;
    if (C > M / 64) C = (int)(M / 64);
    if (C < 1) C = 1;
    const int NBK = (N + WG_NB - 1) / WG_NB;
    float *part = (float *)calloc((size_t)C * N * K, sizeof(float));
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int t = 0; t < C; ++t) {
        for (int nb = 0; nb < NBK; ++nb) {
            float *p = part + (size_t)t * N * K;
            const int64_t m0 = M * t / C, m1 = M * (t + 1) / C;
            const int n0 = nb * WG_NB;
            if (n0 + WG_NB <= N) {
                /* register block: 4 output rows x 24 columns (12 accumulators) summed over the chunk's rows */
                for (int nn = n0; nn < n0 + WG_NB; nn += 4) {
                    int k0 = 0;
                    for (; k0 + 24 <= K; k0 += 24) {
                        __m256 c00 = _mm256_setzero_ps(), c01 = c00, c02 = c00, c10 = c00, c11 = c00, c12 = c00,
                               c20 = c00, c21 = c00, c22 = c00, c30 = c00, c31 = c00, c32 = c00;
                        for (int64_t m = m0; m < m1; ++m) {
                            const float *a = A + m * lda + k0;
                            const float *g = G + m * ldg + g0 + nn;
                            const __m256 a0 = _mm256_loadu_ps(a), a1 = _mm256_loadu_ps(a + 8), a2 = _mm256_loadu_ps(a + 16);
                            __m256 u = _mm256_broadcast_ss(g);
                            c00 = _mm256_fmadd_ps(u, a0, c00); c01 = _mm256_fmadd_ps(u, a1, c01); c02 = _mm256_fmadd_ps(u, a2, c02);
                            u = _mm256_broadcast_ss(g + 1);
                            c10 = _mm256_fmadd_ps(u, a0, c10); c11 = _mm256_fmadd_ps(u, a1, c11); c12 = _mm256_fmadd_ps(u, a2, c12);
                            u = _mm256_broadcast_ss(g + 2);
                            c20 = _mm256_fmadd_ps(u, a0, c20); c21 = _mm256_fmadd_ps(u, a1, c21); c22 = _mm256_fmadd_ps(u, a2, c22);
                            u = _mm256_broadcast_ss(g + 3);
                            c30 = _mm256_fmadd_ps(u, a0, c30); c31 = _mm256_fmadd_ps(u, a1, c31); c32 = _mm256_fmadd_ps(u, a2, c32);
                        }
                        float *r = p + (size_t)nn * K + k0;
                        _mm256_storeu_ps(r, c00); _mm256_storeu_ps(r + 8, c01); _mm256_storeu_ps(r + 16, c02); r += K;
                        _mm256_storeu_ps(r, c10); _mm256_storeu_ps(r + 8, c11); _mm256_storeu_ps(r + 16, c12); r += K;
                        _mm256_storeu_ps(r, c20); _mm256_storeu_ps(r + 8, c21); _mm256_storeu_ps(r + 16, c22); r += K;
                        _mm256_storeu_ps(r, c30); _mm256_storeu_ps(r + 8, c31); _mm256_storeu_ps(r + 16, c32);
                    }
                    for (; k0 < K; ++k0) {                /* the last K % 24 columns */
                        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
                        for (int64_t m = m0; m < m1; ++m) {
                            const float av = A[m * lda + k0];
                            const float *g = G + m * ldg + g0 + nn;
                            s0 += g[0] * av; s1 += g[1] * av; s2 += g[2] * av; s3 += g[3] * av;
                        }
                        p[(size_t)nn * K + k0] = s0; p[(size_t)(nn + 1) * K + k0] = s1;
                        p[(size_t)(nn + 2) * K + k0] = s2; p[(size_t)(nn + 3) * K + k0] = s3;
                    }
                }
            } else {                                      /* the last N % WG_NB output rows */
                for (int64_t m = m0; m < m1; ++m) {
                    const float *a = A + m * lda;
                    for (int n = n0; n < N; ++n) {
                        const float u = G[m * ldg + g0 + n];
                        if (u == 0.f) continue;
                        float *row = p + (size_t)n * K;
#pragma omp simd
                        for (int k = 0; k < K; ++k) row[k] += u * a[k];
                    }
                }
            }
        }
    }
    const int64_t NK = (int64_t)N * K;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < NK; ++i) {
        float sum = 0.f;
        for (int t = 0; t < C; ++t) sum += part[(size_t)t * NK + i];
        dW[i] += sum;
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
        PROF(0, seg_mean(feats, ld, idk[k], size[k], 1, D, X[0][k]));
        PROF(0, seg_mean(feats, ld, idk[k + 1], size[k], fan[k], D, A[0][k]));
    }
    for (int l = 0; l < L; ++l) {
        const int hl = h[l], dw = din[l];
        for (int k = 0; k < L - l; ++k) {
            if (l > 0) {
                A[l][k] = (float *)malloc(sizeof(float) * size[k] * dw);
                PROF(0, seg_mean(X[l][k + 1], dw, NULL, size[k], fan[k], dw, A[l][k]));
            }
            float *O = (float *)malloc(sizeof(float) * size[k] * 2 * hl);
            PROF(1, gemm_nt(size[k], hl, dw, X[l][k], dw, params + oWx[l], dw, O, 2 * hl, 0));
            PROF(1, gemm_nt(size[k], hl, dw, A[l][k], dw, params + oWn[l], dw, O, 2 * hl, hl));
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
    PROF(3, wgrad_tn(B, C, E, dl, C, 0, z, E, grads + oWfc));
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
            PROF(3, wgrad_tn(size[k], hl, dlv, G, 2 * hl, 0, X[l][k], dlv, grads + oWx[l]));
            PROF(3, wgrad_tn(size[k], hl, dlv, G, 2 * hl, hl, A[l][k], dlv, grads + oWn[l]));
            if (l > 0) {
                float *gx = (float *)malloc(sizeof(float) * size[k] * dlv);
                float *ga = (float *)malloc(sizeof(float) * size[k] * dlv);
                PROF(4, gemm_nn(size[k], hl, dlv, G, 2 * hl, 0, params + oWx[l], dlv, gx, dlv));
                PROF(4, gemm_nn(size[k], hl, dlv, G, 2 * hl, hl, params + oWn[l], dlv, ga, dlv));
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

void gso_profile(double *out) { for (int i = 0; i < 8; ++i) out[i] = g_prof[i]; }
int gso_omp_threads(void) { return omp_get_max_threads(); }
void gso_omp_set_threads(int n) { omp_set_num_threads(n); }
