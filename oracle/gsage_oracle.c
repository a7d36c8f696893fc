/*
 * gsage_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the integer half of the reference hot path
 * (bkj/pytorch-graphsage, /root/reference).  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load this library; the product (pytorch-graphsage_amd/)
 * never does and fails loudly when its HIP library is missing.
 *
 * Parity status: PINNED.  Every function below is checked bit-exactly against golden vectors
 * produced by importing the reference itself (tests/golden/gen_golden.py ->
 * tests/golden/{sampler,stream,iterate}_kat.npz) in tests/test_oracle_*.py.
 *
 * What is restated, and from where:
 *   gso_mt_*            the numpy *legacy global* MT19937 stream the reference draws from:
 *                         helpers.py:14-15   np.random.seed(seed)            -> init_genrand
 *                         nn_modules.py:88   np.random.choice(max_deg,(M,n)) -> masked rejection
 *                         problem.py:146     np.random.permutation(idx)      -> Fisher-Yates
 *                       (algorithm = numpy 2.2 `_legacy_seeding` / `random_bounded_uint64_fill`
 *                        / `random_interval`; numpy is a pinned third-party dependency that is
 *                        not part of /root/reference, so its published algorithm is restated
 *                        and anchored on the reference's own call sites via the fixtures.)
 *   gso_sample_csr_sel  nn_modules.py:80-101 SparseUniformNeighborSampler.__call__
 *   gso_degrees         nn_modules.py:72-78  SparseUniformNeighborSampler.__init__
 *   gso_philox4x32_10   Salmon et al. SC'11 Philox4x32-10 (the build's counter-based generator;
 *                       no reference counterpart -- pinned by the Random123 known-answer vectors)
 *   gso_philox_sel      the build's definition of `sel` in counter mode
 *   gso_gather_mean_f32 models.py:76,80 + nn_modules.py:197-198 (feats[ids] -> view -> mean(1))
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>

/* ------------------------------------------------------------------ MT19937 (legacy numpy) */
#define GSO_MT_N 624
#define GSO_MT_M 397

typedef struct {
    uint32_t key[GSO_MT_N];
    int32_t pos;
} gso_mt19937;

/* np.random.seed(int) -> _legacy_seeding -> mt19937_seed(): Knuth's init_genrand. helpers.py:15 */
void gso_mt_seed(gso_mt19937 *st, uint32_t seed)
{
    int i;
    st->key[0] = seed;
    for (i = 1; i < GSO_MT_N; i++)
        st->key[i] = 1812433253u * (st->key[i - 1] ^ (st->key[i - 1] >> 30)) + (uint32_t)i;
    st->pos = GSO_MT_N;
}

static void gso_mt_twist(gso_mt19937 *st)
{
    int kk;
    uint32_t y;
    uint32_t *mt = st->key;
    for (kk = 0; kk < GSO_MT_N - GSO_MT_M; kk++) {
        y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
        mt[kk] = mt[kk + GSO_MT_M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (; kk < GSO_MT_N - 1; kk++) {
        y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
        mt[kk] = mt[kk + (GSO_MT_M - GSO_MT_N)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    y = (mt[GSO_MT_N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
    mt[GSO_MT_N - 1] = mt[GSO_MT_M - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    st->pos = 0;
}

uint32_t gso_mt_next(gso_mt19937 *st)
{
    uint32_t y;
    if (st->pos >= GSO_MT_N)
        gso_mt_twist(st);
    y = st->key[st->pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

static uint64_t gso_mask_of(uint64_t v)
{
    v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; v |= v >> 32;
    return v;
}

/* np.random.choice(high, size) == legacy randint(0, high, size) for an int population, with
 * replacement (nn_modules.py:88).  Masked rejection on 32-bit words; `high-1 == 0` consumes
 * no words.  Returns the number of 32-bit words consumed.  Requires 1 <= high <= 2^32. */
int64_t gso_mt_choice(gso_mt19937 *st, int64_t high, int64_t count, int64_t *out)
{
    uint64_t rng = (uint64_t)(high - 1);
    int64_t i, words = 0;
    if (rng == 0) {
        for (i = 0; i < count; i++) out[i] = 0;
        return 0;
    }
    if (rng == 0xffffffffull) {
        for (i = 0; i < count; i++) out[i] = (int64_t)gso_mt_next(st);
        return count;
    }
    {
        uint32_t mask = (uint32_t)gso_mask_of(rng);
        for (i = 0; i < count; i++) {
            uint32_t v;
            do { v = gso_mt_next(st) & mask; words++; } while (v > rng);
            out[i] = (int64_t)v;
        }
    }
    return words;
}

/* np.random.permutation(np.arange(n)) (problem.py:145-146): copy + legacy shuffle =
 * Fisher-Yates for i = n-1 .. 1 with j = random_interval(i) (masked rejection, 32-bit words). */
void gso_mt_permutation(gso_mt19937 *st, int64_t n, int64_t *out)
{
    int64_t i;
    for (i = 0; i < n; i++) out[i] = i;
    for (i = n - 1; i >= 1; i--) {
        uint32_t mask = (uint32_t)gso_mask_of((uint64_t)i);
        uint32_t v;
        int64_t t;
        do { v = gso_mt_next(st) & mask; } while ((uint64_t)v > (uint64_t)i);
        t = out[i]; out[i] = out[v]; out[v] = t;
    }
}

size_t gso_mt_sizeof(void) { return sizeof(gso_mt19937); }

/* ------------------------------------------------------------------ sampler */
/* SparseUniformNeighborSampler.__init__ (nn_modules.py:72-78): per-row count of stored
 * non-zero entries.  Equals diff(indptr) whenever no explicit zero is stored. */
void gso_degrees(const int64_t *indptr, const int64_t *data, int64_t n_rows, int64_t *deg)
{
    int64_t r, p;
    for (r = 0; r < n_rows; r++) {
        int64_t d = 0;
        for (p = indptr[r]; p < indptr[r + 1]; p++) d += (data[p] != 0);
        deg[r] = d;
    }
}

/* SparseUniformNeighborSampler.__call__ (nn_modules.py:80-101) given the `sel` matrix the
 * reference drew at :88.  CSR is in the reference convention (row i holds its neighbours in
 * columns 0..deg_i-1, so column index == position in the row):
 *     out[i*n+j] = data[indptr[ids[i]] + sel[i,j] % deg_i]      deg_i > 0
 *                = 0                                            deg_i == 0 (numpy x % 0 == 0,
 *                                                               then column 0 of an empty row)
 * Returns 0, or -1 if an id is outside [0, n_rows) (the reference raises IndexError). */
int gso_sample_csr_sel(const int64_t *indptr, const int64_t *data, int64_t n_rows,
                       const int64_t *ids, int64_t M, int64_t n, const int64_t *sel, int64_t *out)
{
    int64_t i, j;
    for (i = 0; i < M; i++) {
        int64_t id = ids[i], beg, deg;
        if (id < 0 || id >= n_rows) return -1;
        beg = indptr[id];
        deg = indptr[id + 1] - beg;
        for (j = 0; j < n; j++)
            out[i * n + j] = deg > 0 ? data[beg + sel[i * n + j] % deg] : 0;
    }
    return 0;
}

/* The whole reference call on its own stream: draw sel with the legacy generator, then sample. */
int gso_sample_csr_mt(gso_mt19937 *st, const int64_t *indptr, const int64_t *data, int64_t n_rows,
                      int64_t max_deg, const int64_t *ids, int64_t M, int64_t n, int64_t *sel,
                      int64_t *out)
{
    gso_mt_choice(st, max_deg, M * n, sel);
    return gso_sample_csr_sel(indptr, data, n_rows, ids, M, n, sel, out);
}

/* ------------------------------------------------------------------ Philox4x32-10 */
static void gso_mulhilo(uint32_t a, uint32_t b, uint32_t *hi, uint32_t *lo)
{
    uint64_t p = (uint64_t)a * (uint64_t)b;
    *hi = (uint32_t)(p >> 32);
    *lo = (uint32_t)p;
}

void gso_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    int r;
    for (r = 0; r < 10; r++) {
        uint32_t hi0, lo0, hi1, lo1;
        gso_mulhilo(0xD2511F53u, c0, &hi0, &lo0);
        gso_mulhilo(0xCD9E8D57u, c2, &hi1, &lo1);
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Counter-mode `sel`: sample with GLOBAL index g (row-major position in the
 * whole job's [M_global, n] matrix of sampler call `call`) uses
 *     word = philox4x32_10(ctr = {lo32(g>>2), hi32(g>>2), lo32(call), hi32(call)},
 *                          key = {lo32(seed), hi32(seed)})[g & 3]
 *     sel  = (word * max_deg) >> 32            in [0, max_deg)
 * so results do not depend on how the seed batch is sharded across GPUs. */
void gso_philox_sel(uint64_t seed, uint64_t call, uint64_t g0, int64_t count, uint32_t max_deg,
                    int64_t *sel)
{
    int64_t t;
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    for (t = 0; t < count; t++) {
        uint64_t g = g0 + (uint64_t)t, blk = g >> 2;
        uint32_t ctr[4] = { (uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)call,
                            (uint32_t)(call >> 32) };
        uint32_t w[4];
        gso_philox4x32_10(ctr, key, w);
        sel[t] = (int64_t)(((uint64_t)w[g & 3] * (uint64_t)max_deg) >> 32);
    }
}

/* ------------------------------------------------------------------ gather + mean (fp32) */
/* feats[ids] (models.py:76,80) -> view(M, n, D).mean(1) (nn_modules.py:197-198), fp32 table
 * with row stride ld.  ids == NULL means rows are already in order (row i*n+j).  Sums in fp64
 * then rounds once, so it bounds (not reproduces) torch's fp32 summation order. */
void gso_gather_mean_f32(const float *table, int64_t ld, const int64_t *ids, int64_t M, int64_t n,
                         int64_t D, float *out)
{
    int64_t i, j, c;
    for (i = 0; i < M; i++)
        for (c = 0; c < D; c++) {
            double s = 0.0;
            for (j = 0; j < n; j++) {
                int64_t r = ids ? ids[i * n + j] : i * n + j;
                s += (double)table[r * ld + c];
            }
            out[i * D + c] = (float)(s / (double)n);
        }
}
