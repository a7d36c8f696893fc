// gsage_sample.hip -- K1: CSR uniform neighbour sampler (gfx950) + the host-side legacy stream.
//
// Replaces SparseUniformNeighborSampler (reference nn_modules.py:52-101): the reference pulls
// ids to the host, slices a scipy CSR (copying every full adjacency row), draws `sel` from
// numpy's global MT19937 and copies the result back (2 D2H + 2 H2D syncs per step).  Here the
// graph is resident in HBM as (rowptr int64, col int32) and one launch per hop does
//     out[i*n+j] = col[rowptr[id_i] + sel[i,j] % deg_i]      (0 when deg_i == 0)
// HBM-latency bound integer work: one lane per sample (sel mode) or per Philox block of four
// samples (counter mode); rowptr[id], rowptr[id+1] are adjacent 8-byte words, the n lanes of a
// parent hit the same two words (TA broadcast), the only scattered traffic is one 4-byte `col`
// read and one 8-byte write per sample.
#include "gsage_common.h"
#include "gsage_sample_dev.h"

#include <string.h>

namespace gsage {

// ---- device code ----------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_sample_sel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, int64_t n_rows,
             const int64_t *__restrict__ ids, int64_t total, uint32_t n,
             const int32_t *__restrict__ sel, int64_t *__restrict__ out, int32_t *err_flag)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += stride) {
        const int64_t i = (total <= 0xffffffffLL) ? (int64_t)((uint32_t)g / n) : g / (int64_t)n;
        out[g] = pick_neighbor(rowptr, col, n_rows, ids[i], (uint32_t)sel[g], err_flag);
    }
}

__global__ void __launch_bounds__(256)
k_sample_philox(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                int64_t n_rows, const int64_t *__restrict__ ids, int64_t total, uint32_t n,
                uint32_t max_deg, uint32_t seed_lo, uint32_t seed_hi,
                const uint64_t *__restrict__ call_ctr, uint64_t call_base, uint64_t g0,
                int64_t *__restrict__ out, int32_t *__restrict__ sel_out, int32_t *err_flag)
{
    const uint64_t call = call_base + (call_ctr ? *call_ctr : 0ull);
    const uint64_t blk0 = g0 >> 2;
    const uint64_t nblk = ((g0 + (uint64_t)total - 1) >> 2) - blk0 + 1;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x; b < nblk; b += stride) {
        const uint64_t blk = blk0 + b;
        const philox4 r = philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)call,
                                        (uint32_t)(call >> 32), seed_lo, seed_hi);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t g = (blk << 2) + (uint64_t)k;
            if (g < g0 || g >= g0 + (uint64_t)total) continue;
            const int64_t t = (int64_t)(g - g0);
            const uint32_t s = (uint32_t)(((uint64_t)r.v[k] * (uint64_t)max_deg) >> 32);
            const int64_t i = (total <= 0xffffffffLL) ? (int64_t)((uint32_t)t / n) : t / (int64_t)n;
            if (sel_out) sel_out[t] = (int32_t)s;
            out[t] = pick_neighbor(rowptr, col, n_rows, ids[i], s, err_flag);
        }
    }
}

// Dense sampler (reference nn_modules.py:19-49): out[i, j] = adj[ids[i], keep[j]] -- `tmp = adj[ids]; tmp[:, perm]
// [:, :n]` of the reference without the [M, K] intermediate (and its second gather).  A parent's n samples sit in
// consecutive lanes: they read n scattered 8-byte words of ONE K * 8-byte row (1 KiB at K = 128, eight lines).
__global__ void __launch_bounds__(256)
k_sample_dense(const int64_t *__restrict__ adj, int64_t ld, int64_t n_rows, const int64_t *__restrict__ ids,
               int64_t total, uint32_t n, const int64_t *__restrict__ keep, int64_t *__restrict__ out,
               int32_t *err_flag)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += stride) {
        const int64_t i = (total <= 0xffffffffLL) ? (int64_t)((uint32_t)g / n) : g / (int64_t)n;
        const int64_t j = g - i * (int64_t)n;
        const int64_t c = keep[j];
        out[g] = (c < 0 || c >= ld) ? pick_dense(adj, ld, n_rows, -1, 0u, err_flag)
                                    : pick_dense(adj, ld, n_rows, ids[i], (uint32_t)c, err_flag);
    }
}

// two small device-to-device copies in ONE launch (a step's seed ids and targets into the buffers a captured step
// reads): as hipMemcpyAsync each was a ~7 us blit on the step's stream
__global__ void __launch_bounds__(256)
k_copy_pair(uint32_t *__restrict__ d0, const uint32_t *__restrict__ s0, int64_t n0, uint32_t *__restrict__ d1,
            const uint32_t *__restrict__ s1, int64_t n1)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n0 + n1; t += stride) {
        if (t < n0) d0[t] = s0[t];
        else d1[t - n0] = s1[t - n0];
    }
}

__global__ void k_counter_add(uint64_t *ctr, uint64_t inc) { *ctr += inc; }

__global__ void __launch_bounds__(256)
k_sample_hops(const HopsParams p)
{
    extern __shared__ int64_t frontier[];
    sample_hops_workgroup(p, blockIdx.x, frontier);
}

// ---- the legacy MT19937 stream ON THE DEVICE ------------------------------------------------------
// Compat mode draws `sel` from numpy's global legacy stream, exactly as the reference does at
// nn_modules.py:88 (np.random.choice(high, size) = masked rejection over 32-bit words: v = word & mask,
// accepted if v <= high - 1).  On the host that is ~1 ms of numpy per 512-seed Reddit batch plus a 560 KB
// H2D copy per step.  MT19937 is a sequential recurrence, but one refill of its 624 words splits into
// three data-parallel phases (word k needs words k+1 and (k+397) mod 624: for k < 227 both are old, for
// 227 <= k < 454 the second one was produced in the first phase, and so on), tempering is per word, and
// the order-preserving rejection is a prefix sum over acceptance flags.  One workgroup owns the stream:
// state and position live in device memory between launches, so a training run hands the stream over
// once per epoch instead of once per sampler call.
//   st[0..623] = state words, st[624] = position (624 = refill before the next word)
constexpr uint32_t MT_N = 624, MT_M = 397;

__device__ __forceinline__ uint32_t mt_mix(uint32_t hi, uint32_t lo)
{
    const uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

__device__ __forceinline__ uint32_t mt_temper(uint32_t y)
{
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// one refill of the 624 state words in LDS (256 threads)
__device__ __forceinline__ void mt_refill(uint32_t *s)
{
    const int t = threadIdx.x;
    // phases [0,227) [227,454) [454,623): every operand of a phase is final before the phase starts
    for (int base = 0; base < 623; base += 227) {
        const int k = base + t;
        const bool on = t < 227 && k < 623;
        uint32_t v = 0;
        if (on) v = s[(k + MT_M) % MT_N] ^ mt_mix(s[k], s[k + 1]);
        __syncthreads();
        if (on) s[k] = v;
        __syncthreads();
    }
    if (t == 0) s[623] = s[396] ^ mt_mix(s[623], s[0]);
    __syncthreads();
}

// `count` accepted values, in stream order, into out; s = the 624 state words in LDS, idx = position.
// wave_tot / cut: LDS scratch.  Returns the new position (uniform across the workgroup).
__device__ __forceinline__ uint32_t mt_fill(uint32_t *s, uint32_t idx, uint32_t top, uint32_t mask, int64_t count,
                                            int32_t *__restrict__ out, int *wave_tot, int *cut_p)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int64_t produced = 0;
    while (produced < count) {
        if (idx >= MT_N) {
            mt_refill(s);
            idx = 0;
        }
        // the next (up to) 256 words, in order
        const uint32_t pos = idx + (uint32_t)t;
        const bool have = pos < MT_N;
        const uint32_t v = have ? (mt_temper(s[pos]) & mask) : 0u;
        const int acc = (have && v <= top) ? 1 : 0;
        int incl = acc;                                          // inclusive prefix inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        if (lane == 63) wave_tot[wave] = incl;
        if (t == 0) *cut_p = -1;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) before += wave_tot[w];
            total += wave_tot[w];
        }
        incl += before;
        const int64_t need = count - produced;
        if (acc && (int64_t)incl <= need) out[produced + incl - 1] = (int32_t)v;
        if (acc && (int64_t)incl == need) *cut_p = t;              // the word that completes the request
        __syncthreads();
        const int c = *cut_p;
        const uint32_t n_here = MT_N - idx < 256u ? MT_N - idx : 256u;
        if ((int64_t)total >= need) {
            idx += (uint32_t)c + 1u;
            produced = count;
        } else {
            idx += n_here;
            produced += total;
        }
        __syncthreads();                                           // wave_tot / cut are reused
    }
    return idx;
}

// n_seg requests served back to back from ONE stream (n_seg == 1, seg_* null: `count` values into out):
// request q puts seg_cnt[q] values at out + seg_off[q].  A training epoch's sampler draws are one launch:
// consecutive np.random.choice calls with the same range consume the stream exactly like this.
__global__ void __launch_bounds__(256)
k_mt_choice(uint32_t *__restrict__ st, uint32_t top, uint32_t mask, int64_t count, int32_t *__restrict__ out,
            const int64_t *__restrict__ seg_off, const int64_t *__restrict__ seg_cnt, int64_t n_seg)
{
    __shared__ uint32_t s[MT_N];
    __shared__ int wave_tot[4];
    __shared__ int cut;
    const int t = threadIdx.x;
    for (int k = t; k < (int)MT_N; k += 256) s[k] = st[k];
    uint32_t idx = st[MT_N];
    __syncthreads();
    if (!seg_off) {
        idx = mt_fill(s, idx, top, mask, count, out, wave_tot, &cut);
    } else {
        for (int64_t q = 0; q < n_seg; ++q)
            idx = mt_fill(s, idx, top, mask, seg_cnt[q], out + seg_off[q], wave_tot, &cut);
    }
    for (int k = t; k < (int)MT_N; k += 256) st[k] = s[k];
    if (t == 0) st[MT_N] = idx;
}

static inline int grid_for(int64_t work_items)
{
    int64_t blocks = ceil_div(work_items, 256);
    if (blocks > 8192) blocks = 8192;          // grid-stride the rest
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

// ---- host-side legacy MT19937 (numpy global stream) -------------------------------------------
// Compat mode only: reproduces what the reference consumes through np.random.seed /
// np.random.choice / np.random.permutation (helpers.py:15, nn_modules.py:88, problem.py:146).
class LegacyStream {
public:
    explicit LegacyStream(uint32_t seed) { reseed(seed); }

    void reseed(uint32_t seed)
    {
        s_[0] = seed;
        for (uint32_t i = 1; i < kN; ++i) s_[i] = 1812433253u * (s_[i - 1] ^ (s_[i - 1] >> 30)) + i;
        idx_ = kN;
    }

    uint32_t next()
    {
        if (idx_ >= kN) refill();
        uint32_t y = s_[idx_++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }

    // uniform on [0, top] by masked rejection of 32-bit words (top < 2^32)
    uint32_t bounded(uint32_t top, uint32_t mask, int64_t *words)
    {
        uint32_t v;
        do {
            v = next() & mask;
            ++*words;
        } while (v > top);
        return v;
    }

    static uint32_t mask_for(uint32_t top)
    {
        uint32_t m = top;
        m |= m >> 1; m |= m >> 2; m |= m >> 4; m |= m >> 8; m |= m >> 16;
        return m;
    }

private:
    static constexpr uint32_t kN = 624, kM = 397;
    uint32_t s_[kN];
    uint32_t idx_;

    static uint32_t mix(uint32_t hi, uint32_t lo)
    {
        const uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
        return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }

    void refill()
    {
        for (uint32_t k = 0; k < kN; ++k)
            s_[k] = s_[(k + kM) % kN] ^ mix(s_[k], s_[(k + 1) % kN]);
        idx_ = 0;
    }
};

}  // namespace gsage

using namespace gsage;

extern "C" {

int gsage_sample_csr_sel(const int64_t *rowptr, const int32_t *col, int64_t n_rows,
                         const int64_t *ids, int64_t M, int32_t n, const int32_t *sel,
                         int64_t *out, int32_t *err_flag, void *stream)
{
    GSAGE_REQUIRE(n > 0, "sample_csr_sel: n_samples must be > 0");      // nn_modules.py:81
    GSAGE_REQUIRE(M >= 0 && n_rows >= 0, "sample_csr_sel: negative size");
    if (M == 0) return GSAGE_OK;
    GSAGE_REQUIRE(rowptr && col && ids && sel && out, "sample_csr_sel: null pointer");
    const int64_t total = M * (int64_t)n;
    launch(k_sample_sel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       rowptr, col, n_rows, ids, total, (uint32_t)n, sel, out, err_flag);
    return check_launch("sample_csr_sel");
}

int gsage_sample_dense(const int64_t *adj, int64_t ld, int64_t n_rows, const int64_t *ids, int64_t M,
                       const int64_t *keep, int32_t n, int64_t *out, int32_t *err_flag, void *stream)
{
    GSAGE_REQUIRE(n >= 0 && M >= 0 && n_rows >= 0 && ld >= 0, "sample_dense: negative size");
    if (M == 0 || n == 0) return GSAGE_OK;
    GSAGE_REQUIRE(adj && ids && keep && out, "sample_dense: null pointer");
    GSAGE_REQUIRE(n <= ld, "sample_dense: more samples than columns (the reference samples without replacement)");
    const int64_t total = M * (int64_t)n;
    launch(k_sample_dense, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, adj, ld, n_rows, ids, total,
           (uint32_t)n, keep, out, err_flag);
    return check_launch("sample_dense");
}

int gsage_sample_csr_philox(const int64_t *rowptr, const int32_t *col, int64_t n_rows,
                            const int64_t *ids, int64_t M, int32_t n, uint32_t max_deg,
                            uint64_t seed, const uint64_t *call_ctr, uint64_t call_base,
                            uint64_t g0, int64_t *out, int32_t *sel_out, int32_t *err_flag,
                            void *stream)
{
    GSAGE_REQUIRE(n > 0, "sample_csr_philox: n_samples must be > 0");
    GSAGE_REQUIRE(M >= 0 && n_rows >= 0, "sample_csr_philox: negative size");
    GSAGE_REQUIRE(max_deg > 0, "sample_csr_philox: max_deg must be > 0");
    if (M == 0) return GSAGE_OK;
    GSAGE_REQUIRE(rowptr && col && ids && out, "sample_csr_philox: null pointer");
    const int64_t total = M * (int64_t)n;
    launch(k_sample_philox, dim3(grid_for(total / 4 + 2)), dim3(256), 0,
                       (hipStream_t)stream, rowptr, col, n_rows, ids, total, (uint32_t)n, max_deg,
                       (uint32_t)seed, (uint32_t)(seed >> 32), call_ctr, call_base, g0, out,
                       sel_out, err_flag);
    return check_launch("sample_csr_philox");
}

int gsage_sample_hops_philox(const int64_t *rowptr, const int32_t *col, int64_t n_rows, int64_t *ids,
                             int64_t B, int32_t n_hops, const int32_t *fan, uint32_t max_deg,
                             uint64_t seed, const uint64_t *call_ctr, uint64_t call_base,
                             uint64_t rank, const int64_t *seed_queue, const int64_t *batch_idx,
                             int64_t n_batches, int32_t *err_flag, void *stream)
{
    gsage_hops_desc d;
    d.rowptr = rowptr; d.col = col; d.n_rows = n_rows; d.ids = ids; d.B = B; d.n_hops = n_hops;
    for (int k = 0; k < 5; ++k) d.fan[k] = (fan && k < n_hops && n_hops <= 5) ? fan[k] : 1;
    GSAGE_REQUIRE(fan, "sample_hops_philox: null pointer");
    d.max_deg = max_deg; d.seed = seed; d.call_ctr = call_ctr; d.call_base = call_base; d.rank = rank;
    d.seed_queue = seed_queue; d.batch_idx = batch_idx; d.batch_base = 0; d.n_batches = n_batches;
    d.err_flag = err_flag; d.sel = nullptr; d.sel_stride = 0; d.dense_adj = nullptr; d.dense_ld = 0;
    return gsage_sample_hops(&d, stream);
}

int gsage_sample_hops(const gsage_hops_desc *hops, void *stream)
{
    GSAGE_REQUIRE(hops, "sample_hops: null descriptor");
    HopsParams p;
    size_t lds = 0;
    int rc = fill_hops(p, lds, *hops);
    if (rc != GSAGE_OK || hops->B == 0) return rc;
    launch(k_sample_hops, dim3((unsigned)ceil_div(hops->B, p.spw)), dim3(256), lds, (hipStream_t)stream, p);
    return check_launch("sample_hops");
}

int gsage_copy_pair(void *dst0, const void *src0, int64_t bytes0, void *dst1, const void *src1, int64_t bytes1,
                    void *stream)
{
    GSAGE_REQUIRE(bytes0 >= 0 && bytes1 >= 0 && (bytes0 == 0 || (dst0 && src0)) && (bytes1 == 0 || (dst1 && src1)),
                  "copy_pair: bad arguments");
    GSAGE_REQUIRE(bytes0 % 4 == 0 && bytes1 % 4 == 0 &&
                  ((((uintptr_t)dst0 | (uintptr_t)src0 | (uintptr_t)dst1 | (uintptr_t)src1) & 3) == 0),
                  "copy_pair: 4-byte granularity");
    const int64_t words = (bytes0 + bytes1) / 4;
    if (words == 0) return GSAGE_OK;
    int64_t blocks = ceil_div(words, (int64_t)256);
    if (blocks > 1024) blocks = 1024;
    launch(k_copy_pair, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (uint32_t *)dst0,
           (const uint32_t *)src0, bytes0 / 4, (uint32_t *)dst1, (const uint32_t *)src1, bytes1 / 4);
    return check_launch("copy_pair");
}

int gsage_counter_add(uint64_t *ctr, uint64_t inc, void *stream)
{
    GSAGE_REQUIRE(ctr, "counter_add: null pointer");
    launch(k_counter_add, dim3(1), dim3(1), 0, (hipStream_t)stream, ctr, inc);
    return check_launch("counter_add");
}

void *gsage_mt_create(uint32_t seed) { return new LegacyStream(seed); }
void gsage_mt_destroy(void *mt) { delete static_cast<LegacyStream *>(mt); }
void gsage_mt_seed(void *mt, uint32_t seed) { static_cast<LegacyStream *>(mt)->reseed(seed); }

int64_t gsage_mt_choice_i32(void *mt, int64_t high, int64_t count, int32_t *out)
{
    LegacyStream *st = static_cast<LegacyStream *>(mt);
    int64_t words = 0;
    if (high <= 1) {                               // range of one value: numpy draws nothing
        for (int64_t i = 0; i < count; ++i) out[i] = 0;
        return 0;
    }
    const uint32_t top = (uint32_t)(high - 1);
    const uint32_t mask = LegacyStream::mask_for(top);
    for (int64_t i = 0; i < count; ++i) out[i] = (int32_t)st->bounded(top, mask, &words);
    return words;
}

int gsage_mt_choice_device(uint32_t *state, int64_t high, int64_t count, int32_t *out, void *stream)
{
    GSAGE_REQUIRE(state && (out || count == 0) && count >= 0, "mt_choice_device: null pointer / negative count");
    GSAGE_REQUIRE(high >= 1 && high <= 0x100000000LL, "mt_choice_device: high must be in [1, 2^32]");
    if (count == 0) return GSAGE_OK;
    if (high == 1) {                               // numpy draws nothing for a range of one value
        GSAGE_REQUIRE(hipMemsetAsync(out, 0, sizeof(int32_t) * (size_t)count, (hipStream_t)stream) == hipSuccess,
                      "mt_choice_device: memset failed");
        return GSAGE_OK;
    }
    const uint32_t top = (uint32_t)(high - 1);
    launch(k_mt_choice, dim3(1), dim3(256), 0, (hipStream_t)stream, state, top, LegacyStream::mask_for(top), count,
           out, (const int64_t *)nullptr, (const int64_t *)nullptr, (int64_t)0);
    return check_launch("mt_choice_device");
}

int gsage_mt_choice_segments(uint32_t *state, int64_t high, int64_t n_seg, const int64_t *seg_off,
                             const int64_t *seg_cnt, int32_t *out, void *stream)
{
    GSAGE_REQUIRE(state && n_seg >= 0 && (n_seg == 0 || (seg_off && seg_cnt && out)), "mt_choice_segments: null pointer");
    GSAGE_REQUIRE(high >= 2 && high <= 0x100000000LL, "mt_choice_segments: high must be in [2, 2^32]");
    if (n_seg == 0) return GSAGE_OK;
    const uint32_t top = (uint32_t)(high - 1);
    launch(k_mt_choice, dim3(1), dim3(256), 0, (hipStream_t)stream, state, top, LegacyStream::mask_for(top), (int64_t)0,
           out, seg_off, seg_cnt, n_seg);
    return check_launch("mt_choice_segments");
}

void gsage_mt_permutation(void *mt, int64_t n, int64_t *out)
{
    LegacyStream *st = static_cast<LegacyStream *>(mt);
    int64_t words = 0;
    for (int64_t i = 0; i < n; ++i) out[i] = i;
    for (int64_t i = n - 1; i >= 1; --i) {
        const uint32_t j = st->bounded((uint32_t)i, LegacyStream::mask_for((uint32_t)i), &words);
        const int64_t t = out[i];
        out[i] = out[j];
        out[j] = t;
    }
}

}  // extern "C"
