// gsage_mtjump.hip -- numpy's legacy MT19937 stream consumed by MANY workgroups at once (jump-ahead).
//
// The reference draws every sampler call's `sel` from ONE sequential generator (nn_modules.py:88:
// np.random.choice(adj.shape[1], size) on numpy's global legacy stream).  k_mt_choice (gsage_sample.hip) consumes
// that stream on the device with one workgroup -- ~0.47 accepted values per ns, i.e. ~90 ms for the 4.2e7 draws of
// a Reddit-sized epoch, more than the epoch's training steps take.  MT19937 is linear over GF(2): the state after J
// steps is g(A) s with g(x) = x^J mod phi(x) (phi = the characteristic polynomial of the one-word step A, degree
// 19937), and for a linear recurrence g(A) s is a CONVOLUTION of g's coefficient bits with the raw word stream of s:
//
//     window_J[w] = XOR over { i : g_i = 1 } of x[i + w],    w = 0 .. 623,   x = raw words from state s
//
// so a workgroup reaches "its" part of the stream with two table look-ups (x^(a * 64 U) and x^(b U) blocks, a, b < 64,
// U = 64 refills: the polynomials do not depend on the seed) and ~4e4 XORs per state word, all in LDS -- no
// sequential stepping.  The table is computed once (Berlekamp-Massey on 2 x 19937 output bits gives phi, then
// square-and-multiply; ~1 s on the host) and cached by the Python side.
//
// Order-preserving rejection (numpy keeps v = word & mask when v <= top) over many workgroups: pass A counts the
// accepted words of every chunk, a one-workgroup scan turns the counts into offsets, pass B regenerates the chunks
// and stores value #n of the whole request at its place; the workgroup that holds the request's last value leaves
// state and position exactly where numpy would (stream_kat.npz is the test).  A serial finisher (k_mt_par_finish,
// normally a no-op) serves whatever an unlucky estimate of the acceptance rate left over.
#include "gsage_common.h"
#include <string.h>
#include <mutex>

namespace gsage {

constexpr int MTJ_N = 624, MTJ_M = 397;
constexpr int MTJ_DEG = 19937;
constexpr int MTJ_WORDS = (MTJ_DEG + 63) / 64;            // 312 x 64 bits per polynomial
constexpr int MTJ_UNIT = 64;                              // refills per jump unit
constexpr int MTJ_TABLE = 128;                            // lo[0..63] | hi[0..63]
constexpr int MTJ_STREAM_BLOCKS = 33;                     // 33 x 624 = 20 592 >= 19 937 + 623 raw words
constexpr int MTJ_T = 640;                                // threads per workgroup: thread k <-> state word k

// ---------------------------------------------------------------------------------------------------------------
// host: GF(2)[x] arithmetic on bit vectors (bit i of word i / 64 = coefficient of x^i)
// ---------------------------------------------------------------------------------------------------------------
typedef std::vector<uint64_t> Poly;

static inline bool pbit(const Poly &p, int i) { return (p[(size_t)i >> 6] >> (i & 63)) & 1u; }

static inline uint32_t mtj_mix(uint32_t hi, uint32_t lo)
{
    const uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

static void mtj_refill_host(uint32_t *s)
{
    for (int k = 0; k < MTJ_N; ++k) s[k] = s[(k + MTJ_M) % MTJ_N] ^ mtj_mix(s[k], s[(k + 1) % MTJ_N]);
}

// dst ^= src << sh (bits), both `n` words long (bits shifted past the end are dropped)
static void xor_shifted(uint64_t *dst, const uint64_t *src, int n_src, int n_dst, int sh)
{
    const int ws = sh >> 6, bs = sh & 63;
    for (int i = 0; i < n_src && i + ws < n_dst; ++i) {
        dst[i + ws] ^= src[i] << bs;
        if (bs && i + ws + 1 < n_dst) dst[i + ws + 1] ^= src[i] >> (64 - bs);
    }
}

// the characteristic polynomial of the one-word step, by Berlekamp-Massey on bit 0 of the raw word stream
static Poly mtj_char_poly()
{
    const int NBITS = 2 * MTJ_DEG + 64;
    std::vector<uint8_t> seq((size_t)NBITS);
    {
        uint32_t s[MTJ_N];
        s[0] = 5489u;
        for (int i = 1; i < MTJ_N; ++i) s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + (uint32_t)i;
        int k = MTJ_N;
        for (int n = 0; n < NBITS; ++n) {
            if (k == MTJ_N) { mtj_refill_host(s); k = 0; }
            seq[(size_t)n] = (uint8_t)(s[k++] & 1u);
        }
    }
    const int W = (MTJ_DEG + 2 + 63) / 64 + 1;
    Poly C((size_t)W, 0), Bp((size_t)W, 0), T((size_t)W, 0), sr((size_t)W, 0);
    C[0] = 1; Bp[0] = 1;
    int L = 0, m = 1;
    for (int n = 0; n < NBITS; ++n) {
        // sr: bit i = seq[n - i]
        for (int i = W - 1; i > 0; --i) sr[(size_t)i] = (sr[(size_t)i] << 1) | (sr[(size_t)i - 1] >> 63);
        sr[0] = (sr[0] << 1) | seq[(size_t)n];
        uint64_t acc = 0;
        const int lw = (L >> 6) + 1;
        for (int i = 0; i < lw && i < W; ++i) acc ^= C[(size_t)i] & sr[(size_t)i];
        if (!(__builtin_popcountll(acc) & 1)) { ++m; continue; }
        if (2 * L <= n) {
            T = C;
            xor_shifted(C.data(), Bp.data(), W, W, m);
            L = n + 1 - L;
            Bp = T;
            m = 1;
        } else {
            xor_shifted(C.data(), Bp.data(), W, W, m);
            ++m;
        }
    }
    Poly phi((size_t)MTJ_WORDS + 1, 0);
    if (L != MTJ_DEG) return Poly();                      // (cannot happen: MT19937's period polynomial is primitive)
    for (int i = 0; i <= L; ++i)                          // phi(x) = x^L C(1/x)
        if (pbit(C, i)) phi[(size_t)(L - i) >> 6] |= 1ull << ((L - i) & 63);
    return phi;
}

// (a * b) mod phi; a, b of degree < 19937
static Poly mtj_mulmod(const Poly &a, const Poly &b, const Poly &phi)
{
    const int W2 = 2 * MTJ_WORDS + 2;
    Poly r((size_t)W2, 0);
    for (int i = 0; i < MTJ_DEG; ++i)
        if (pbit(a, i)) xor_shifted(r.data(), b.data(), MTJ_WORDS, W2, i);
    for (int k = 2 * MTJ_DEG; k >= MTJ_DEG; --k)
        if (pbit(r, k)) xor_shifted(r.data(), phi.data(), MTJ_WORDS + 1, W2, k - MTJ_DEG);
    r.resize((size_t)MTJ_WORDS);
    r[(size_t)MTJ_WORDS - 1] &= (1ull << (MTJ_DEG & 63)) - 1;
    return r;
}

static Poly mtj_powx(uint64_t e, const Poly &phi)         // x^e mod phi
{
    Poly result((size_t)MTJ_WORDS, 0), base((size_t)MTJ_WORDS, 0);
    result[0] = 1;
    base[0] = 2;                                          // x
    while (e) {
        if (e & 1) result = mtj_mulmod(result, base, phi);
        e >>= 1;
        if (e) base = mtj_mulmod(base, base, phi);
    }
    return result;
}

static std::mutex g_table_mutex;
static std::vector<uint64_t> g_table;                     // [MTJ_TABLE][MTJ_WORDS]

static int mtj_build_table()
{
    const Poly phi = mtj_char_poly();
    if (phi.empty()) return -1;
    std::vector<Poly> t((size_t)MTJ_TABLE);
    Poly one((size_t)MTJ_WORDS, 0);
    one[0] = 1;
    t[0] = one;
    t[1] = mtj_powx((uint64_t)MTJ_UNIT * MTJ_N, phi);     // one unit = 64 refills = 39 936 word steps
    for (int b = 2; b < 64; ++b) t[(size_t)b] = mtj_mulmod(t[(size_t)b - 1], t[1], phi);
    t[64] = one;
    t[65] = mtj_mulmod(t[63], t[1], phi);                 // 64 units
    for (int a = 2; a < 64; ++a) t[(size_t)64 + a] = mtj_mulmod(t[(size_t)64 + a - 1], t[65], phi);
    g_table.assign((size_t)MTJ_TABLE * MTJ_WORDS, 0);
    for (int i = 0; i < MTJ_TABLE; ++i) memcpy(&g_table[(size_t)i * MTJ_WORDS], t[(size_t)i].data(), sizeof(uint64_t) * MTJ_WORDS);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// device
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mtd_mix(uint32_t hi, uint32_t lo)
{
    const uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ uint32_t mtd_temper(uint32_t y)
{
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// one refill of the 624 state words in LDS; thread k <-> word k (640 threads: 16 of them idle)
__device__ __forceinline__ void mtd_refill(uint32_t *s)
{
    const int k = threadIdx.x;
    uint32_t v = 0;
    if (k < 227) v = s[k + MTJ_M] ^ mtd_mix(s[k], s[k + 1]);
    __syncthreads();
    if (k < 227) s[k] = v;
    __syncthreads();
    if (k >= 227 && k < 454) v = s[k - 227] ^ mtd_mix(s[k], s[k + 1]);
    __syncthreads();
    if (k >= 227 && k < 454) s[k] = v;
    __syncthreads();
    if (k >= 454 && k < 623) v = s[k - 227] ^ mtd_mix(s[k], s[k + 1]);
    __syncthreads();
    if (k >= 454 && k < 623) s[k] = v;
    __syncthreads();
    if (k == 0) s[623] = s[396] ^ mtd_mix(s[623], s[0]);
    __syncthreads();
}

// the raw word stream of state `s` (624 words in LDS, consumed): x[0 .. 33 * 624) into `x` (LDS or global)
__device__ __forceinline__ void mtd_stream(uint32_t *s, uint32_t *x)
{
    const int k = threadIdx.x;
    for (int b = 0; b < MTJ_STREAM_BLOCKS; ++b) {
        if (k < MTJ_N) x[b * MTJ_N + k] = s[k];
        if (b + 1 < MTJ_STREAM_BLOCKS) mtd_refill(s);
    }
    __syncthreads();
}

// s[w] = XOR over the set bits i of `poly` of x[i + w]   (x: the stream of the state being jumped, in LDS)
__device__ __forceinline__ void mtd_convolve(const uint32_t *x, const uint64_t *__restrict__ poly, uint32_t *s)
{
    const int w = threadIdx.x < MTJ_N ? threadIdx.x : 0;
    uint32_t acc = 0;
    for (int j = 0; j < MTJ_WORDS; ++j) {
        const uint64_t bits = poly[j];                     // (wave-uniform: a scalar load)
        const uint32_t *xp = x + 64 * j + w;
        // every bit of the word, selected by a mask: 64 independent LDS reads the scheduler can keep in flight
        // (a loop over the SET bits is a chain of dependent iterations: ~4x slower)
#pragma unroll
        for (int i = 0; i < 64; ++i) acc ^= xp[i] & (uint32_t)(0u - (uint32_t)((bits >> i) & 1ull));
    }
    __syncthreads();
    if (threadIdx.x < MTJ_N) s[threadIdx.x] = acc;
    __syncthreads();
}

struct MtPar {
    uint32_t *st;                  // [625] numpy's state + position (updated)
    const uint64_t *table;         // [128][312] jump polynomials (device)
    uint32_t *xbase;               // [33 * 624] raw stream of the state at entry (written by k_mt_par_stream)
    uint32_t *wstate;              // [n_wg][624] every chunk's start state (pass A -> pass B)
    int64_t *cnt;                  // [n_wg + 1] accepted words per chunk (slot n_wg: the current block's unread words)
    int64_t *base;                 // [n_wg + 3] exclusive prefix (scan); [n_wg + 1] = produced so far; [n_wg + 2] = the
                                   // position at entry (pass B must not read st: the chunk that ends the request writes it)
    const int64_t *seg_cum;        // [n_seg + 1] prefix of the requests' counts
    const int64_t *seg_off;        // [n_seg]
    int32_t *out;
    int64_t n_total, n_seg;
    uint32_t top, mask;
    int32_t n_wg, units_per_wg;    // a chunk = units_per_wg x 64 refills
};

// slot of accepted value number n of the whole request: out[seg_off[q] + n - seg_cum[q]], seg_cum[q] <= n < seg_cum[q+1]
__device__ __forceinline__ int64_t mtd_slot(const MtPar &p, int64_t n, int64_t &q)
{
    while (n >= p.seg_cum[q + 1]) ++q;                     // (q only moves forward inside a chunk)
    return p.seg_off[q] + (n - p.seg_cum[q]);
}

__device__ __forceinline__ int64_t mtd_find_seg(const MtPar &p, int64_t n)
{
    int64_t lo = 0, hi = p.n_seg - 1;                      // largest q with seg_cum[q] <= n
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (p.seg_cum[mid] <= n) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ void __launch_bounds__(MTJ_T)
k_mt_par_stream(const MtPar p)
{
    __shared__ uint32_t s[MTJ_N];
    if (threadIdx.x < MTJ_N) s[threadIdx.x] = p.st[threadIdx.x];
    __syncthreads();
    mtd_stream(s, p.xbase);
}

// number of accepted words among s[from .. 624) (uniform result); red: LDS scratch of 16 ints
__device__ __forceinline__ int mtd_count_block(const uint32_t *s, int from, uint32_t top, uint32_t mask, int *red)
{
    const int k = threadIdx.x;
    const bool acc = k >= from && k < MTJ_N && (mtd_temper(s[k]) & mask) <= top;
    const unsigned long long b = __ballot(acc);
    if ((k & 63) == 0) red[k >> 6] = __popcll(b);
    __syncthreads();
    int tot = 0;
#pragma unroll
    for (int w = 0; w < MTJ_T / 64; ++w) tot += red[w];
    __syncthreads();
    return tot;
}

// pass A: jump to the chunk's start state, keep it, count the chunk's accepted words
__global__ void __launch_bounds__(MTJ_T)
k_mt_par_count(const MtPar p)
{
    extern __shared__ uint32_t xs[];                       // [33 * 624] stream of the state being jumped
    __shared__ uint32_t s[MTJ_N];
    __shared__ int red[16];
    const int j = blockIdx.x, k = threadIdx.x;
    const int unit = j * p.units_per_wg;                   // jump distance in units of 64 refills
    const int a = unit >> 6, b = unit & 63;
    if (k < MTJ_N) s[k] = p.st[k];
    __syncthreads();
    if (j == 0) {                                          // the unread rest of the current block belongs to chunk 0
        const int tot0 = mtd_count_block(s, (int)p.st[MTJ_N], p.top, p.mask, red);
        if (k == 0) p.cnt[p.n_wg] = tot0;
    }
    if (a > 0) {
        for (int i = k; i < MTJ_STREAM_BLOCKS * MTJ_N; i += MTJ_T) xs[i] = p.xbase[i];
        __syncthreads();
        mtd_convolve(xs, p.table + (size_t)(64 + a) * MTJ_WORDS, s);
    }
    if (b > 0) {
        mtd_stream(s, xs);                                 // (s is consumed: the convolution rewrites it)
        mtd_convolve(xs, p.table + (size_t)b * MTJ_WORDS, s);
    }
    if (k < MTJ_N) p.wstate[(size_t)j * MTJ_N + k] = s[k];
    int64_t total = 0;
    const int n_blocks = p.units_per_wg * MTJ_UNIT;
    for (int blk = 0; blk < n_blocks; ++blk) {
        mtd_refill(s);
        total += mtd_count_block(s, 0, p.top, p.mask, red);
    }
    if (k == 0) p.cnt[j] = total;
}

__global__ void __launch_bounds__(256)
k_mt_par_scan(const MtPar p)
{
    if (threadIdx.x == 0) {                                // (<= 1 024 chunks: a serial scan is a few microseconds)
        int64_t run = p.cnt[p.n_wg];                       // the current block's rest comes first
        for (int j = 0; j < p.n_wg; ++j) {
            p.base[j] = run;
            run += p.cnt[j];
        }
        p.base[p.n_wg] = run;
        p.base[p.n_wg + 1] = run < p.n_total ? run : p.n_total;
        p.base[p.n_wg + 2] = (int64_t)p.st[MTJ_N];
    }
}

// accepted words of s[from .. 624) -> their slots; n0 = request-wide number of the block's first accepted word.
// Returns the block's accepted count; when value n_total - 1 lies in this block, *cut (LDS) = its word index.
__device__ __forceinline__ int mtd_emit_block(const MtPar &p, const uint32_t *s, int from, int64_t n0, int64_t &q,
                                              int *red, int *cut)
{
    const int k = threadIdx.x, lane = k & 63, wave = k >> 6;
    const uint32_t v = k < MTJ_N ? (mtd_temper(s[k]) & p.mask) : 0u;
    const bool acc = k >= from && k < MTJ_N && v <= p.top;
    const unsigned long long b = __ballot(acc);
    if (lane == 0) red[wave] = __popcll(b);
    __syncthreads();
    int before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < MTJ_T / 64; ++w) {
        if (w < wave) before += red[w];
        tot += red[w];
    }
    const int rank = before + __popcll(b & ((1ull << lane) - 1ull));
    const int64_t n = n0 + rank;
    if (acc && n < p.n_total) {
        p.out[mtd_slot(p, n, q)] = (int32_t)v;
        if (n == p.n_total - 1) *cut = k;
    }
    __syncthreads();
    return tot;
}

// pass B: regenerate the chunk, store its accepted words, leave the stream where the request ends
__global__ void __launch_bounds__(MTJ_T)
k_mt_par_write(const MtPar p)
{
    __shared__ uint32_t s[MTJ_N];
    __shared__ int red[16];
    __shared__ int cut;
    const int j = blockIdx.x, k = threadIdx.x;
    int64_t n0 = p.base[j];
    if (k == 0) cut = -1;
    __syncthreads();
    if (j == 0 && p.cnt[p.n_wg] > 0) {                     // the unread rest of the current block (numbers 0 ..)
        if (k < MTJ_N) s[k] = p.xbase[k];                  // (= the state at entry, kept by k_mt_par_stream)
        __syncthreads();
        int64_t q = 0;
        mtd_emit_block(p, s, (int)p.base[p.n_wg + 2], 0, q, red, &cut);
        if (cut >= 0) {                                    // the request ends inside the current block
            if (k == 0) p.st[MTJ_N] = (uint32_t)cut + 1u;
            return;
        }
    }
    if (n0 >= p.n_total) return;                           // an earlier chunk completes the request
    if (k < MTJ_N) s[k] = p.wstate[(size_t)j * MTJ_N + k];
    __syncthreads();
    int64_t q = mtd_find_seg(p, n0);
    const int n_blocks = p.units_per_wg * MTJ_UNIT;
    for (int blk = 0; blk < n_blocks && n0 < p.n_total; ++blk) {
        mtd_refill(s);
        n0 += mtd_emit_block(p, s, 0, n0, q, red, &cut);
        if (cut >= 0) {                                    // value n_total - 1 was word `cut` of this block
            if (k < MTJ_N) p.st[k] = s[k];
            if (k == 0) p.st[MTJ_N] = (uint32_t)cut + 1u;
            return;
        }
    }
    if (j == p.n_wg - 1 && n0 < p.n_total) {               // short of the request: the finisher continues from here
        if (k < MTJ_N) p.st[k] = s[k];
        if (k == 0) p.st[MTJ_N] = MTJ_N;
    }
}

// whatever the chunks did not cover (their size comes from the EXPECTED acceptance rate), one block at a time
__global__ void __launch_bounds__(MTJ_T)
k_mt_par_finish(const MtPar p)
{
    __shared__ uint32_t s[MTJ_N];
    __shared__ int red[16];
    __shared__ int cut;
    int64_t n0 = p.base[p.n_wg + 1];
    if (n0 >= p.n_total) return;
    const int k = threadIdx.x;
    if (k < MTJ_N) s[k] = p.st[k];
    if (k == 0) cut = -1;
    __syncthreads();
    int64_t q = mtd_find_seg(p, n0);
    for (;;) {
        mtd_refill(s);
        n0 += mtd_emit_block(p, s, 0, n0, q, red, &cut);
        if (cut >= 0) break;
    }
    if (k < MTJ_N) p.st[k] = s[k];
    if (k == 0) p.st[MTJ_N] = (uint32_t)cut + 1u;
}

}  // namespace gsage

using namespace gsage;

extern "C" {

int64_t gsage_mt_jump_table_words(void) { return (int64_t)MTJ_TABLE * MTJ_WORDS; }

int gsage_mt_jump_table(uint64_t *out, int64_t words)
{
    GSAGE_REQUIRE(out && words == (int64_t)MTJ_TABLE * MTJ_WORDS, "mt_jump_table: needs %d x %d words", MTJ_TABLE, MTJ_WORDS);
    std::lock_guard<std::mutex> lock(g_table_mutex);
    if (g_table.empty() && mtj_build_table() != 0) {
        set_error("mt_jump_table: Berlekamp-Massey did not find a degree-19937 polynomial");
        return GSAGE_EINVAL;
    }
    memcpy(out, g_table.data(), sizeof(uint64_t) * (size_t)words);
    return GSAGE_OK;
}

int gsage_mt_jump_host(const uint32_t *state, const uint64_t *poly, uint32_t *out)
{
    GSAGE_REQUIRE(state && poly && out, "mt_jump_host: null pointer");
    std::vector<uint32_t> x((size_t)MTJ_STREAM_BLOCKS * MTJ_N);
    uint32_t s[MTJ_N];
    memcpy(s, state, sizeof(s));
    for (int b = 0; b < MTJ_STREAM_BLOCKS; ++b) {
        memcpy(&x[(size_t)b * MTJ_N], s, sizeof(s));
        mtj_refill_host(s);
    }
    for (int w = 0; w < MTJ_N; ++w) {
        uint32_t acc = 0;
        for (int i = 0; i < MTJ_DEG; ++i)
            if ((poly[i >> 6] >> (i & 63)) & 1ull) acc ^= x[(size_t)i + w];
        out[w] = acc;
    }
    return GSAGE_OK;
}

int64_t gsage_mt_choice_par_scratch(int32_t n_wg)
{
    // xbase | wstate (uint32) | cnt | base (int64): in bytes, 16-byte aligned pieces
    const int64_t a = ((int64_t)MTJ_STREAM_BLOCKS * MTJ_N * 4 + 15) & ~15ll;
    const int64_t b = ((int64_t)n_wg * MTJ_N * 4 + 15) & ~15ll;
    return a + b + 8 * ((int64_t)n_wg + 1) + 8 * ((int64_t)n_wg + 3) + 32;
}

int gsage_mt_choice_par(uint32_t *state, int64_t high, int64_t n_seg, const int64_t *seg_cum, const int64_t *seg_off,
                        int64_t n_total, int32_t *out, const uint64_t *table, void *scratch, int64_t scratch_bytes,
                        int32_t n_wg, int32_t units_per_wg, void *stream)
{
    GSAGE_REQUIRE(state && seg_cum && seg_off && out && table && scratch && n_seg >= 1 && n_total >= 1,
                  "mt_choice_par: null pointer / empty request");
    GSAGE_REQUIRE(high >= 2 && high <= 0x100000000LL, "mt_choice_par: high must be in [2, 2^32]");
    GSAGE_REQUIRE(n_wg >= 1 && n_wg <= 1024 && units_per_wg >= 1 && (int64_t)n_wg * units_per_wg <= 4096,
                  "mt_choice_par: 1..1024 chunks of whole 64-refill units, at most 4096 units in all (the table's reach)");
    GSAGE_REQUIRE(scratch_bytes >= gsage_mt_choice_par_scratch(n_wg) && ((uintptr_t)scratch & 15) == 0,
                  "mt_choice_par: scratch too small or misaligned");
    MtPar p;
    p.st = state; p.table = table;
    char *sc = (char *)scratch;
    p.xbase = (uint32_t *)sc;
    sc += ((int64_t)MTJ_STREAM_BLOCKS * MTJ_N * 4 + 15) & ~15ll;
    p.wstate = (uint32_t *)sc;
    sc += ((int64_t)n_wg * MTJ_N * 4 + 15) & ~15ll;
    p.cnt = (int64_t *)sc;
    sc += 8 * ((int64_t)n_wg + 1);
    p.base = (int64_t *)sc;
    p.seg_cum = seg_cum; p.seg_off = seg_off; p.out = out; p.n_total = n_total; p.n_seg = n_seg;
    p.top = (uint32_t)(high - 1);
    {
        uint32_t m = p.top;
        m |= m >> 1; m |= m >> 2; m |= m >> 4; m |= m >> 8; m |= m >> 16;
        p.mask = m;
    }
    p.n_wg = n_wg; p.units_per_wg = units_per_wg;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = sizeof(uint32_t) * (size_t)MTJ_STREAM_BLOCKS * MTJ_N;
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute((const void *)k_mt_par_count, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            set_error("mt_choice_par: cannot raise the dynamic LDS limit");
            return GSAGE_ELAUNCH;
        }
        raised = true;
    }
    launch(k_mt_par_stream, dim3(1), dim3(MTJ_T), 0, s, p);
    int rc = check_launch("mt_par_stream");
    if (rc != GSAGE_OK) return rc;
    launch(k_mt_par_count, dim3((unsigned)n_wg), dim3(MTJ_T), lds, s, p);
    rc = check_launch("mt_par_count");
    if (rc != GSAGE_OK) return rc;
    launch(k_mt_par_scan, dim3(1), dim3(256), 0, s, p);
    rc = check_launch("mt_par_scan");
    if (rc != GSAGE_OK) return rc;
    launch(k_mt_par_write, dim3((unsigned)n_wg), dim3(MTJ_T), 0, s, p);
    rc = check_launch("mt_par_write");
    if (rc != GSAGE_OK) return rc;
    launch(k_mt_par_finish, dim3(1), dim3(MTJ_T), 0, s, p);
    return check_launch("mt_par_finish");
}

}  // extern "C"
