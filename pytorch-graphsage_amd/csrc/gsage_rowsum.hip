// gsage_rowsum.hip -- deterministic gradient of a trainable embedding table (include/gsage.h, "Deterministic
// gradient of a trainable table").
//
// Reference: nn_modules.py:131-155 under autograd -- `nn.Embedding`'s dense gradient is the sum of the gradient rows
// of every occurrence of a node in the frontier (index_add into a [n_nodes + 1, 64] tensor).  K6
// (gsage_scatter_add_rows) forms that sum with fp32 atomics, i.e. in an order that differs from run to run and, in a
// data-parallel run, from rank to rank: replicas that apply "the same" update would drift apart bit by bit.  Here
// the frontier's ids are sorted once and every run of equal ids is summed IN LIST ORDER by one group of lanes and
// stored -- no atomics, no zero-fill, the same bits on every rank.  HBM-bound integer / byte work: 8 + 4 bytes per
// entry through the sort passes, each gradient row read once, each distinct row written once.
//
// The sort (round 6: the library's own, rocPRIM's radix sort until then) is a stable least-significant-digit radix
// sort over the bits a node id needs, 8 bits per pass (Pokec: 21 bits, three passes), three launches per pass:
//   k_rs_hist     a workgroup counts the digits of its block of 2 048 entries        -> hist[digit][block]
//   k_rs_scan     a workgroup per digit: exclusive prefix sums over the digit's per-block counts + the digit's total
//                 (where a digit starts in the output: the prefix over the 256 totals, formed by every k_rs_scatter
//                 workgroup for itself)
//   k_rs_scatter  a workgroup places its entries: position = start[digit][block] + the number of entries of the same
//                 digit BEFORE it in the block.  The block is walked in eight rounds of 256 (entry = round * 256 +
//                 thread: coalesced), the rank inside a round comes from eight wave ballots (the lanes that share all
//                 eight digit bits) and per-wave counts in LDS -- original order is kept among equal digits, so the
//                 sort is stable and its result depends on the keys alone.
// Every launch goes through launch(): the sort is recorded into command lists like any other kernel.
#include "gsage_common.h"

namespace gsage {

constexpr int RS_BLOCK = 2048;        // entries per workgroup
constexpr int RS_ROUNDS = RS_BLOCK / 256;

// keys of the first pass: a frontier's ids followed by n_tail entries of one spare row; values: positions
struct RsIn {
    const int64_t *ids0;
    int64_t n0, tail_id;
    const int64_t *keys;              // later passes: the previous pass's output (ids0 == NULL)
    const int32_t *vals;
    __device__ __forceinline__ int64_t key(int64_t i) const { return keys ? keys[i] : (i < n0 ? ids0[i] : tail_id); }
    __device__ __forceinline__ int32_t val(int64_t i) const { return vals ? vals[i] : (int32_t)i; }
};

__global__ void __launch_bounds__(256)
k_rs_hist(const RsIn in, int64_t n, int shift, int32_t *__restrict__ hist, int64_t n_blocks)
{
    __shared__ int32_t cnt[256];
    cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_BLOCK;
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const int64_t i = base + r * 256 + threadIdx.x;
        if (i < n) atomicAdd(&cnt[(int)((in.key(i) >> shift) & 255)], 1);
    }
    __syncthreads();
    hist[(int64_t)threadIdx.x * n_blocks + blockIdx.x] = cnt[threadIdx.x];
}

// sum over the workgroup's 256 threads of v, exclusive (red: 4 ints of LDS); *total = the sum
__device__ __forceinline__ int32_t rs_block_exclusive(int32_t v, int32_t *red, int32_t *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    __syncthreads();                                       // (red may still be read from an earlier call)
    if (lane == 63) red[wave] = inc;
    __syncthreads();
    int32_t before = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w)
        if (w < wave) before += red[w];
    *total = (red[0] + red[1]) + (red[2] + red[3]);
    return before + inc - v;
}

// One workgroup per DIGIT: exclusive prefix sums over that digit's per-block counts, in place (hist[digit][block] =
// entries of the digit in earlier blocks), and the digit's total.  Where a digit starts in the output -- the prefix
// over the 256 totals -- is formed by every k_rs_scatter workgroup for itself (256 values).  (Round 6, first version:
// ONE workgroup walked all 256 x blocks counts, a strided piece per thread: 24.8 us per pass at Pokec's 164 k entries.)
__global__ void __launch_bounds__(256)
k_rs_scan(int32_t *__restrict__ hist, int64_t n_blocks, int32_t *__restrict__ totals)
{
    __shared__ int32_t red[4];
    int32_t *h = hist + (int64_t)blockIdx.x * n_blocks;
    int32_t carry = 0;
    for (int64_t b0 = 0; b0 < n_blocks; b0 += 1024) {       // 1 024 blocks per trip, four consecutive ones per thread
        int32_t c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t b = b0 + 4 * threadIdx.x + u;
            c[u] = b < n_blocks ? h[b] : 0;
        }
        int32_t tot;
        int32_t run = carry + rs_block_exclusive((c[0] + c[1]) + (c[2] + c[3]), red, &tot);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t b = b0 + 4 * threadIdx.x + u;
            if (b < n_blocks) h[b] = run;
            run += c[u];
        }
        carry += tot;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

__global__ void __launch_bounds__(256)
k_rs_scatter(const RsIn in, int64_t n, int shift, const int32_t *__restrict__ start, const int32_t *__restrict__ totals,
             int64_t n_blocks, int64_t *__restrict__ keys_out, int32_t *__restrict__ vals_out)
{
    __shared__ int32_t base[256];             // entries of a digit placed so far (earlier rounds) + the block's start
    __shared__ int32_t wave_cnt[4][256];      // this round: entries of a digit in each wave
    __shared__ int32_t red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        // start of digit t in the output (prefix over the 256 digit totals) + its entries in earlier blocks
        int32_t unused;
        const int32_t digit_start = rs_block_exclusive(totals[threadIdx.x], red, &unused);
        base[threadIdx.x] = digit_start + start[(int64_t)threadIdx.x * n_blocks + blockIdx.x];
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) wave_cnt[w][threadIdx.x] = 0;
    __syncthreads();
    const int64_t first = (int64_t)blockIdx.x * RS_BLOCK;
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const int64_t i = first + r * 256 + threadIdx.x;
        const bool live = i < n;
        const int64_t key = live ? in.key(i) : 0;
        const int32_t val = live ? in.val(i) : 0;
        const int d = (int)((key >> shift) & 255);
        // the lanes of this wave with the same digit (dead lanes match nobody)
        uint64_t peers = __ballot(live);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t m = __ballot((d >> b) & 1);
            peers &= ((d >> b) & 1) ? m : ~m;
        }
        const int before = __popcll(peers & ((1ull << lane) - 1));
        if (live && before == 0) wave_cnt[wave][d] = __popcll(peers);        // (the first of its peers)
        __syncthreads();
        if (live) {
            int32_t pos = base[d] + before;
            for (int w = 0; w < wave; ++w) pos += wave_cnt[w][d];
            keys_out[pos] = key;
            vals_out[pos] = val;
        }
        __syncthreads();
        {
            const int t = threadIdx.x;
            base[t] += (wave_cnt[0][t] + wave_cnt[1][t]) + (wave_cnt[2][t] + wave_cnt[3][t]);
#pragma unroll
            for (int w = 0; w < 4; ++w) wave_cnt[w][t] = 0;
        }
        __syncthreads();
    }
}

static inline int64_t rs_align(int64_t b) { return (b + 255) / 256 * 256; }

// One group of lpr lanes (16 for 64-wide rows: E / 4 rounded up to a power of two) per list entry; groups whose entry is
// not the first of its run leave at once.  A run is walked in list order, a WINDOW of up to 16 entries at a time: lane u of
// the group looks at entry j + u (its id and position: one request each, the whole window's in flight together), a ballot
// gives the number of leading entries that still belong to the run, their rows are requested together and added in
// list order -- and the next window's ids / positions are requested before this window's rows, so a long run costs about
// one round trip per 16 entries.  (First version: four entries per trip, each trip an id check and then the rows: the
// 300-entry runs a degree-1 seed produces at Pokec's fan-out 20 / 15 -- every batch has some -- made the launch 65-190 us.)
constexpr int SS_WIN = 16;

__global__ void __launch_bounds__(256)
k_segment_sum_rows(const int64_t *__restrict__ ids, const int32_t *__restrict__ pos, int64_t n,
                   const float *__restrict__ rows0, int64_t ld0, int64_t n0, const float *__restrict__ rows1,
                   int64_t ld1, int32_t lpr, int32_t chunks, float scale, float *__restrict__ table, int64_t ldt)
{
    typedef float v4 __attribute__((ext_vector_type(4)));
    const int64_t gpb = 256 / lpr;                                    // groups per workgroup
    const int64_t g = (int64_t)blockIdx.x * gpb + threadIdx.x / lpr;
    const int sub = threadIdx.x % lpr;
    if (g >= n) return;
    const int lane = threadIdx.x & 63;
    const int gbase = lane - sub;                                     // first lane of this group in its wave
    const int win = lpr < SS_WIN ? lpr : SS_WIN;
    const uint64_t gmask = (win >= 64 ? ~0ull : ((1ull << win) - 1)) << gbase;
    // entry g, its predecessor, and the first window behind it: all requested together
    const int64_t id = ids[g];
    const int64_t prev = g > 0 ? ids[g - 1] : ~id;
    const int64_t p_first = pos[g];
    auto look = [&](int64_t j, int64_t &wid, int32_t &wpos) {         // lane u < win: entry j + u
        const int64_t e = j + sub;
        const bool in = sub < win && e < n;
        wid = in ? ids[e] : ~id;
        wpos = in ? pos[e] : 0;
    };
    int64_t wid;
    int32_t wpos;
    look(g + 1, wid, wpos);
    if (prev == id) return;                                           // (the whole group: not the first of its run)
    auto row = [&](int64_t p) -> const float * { return p < n0 ? rows0 + p * ld0 : rows1 + (p - n0) * ld1; };
    const bool owner = sub < chunks;
    v4 acc = {0.f, 0.f, 0.f, 0.f};
    if (owner) acc = *reinterpret_cast<const v4 *>(row(p_first) + sub * 4);
    int64_t j = g + 1;
    for (;;) {
        // leading entries of the window that still carry the run's id
        const uint64_t same = (__ballot(wid == id) & gmask) >> gbase;
        const int m = same == (win >= 64 ? ~0ull : ((1ull << win) - 1)) ? win : __builtin_ctzll(~same);
        if (m == 0) break;
        const int32_t cur_pos = wpos;
        const bool more = m == win;
        if (more) look(j + win, wid, wpos);                            // (requested before this window's rows)
        v4 t[SS_WIN];
#pragma unroll
        for (int u = 0; u < SS_WIN; ++u) {
            const int32_t pu = __shfl(cur_pos, gbase + (u < win ? u : 0), 64);
            if (u < m && owner) t[u] = *reinterpret_cast<const v4 *>(row(pu) + sub * 4);
        }
#pragma unroll
        for (int u = 0; u < SS_WIN; ++u)
            if (u < m && owner) acc += t[u];
        if (!more) break;
        j += win;
    }
    if (owner) *reinterpret_cast<v4 *>(table + id * ldt + sub * 4) = acc * scale;
}

}  // namespace gsage

using namespace gsage;

extern "C" {

int64_t gsage_sort_rows_temp_bytes(int64_t n, int32_t key_bits)
{
    if (n <= 0 || key_bits <= 0 || key_bits > 63) return -1;
    const int64_t n_blocks = ceil_div(n, RS_BLOCK);
    // a second (key, position) buffer for the passes to alternate with + the digit counts of every block
    return rs_align(n * 8) + rs_align(n * 4) + rs_align(256 * n_blocks * 4) + rs_align(256 * 4);
}

int gsage_sort_rows(const int64_t *ids0, int64_t n0, int64_t tail_id, int64_t n_tail, int32_t key_bits,
                    int64_t *ids_sorted, int32_t *pos_sorted, void *temp, int64_t temp_bytes, void *stream)
{
    const int64_t n = n0 + n_tail;
    GSAGE_REQUIRE(n0 >= 0 && n_tail >= 0 && n > 0 && n < ((int64_t)1 << 31), "sort_rows: bad sizes");
    GSAGE_REQUIRE((ids0 || n0 == 0) && ids_sorted && pos_sorted && temp, "sort_rows: null pointer");
    GSAGE_REQUIRE(key_bits > 0 && key_bits <= 63 && (tail_id >> key_bits) == 0, "sort_rows: ids must fit key_bits");
    GSAGE_REQUIRE(temp_bytes >= gsage_sort_rows_temp_bytes(n, key_bits) && ((uintptr_t)temp & 15) == 0,
                  "sort_rows: temp storage too small or misaligned");
    const int64_t n_blocks = ceil_div(n, RS_BLOCK);
    int64_t *keys_t = (int64_t *)temp;
    int32_t *vals_t = (int32_t *)((char *)temp + rs_align(n * 8));
    int32_t *hist = (int32_t *)((char *)temp + rs_align(n * 8) + rs_align(n * 4));
    int32_t *totals = (int32_t *)((char *)hist + rs_align(256 * n_blocks * 4));
    const int passes = (key_bits + 7) / 8;
    hipStream_t s = (hipStream_t)stream;
    // the passes alternate between the temp pair and the output pair so that the LAST one writes the output
    bool to_out = (passes & 1) != 0;
    RsIn in{ids0, n0, tail_id, nullptr, nullptr};
    for (int p = 0; p < passes; ++p) {
        int64_t *ko = to_out ? ids_sorted : keys_t;
        int32_t *vo = to_out ? pos_sorted : vals_t;
        launch(k_rs_hist, dim3((unsigned)n_blocks), dim3(256), 0, s, in, n, 8 * p, hist, n_blocks);
        launch(k_rs_scan, dim3(256), dim3(256), 0, s, hist, n_blocks, totals);
        launch(k_rs_scatter, dim3((unsigned)n_blocks), dim3(256), 0, s, in, n, 8 * p, (const int32_t *)hist,
               (const int32_t *)totals, n_blocks, ko, vo);
        in = RsIn{nullptr, 0, 0, ko, vo};
        to_out = !to_out;
    }
    return check_launch("sort_rows");
}

int gsage_segment_sum_rows(const int64_t *ids_sorted, const int32_t *pos_sorted, int64_t n, const float *rows0,
                           int64_t ld0, int64_t n0, const float *rows1, int64_t ld1, int32_t E, float scale,
                           float *table, int64_t ldt, void *stream)
{
    GSAGE_REQUIRE(ids_sorted && pos_sorted && table && (rows0 || n0 == 0) && (rows1 || n0 >= n),
                  "segment_sum_rows: null pointer");
    GSAGE_REQUIRE(n > 0 && n0 >= 0 && n0 <= n, "segment_sum_rows: bad sizes");
    GSAGE_REQUIRE(E > 0 && E % 4 == 0 && E <= 256 && ld0 % 4 == 0 && ld1 % 4 == 0 && ldt % 4 == 0 && ldt >= E,
                  "segment_sum_rows: rows of whole 16-byte chunks, E <= 256");
    const int chunks = E / 4;
    int lpr = 1;
    while (lpr < chunks) lpr *= 2;
    const int64_t gpb = 256 / lpr;
    launch(k_segment_sum_rows, dim3((unsigned)ceil_div(n, gpb)), dim3(256), 0, (hipStream_t)stream, ids_sorted,
           pos_sorted, n, rows0, ld0, n0, rows1 ? rows1 : rows0, ld1, (int32_t)lpr, (int32_t)chunks, scale, table, ldt);
    return check_launch("segment_sum_rows");
}

}  // extern "C"
