// gsage_gather.hip -- K2 gather+mean, its backward, and K6 scatter-add (gfx950).
//
// Replaces feats[ids] (reference models.py:76,80: an index_select that MATERIALISES the
// [B*f1*f2, D] frontier, 308 MB fp32 per 512-seed batch at Reddit shapes) followed by
// neibs.view(M,-1,D).mean(1) (nn_modules.py:197-198) that re-reads it.  Here each sampled
// row is read from the HBM-resident table exactly once and only the [M, D] means are written.
//
// HBM-bandwidth bound.  Work item = one 16-byte column chunk of one OUTPUT row: lanes of a wave
// cover consecutive chunks of the same row (coalesced 1 KiB per wave-instruction for D >= 512),
// each lane streams the n neighbour rows of its output row with the loop unrolled so that
// >= 4 independent 16-B loads are in flight per lane, accumulates in fp32 registers and writes
// one 16-B chunk.  No LDS: there is no reuse inside a workgroup to stage (every neighbour row is
// needed by exactly one output row); L2 / Infinity Cache absorb the duplicates that sampling
// with replacement produces.
#include "gsage_common.h"
#include <stdlib.h>
#include "gsage_optim_dev.h"
#include "gsage_sample_dev.h"

namespace gsage {

template <typename T, int VEC>
struct chunk_io;

// bf16 storage: VEC bf16 per chunk
template <int VEC>
struct chunk_io<uint16_t, VEC> {
    static constexpr int kWords = (VEC * 2 + 3) / 4;
    struct __attribute__((aligned(VEC * 2))) raw { uint16_t h[VEC]; };
    __device__ static __forceinline__ void accumulate(const raw &r, float (&acc)[VEC])
    {
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += bf16_to_f32(r.h[e]);
    }
    __device__ static __forceinline__ raw pack(const float (&v)[VEC])
    {
        raw r;
#pragma unroll
        for (int e = 0; e < VEC; ++e) r.h[e] = f32_to_bf16(v[e]);
        return r;
    }
};

template <int VEC>
struct chunk_io<float, VEC> {
    struct __attribute__((aligned(VEC * 4))) raw { float h[VEC]; };
    __device__ static __forceinline__ void accumulate(const raw &r, float (&acc)[VEC])
    {
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += r.h[e];
    }
    __device__ static __forceinline__ raw pack(const float (&v)[VEC])
    {
        raw r;
#pragma unroll
        for (int e = 0; e < VEC; ++e) r.h[e] = v[e];
        return r;
    }
};

// One work item: 16-byte column chunk c0 of output row `row` = mean of n table rows.
// TI = table element type, TO = output element type, VEC elements per chunk (both sides).
// BATCH = rows in flight per lane (8; 16 in k_gather_multi_adam_wide): the summation order j = 0 .. n-1 is the same.
template <typename TI, typename TO, int VEC, int BATCH = 8>
__device__ __forceinline__ void gather_mean_chunk(const TI *__restrict__ table, int64_t ld,
                                                  const int64_t *__restrict__ ids, int64_t row,
                                                  int32_t n, int32_t D, int32_t c0,
                                                  TO *__restrict__ out, int64_t out_ld)
{
    using in_io = chunk_io<TI, VEC>;
    using out_io = chunk_io<TO, VEC>;
    using in_raw = typename in_io::raw;
    using out_raw = typename out_io::raw;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    const int64_t base = row * (int64_t)n;
    if (ids) {
        // Batches of 8 rows, software-pipelined: the ids of batch b+1 are requested BEFORE the rows of batch b
        // (requests return in order: ids asked for behind 8 row requests would wait for those rows), every batch's
        // ids and rows are in flight together, and ONE branch covers a batch's id loads.  Written as
        // `r[u] = ids ? ids[..] : ..` per element, every id load sat in a branch of its own with a vmcnt(0) behind
        // it: 8 dependent round trips per batch before the first row request, and the < 8 remainder went id -> row
        // -> id -> row (n = 10: 13 dependent round trips per work item; now 3).
        // Ids are node ids: < 2^31 (the adjacency's neighbour array is int32), so the low dword of each int64 is
        // read -- 8 instead of 16 registers per batch in flight, which is what lets two batches' ids coexist under
        // the register cap of k_gather_multi_adam.  A short last batch repeats its last row (same lines: cache
        // hits) and drops the repeats at the accumulation: loads stay unconditional, the order j = 0 .. n-1 and
        // with it every bit of the mean is unchanged.
        const int32_t *ids32 = reinterpret_cast<const int32_t *>(ids + base);
        int32_t rc[BATCH], rn[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) rc[u] = ids32[2 * min(u, n - 1)];
        for (int32_t j = 0; j < n; j += BATCH) {
            const int32_t m = min(BATCH, n - j);
            if (j + BATCH < n) {
                const int32_t m2 = min(BATCH, n - j - BATCH);
#pragma unroll
                for (int u = 0; u < BATCH; ++u) rn[u] = ids32[2 * (j + BATCH + min(u, m2 - 1))];
            }
            in_raw v[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) v[u] = *reinterpret_cast<const in_raw *>(table + (int64_t)rc[u] * ld + c0);
            if (m == BATCH) {
#pragma unroll
                for (int u = 0; u < BATCH; ++u) in_io::accumulate(v[u], acc);
            } else {
#pragma unroll
                for (int u = 0; u < BATCH; ++u)
                    if (u < m) in_io::accumulate(v[u], acc);
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) rc[u] = rn[u];
        }
    } else {
        int32_t j = 0;
        for (; j + 8 <= n; j += 8) {
            in_raw v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const in_raw *>(table + (base + j + u) * ld + c0);
#pragma unroll
            for (int u = 0; u < 8; ++u) in_io::accumulate(v[u], acc);
        }
        for (; j + 4 <= n; j += 4) {
            in_raw v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const in_raw *>(table + (base + j + u) * ld + c0);
#pragma unroll
            for (int u = 0; u < 4; ++u) in_io::accumulate(v[u], acc);
        }
        for (; j < n; ++j) {
            const in_raw a = *reinterpret_cast<const in_raw *>(table + (base + j) * ld + c0);
            in_io::accumulate(a, acc);
        }
    }
    const float fn = (float)n;
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = (c0 + e < D) ? acc[e] / fn : 0.f;   // n == 1: exact
    *reinterpret_cast<out_raw *>(out + row * out_ld + c0) = out_io::pack(acc);
}

template <typename TI, typename TO, int VEC>
__global__ void __launch_bounds__(256)
k_gather_mean(const TI *__restrict__ table, int64_t ld, const int64_t *__restrict__ ids, int64_t M,
              int32_t n, int32_t D, int32_t chunks, TO *__restrict__ out, int64_t out_ld)
{
    const int64_t total = M * (int64_t)chunks;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
        const int64_t row = (total <= 0xffffffffLL) ? (int64_t)((uint32_t)t / (uint32_t)chunks)
                                                    : t / chunks;
        const int32_t c0 = (int32_t)(t - row * chunks) * VEC;
        gather_mean_chunk<TI, TO, VEC>(table, ld, ids, row, n, D, c0, out, out_ld);
    }
}

// Several gather+mean problems in ONE launch (all the hops of a level): segment s covers work
// items [first[s], first[s+1]).  Small segments ride along with the big one instead of paying
// their own launch + tail.
struct MultiSeg {
    const void *table[8];
    const int64_t *ids[8];
    void *out[8];
    int64_t M[8];
    int64_t first[9];
    int32_t n[8];
    int32_t rpi[8];         // rows per work item: 4 for n == 1 segments of a mixed launch (a row copy has
                            // ONE load per row, so a lane takes the same chunk of four rows), else 1
    int32_t n_seg;
    int32_t all_single;     // every segment has n == 1 (plain row copies): 4 rows in flight per lane
};

template <typename TI, typename TO, int VEC, int BATCH = 8>
__device__ __forceinline__ void gather_multi_workgroup(const MultiSeg &q, int64_t ld, int32_t D,
                                                       int32_t chunks, int64_t out_ld, int bx, int gx)
{
    const int64_t total = q.first[q.n_seg];
    const int64_t stride = (int64_t)gx * 256;
    if (q.all_single && sizeof(TI) == 2 && sizeof(TO) == 2 && VEC == 8) {
        // n == 1 everywhere: a row copy has ONE load per work item, so a lane takes four work items at
        // a time (ids, then rows, then stores) -- 28 KB in flight per CU held this at ~3.2 TB/s
        for (int64_t t0 = (int64_t)bx * 1024 + threadIdx.x; t0 < total; t0 += stride * 4) {
            int sg[4];
            int64_t row[4], src[4];
            int32_t c0[4];
            bool ok[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int64_t t = t0 + k * 256;
                ok[k] = t < total;
                if (!ok[k]) t = total - 1;
                int s = 0;
#pragma unroll
                for (int j = 1; j < 8; ++j)
                    if (j < q.n_seg && t >= q.first[j]) s = j;
                const int64_t u = t - q.first[s];
                sg[k] = s;
                row[k] = u / chunks;
                c0[k] = (int32_t)(u - row[k] * chunks) * VEC;
                src[k] = row[k];
            }
            {   // the four ids in flight together: one branch around all of them (a null check per load ends in a
                // wait per load), the mixed case (some segments without a row list) keeps the per-element form
                const int64_t *ip[4];
                bool all = true;
#pragma unroll
                for (int k = 0; k < 4; ++k) { ip[k] = q.ids[sg[k]]; all = all && ip[k] != nullptr; }
                if (all) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) src[k] = ip[k][row[k]];
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (ip[k]) src[k] = ip[k][row[k]];
                }
            }
            vec16 raw[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                raw[k] = *reinterpret_cast<const vec16 *>((const uint16_t *)q.table[sg[k]] + src[k] * ld + c0[k]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!ok[k]) continue;
                vec16 v = raw[k];
                if (c0[k] + VEC > D) {                         // last chunk of a row: columns >= D are written as zero
#pragma unroll
                    for (int e = 0; e < VEC; ++e)
                        if (c0[k] + e >= D) v[e >> 1] &= (e & 1) ? 0x0000ffffu : 0xffff0000u;
                }
                *reinterpret_cast<vec16 *>((uint16_t *)q.out[sg[k]] + row[k] * out_ld + c0[k]) = v;
            }
        }
        return;
    }
    for (int64_t t = (int64_t)bx * 256 + threadIdx.x; t < total; t += stride) {
        int s = 0;
#pragma unroll
        for (int j = 1; j < 8; ++j)
            if (j < q.n_seg && t >= q.first[j]) s = j;
        const int64_t u = t - q.first[s];
        const int64_t row = u / chunks;
        const int32_t c0 = (int32_t)(u - row * chunks) * VEC;
        if (q.rpi[s] == 4 && sizeof(TI) == 2 && sizeof(TO) == 2 && VEC == 8) {
            // row copies: rows 4*row .. 4*row+3, same chunk; ids, then rows, then stores
            const int64_t M = q.M[s];
            int64_t src[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) src[k] = min(row * 4 + k, M - 1);
            if (q.ids[s]) {                                    // (one branch around the four id loads)
                const int64_t *ip = q.ids[s];
#pragma unroll
                for (int k = 0; k < 4; ++k) src[k] = ip[src[k]];
            }
            vec16 raw[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                raw[k] = *reinterpret_cast<const vec16 *>((const uint16_t *)q.table[s] + src[k] * ld + c0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (row * 4 + k >= M) continue;
                vec16 v = raw[k];
                if (c0 + VEC > D) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e)
                        if (c0 + e >= D) v[e >> 1] &= (e & 1) ? 0x0000ffffu : 0xffff0000u;
                }
                *reinterpret_cast<vec16 *>((uint16_t *)q.out[s] + (row * 4 + k) * out_ld + c0) = v;
            }
            continue;
        }
        gather_mean_chunk<TI, TO, VEC, BATCH>((const TI *)q.table[s], ld, q.ids[s], row, q.n[s], D, c0,
                                              (TO *)q.out[s], out_ld);
    }
}

template <typename TI, typename TO, int VEC>
__global__ void __launch_bounds__(256)
k_gather_mean_multi(const MultiSeg q, int64_t ld, int32_t D, int32_t chunks, int64_t out_ld)
{
    gather_multi_workgroup<TI, TO, VEC>(q, ld, D, chunks, out_ld, blockIdx.x, gridDim.x);
}

// The level-0 gathers of batch i+1, the clip + Adam update of batch i and the frontier sampling of
// batch i+2 side by side.  The three touch disjoint data: the gather reads features and the ids of
// batch i+1, Adam reads and writes the parameter / gradient / moment buckets and the bf16 operand
// copies, the sampler reads the graph and writes the OTHER frontier buffer.  The two short jobs
// (~10 us and ~6 us of dependent-load latency, a few hundred workgroups) sit in the MIDDLE of the
// grid: the gather's pipeline is in steady state by then and both are done long before it drains.
// Neither side job may advance a counter the sampler reads (batch index, Philox call counter):
// those are ticked by the gradient finalisation that precedes this launch.
// (waves_per_eu: the Adam role's powf/sqrtf would otherwise raise the register count and cost the
// HBM-bound gather role two of its seven waves per SIMD)
template <typename TI, typename TO, int VEC>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 8)))
k_gather_multi_adam(const MultiSeg q, int64_t ld, int32_t D, int32_t chunks, int64_t out_ld,
                    int n_adam, const AdamParams a, int n_smp, const HopsParams h, int first)
{
    extern __shared__ int64_t frontier[];
    __shared__ float red[4];
    const int n_side = n_adam + n_smp;
    const int n_gather = (int)gridDim.x - n_side;
    const int bx = (int)blockIdx.x;
    if (bx >= first && bx < first + n_adam)
        adam_workgroup<false>(a, bx - first, n_adam, red);
    else if (bx >= first + n_adam && bx < first + n_side)
        sample_hops_workgroup<false>(h, bx - first - n_adam, frontier);
    else
        gather_multi_workgroup<TI, TO, VEC>(q, ld, D, chunks, out_ld, bx < first ? bx : bx - n_side, n_gather);
}

// The same launch when its gather role is LIGHT (a few hundred workgroups of means with fan-outs above 8, e.g. config 2's
// hop-1 means once the seed-level launch has taken the whole last hop): the launch is then as long as a lane's chain of
// dependent row trips -- fan-out 25 is four trips of eight rows under the 72-register cap above -- so this variant
// trades occupancy it does not need (3-4 waves per SIMD) for 16 rows in flight per lane: two trips.  Same sums, bit
// for bit (gather_mean_chunk's order does not depend on the batch).
#ifndef GSAGE_WIDE_BATCH
#define GSAGE_WIDE_BATCH 16
#endif
template <typename TI, typename TO, int VEC>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GSAGE_WIDE_BATCH > 16 ? 2 : 3, GSAGE_WIDE_BATCH > 16 ? 2 : 4)))
k_gather_multi_adam_wide(const MultiSeg q, int64_t ld, int32_t D, int32_t chunks, int64_t out_ld,
                         int n_adam, const AdamParams a, int n_smp, const HopsParams h, int first)
{
    extern __shared__ int64_t frontier[];
    __shared__ float red[4];
    const int n_side = n_adam + n_smp;
    const int n_gather = (int)gridDim.x - n_side;
    const int bx = (int)blockIdx.x;
    if (bx >= first && bx < first + n_adam)
        adam_workgroup<false, 4, true>(a, bx - first, n_adam, red);
    else if (bx >= first + n_adam && bx < first + n_side)
        sample_hops_workgroup<false>(h, bx - first - n_adam, frontier);
    else
        gather_multi_workgroup<TI, TO, VEC, GSAGE_WIDE_BATCH>(q, ld, D, chunks, out_ld, bx < first ? bx : bx - n_side, n_gather);
}

// dneibs[i*n+j, :] = dagg[i, :] / n     (fp32, 16-byte chunks when aligned)
template <int VEC>
__global__ void __launch_bounds__(256)
k_segment_mean_bwd(const float *__restrict__ dagg, int64_t ld, int64_t M, int32_t n, int32_t D,
                   int32_t chunks, float *__restrict__ dneibs, int64_t out_ld)
{
    using io = chunk_io<float, VEC>;
    using raw = typename io::raw;
    const int64_t total = M * (int64_t)n * chunks;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const float fn = (float)n;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
        const int64_t orow = t / chunks;
        const int32_t c0 = (int32_t)(t - orow * chunks) * VEC;
        const int64_t i = orow / n;
        float v[VEC];
        if (VEC > 1) {
            const raw r = *reinterpret_cast<const raw *>(dagg + i * ld + c0);
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = r.h[e] / fn;
            *reinterpret_cast<raw *>(dneibs + orow * out_ld + c0) = io::pack(v);
        } else {
            if (c0 < D) dneibs[orow * out_ld + c0] = dagg[i * ld + c0] / fn;
        }
    }
}

// table_grad[ids[r], c] += scale * rows[r / n, c]
__global__ void __launch_bounds__(256)
k_scatter_add_rows(const float *__restrict__ rows, int64_t ld, const int64_t *__restrict__ ids,
                   int64_t total_rows, int32_t n, int32_t D, float scale,
                   float *__restrict__ table_grad, int64_t table_ld)
{
    const int64_t total = total_rows * (int64_t)D;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
        const int64_t r = t / D;
        const int32_t c = (int32_t)(t - r * D);
        const float v = rows[(r / n) * ld + c] * scale;
        atomicAdd(table_grad + ids[r] * table_ld + c, v);      // device-scope fp32 add
    }
}

static inline int grid_for(int64_t work_items)
{
    int64_t blocks = ceil_div(work_items, 256);
    if (blocks > 16384) blocks = 16384;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

static inline bool aligned_to(const void *p, size_t a) { return ((uintptr_t)p % a) == 0; }

template <typename TI, typename TO, int VEC>
static int launch_gather_mean(const void *table, int64_t ld, const int64_t *ids, int64_t M,
                              int32_t n, int64_t D, void *out, int64_t out_ld, hipStream_t s)
{
    const int32_t chunks = (int32_t)ceil_div(D, VEC);
    launch(k_gather_mean<TI, TO, VEC>, dim3(grid_for(M * chunks)), dim3(256), 0, s,
                       (const TI *)table, ld, ids, M, n, (int32_t)D, chunks, (TO *)out, out_ld);
    return check_launch("gather_mean");
}

template <typename TI, typename TO>
static int dispatch_vec(const void *table, int64_t ld, const int64_t *ids, int64_t M, int32_t n,
                        int64_t D, void *out, int64_t out_ld, hipStream_t s)
{
    // widest chunk both sides can take: VEC elements must be a 16/8/4/2-byte aligned unit in
    // both the table and the output, and round_up(D, VEC) must fit in both leading dimensions.
    constexpr int kMax = (sizeof(TI) == 2 && sizeof(TO) == 2) ? 8 : 4;
    auto ok = [&](int vec) {
        return ld % vec == 0 && out_ld % vec == 0 && aligned_to(table, vec * sizeof(TI)) &&
               aligned_to(out, vec * sizeof(TO)) && ceil_div(D, vec) * vec <= ld &&
               ceil_div(D, vec) * vec <= out_ld;
    };
    if (kMax == 8 && ok(8)) return launch_gather_mean<TI, TO, (kMax == 8 ? 8 : 4)>(table, ld, ids, M, n, D, out, out_ld, s);
    if (ok(4)) return launch_gather_mean<TI, TO, 4>(table, ld, ids, M, n, D, out, out_ld, s);
    if (ok(2)) return launch_gather_mean<TI, TO, 2>(table, ld, ids, M, n, D, out, out_ld, s);
    return launch_gather_mean<TI, TO, 1>(table, ld, ids, M, n, D, out, out_ld, s);
}

}  // namespace gsage

using namespace gsage;

extern "C" {

int gsage_gather_mean(const void *table, int dtype, int64_t ld, const int64_t *ids, int64_t M,
                      int32_t n, int64_t D, void *out, int out_dtype, int64_t out_ld, void *stream)
{
    GSAGE_REQUIRE(n > 0 && M >= 0 && D > 0, "gather_mean: bad sizes M=%lld n=%d D=%lld",
                  (long long)M, n, (long long)D);
    GSAGE_REQUIRE(ld >= D && out_ld >= D, "gather_mean: leading dimension smaller than D");
    GSAGE_REQUIRE(D <= 0x7fffffff, "gather_mean: D too large");
    if (M == 0) return GSAGE_OK;
    GSAGE_REQUIRE(table && out, "gather_mean: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GSAGE_BF16 && out_dtype == GSAGE_BF16)
        return dispatch_vec<uint16_t, uint16_t>(table, ld, ids, M, n, D, out, out_ld, s);
    if (dtype == GSAGE_BF16 && out_dtype == GSAGE_F32)
        return dispatch_vec<uint16_t, float>(table, ld, ids, M, n, D, out, out_ld, s);
    if (dtype == GSAGE_F32 && out_dtype == GSAGE_F32)
        return dispatch_vec<float, float>(table, ld, ids, M, n, D, out, out_ld, s);
    if (dtype == GSAGE_F32 && out_dtype == GSAGE_BF16)
        return dispatch_vec<float, uint16_t>(table, ld, ids, M, n, D, out, out_ld, s);
    set_error("gather_mean: unsupported dtype pair %d -> %d", dtype, out_dtype);
    return GSAGE_EINVAL;
}

static int fill_multi(MultiSeg &q, int32_t &chunks, int32_t n_seg, const void *const *tables,
                      const int64_t *const *ids, void *const *outs, const int64_t *M, const int32_t *n,
                      int dtype, int64_t ld, int64_t D, int out_dtype, int64_t out_ld)
{
    GSAGE_REQUIRE(n_seg >= 1 && n_seg <= 8, "gather_mean_multi: 1..8 segments");
    GSAGE_REQUIRE(tables && ids && outs && M && n, "gather_mean_multi: null pointer");
    GSAGE_REQUIRE((dtype == GSAGE_BF16 || dtype == GSAGE_F32) && (out_dtype == dtype || out_dtype == GSAGE_BF16),
                  "gather_mean_multi: bf16 -> bf16, fp32 -> fp32 (the exact-arithmetic parity mode) or fp32 -> bf16 "
                  "(fp32 embedding rows as a bf16 operand)");
    const int vec = dtype == GSAGE_BF16 ? 8 : 4;
    GSAGE_REQUIRE(D > 0 && ld % vec == 0 && out_ld % vec == 0 && ceil_div(D, vec) * vec <= ld &&
                  ceil_div(D, vec) * vec <= out_ld, "gather_mean_multi: needs 16-byte row chunks");
    chunks = (int32_t)ceil_div(D, vec);
    q.n_seg = n_seg;
    q.first[0] = 0;
    for (int s = 0; s < 8; ++s) {
        const bool live = s < n_seg;
        q.table[s] = live ? tables[s] : nullptr;
        q.ids[s] = live ? ids[s] : nullptr;
        q.out[s] = live ? outs[s] : nullptr;
        q.M[s] = live ? M[s] : 0;
        q.n[s] = live ? n[s] : 1;
        if (live) {
            GSAGE_REQUIRE(q.table[s] && q.out[s] && q.M[s] >= 0 && q.n[s] > 0 &&
                          aligned_to(q.table[s], 16) && aligned_to(q.out[s], 16),
                          "gather_mean_multi: bad segment %d", s);
        }
    }
    q.all_single = 1;
    for (int s = 0; s < n_seg; ++s)
        if (q.n[s] != 1) q.all_single = 0;
    for (int s = 0; s < 8; ++s) {
        q.rpi[s] = (dtype == GSAGE_BF16 && !q.all_single && s < n_seg && q.n[s] == 1) ? 4 : 1;
        q.first[s + 1] = q.first[s] + ceil_div(q.M[s], (int64_t)q.rpi[s]) * chunks;
    }
    return GSAGE_OK;
}

int gsage_gather_mean_multi(int32_t n_seg, const void *const *tables, const int64_t *const *ids,
                            void *const *outs, const int64_t *M, const int32_t *n, int dtype,
                            int64_t ld, int64_t D, int out_dtype, int64_t out_ld, void *stream)
{
    MultiSeg q;
    int32_t chunks = 0;
    int rc = fill_multi(q, chunks, n_seg, tables, ids, outs, M, n, dtype, ld, D, out_dtype, out_ld);
    if (rc != GSAGE_OK) return rc;
    if (q.first[n_seg] == 0) return GSAGE_OK;
    if (dtype == GSAGE_F32 && out_dtype == GSAGE_BF16)
        launch(k_gather_mean_multi<float, uint16_t, 4>, dim3(grid_for(q.first[n_seg])),
               dim3(256), 0, (hipStream_t)stream, q, ld, (int32_t)D, chunks, out_ld);
    else if (dtype == GSAGE_F32)
        launch(k_gather_mean_multi<float, float, 4>, dim3(grid_for(q.first[n_seg])),
               dim3(256), 0, (hipStream_t)stream, q, ld, (int32_t)D, chunks, out_ld);
    else
        launch(k_gather_mean_multi<uint16_t, uint16_t, 8>, dim3(grid_for(q.first[n_seg])),
               dim3(256), 0, (hipStream_t)stream, q, ld, (int32_t)D, chunks, out_ld);
    return check_launch("gather_mean_multi");
}

// Workgroups of k_gather_multi_adam that can be RESIDENT at once with `lds_bytes` of dynamic LDS (the sampler role's
// frontier buffers): what bounds the in-launch gradient norm, a meeting of the update's workgroups (every one of
// them polls the slots of all the others: one that is not resident yet would be waited for forever).  The occupancy
// API's answer can be one workgroup per CU too high (MI355X_MICROARCH.md, "Correctness boundaries"), and a workgroup
// of another role may hold a slot when the update's are dispatched: one per CU is kept as margin.
static int gather_adam_capacity(int dtype, size_t lds_bytes)
{
    int dev = 0, per_cu = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    hipError_t e = dtype == GSAGE_F32
        ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_gather_multi_adam<float, float, 4>, 256, lds_bytes)
        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_gather_multi_adam<uint16_t, uint16_t, 8>, 256, lds_bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    if (per_cu > 8) per_cu = 8;
    return (per_cu - 1) * prop.multiProcessorCount;
}

int gsage_gather_adam_capacity(int dtype, int64_t lds_bytes)
{
    GSAGE_REQUIRE((dtype == GSAGE_BF16 || dtype == GSAGE_F32) && lds_bytes >= 0, "gather_adam_capacity: bad arguments");
    return gather_adam_capacity(dtype, (size_t)lds_bytes);
}

int gsage_gather_mean_multi_adam(int32_t n_seg, const void *const *tables, const int64_t *const *ids,
                                 void *const *outs, const int64_t *M, const int32_t *n, int dtype,
                                 int64_t ld, int64_t D, int out_dtype, int64_t out_ld,
                                 const gsage_adam_desc *adam, const gsage_hops_desc *hops, void *stream)
{
    MultiSeg q;
    int32_t chunks = 0;
    int rc = fill_multi(q, chunks, n_seg, tables, ids, outs, M, n, dtype, ld, D, out_dtype, out_ld);
    if (rc != GSAGE_OK) return rc;
    GSAGE_REQUIRE(out_dtype == dtype, "gather_mean_multi_adam: table and output share a type");
    GSAGE_REQUIRE((adam || hops) && q.first[n_seg] > 0,
                  "gather_mean_multi_adam: needs an Adam or a sampler descriptor and a non-empty gather");
    GSAGE_REQUIRE(!adam || adam->step_is_current, "gather_mean_multi_adam: Adam descriptor needs step_is_current");
    AdamParams a = {};
    int n_adam = 0, n_smp = 0;
    if (adam) {
        rc = fill_adam(a, *adam);
        if (rc != GSAGE_OK) return rc;
        n_adam = adam_grid(a.n_prep > 0 ? ceil_div(adam->n, 4) : adam->n, 2048);
        // (the in-launch norm is a meeting of the update's workgroups: all of them must be resident at once)
        GSAGE_REQUIRE(!a.norm_slots || (n_adam <= 1024 && a.n_prep > 0 && (int64_t)n_adam * 1024 >= adam->n),
                      "gather_mean_multi_adam: the in-launch norm needs <= 1024 update workgroups covering the bucket in "
                      "one trip (got %d); pass norm partials instead", n_adam);
        GSAGE_REQUIRE(!hops || (adam->tick1 != (int64_t *)hops->call_ctr && adam->tick2 != (int64_t *)hops->batch_idx) ||
                      (!adam->tick1 && !adam->tick2),
                      "gather_mean_multi_adam: the update may not tick a counter the sampler reads");
    }
    HopsParams h = {};
    size_t lds = 0;
    if (hops) {
        GSAGE_REQUIRE(!hops->dense_adj, "gather_mean_multi_adam: the sampler role walks a CSR; a dense adjacency is sampled "
                                        "by gsage_sample_hops (a launch of its own)");
        rc = fill_hops(h, lds, *hops);
        if (rc != GSAGE_OK) return rc;
        n_smp = (int)ceil_div(hops->B, h.spw);
    }
    if (a.norm_slots) {
        const int cap = gather_adam_capacity(dtype, lds);
        GSAGE_REQUIRE(n_adam <= cap, "gather_mean_multi_adam: the in-launch norm is a meeting of the %d update workgroups, but "
                      "only %d workgroups of this launch (%zu bytes of LDS) are resident at once on this device; pass norm "
                      "partials instead (gsage_gather_adam_capacity)", n_adam, cap, lds);
    }
    // where the side roles sit in the grid (fraction of the gather workgroups dispatched before them).
    // Measured in-step at config 2 (tools/role_sweep.sh): 0.0 -> 33.1 us, 0.5 -> 34.8 us, 1.0 -> 37.2 us:
    // the Adam/sampler workgroups have a long serial latency, so they go first.
    static const double side_pos = [] { const char *e = getenv("GSAGE_SIDE_ROLE_POS"); return e ? atof(e) : 0.0; }();
    const int n_gather = grid_for(q.first[n_seg]);
    int first = (int)(n_gather * side_pos);
    first = first < 0 ? 0 : (first > n_gather ? n_gather : first);
    // a LIGHT gather role whose means have more than eight rows: the variant with 16 rows in flight per lane (the
    // launch is as long as a lane's chain of row trips; GSAGE_GATHER_WIDE=0: never).  Not with the in-launch norm:
    // its capacity check above is the narrow kernel's.
    static const bool wide_ok = [] { const char *e = getenv("GSAGE_GATHER_WIDE"); return !e || atoi(e) != 0; }();
    int32_t max_n = 0;
    for (int sidx = 0; sidx < n_seg; ++sidx)
        if (q.ids[sidx] && q.n[sidx] > max_n) max_n = q.n[sidx];
    const bool wide = wide_ok && dtype == GSAGE_BF16 && !a.norm_slots && max_n > 8 && q.first[n_seg] <= 256LL * 1024;
    if (dtype == GSAGE_F32)
        launch(k_gather_multi_adam<float, float, 4>, dim3(n_gather + n_adam + n_smp),
               dim3(256), lds, (hipStream_t)stream, q, ld, (int32_t)D, chunks, out_ld, n_adam, a, n_smp, h, first);
    else if (wide)
        launch(k_gather_multi_adam_wide<uint16_t, uint16_t, 8>, dim3(n_gather + n_adam + n_smp),
               dim3(256), lds, (hipStream_t)stream, q, ld, (int32_t)D, chunks, out_ld, n_adam, a, n_smp, h, first);
    else
        launch(k_gather_multi_adam<uint16_t, uint16_t, 8>, dim3(n_gather + n_adam + n_smp),
               dim3(256), lds, (hipStream_t)stream, q, ld, (int32_t)D, chunks, out_ld, n_adam, a, n_smp, h, first);
    return check_launch("gather_mean_multi_adam");
}

int gsage_segment_mean_bwd(const float *dagg, int64_t ld, int64_t M, int32_t n, int64_t D,
                           float *dneibs, int64_t out_ld, void *stream)
{
    GSAGE_REQUIRE(n > 0 && M >= 0 && D > 0, "segment_mean_bwd: bad sizes");
    GSAGE_REQUIRE(ld >= D && out_ld >= D, "segment_mean_bwd: leading dimension smaller than D");
    if (M == 0) return GSAGE_OK;
    GSAGE_REQUIRE(dagg && dneibs, "segment_mean_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const bool v4 = D % 4 == 0 && ld % 4 == 0 && out_ld % 4 == 0 && aligned_to(dagg, 16) &&
                    aligned_to(dneibs, 16);
    if (v4) {
        const int32_t chunks = (int32_t)(D / 4);
        launch(k_segment_mean_bwd<4>, dim3(grid_for(M * n * chunks)), dim3(256), 0, s,
                           dagg, ld, M, n, (int32_t)D, chunks, dneibs, out_ld);
    } else {
        launch(k_segment_mean_bwd<1>, dim3(grid_for(M * n * D)), dim3(256), 0, s,
                           dagg, ld, M, n, (int32_t)D, (int32_t)D, dneibs, out_ld);
    }
    return check_launch("segment_mean_bwd");
}

int gsage_scatter_add_rows(const float *rows, int64_t ld, const int64_t *ids, int64_t M, int32_t n,
                           int64_t D, float scale, float *table_grad, int64_t table_ld,
                           void *stream)
{
    GSAGE_REQUIRE(n > 0 && M >= 0 && D > 0, "scatter_add_rows: bad sizes");
    GSAGE_REQUIRE(ld >= D && table_ld >= D, "scatter_add_rows: leading dimension smaller than D");
    if (M == 0) return GSAGE_OK;
    GSAGE_REQUIRE(rows && ids && table_grad, "scatter_add_rows: null pointer");
    const int64_t total_rows = M * (int64_t)n;
    launch(k_scatter_add_rows, dim3(grid_for(total_rows * D)), dim3(256), 0,
                       (hipStream_t)stream, rows, ld, ids, total_rows, n, (int32_t)D, scale,
                       table_grad, table_ld);
    return check_launch("scatter_add_rows");
}

}  // extern "C"
