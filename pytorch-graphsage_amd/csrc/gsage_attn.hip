// gsage_attn.hip -- K4: attention weighting over the sampled fanout (gfx950).
//
// Replaces AttentionAggregator.forward's middle section (reference nn_modules.py:309-315):
//     scores = bmm(att(neibs).view(M,n,Ha), att(x).view(M,Ha,1)).squeeze()
//     ws     = softmax(scores)            (legacy implicit dim -> over the fanout)
//     agg    = sum_r ws[:, r] * neibs.view(M,n,D)[:, r, :]
// One wavefront per parent row: the Ha-wide dots are wave-shuffle (DPP/xor) reductions, the
// n <= 64 scores live one per lane, softmax is two more wave reductions, and the weighted sum
// streams the n RAW neighbour rows (gathered through ids when given -- feats[ids] is never
// materialised) with lanes across 16-byte column chunks.  HBM-gather bound.
#include "gsage_common.h"

namespace gsage {

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

template <typename T>
__device__ __forceinline__ float load_elem(const T *p);
template <>
__device__ __forceinline__ float load_elem<float>(const float *p) { return *p; }
template <>
__device__ __forceinline__ float load_elem<uint16_t>(const uint16_t *p) { return bf16_to_f32(*p); }

template <typename T>
__global__ void __launch_bounds__(256)
k_attn_aggregate(const float *__restrict__ na, int64_t na_ld, const float *__restrict__ xa,
                 int64_t xa_ld, const T *__restrict__ table, int64_t ld,
                 const int64_t *__restrict__ ids, int64_t M, int32_t n, int32_t Ha, int32_t D,
                 float *__restrict__ agg, int64_t agg_ld, float *__restrict__ ws)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);     // one wave per parent row
    if (i >= M) return;                                                 // wave-uniform exit

    // scores: lane r ends up holding s[r]
    float mine = -INFINITY;
    for (int r = 0; r < n; ++r) {
        float part = 0.f;
        for (int h = lane; h < Ha; h += 64)
            part += na[(i * n + r) * na_ld + h] * xa[i * xa_ld + h];
        const float s = wave_sum(part);
        if (lane == r) mine = s;
    }
    const float mx = wave_max(mine);
    const float e = (lane < n) ? expf(mine - mx) : 0.f;
    const float denom = wave_sum(e);
    const float w = e / denom;
    if (lane < n) ws[i * n + lane] = w;

    // weighted sum of the raw rows; lanes stride over columns.  The trip count is wave-uniform
    // (c0, not c, bounds the loop) so every lane takes part in the weight broadcast.
    for (int c0 = 0; c0 < D; c0 += 64) {
        const int c = c0 + lane;
        const bool live = c < D;
        float acc = 0.f;
        for (int r = 0; r < n; ++r) {
            const float wr = __shfl(w, r, 64);
            const int64_t row = ids ? ids[i * n + r] : i * n + r;
            if (live) acc += wr * load_elem<T>(table + row * ld + c);
        }
        if (live) agg[i * agg_ld + c] = acc;
    }
}

// ---- wide variants: 16-byte lanes, all of a batch of neighbour rows in flight ------------------------
// The kernel above is the general fallback (any D / ld / alignment): one 2-byte or 4-byte column per
// lane and one dependent (id -> row) round trip per neighbour, ~10 % of the HBM roofline at Reddit
// shapes.  Whenever rows are whole 16-byte chunks (FeatureStore tables, hidden activations) the
// kernels below run instead: still one wavefront per parent row (its softmax lives in the lanes), but
// lanes own 16-byte column chunks and up to 8 neighbour rows are requested before any is consumed.
template <typename T, int VEC>
__device__ __forceinline__ void chunk_to_f32(const vec16 &raw, float (&f)[VEC]);
template <>
__device__ __forceinline__ void chunk_to_f32<uint16_t, 8>(const vec16 &raw, float (&f)[8])
{
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint32_t w = raw[e >> 1];
        f[e] = __uint_as_float((e & 1) ? (w & 0xffff0000u) : (w << 16));
    }
}
template <>
__device__ __forceinline__ void chunk_to_f32<float, 4>(const vec16 &raw, float (&f)[4])
{
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = __uint_as_float(raw[e]);
}

// lane r < n: s[r] = <na[i*n + r, :], xa[i, :]>, then w = softmax over the n lanes
__device__ __forceinline__ float attn_weights(const float *__restrict__ na, int64_t na_ld,
                                              const float *__restrict__ xa, int64_t xa_ld, int64_t i, int32_t n,
                                              int32_t Ha, int lane)
{
    float mine = -INFINITY;
    if (lane < n) {
        const float *a = na + (i * n + lane) * na_ld;
        const float *x = xa + i * xa_ld;
        float s = 0.f;
        if ((Ha & 3) == 0 && (na_ld & 3) == 0 && (xa_ld & 3) == 0 && (((uintptr_t)na | (uintptr_t)xa) & 15) == 0) {
            for (int h = 0; h < Ha; h += 4) {
                const float4 av = *reinterpret_cast<const float4 *>(a + h);
                const float4 xv = *reinterpret_cast<const float4 *>(x + h);
                s += av.x * xv.x + av.y * xv.y + av.z * xv.z + av.w * xv.w;
            }
        } else {
            for (int h = 0; h < Ha; ++h) s += a[h] * x[h];
        }
        mine = s;
    }
    const float mx = wave_max(mine);
    const float e = (lane < n) ? expf(mine - mx) : 0.f;
    return e / wave_sum(e);
}

template <typename T, int VEC>
__global__ void __launch_bounds__(256)
k_attn_aggregate_wide(const float *__restrict__ na, int64_t na_ld, const float *__restrict__ xa,
                      int64_t xa_ld, const T *__restrict__ table, int64_t ld,
                      const int64_t *__restrict__ ids, int64_t M, int32_t n, int32_t Ha, int32_t D,
                      float *__restrict__ agg, int64_t agg_ld, float *__restrict__ ws)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= M) return;                                                 // wave-uniform exit
    const float w = attn_weights(na, na_ld, xa, xa_ld, i, n, Ha, lane);
    if (lane < n) ws[i * n + lane] = w;
    const int chunks = (D + VEC - 1) / VEC;
    for (int c0 = 0; c0 < chunks; c0 += 64) {                            // wave-uniform trip count
        const int c = c0 + lane < chunks ? c0 + lane : chunks - 1;       // clamped: loads stay unconditional
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
        for (int j0 = 0; j0 < n; j0 += 8) {
            int64_t row[8];
            vec16 raw[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u < n ? j0 + u : n - 1;
                row[u] = ids ? ids[i * n + j] : i * n + j;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) raw[u] = *reinterpret_cast<const vec16 *>(table + row[u] * ld + c * VEC);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float wj = j0 + u < n ? __shfl(w, j0 + u < n ? j0 + u : 0, 64) : 0.f;
                float f[VEC];
                chunk_to_f32<T, VEC>(raw[u], f);
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] += wj * f[e];
            }
        }
        if (c0 + lane < chunks) {
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                if (c * VEC + e < D) agg[i * agg_ld + c * VEC + e] = acc[e];
        }
    }
}

// Backward of the weighting in one launch (autograd of nn_modules.py:309-315 w.r.t. att(neibs) and
// att(x)): dws[j] = <row_j, g_i>, ds = softmax backward, dxa[i,:] = sum_j ds[j] na[i,j,:],
// dna[i,j,:] = ds[j] xa[i,:].  n <= NMAX (per-neighbour partial dots live in registers).
template <typename T, int VEC, int NMAX>
__global__ void __launch_bounds__(256)
k_attn_bwd_wide(const float *__restrict__ g, int64_t g_ld, const float *__restrict__ ws,
                const float *__restrict__ na, int64_t na_ld, const float *__restrict__ xa, int64_t xa_ld,
                const T *__restrict__ table, int64_t ld, const int64_t *__restrict__ ids, int64_t M,
                int32_t n, int32_t Ha, int32_t D, float *__restrict__ dna, int64_t dna_ld,
                float *__restrict__ dxa, int64_t dxa_ld)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= M) return;
    float p[NMAX];
#pragma unroll
    for (int j = 0; j < NMAX; ++j) p[j] = 0.f;
    const int chunks = (D + VEC - 1) / VEC;
    for (int c0 = 0; c0 < chunks; c0 += 64) {
        const bool live = c0 + lane < chunks;
        const int c = live ? c0 + lane : chunks - 1;
        float gv[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) gv[e] = (live && c * VEC + e < D) ? g[i * g_ld + c * VEC + e] : 0.f;
#pragma unroll
        for (int j0 = 0; j0 < NMAX; j0 += 8) {
            if (j0 < n) {                                               // wave-uniform
                int64_t row[8];
                vec16 raw[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = j0 + u < n ? j0 + u : n - 1;
                    row[u] = ids ? ids[i * n + j] : i * n + j;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) raw[u] = *reinterpret_cast<const vec16 *>(table + row[u] * ld + c * VEC);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    float f[VEC], d = 0.f;
                    chunk_to_f32<T, VEC>(raw[u], f);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) d += f[e] * gv[e];
                    p[j0 + u] += d;                                     // (rows >= n repeat row n-1: never read)
                }
            }
        }
    }
    // lane j keeps dws[j]; softmax backward
    float dws = 0.f;
#pragma unroll
    for (int j = 0; j < NMAX; ++j)
        if (j < n) {
            const float t = wave_sum(p[j]);
            if (lane == j) dws = t;
        }
    const float w = lane < n ? ws[i * n + lane] : 0.f;
    const float dot = wave_sum(dws * w);
    const float ds = w * (dws - dot);
    for (int h = lane; h < Ha; h += 64) {
        const float xh = xa[i * xa_ld + h];
        float acc = 0.f;
        for (int j = 0; j < n; ++j) {
            const float dsj = __shfl(ds, j, 64);
            acc += dsj * na[(i * n + j) * na_ld + h];
            dna[(i * n + j) * dna_ld + h] = dsj * xh;
        }
        dxa[i * dxa_ld + h] = acc;
    }
}

__device__ __forceinline__ void store_as(uint16_t *p, float v) { *p = f32_to_bf16(v); }
__device__ __forceinline__ void store_as(float *p, float v) { *p = v; }
__device__ __forceinline__ float load_as(const uint16_t *p) { return bf16_to_f32(*p); }
__device__ __forceinline__ float load_as(const float *p) { return *p; }

// ---- grouped variants: lanes in groups of `lpc`, one child row per group ---------------------------------
// The wide kernels give every lane one 16-byte chunk of EVERY child row: with 76 chunks (Reddit, bf16) the second
// pass has 12 of 64 lanes busy, with 8 chunks (64-d embeddings) one lane in eight ever works, and the unrolled
// batches of 8 rows load padding rows (n = 10: 16 loads for 10 rows).  Here a group of lpc lanes (8 / 16 / 32:
// the power of two that covers the row in <= TMAX strided chunks per lane) takes ONE child at a time and the
// 64 / lpc groups take different children; two children per group are in flight.  Forward: per-lane partial sums
// over the group's children, added across groups at the end; backward: per-child dot products reduced inside the
// group, parked in LDS for the softmax lanes.
// NF = children per group in flight (per trip).  The ids of a trip are loaded first (all in flight, ONE branch on
// `ids`), then every row request of the trip is issued before anything waits, and the FIRST trip's requests go out
// before the attention weights are computed (the rows do not depend on them).  Written as "for each child: id ->
// row chunks" the compiler put a vmcnt(0) in front of every child's rows: a dependent round trip per child
// (59.8 -> 54.3 us at Reddit's last hop once hoisted).  Sums run in the same order for every NF: bit-identical.
// WPP = waves per parent.  1: a workgroup's four waves take four parents.  4: they share ONE parent (its children go
// round the 4 * 64 / lpc groups of the workgroup, partial sums meet in LDS in wave order) -- for hops with few parents
// and long fan-outs, where a parent's chain of dependent trips is all a launch consists of (Reddit's first hop: 512
// parents x 25 children on 512 waves, seven trips each: 13 us forward / 24 us backward for 15 MB of rows).
template <typename T, int VEC, int TMAX, int NF, int WPP>
__global__ void __launch_bounds__(256)
k_attn_aggregate_grp(const float *__restrict__ na, int64_t na_ld, const float *__restrict__ xa, int64_t xa_ld,
                     const T *__restrict__ table, int64_t ld, const int64_t *__restrict__ ids, int64_t M, int32_t n,
                     int32_t Ha, int32_t D, float *__restrict__ agg, int64_t agg_ld, float *__restrict__ ws, int32_t lpc,
                     T *__restrict__ agg_lp, int64_t lp_ld)
{
    __shared__ float part_s[WPP > 1 ? (WPP - 1) * 32 * TMAX * VEC : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = WPP == 1 ? (int64_t)blockIdx.x * 4 + wave : (int64_t)blockIdx.x;
    if (i >= M) return;                                                 // wave-uniform (WPP > 1: block-uniform) exit
    const int sub = lane & (lpc - 1);
    const int G = (64 / lpc) * WPP;                                     // groups working on this parent
    const int grp = lane / lpc + (WPP == 1 ? 0 : wave * (64 / lpc));
    const int chunks = (D + VEC - 1) / VEC;
    constexpr int Tn = TMAX;                                            // == ceil(chunks / lpc) (host picks TMAX)
    int cc[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        const int c = sub + lpc * t;
        cc[t] = c < chunks ? c : chunks - 1;                            // clamped: loads stay unconditional
    }
    vec16 raw[NF][TMAX];
    // (the ids of a trip first, all in flight together, then every row request: written as one loop, each id ->
    //  row pair was a dependent round trip of its own -- the compiler waits for vmcnt(0) in front of each)
    auto load_trip = [&](int j0) {
        int64_t row[NF];
#pragma unroll
        for (int u = 0; u < NF; ++u) {
            const int j = j0 + u * G + grp;
            row[u] = i * n + (j < n ? j : n - 1);
        }
        if (ids) {                                           // (ONE branch around all NF loads: a branch per load
#pragma unroll                                               //  ends in a wait per load)
            for (int u = 0; u < NF; ++u) row[u] = ids[row[u]];
        }
#pragma unroll
        for (int u = 0; u < NF; ++u)
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
                raw[u][t] = *reinterpret_cast<const vec16 *>(table + row[u] * ld + cc[t] * VEC);
    };
    load_trip(0);
    const float w = attn_weights(na, na_ld, xa, xa_ld, i, n, Ha, lane);
    if (lane < n && (WPP == 1 || wave == 0)) ws[i * n + lane] = w;
    float acc[TMAX][VEC];
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[t][e] = 0.f;
    for (int j0 = 0; j0 < n; j0 += NF * G) {                            // wave-uniform trip count
        if (j0 > 0) load_trip(j0);
#pragma unroll
        for (int u = 0; u < NF; ++u) {
            const int j = j0 + u * G + grp;
            const int jj = j < n ? j : n - 1;
            const float wv = __shfl(w, jj, 64);
            const float wj = j < n ? wv : 0.f;
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
                if (t < Tn) {
                    float f[VEC];
                    chunk_to_f32<T, VEC>(raw[u][t], f);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[t][e] += wj * f[e];
                }
        }
    }
    for (int off = lpc; off < 64; off <<= 1)                            // add the wave's groups' partial sums
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (t < Tn)
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[t][e] += __shfl_xor(acc[t][e], off, 64);
    if (WPP > 1) {                                                      // ... and the other waves', in wave order
        if (wave > 0 && lane < lpc)
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
#pragma unroll
                for (int e = 0; e < VEC; ++e) part_s[((wave - 1) * 32 + sub) * (TMAX * VEC) + t * VEC + e] = acc[t][e];
        __syncthreads();
        if (wave > 0) return;
        if (lane < lpc)
            for (int q = 0; q < WPP - 1; ++q)
#pragma unroll
                for (int t = 0; t < TMAX; ++t)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[t][e] += part_s[(q * 32 + sub) * (TMAX * VEC) + t * VEC + e];
    }
    if (lane < lpc) {
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            const int c = sub + lpc * t;
            if (t < Tn && c < chunks) {
                if (agg) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e)
                        if (c * VEC + e < D) agg[i * agg_ld + c * VEC + e] = acc[t][e];
                }
                if (agg_lp) {            // the operand copy the fc_neib projection and its weight gradient read
#pragma unroll
                    for (int e = 0; e < VEC; ++e)
                        if (c * VEC + e < lp_ld) store_as(agg_lp + i * lp_ld + c * VEC + e, c * VEC + e < D ? acc[t][e] : 0.f);
                }
            }
        }
    }
}

template <typename T, int VEC, int TMAX, int NF, int WPP>
__global__ void __launch_bounds__(256)
k_attn_bwd_grp(const float *__restrict__ g, int64_t g_ld, const float *__restrict__ ws, const float *__restrict__ na,
               int64_t na_ld, const float *__restrict__ xa, int64_t xa_ld, const T *__restrict__ table, int64_t ld,
               const int64_t *__restrict__ ids, int64_t M, int32_t n, int32_t Ha, int32_t D, float *__restrict__ dna,
               int64_t dna_ld, float *__restrict__ dxa, int64_t dxa_ld, int32_t lpc)
{
    __shared__ float dws_s[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = WPP == 1 ? (int64_t)blockIdx.x * 4 + wave : (int64_t)blockIdx.x;
    if (i >= M) return;
    const int sub = lane & (lpc - 1);
    const int G = (64 / lpc) * WPP;                                     // groups working on this parent
    const int grp = lane / lpc + (WPP == 1 ? 0 : wave * (64 / lpc));
    const int chunks = (D + VEC - 1) / VEC;
    constexpr int Tn = TMAX;                                            // == ceil(chunks / lpc) (host picks TMAX)
    int cc[TMAX];
    float gv[TMAX][VEC];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        const int c = sub + lpc * t;
        const bool live = t < Tn && c < chunks;
        cc[t] = c < chunks ? c : chunks - 1;
#pragma unroll
        for (int e = 0; e < VEC; ++e) gv[t][e] = (live && c * VEC + e < D) ? g[i * g_ld + c * VEC + e] : 0.f;
    }
    volatile float *mine = dws_s[WPP == 1 ? wave : 0];
    for (int j0 = 0; j0 < n; j0 += NF * G) {
        vec16 raw[NF][TMAX];
        int jv[NF];
        int64_t row[NF];
#pragma unroll
        for (int u = 0; u < NF; ++u) {                       // ids first, all in flight (see the forward kernel)
            const int j = j0 + u * G + grp;
            jv[u] = j;
            row[u] = i * n + (j < n ? j : n - 1);
        }
        if (ids) {
#pragma unroll
            for (int u = 0; u < NF; ++u) row[u] = ids[row[u]];
        }
#pragma unroll
        for (int u = 0; u < NF; ++u)
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
                raw[u][t] = *reinterpret_cast<const vec16 *>(table + row[u] * ld + cc[t] * VEC);
#pragma unroll
        for (int u = 0; u < NF; ++u) {
            float d = 0.f;
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
                if (t < Tn) {
                    float f[VEC];
                    chunk_to_f32<T, VEC>(raw[u][t], f);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) d += f[e] * gv[t][e];       // (clamped chunks meet gv = 0)
                }
            for (int off = lpc >> 1; off > 0; off >>= 1) d += __shfl_xor(d, off, 64);
            if (sub == 0 && jv[u] < n) mine[jv[u]] = d;
        }
    }
    if (WPP == 1) __builtin_amdgcn_wave_barrier();
    else {
        __syncthreads();
        if (wave > 0) return;
    }
    // lane j keeps dws[j]; softmax backward
    const float dws = lane < n ? mine[lane] : 0.f;
    const float w = lane < n ? ws[i * n + lane] : 0.f;
    const float dot = wave_sum(dws * w);
    const float ds = w * (dws - dot);
    for (int h = lane; h < Ha; h += 64) {
        const float xh = xa[i * xa_ld + h];
        float acc = 0.f;
        for (int j = 0; j < n; ++j) {
            const float dsj = __shfl(ds, j, 64);
            acc += dsj * na[(i * n + j) * na_ld + h];
            dna[(i * n + j) * dna_ld + h] = dsj * xh;
        }
        dxa[i * dxa_ld + h] = acc;
    }
}

// ---- second layer of the att MLP (32 -> 32, nn_modules.py:292-296), forward and backward --------------------
// A [M x 32] x [32 x 32] product is 145 MFLOP at Reddit's frontier: as a GEMM launch it cost 25 us (and its
// backward needed a cast, a GEMM and a tanh-backward launch, 50 us); here lane j of a half-wave owns output
// column j (its weight column in registers), a row's 32 inputs reach every lane through LDS, and the element-wise
// neighbours of the product are fused in.  The roundings are the engine's: operands in T, fp32 sums.
constexpr int MLP2_RT = 16;       // rows per wave and trip
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// 16 rows x 32 elements of T (ld elements apart) -> LDS as floats; 16-byte lane loads
template <typename T>
__device__ __forceinline__ void mlp2_stage_rows(const T *__restrict__ src, int64_t ld, int64_t m0, int64_t M, int lane,
                                                float (*dst)[32])
{
    constexpr int EPC = 16 / (int)sizeof(T);                 // 8 (bf16) / 4 (fp32) elements per chunk
    constexpr int CPR = 32 / EPC;                            // chunks per row
    for (int q = lane; q < MLP2_RT * CPR; q += 64) {
        const int r = q / CPR, c = q - r * CPR;
        vec16 raw = {0u, 0u, 0u, 0u};
        if (m0 + r < M) raw = *reinterpret_cast<const vec16 *>(src + (m0 + r) * ld + c * EPC);
        float f[EPC];
        chunk_to_f32<T, EPC>(raw, f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) dst[r][c * EPC + e] = f[e];
    }
}

// 16 rows x 32 floats in LDS -> global rows of T; 16-byte lane stores
template <typename T>
__device__ __forceinline__ void mlp2_store_rows(T *__restrict__ dstp, int64_t ld, int64_t m0, int64_t M, int lane,
                                                const float (*srcs)[32])
{
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int CPR = 32 / EPC;
    for (int q = lane; q < MLP2_RT * CPR; q += 64) {
        const int r = q / CPR, c = q - r * CPR;
        if (m0 + r >= M) continue;
        T out[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) store_as(&out[e], srcs[r][c * EPC + e]);
        *reinterpret_cast<vec16 *>(dstp + (m0 + r) * ld + c * EPC) = *reinterpret_cast<const vec16 *>(out);
    }
}

// lane (half, j): out[r][j] = sum_k x[r][k] w[k] for the rows r = half, half + 2, ... of the staged tile
#define MLP2_DOT(xs, r, acc)                                                                            \
    do {                                                                                                \
        _Pragma("unroll") for (int k4 = 0; k4 < 8; ++k4) {                                              \
            const float4 v = *reinterpret_cast<const float4 *>(&(xs)[r][4 * k4]);                       \
            acc += v.x * w[4 * k4] + v.y * w[4 * k4 + 1] + v.z * w[4 * k4 + 2] + v.w * w[4 * k4 + 3];   \
        }                                                                                               \
    } while (0)

template <typename T>
__global__ void __launch_bounds__(256)
k_attn_mlp2_fwd(const T *__restrict__ hid, int64_t ldh, const T *__restrict__ W2, int64_t ldw, float *__restrict__ a,
                int64_t lda, int64_t M)
{
    __shared__ __attribute__((aligned(16))) float xs[4][MLP2_RT][32], os[4][MLP2_RT][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, half = lane >> 5;
    float w[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) w[k] = load_as(W2 + j * ldw + k);          // a[m, j] = sum_k hid[m, k] W2[j, k]
    const int64_t n_waves = (int64_t)gridDim.x * 4, wv = (int64_t)blockIdx.x * 4 + wave;
    for (int64_t m0 = wv * MLP2_RT; m0 < M; m0 += n_waves * MLP2_RT) {
        mlp2_stage_rows<T>(hid, ldh, m0, M, lane, xs[wave]);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < MLP2_RT; r += 2) {
            float acc = 0.f;
            MLP2_DOT(xs[wave], r + half, acc);
            os[wave][r + half][j] = acc;
        }
        __builtin_amdgcn_wave_barrier();
        mlp2_store_rows<float>(a, lda, m0, M, lane, os[wave]);
        __builtin_amdgcn_wave_barrier();
    }
}

// da = T(dan + dax);   dhid = T( (da W2) * (1 - hid^2) )
template <typename T>
__global__ void __launch_bounds__(256)
k_attn_mlp2_bwd(const float *__restrict__ dan, int64_t ldn, const float *__restrict__ dax, int64_t ldx,
                const T *__restrict__ hid, int64_t ldh, const T *__restrict__ W2T, int64_t ldw, T *__restrict__ da,
                int64_t ldda, T *__restrict__ dhid, int64_t lddh, int64_t M)
{
    __shared__ __attribute__((aligned(16))) float xs[4][MLP2_RT][32], hs[4][MLP2_RT][32], os[4][MLP2_RT][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, k = lane & 31, half = lane >> 5;
    float w[32];
#pragma unroll
    for (int jj = 0; jj < 32; ++jj) w[jj] = load_as(W2T + k * ldw + jj);    // dhg[m, k] = sum_j da[m, j] W2[j, k]
    const int64_t n_waves = (int64_t)gridDim.x * 4, wv = (int64_t)blockIdx.x * 4 + wave;
    for (int64_t m0 = wv * MLP2_RT; m0 < M; m0 += n_waves * MLP2_RT) {
        // d a = dan + dax, rounded to T as the next GEMMs see it: 16 rows x 8 chunks of four floats
        for (int q = lane; q < MLP2_RT * 8; q += 64) {
            const int r = q >> 3, c = q & 7;
            f32x4_t s4 = {0.f, 0.f, 0.f, 0.f};
            if (m0 + r < M) {
                const f32x4_t u = *reinterpret_cast<const f32x4_t *>(dan + (m0 + r) * ldn + 4 * c);
                const f32x4_t v = *reinterpret_cast<const f32x4_t *>(dax + (m0 + r) * ldx + 4 * c);
                s4 = u + v;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                T rounded;
                store_as(&rounded, s4[e]);
                xs[wave][r][4 * c + e] = load_as(&rounded);
            }
        }
        mlp2_stage_rows<T>(hid, ldh, m0, M, lane, hs[wave]);
        __builtin_amdgcn_wave_barrier();
        mlp2_store_rows<T>(da, ldda, m0, M, lane, xs[wave]);                 // (exact: the values are T already)
#pragma unroll
        for (int r = 0; r < MLP2_RT; r += 2) {
            float acc = 0.f;
            MLP2_DOT(xs[wave], r + half, acc);
            const float h = hs[wave][r + half][k];
            os[wave][r + half][k] = acc * (1.f - h * h);
        }
        __builtin_amdgcn_wave_barrier();
        mlp2_store_rows<T>(dhid, lddh, m0, M, lane, os[wave]);
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- glue of the native attention train step (engine.FusedAttnTrainStep) ------------------------------
// Element-wise kernels between K4 / K5 / K5b: what autograd ran as a cast, an add, a tanh backward and
// four scatter / expand kernels per level (52 casts and 34 adds per Pokec-shaped step).

// dst[m, c] = T(a[m, c] (+ b[m, c]))
template <typename T>
__global__ void __launch_bounds__(256)
k_add_cast(const float *__restrict__ a, int64_t lda, const float *__restrict__ b, int64_t ldb, T *__restrict__ dst,
           int64_t ldd, int64_t M, int32_t D)
{
    const int64_t total = M * D, stride = (int64_t)gridDim.x * 256;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
        const int64_t m = t / D;
        const int c = (int)(t - m * D);
        float v = a[m * lda + c];
        if (b) v += b[m * ldb + c];
        store_as(dst + m * ldd + c, v);
    }
}

// out[m, c] = T(g[m, c] * (1 - hid[m, c]^2))      (backward of the att MLP's tanh, nn_modules.py:294)
template <typename T>
__global__ void __launch_bounds__(256)
k_tanh_bwd(const float *__restrict__ g, int64_t ldg, const T *__restrict__ hid, int64_t ldh, T *__restrict__ out,
           int64_t ldo, int64_t M, int32_t D)
{
    const int64_t total = M * D, stride = (int64_t)gridDim.x * 256;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
        const int64_t m = t / D;
        const int c = (int)(t - m * D);
        const float h = load_as(hid + m * ldh + c);
        store_as(out + m * ldo + c, g[m * ldg + c] * (1.f - h * h));
    }
}

// Input gradient of an attention level, all of its sources in one pass (autograd of nn_modules.py:307-317
// w.r.t. x and neibs): rows are the hops concatenated (hop k starts at off[k], fan[k] children per parent)
//   dIn[m] = mask(m) * ( DATT[m]                                   through att(.) -- every row
//                      + (m < r_x ? DX[m] : 0)                     through fc_x -- rows that were "x"
//                      + (hop(m) >= 1 ? ws[m - off[1]] * DAGG[parent(m)] : 0) )   the weighted sum of raw rows
//   mask(m) = H ? (H[m, c] > 0) : 1     (ReLU of the level below; none for a prep output)
struct AttnMergeParams {
    const void *H;
    const float *DATT, *DX, *DAGG, *ws;
    void *out;
    uint16_t *out2;              // optional second copy of the result, bf16 (the operand the next GEMMs read)
    int64_t ldo2;
    int64_t ldh, ldatt, ldx, ldagg, ldo;
    int64_t R, r_x;
    int32_t D, n_hops;
    int64_t off[6];
    int32_t fan[6];
};

template <typename TH, typename TO>
__global__ void __launch_bounds__(256)
k_attn_merge_bwd(const AttnMergeParams q)
{
    const int64_t total = q.R * q.D, stride = (int64_t)gridDim.x * 256;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
        const int64_t m = t / q.D;
        const int c = (int)(t - m * q.D);
        float v = q.DATT ? q.DATT[m * q.ldatt + c] : 0.f;
        if (m < q.r_x && q.DX) v += q.DX[m * q.ldx + c];
        int k = 0;
#pragma unroll
        for (int j = 1; j < 6; ++j)
            if (j < q.n_hops && m >= q.off[j]) k = j;
        if (k >= 1) {
            const int64_t parent = q.off[k - 1] + (m - q.off[k]) / q.fan[k];
            v += (q.ws ? q.ws[m - q.off[1]] : 1.f / (float)q.fan[k]) * q.DAGG[parent * q.ldagg + c];
        }
        if (q.H && !(load_as((const TH *)q.H + m * q.ldh + c) > 0.f)) v = 0.f;
        store_as((TO *)q.out + m * q.ldo + c, v);
        if (q.out2) store_as(q.out2 + m * q.ldo2 + c, v);
    }
}

// four columns per thread (16-byte loads of the fp32 inputs, 16 / 8-byte stores): the element-per-thread kernel above
// moved 4 + 2 bytes per lane and instruction and took 39 us for Pokec's 164 k x 64 level-0 gradient
template <typename TH, typename TO>
__global__ void __launch_bounds__(256)
k_attn_merge_bwd_v4(const AttnMergeParams q)
{
    const int D4 = q.D >> 2;
    const int64_t total = q.R * D4, stride = (int64_t)gridDim.x * 256;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
        const int64_t m = t / D4;
        const int c = (int)(t - m * D4) * 4;
        f32x4_t v = {0.f, 0.f, 0.f, 0.f};
        if (q.DATT) v = *reinterpret_cast<const f32x4_t *>(q.DATT + m * q.ldatt + c);
        if (m < q.r_x && q.DX) v += *reinterpret_cast<const f32x4_t *>(q.DX + m * q.ldx + c);
        int k = 0;
#pragma unroll
        for (int j = 1; j < 6; ++j)
            if (j < q.n_hops && m >= q.off[j]) k = j;
        if (k >= 1) {
            const int64_t parent = q.off[k - 1] + (m - q.off[k]) / q.fan[k];
            v += (q.ws ? q.ws[m - q.off[1]] : 1.f / (float)q.fan[k]) *
                 *reinterpret_cast<const f32x4_t *>(q.DAGG + parent * q.ldagg + c);
        }
        if (q.H) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (!(load_as((const TH *)q.H + m * q.ldh + c + e) > 0.f)) v[e] = 0.f;
        }
        TO o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) store_as(&o[e], v[e]);
        typedef uint32_t out_vec __attribute__((ext_vector_type(sizeof(TO))));      // 8 or 16 bytes
        *reinterpret_cast<out_vec *>((TO *)q.out + m * q.ldo + c) = *reinterpret_cast<const out_vec *>(o);
        if (q.out2) {
            uint16_t o2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o2[e] = f32_to_bf16(v[e]);
            *reinterpret_cast<uint2 *>(q.out2 + m * q.ldo2 + c) = *reinterpret_cast<const uint2 *>(o2);
        }
    }
}

static inline int ew_grid(int64_t items)
{
    int64_t b = ceil_div(items, 256);
    return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace gsage

using namespace gsage;

extern "C" int gsage_add_cast(const float *a, int64_t lda, const float *b, int64_t ldb, void *dst, int dst_dtype,
                              int64_t ldd, int64_t M, int64_t D, void *stream)
{
    GSAGE_REQUIRE(a && dst && M >= 0 && D > 0 && lda >= D && ldd >= D && (!b || ldb >= D), "add_cast: bad arguments");
    GSAGE_REQUIRE(dst_dtype == GSAGE_BF16 || dst_dtype == GSAGE_F32, "add_cast: bad dtype");
    if (M == 0) return GSAGE_OK;
    if (dst_dtype == GSAGE_BF16)
        launch(k_add_cast<uint16_t>, dim3(ew_grid(M * D)), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb,
               (uint16_t *)dst, ldd, M, (int32_t)D);
    else
        launch(k_add_cast<float>, dim3(ew_grid(M * D)), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb,
               (float *)dst, ldd, M, (int32_t)D);
    return check_launch("add_cast");
}

extern "C" int gsage_tanh_bwd(const float *g, int64_t ldg, const void *hid, int dtype, int64_t ldh, void *out,
                              int64_t ldo, int64_t M, int64_t D, void *stream)
{
    GSAGE_REQUIRE(g && hid && out && M >= 0 && D > 0 && ldg >= D && ldh >= D && ldo >= D, "tanh_bwd: bad arguments");
    GSAGE_REQUIRE(dtype == GSAGE_BF16 || dtype == GSAGE_F32, "tanh_bwd: bad dtype");
    if (M == 0) return GSAGE_OK;
    if (dtype == GSAGE_BF16)
        launch(k_tanh_bwd<uint16_t>, dim3(ew_grid(M * D)), dim3(256), 0, (hipStream_t)stream, g, ldg,
               (const uint16_t *)hid, ldh, (uint16_t *)out, ldo, M, (int32_t)D);
    else
        launch(k_tanh_bwd<float>, dim3(ew_grid(M * D)), dim3(256), 0, (hipStream_t)stream, g, ldg, (const float *)hid,
               ldh, (float *)out, ldo, M, (int32_t)D);
    return check_launch("tanh_bwd");
}

extern "C" int gsage_attn_merge_bwd2(const void *H, int h_dtype, int64_t ldh, const float *DATT, int64_t ldatt,
                                     const float *DX, int64_t ldx, int64_t r_x, const float *DAGG, int64_t ldagg,
                                     const float *ws, void *out, int out_dtype, int64_t ldo, int64_t R, int32_t D,
                                     int32_t n_hops, const int64_t *off, const int32_t *fan, void *out2_bf16,
                                     int64_t ldo2, void *stream);

extern "C" int gsage_attn_merge_bwd(const void *H, int h_dtype, int64_t ldh, const float *DATT, int64_t ldatt,
                                    const float *DX, int64_t ldx, int64_t r_x, const float *DAGG, int64_t ldagg,
                                    const float *ws, void *out, int out_dtype, int64_t ldo, int64_t R, int32_t D,
                                    int32_t n_hops, const int64_t *off, const int32_t *fan, void *stream)
{
    return gsage_attn_merge_bwd2(H, h_dtype, ldh, DATT, ldatt, DX, ldx, r_x, DAGG, ldagg, ws, out, out_dtype, ldo, R, D,
                                 n_hops, off, fan, nullptr, 0, stream);
}

extern "C" int gsage_attn_merge_bwd2(const void *H, int h_dtype, int64_t ldh, const float *DATT, int64_t ldatt,
                                     const float *DX, int64_t ldx, int64_t r_x, const float *DAGG, int64_t ldagg,
                                     const float *ws, void *out, int out_dtype, int64_t ldo, int64_t R, int32_t D,
                                     int32_t n_hops, const int64_t *off, const int32_t *fan, void *out2_bf16,
                                     int64_t ldo2, void *stream)
{
    GSAGE_REQUIRE(!out2_bf16 || ldo2 >= D, "attn_merge_bwd: leading dimension too small");
    GSAGE_REQUIRE(DAGG && out && off && fan, "attn_merge_bwd: null pointer");
    GSAGE_REQUIRE(n_hops >= 2 && n_hops <= 6 && R >= 0 && r_x >= 0 && r_x <= R && D > 0, "attn_merge_bwd: bad sizes");
    GSAGE_REQUIRE((out_dtype == GSAGE_BF16 || out_dtype == GSAGE_F32) && (!H || h_dtype == GSAGE_BF16 || h_dtype == GSAGE_F32),
                  "attn_merge_bwd: bad dtype");
    if (R == 0) return GSAGE_OK;
    AttnMergeParams q;
    q.H = H; q.DATT = DATT; q.DX = DX; q.DAGG = DAGG; q.ws = ws; q.out = out; q.out2 = (uint16_t *)out2_bf16; q.ldo2 = ldo2;
    q.ldh = ldh; q.ldatt = ldatt; q.ldx = ldx; q.ldagg = ldagg; q.ldo = ldo; q.R = R; q.r_x = r_x; q.D = D;
    q.n_hops = n_hops;
    for (int i = 0; i < 6; ++i) { q.off[i] = i < n_hops ? off[i] : 0; q.fan[i] = i < n_hops ? fan[i] : 1; }
    hipStream_t s = (hipStream_t)stream;
    const bool hb = H && h_dtype == GSAGE_BF16;
    const bool v4 = D % 4 == 0 && ldatt % 4 == 0 && ldagg % 4 == 0 && ldo % 4 == 0 && (!DX || ldx % 4 == 0) &&
                    (!out2_bf16 || ldo2 % 4 == 0) &&
                    ((((uintptr_t)DATT | (uintptr_t)DAGG | (uintptr_t)DX | (uintptr_t)out) & 15) == 0) &&
                    ((uintptr_t)out2_bf16 & 7) == 0;
    if (v4) {
        const dim3 grid4(ew_grid(R * (D / 4)));
        if (out_dtype == GSAGE_BF16) {
            if (hb || !H) launch(k_attn_merge_bwd_v4<uint16_t, uint16_t>, grid4, dim3(256), 0, s, q);
            else launch(k_attn_merge_bwd_v4<float, uint16_t>, grid4, dim3(256), 0, s, q);
        } else {
            if (hb) launch(k_attn_merge_bwd_v4<uint16_t, float>, grid4, dim3(256), 0, s, q);
            else launch(k_attn_merge_bwd_v4<float, float>, grid4, dim3(256), 0, s, q);
        }
        return check_launch("attn_merge_bwd");
    }
    const dim3 grid(ew_grid(R * D));
    if (out_dtype == GSAGE_BF16) {
        if (hb || !H) launch(k_attn_merge_bwd<uint16_t, uint16_t>, grid, dim3(256), 0, s, q);
        else launch(k_attn_merge_bwd<float, uint16_t>, grid, dim3(256), 0, s, q);
    } else {
        if (hb) launch(k_attn_merge_bwd<uint16_t, float>, grid, dim3(256), 0, s, q);
        else launch(k_attn_merge_bwd<float, float>, grid, dim3(256), 0, s, q);
    }
    return check_launch("attn_merge_bwd");
}


// lanes per child row of the grouped K4 kernels (0: the row is too wide for TMAX chunks per lane -> wide kernels)
constexpr int ATTN_TMAX = 3;
static int attn_group_lanes(int64_t D, int vec)
{
    const int64_t chunks = ceil_div(D, (int64_t)vec);
    if (chunks > 32 * ATTN_TMAX) return 0;
    return chunks <= 8 ? 8 : chunks <= 16 ? 16 : 32;
}

// children per group in flight (NF of the grouped kernels).  Measured at Reddit's last hop (n = 10, two groups of
// 32 lanes, 154 MB): NF = 2 (80 VGPRs, six waves per SIMD) 54 us forward / 61 us backward, NF = 5 (one trip per
// parent, 136 VGPRs, three waves per SIMD) 65 / 63 us -- the launch is bound by how many requests a CU keeps in
// flight across ALL its waves, not by a parent's chain of round trips, so occupancy wins.  GSAGE_ATTN_NF=5 selects
// the deep variant (kept for fan-outs where a parent's chain is long and rows are short).
static int attn_in_flight(int32_t n, int lpc)
{
    static const int force = [] { const char *e = getenv("GSAGE_ATTN_NF"); return e ? atoi(e) : 0; }();
    (void)n; (void)lpc;
    return force == 5 ? 5 : 2;
}

// waves per parent: one parent per workgroup when a launch has few parents whose children need several trips
static int attn_waves_per_parent(int64_t M, int32_t n, int lpc, int nf)
{
    static const int force = [] { const char *e = getenv("GSAGE_ATTN_WPP"); return e ? atoi(e) : 0; }();
    if (force == 1 || force == 4) return nf == 2 ? force : 1;
    return (nf == 2 && M <= 4096 && n > nf * (64 / lpc)) ? 4 : 1;
}

// the grouped kernels are instantiated on the exact number of strided chunks per lane (1..3), on NF and on WPP
template <typename T, int VEC, typename... A>
static void launch_attn_fwd(int tn, int nf, int wpp, int64_t M, hipStream_t s, A... a)
{
    const dim3 grid((unsigned)(wpp == 1 ? ceil_div(M, 4) : M));
#define GSAGE_ATTN_CASE(TN, NF, WPP) \
    if (tn == TN && nf == NF && wpp == WPP) { launch(k_attn_aggregate_grp<T, VEC, TN, NF, WPP>, grid, dim3(256), 0, s, a...); return; }
    GSAGE_ATTN_CASE(1, 2, 1) GSAGE_ATTN_CASE(2, 2, 1) GSAGE_ATTN_CASE(3, 2, 1)
    GSAGE_ATTN_CASE(1, 2, 4) GSAGE_ATTN_CASE(2, 2, 4) GSAGE_ATTN_CASE(3, 2, 4)
    GSAGE_ATTN_CASE(1, 5, 1) GSAGE_ATTN_CASE(2, 5, 1) GSAGE_ATTN_CASE(3, 5, 1)
#undef GSAGE_ATTN_CASE
}

template <typename T, int VEC, typename... A>
static void launch_attn_bwd(int tn, int nf, int wpp, int64_t M, hipStream_t s, A... a)
{
    const dim3 grid((unsigned)(wpp == 1 ? ceil_div(M, 4) : M));
#define GSAGE_ATTN_CASE(TN, NF, WPP) \
    if (tn == TN && nf == NF && wpp == WPP) { launch(k_attn_bwd_grp<T, VEC, TN, NF, WPP>, grid, dim3(256), 0, s, a...); return; }
    GSAGE_ATTN_CASE(1, 2, 1) GSAGE_ATTN_CASE(2, 2, 1) GSAGE_ATTN_CASE(3, 2, 1)
    GSAGE_ATTN_CASE(1, 2, 4) GSAGE_ATTN_CASE(2, 2, 4) GSAGE_ATTN_CASE(3, 2, 4)
    GSAGE_ATTN_CASE(1, 5, 1) GSAGE_ATTN_CASE(2, 5, 1) GSAGE_ATTN_CASE(3, 5, 1)
#undef GSAGE_ATTN_CASE
}

static int attn_chunks_per_lane(int64_t D, int vec, int lpc) { return (int)ceil_div(ceil_div(D, (int64_t)vec), (int64_t)lpc); }

template <typename T, int VEC>
static bool attn_wide_ok(const void *table, int64_t ld, int64_t D)
{
    return ld % VEC == 0 && ((uintptr_t)table % 16) == 0 && ceil_div(D, VEC) * VEC <= ld;
}

extern "C" int gsage_attn_bwd(const float *g, int64_t g_ld, const float *ws, const float *na, int64_t na_ld,
                              const float *xa, int64_t xa_ld, const void *table, int dtype, int64_t ld,
                              const int64_t *ids, int64_t M, int32_t n, int64_t Ha, int64_t D, float *dna,
                              int64_t dna_ld, float *dxa, int64_t dxa_ld, void *stream)
{
    GSAGE_REQUIRE(n >= 1 && n <= 32, "attn_bwd: fanout must be in [1, 32]");
    GSAGE_REQUIRE(M >= 0 && Ha > 0 && D > 0, "attn_bwd: bad sizes");
    GSAGE_REQUIRE(g_ld >= D && na_ld >= Ha && xa_ld >= Ha && dna_ld >= Ha && dxa_ld >= Ha && ld >= D,
                  "attn_bwd: leading dimension too small");
    if (M == 0) return GSAGE_OK;
    GSAGE_REQUIRE(g && ws && na && xa && table && dna && dxa, "attn_bwd: null pointer");
    dim3 grid((unsigned)ceil_div(M, 4));
    hipStream_t s = (hipStream_t)stream;
    const int lpc_b = attn_group_lanes(D, 8), lpc_f = attn_group_lanes(D, 4);
    if (dtype == GSAGE_BF16 && attn_wide_ok<uint16_t, 8>(table, ld, D) && lpc_b)
        launch_attn_bwd<uint16_t, 8>(attn_chunks_per_lane(D, 8, lpc_b), attn_in_flight(n, lpc_b),
                                     attn_waves_per_parent(M, n, lpc_b, attn_in_flight(n, lpc_b)), M, s, g, g_ld, ws, na,
                                     na_ld, xa, xa_ld, (const uint16_t *)table, ld, ids, M, n, (int32_t)Ha, (int32_t)D, dna,
                                     dna_ld, dxa, dxa_ld, (int32_t)lpc_b);
    else if (dtype == GSAGE_F32 && attn_wide_ok<float, 4>(table, ld, D) && lpc_f)
        launch_attn_bwd<float, 4>(attn_chunks_per_lane(D, 4, lpc_f), attn_in_flight(n, lpc_f),
                                  attn_waves_per_parent(M, n, lpc_f, attn_in_flight(n, lpc_f)), M, s, g, g_ld, ws, na, na_ld,
                                  xa, xa_ld, (const float *)table, ld, ids, M, n, (int32_t)Ha, (int32_t)D, dna, dna_ld, dxa,
                                  dxa_ld, (int32_t)lpc_f);
    else if (dtype == GSAGE_BF16 && attn_wide_ok<uint16_t, 8>(table, ld, D))
        launch(k_attn_bwd_wide<uint16_t, 8, 32>, grid, dim3(256), 0, s, g, g_ld, ws, na, na_ld, xa, xa_ld,
               (const uint16_t *)table, ld, ids, M, n, (int32_t)Ha, (int32_t)D, dna, dna_ld, dxa, dxa_ld);
    else if (dtype == GSAGE_F32 && attn_wide_ok<float, 4>(table, ld, D))
        launch(k_attn_bwd_wide<float, 4, 32>, grid, dim3(256), 0, s, g, g_ld, ws, na, na_ld, xa, xa_ld,
               (const float *)table, ld, ids, M, n, (int32_t)Ha, (int32_t)D, dna, dna_ld, dxa, dxa_ld);
    else {
        set_error("attn_bwd: rows must be whole 16-byte chunks of bf16 / fp32 (ld %% %d == 0, aligned)",
                  dtype == GSAGE_BF16 ? 8 : 4);
        return GSAGE_EINVAL;
    }
    return check_launch("attn_bwd");
}

extern "C" int gsage_attn_aggregate_lp(const float *na, int64_t na_ld, const float *xa, int64_t xa_ld, const void *table,
                                       int dtype, int64_t ld, const int64_t *ids, int64_t M, int32_t n, int64_t Ha,
                                       int64_t D, float *agg, int64_t agg_ld, float *ws, void *agg_lp, int64_t agg_lp_ld,
                                       void *stream);

extern "C" int gsage_attn_aggregate(const float *na, int64_t na_ld, const float *xa, int64_t xa_ld,
                                    const void *table, int dtype, int64_t ld, const int64_t *ids,
                                    int64_t M, int32_t n, int64_t Ha, int64_t D, float *agg,
                                    int64_t agg_ld, float *ws, void *stream)
{
    return gsage_attn_aggregate_lp(na, na_ld, xa, xa_ld, table, dtype, ld, ids, M, n, Ha, D, agg, agg_ld, ws, nullptr, 0,
                                   stream);
}

extern "C" int gsage_attn_mlp2_fwd(const void *hid, int dtype, int64_t ldh, const void *W2, int64_t ldw, float *a,
                                   int64_t lda, int64_t M, int32_t Ha, void *stream)
{
    GSAGE_REQUIRE(hid && W2 && a && M >= 0 && Ha == 32 && ldh >= 32 && ldw >= 32 && lda >= 32,
                  "attn_mlp2_fwd: needs the reference's 32-wide att MLP");
    GSAGE_REQUIRE(dtype == GSAGE_BF16 || dtype == GSAGE_F32, "attn_mlp2_fwd: bad dtype");
    if (M == 0) return GSAGE_OK;
    const dim3 grid(ew_grid(ceil_div(M, (int64_t)MLP2_RT) * 64));
    if (dtype == GSAGE_BF16)
        launch(k_attn_mlp2_fwd<uint16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t *)hid, ldh,
               (const uint16_t *)W2, ldw, a, lda, M);
    else
        launch(k_attn_mlp2_fwd<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float *)hid, ldh, (const float *)W2,
               ldw, a, lda, M);
    return check_launch("attn_mlp2_fwd");
}

extern "C" int gsage_attn_mlp2_bwd(const float *dan, int64_t ldn, const float *dax, int64_t ldx, const void *hid, int dtype,
                                   int64_t ldh, const void *W2T, int64_t ldw, void *da, int64_t ldda, void *dhid,
                                   int64_t lddh, int64_t M, int32_t Ha, void *stream)
{
    GSAGE_REQUIRE(dan && dax && hid && W2T && da && dhid && M >= 0 && Ha == 32 && ldn >= 32 && ldx >= 32 && ldh >= 32 &&
                  ldw >= 32 && ldda >= 32 && lddh >= 32, "attn_mlp2_bwd: needs the reference's 32-wide att MLP");
    GSAGE_REQUIRE(dtype == GSAGE_BF16 || dtype == GSAGE_F32, "attn_mlp2_bwd: bad dtype");
    if (M == 0) return GSAGE_OK;
    const dim3 grid(ew_grid(ceil_div(M, (int64_t)MLP2_RT) * 64));
    if (dtype == GSAGE_BF16)
        launch(k_attn_mlp2_bwd<uint16_t>, grid, dim3(256), 0, (hipStream_t)stream, dan, ldn, dax, ldx, (const uint16_t *)hid,
               ldh, (const uint16_t *)W2T, ldw, (uint16_t *)da, ldda, (uint16_t *)dhid, lddh, M);
    else
        launch(k_attn_mlp2_bwd<float>, grid, dim3(256), 0, (hipStream_t)stream, dan, ldn, dax, ldx, (const float *)hid, ldh,
               (const float *)W2T, ldw, (float *)da, ldda, (float *)dhid, lddh, M);
    return check_launch("attn_mlp2_bwd");
}

extern "C" int gsage_attn_aggregate_lp(const float *na, int64_t na_ld, const float *xa, int64_t xa_ld, const void *table,
                                       int dtype, int64_t ld, const int64_t *ids, int64_t M, int32_t n, int64_t Ha,
                                       int64_t D, float *agg, int64_t agg_ld, float *ws, void *agg_lp, int64_t agg_lp_ld,
                                       void *stream)
{
    GSAGE_REQUIRE(agg || agg_lp, "attn_aggregate: no output");
    GSAGE_REQUIRE(!agg_lp || agg_lp_ld >= D, "attn_aggregate: leading dimension too small");
    if (!agg) agg_ld = D;
    GSAGE_REQUIRE(n >= 1 && n <= 64, "attn_aggregate: fanout must be in [1, 64]");
    GSAGE_REQUIRE(M >= 0 && Ha > 0 && D > 0, "attn_aggregate: bad sizes");
    GSAGE_REQUIRE(na_ld >= Ha && xa_ld >= Ha && ld >= D && agg_ld >= D,
                  "attn_aggregate: leading dimension too small");
    if (M == 0) return GSAGE_OK;
    GSAGE_REQUIRE(na && xa && table && ws, "attn_aggregate: null pointer");
    dim3 grid((unsigned)ceil_div(M, 4));
    const int lpc_b = attn_group_lanes(D, 8), lpc_f = attn_group_lanes(D, 4);
    if (dtype == GSAGE_BF16 && attn_wide_ok<uint16_t, 8>(table, ld, D) && lpc_b && n <= 64) {
        launch_attn_fwd<uint16_t, 8>(attn_chunks_per_lane(D, 8, lpc_b), attn_in_flight(n, lpc_b),
                                     attn_waves_per_parent(M, n, lpc_b, attn_in_flight(n, lpc_b)), M, (hipStream_t)stream,
                                     na, na_ld, xa, xa_ld, (const uint16_t *)table, ld, ids, M, n, (int32_t)Ha, (int32_t)D,
                                     agg, agg_ld, ws, (int32_t)lpc_b, (uint16_t *)agg_lp, agg_lp_ld);
        return check_launch("attn_aggregate");
    }
    if (dtype == GSAGE_F32 && attn_wide_ok<float, 4>(table, ld, D) && lpc_f && n <= 64) {
        launch_attn_fwd<float, 4>(attn_chunks_per_lane(D, 4, lpc_f), attn_in_flight(n, lpc_f),
                                  attn_waves_per_parent(M, n, lpc_f, attn_in_flight(n, lpc_f)), M, (hipStream_t)stream, na,
                                  na_ld, xa, xa_ld, (const float *)table, ld, ids, M, n, (int32_t)Ha, (int32_t)D, agg, agg_ld,
                                  ws, (int32_t)lpc_f, (float *)agg_lp, agg_lp_ld);
        return check_launch("attn_aggregate");
    }
    // the kernels below write fp32 only: the copy in the table's type follows as a cast launch
    GSAGE_REQUIRE(agg, "attn_aggregate: rows this wide need the fp32 output (the low-precision copy is derived from it)");
    if (agg_lp) {
        int rc = gsage_attn_aggregate_lp(na, na_ld, xa, xa_ld, table, dtype, ld, ids, M, n, Ha, D, agg, agg_ld, ws, nullptr,
                                         0, stream);
        if (rc != GSAGE_OK) return rc;
        return gsage_add_cast(agg, agg_ld, nullptr, 0, agg_lp, dtype, agg_lp_ld, M, (int32_t)D, stream);
    }
    if (dtype == GSAGE_BF16 && attn_wide_ok<uint16_t, 8>(table, ld, D)) {
        launch(k_attn_aggregate_wide<uint16_t, 8>, grid, dim3(256), 0, (hipStream_t)stream, na, na_ld, xa, xa_ld,
               (const uint16_t *)table, ld, ids, M, n, (int32_t)Ha, (int32_t)D, agg, agg_ld, ws);
        return check_launch("attn_aggregate");
    }
    if (dtype == GSAGE_F32 && attn_wide_ok<float, 4>(table, ld, D)) {
        launch(k_attn_aggregate_wide<float, 4>, grid, dim3(256), 0, (hipStream_t)stream, na, na_ld, xa, xa_ld,
               (const float *)table, ld, ids, M, n, (int32_t)Ha, (int32_t)D, agg, agg_ld, ws);
        return check_launch("attn_aggregate");
    }
    if (dtype == GSAGE_F32)
        launch(k_attn_aggregate<float>, grid, dim3(256), 0, (hipStream_t)stream, na,
                           na_ld, xa, xa_ld, (const float *)table, ld, ids, M, n, (int32_t)Ha,
                           (int32_t)D, agg, agg_ld, ws);
    else if (dtype == GSAGE_BF16)
        launch(k_attn_aggregate<uint16_t>, grid, dim3(256), 0, (hipStream_t)stream,
                           na, na_ld, xa, xa_ld, (const uint16_t *)table, ld, ids, M, n,
                           (int32_t)Ha, (int32_t)D, agg, agg_ld, ws);
    else {
        set_error("attn_aggregate: bad dtype %d", dtype);
        return GSAGE_EINVAL;
    }
    return check_launch("attn_aggregate");
}
