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

}  // namespace gsage

using namespace gsage;

extern "C" int gsage_attn_aggregate(const float *na, int64_t na_ld, const float *xa, int64_t xa_ld,
                                    const void *table, int dtype, int64_t ld, const int64_t *ids,
                                    int64_t M, int32_t n, int64_t Ha, int64_t D, float *agg,
                                    int64_t agg_ld, float *ws, void *stream)
{
    GSAGE_REQUIRE(n >= 1 && n <= 64, "attn_aggregate: fanout must be in [1, 64]");
    GSAGE_REQUIRE(M >= 0 && Ha > 0 && D > 0, "attn_aggregate: bad sizes");
    GSAGE_REQUIRE(na_ld >= Ha && xa_ld >= Ha && ld >= D && agg_ld >= D,
                  "attn_aggregate: leading dimension too small");
    if (M == 0) return GSAGE_OK;
    GSAGE_REQUIRE(na && xa && table && agg && ws, "attn_aggregate: null pointer");
    dim3 grid((unsigned)ceil_div(M, 4));
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
