// gsage_gather_dev.h -- the "gather role": workgroups of a latency- or occupancy-limited launch that spend their
// time on part of the NEXT batch's level-0 gather (rows [0, rows) of one gather-mean segment of fan-out N), which
// the gather launch then skips.  Played by the spare workgroups of the seed-level launch (gsage_tail.hip: B / 4
// workgroups of dependent phases leave half the chip idle) and of the level-0 projection (gsage_packed.hip: 416
// workgroups where 768 fit).  Gathering is bound by what a CU can keep in flight towards HBM (~17 GB/s per CU on
// 1.2 KB rows whatever the kernel), so the time of CUs that wait on something else is the resource.
// Sums run in neighbour order like gather_mean_chunk (gsage_gather.hip): bit-identical means.
#pragma once
#include "gsage_common.h"

namespace gsage {

// element e (0..7) of a 16-byte vector of bf16
__device__ __forceinline__ float tail_elem(const vec16 v, int e)
{
    const uint32_t w = v[e >> 1];
    return __uint_as_float((e & 1) ? (w & 0xffff0000u) : (w << 16));
}

struct TailGather {
    const uint16_t *table;
    const int64_t *ids;
    uint16_t *out;
    int64_t ld, out_ld;
    int32_t rows, D, chunks, n_wg;       // n_wg = 0: no gather role in this launch
};

// bx: index of this workgroup among the n_wg that play the role; U work items (16-byte chunks of an output row) per
// lane and trip = U * N row requests in flight (U = 4 where the launch leaves the registers, 2 inside K5)
// THREADS: workgroup size of the launch that plays the role (the matrix-core seed level runs 512-thread workgroups)
template <int N, int U, int THREADS = 256>
__device__ __forceinline__ void gather_role(const TailGather &g, int bx)
{
    const int64_t total = (int64_t)g.rows * g.chunks;
    const int64_t S = (int64_t)g.n_wg * THREADS;
    for (int64_t t0 = (int64_t)bx * THREADS + threadIdx.x; t0 < total; t0 += U * S) {
        int64_t row[U];
        int32_t c0[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int64_t t = t0 + u * S;
            ok[u] = t < total;
            if (!ok[u]) t = total - 1;
            row[u] = t / g.chunks;
            c0[u] = (int32_t)(t - row[u] * g.chunks) * 8;
        }
        // (node ids are < 2^31 -- the adjacency's neighbour array is int32 --: the low dword of each int64, as in
        // gather_mean_chunk; half the id registers, which is what lets a 512-thread launch keep U = 4 items in flight)
        int32_t id[U][N];
        const int32_t *ids32 = reinterpret_cast<const int32_t *>(g.ids);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < N; ++j) id[u][j] = ids32[2 * (row[u] * N + j)];
        vec16 v[U][N];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < N; ++j)
                v[u][j] = *reinterpret_cast<const vec16 *>(g.table + (int64_t)id[u][j] * g.ld + c0[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
            for (int j = 0; j < N; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += tail_elem(v[u][j], e);
            vec16 o;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const float a = (c0[u] + e < g.D) ? acc[e] / (float)N : 0.f;
                const float b = (c0[u] + e + 1 < g.D) ? acc[e + 1] / (float)N : 0.f;
                o[e >> 1] = (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16);
            }
            if (ok[u]) *reinterpret_cast<vec16 *>(g.out + row[u] * g.out_ld + c0[u]) = o;
        }
    }
}

// [host] validate a gsage_tail_gather_desc and turn it into kernel parameters
inline int fill_gather_role(TailGather &tg, const gsage_tail_gather_desc &d, const char *who)
{
    GSAGE_REQUIRE(d.n == 5 || d.n == 10 || d.n == 15, "%s: the gather role is built for fan-outs 5, 10 and 15", who);
    GSAGE_REQUIRE(d.table && d.ids && d.out && d.D > 0 && d.ld % 8 == 0 && d.out_ld % 8 == 0 &&
                  ceil_div(d.D, 8) * 8 <= d.ld && ceil_div(d.D, 8) * 8 <= d.out_ld && d.n_workgroups > 0 &&
                  (((uintptr_t)d.table | (uintptr_t)d.out) & 15) == 0, "%s: bad gather descriptor", who);
    tg.table = (const uint16_t *)d.table; tg.ids = d.ids; tg.out = (uint16_t *)d.out;
    tg.ld = d.ld; tg.out_ld = d.out_ld; tg.rows = (int32_t)d.rows;
    tg.D = (int32_t)d.D; tg.chunks = (int32_t)ceil_div(d.D, 8); tg.n_wg = d.n_workgroups;
    return GSAGE_OK;
}

}  // namespace gsage
