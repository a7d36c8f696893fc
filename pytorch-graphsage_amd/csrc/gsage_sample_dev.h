// gsage_sample_dev.h -- device body of the fused multi-hop sampler (K1), shared by the stand-alone
// kernel (gsage_sample.hip) and by k_gather_multi_adam (gsage_gather.hip), which samples a LATER
// batch's frontier inside the launch that gathers the next batch's rows.
#pragma once
#include "gsage_common.h"

namespace gsage {

__device__ __forceinline__ int64_t pick_neighbor(const int64_t *__restrict__ rowptr,
                                                 const int32_t *__restrict__ col, int64_t n_rows,
                                                 int64_t id, uint32_t s, int32_t *err_flag)
{
    if ((uint64_t)id >= (uint64_t)n_rows) {          // the reference raises IndexError here
        if (err_flag) *err_flag = 1;
        return 0;
    }
    const int64_t beg = rowptr[id];
    const int64_t deg = rowptr[id + 1] - beg;
    if (deg <= 0) return 0;                          // numpy: x % 0 == 0 -> column 0 of an empty row
    const uint64_t off = (deg <= 0xffffffffLL) ? (uint64_t)(s % (uint32_t)deg) : (uint64_t)s;
    return (int64_t)col[beg + (int64_t)off];
}

// Dense adjacency (reference nn_modules.py:19-49: UniformNeighborSampler over an int64 [n_rows, K] table whose
// rows were pre-sampled to exactly K neighbours): neighbour `column` of node id
__device__ __forceinline__ int64_t pick_dense(const int64_t *__restrict__ adj, int64_t ld, int64_t n_rows,
                                              int64_t id, uint32_t column, int32_t *err_flag)
{
    if ((uint64_t)id >= (uint64_t)n_rows || (int64_t)column >= ld) {     // torch indexing raises IndexError here
        if (err_flag) *err_flag = 1;
        return 0;
    }
    return adj[id * ld + (int64_t)column];
}


// Either of the two inside the fused multi-hop kernel: a dense row is a CSR row with beg = id * ld, deg = ld
// (one code path, two selects: the kernel shares its registers with the HBM-bound gather role, gsage_gather.hip)
__device__ __forceinline__ int64_t pick_any(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                            const int64_t *__restrict__ dense, int64_t dense_ld, int64_t n_rows,
                                            int64_t id, uint32_t s, int32_t *err_flag)
{
    if ((uint64_t)id >= (uint64_t)n_rows) {
        if (err_flag) *err_flag = 1;
        return 0;
    }
    const int64_t beg = dense ? id * dense_ld : rowptr[id];
    const int64_t deg = dense ? dense_ld : rowptr[id + 1] - beg;
    if (deg <= 0) return 0;
    const uint64_t off = (deg <= 0xffffffffLL) ? (uint64_t)(s % (uint32_t)deg) : (uint64_t)s;
    return dense ? dense[beg + (int64_t)off] : (int64_t)col[beg + (int64_t)off];
}

// All hops of a frontier in ONE launch.  A workgroup owns SPW consecutive seeds and walks their
// whole sub-tree: the children of hop k are produced by the same workgroup that consumes them at
// hop k+1, so the only synchronisation is __syncthreads() and the previous hop's ids sit in LDS.
// Sample (hop k, global index g) uses exactly the Philox word the per-hop kernel would use
// (call index call_base + k - 1, counter g), so results are identical to L separate launches.
struct HopsParams {
    const int64_t *rowptr;
    const int32_t *col;
    int64_t *ids;                 // [hop 0 | hop 1 | ... | hop L], hop 0 filled by the caller
    const uint64_t *call_ctr;
    int32_t *err_flag;
    const int64_t *seed_queue;    // optional [n_batches, B] device-resident seed batches ...
    const int64_t *batch_idx;     // ... and the device word selecting the current one
    int64_t n_batches;
    int64_t batch_base;           // added to *batch_idx (sampling AHEAD of the counters, see k_gather_multi_adam)
    const int32_t *sel;           // optional: caller-supplied sel [hop 1 | hop 2 | ...] instead of Philox
    int64_t sel_stride;           // with a seed queue: sel of batch b starts at sel + b * sel_stride
    const int64_t *dense_adj;     // optional: dense [n_rows, dense_ld] adjacency instead of the CSR; sel then holds the
    int64_t dense_ld;             // columns every parent of a hop keeps: [fan[1] of hop 1 | fan[2] of hop 2 | ...]
    int64_t n_rows;
    int64_t off[6];               // first element of hop k in ids
    uint64_t g0[6];               // global sample index of this rank's first sample of hop k
    uint64_t call_base;
    int32_t fan[6];               // fan[k]: samples per parent at hop k (k >= 1)
    int32_t n_hops, B;
    uint32_t max_deg, seed_lo, seed_hi;
    int32_t spw;                  // seeds per workgroup (hops_spw())
};

// Seeds per workgroup of the fused multi-hop sampler.  1 (default): hop 2 of a 25x10 frontier is one pass of 250
// lanes, so a seed's whole sub-tree costs two dependent (rowptr -> col) round trips.  GSAGE_HOPS_SPW=2 / 4 (read once):
// fewer, fatter workgroups -- as a ROLE of k_gather_multi_adam the sampler's B workgroups hold B of the launch's 1 792
// resident slots while their chains run (what the finalisation role ran into).
inline int hops_spw()
{
    static const int v = [] { const char *e = getenv("GSAGE_HOPS_SPW"); const int x = e ? atoi(e) : 1;
                              return x >= 1 && x <= 16 ? x : 1; }();
    return v;
}

// frontier: two ping-pong LDS buffers of the widest hop; wg: which group of p.spw seeds.
// DENSE_OK = false compiles the dense-adjacency mode out: k_gather_multi_adam (gsage_gather.hip) runs this body
// beside the HBM-bound gather role under a 72-VGPR cap it already sits on -- two more live values spill.
template <bool DENSE_OK = true>
__device__ __forceinline__ void sample_hops_workgroup(const HopsParams &p, int wg, int64_t *frontier)
{
    const int seed0 = wg * p.spw;
    const int nseed = min(p.spw, p.B - seed0);
    if (nseed <= 0) return;
    int width = 1, widest = 1;
    for (int k = 1; k <= p.n_hops; ++k) { width *= p.fan[k]; widest = max(widest, width); }
    int64_t *cur = frontier, *nxt = frontier + (int64_t)p.spw * widest;
    const int32_t *sel = p.sel;
    if (p.seed_queue) {           // take the seeds from the queue (and publish them as hop 0)
        const int64_t b = (int64_t)((uint64_t)(*p.batch_idx + p.batch_base) % (uint64_t)p.n_batches);
        if (sel) sel += b * p.sel_stride;
        for (int t = threadIdx.x; t < nseed; t += 256) {
            const int64_t v = p.seed_queue[b * p.B + seed0 + t];
            cur[t] = v;
            p.ids[p.off[0] + seed0 + t] = v;
        }
    } else {
        for (int t = threadIdx.x; t < nseed; t += 256) cur[t] = p.ids[p.off[0] + seed0 + t];
    }
    lds_barrier();              // LDS only: the ids stored to HBM are not read back in this launch
    const uint64_t ctr = p.call_ctr ? *p.call_ctr : 0ull;
    int64_t per_seed = 1;                            // nodes of hop k per seed
    for (int k = 1; k <= p.n_hops; ++k) {
        const uint32_t n = (uint32_t)p.fan[k];
        const int64_t parents = per_seed * nseed;
        per_seed *= n;
        const int64_t count = per_seed * nseed;
        const uint64_t call = p.call_base + ctr + (uint64_t)(k - 1);
        const int64_t local0 = (int64_t)seed0 * per_seed;             // first sample of this WG in hop k
        // dense sampler: ONE column permutation per sampler call, shared by every parent of the hop
        // (nn_modules.py:44-48): sample j of a parent reads column keep_k[j], keep_k = sel[kbase ..]
        int64_t kbase = 0;
        if (DENSE_OK)
            for (int q = 1; q < k; ++q) kbase += p.fan[q];
        const bool dense = DENSE_OK && p.dense_adj != nullptr;
        for (int64_t t = threadIdx.x; t < count; t += 256) {
            uint32_t s;
            if (sel) {               // parity level 1: the reference's own draws (nn_modules.py:88), replayed
                s = (uint32_t)sel[dense ? kbase + (int64_t)((uint32_t)t % n) : p.off[k] - p.off[1] + local0 + t];
            } else {
                const uint64_t g = p.g0[k] + (uint64_t)(local0 + t);
                const uint64_t blk = g >> 2;
                const philox4 r = philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)call,
                                                (uint32_t)(call >> 32), p.seed_lo, p.seed_hi);
                const uint32_t sel4 = (uint32_t)g & 3u;          // selects, not r.v[g & 3]: no scratch
                const uint32_t w = sel4 == 0 ? r.v[0] : sel4 == 1 ? r.v[1] : sel4 == 2 ? r.v[2] : r.v[3];
                s = (uint32_t)(((uint64_t)w * (uint64_t)p.max_deg) >> 32);
            }
            const int64_t parent = cur[(uint32_t)t / n];
            const int64_t v = DENSE_OK
                ? pick_any(p.rowptr, p.col, p.dense_adj, p.dense_ld, p.n_rows, parent, s, p.err_flag)
                : pick_neighbor(p.rowptr, p.col, p.n_rows, parent, s, p.err_flag);
            nxt[t] = v;
            p.ids[p.off[k] + local0 + t] = v;
        }
        (void)parents;
        lds_barrier();
        int64_t *tmp = cur; cur = nxt; nxt = tmp;
    }
}


// The same sub-trees walked by a WIDE workgroup (THREADS lanes, U samples per lane and trip, CSR only): the role of a
// launch whose workgroups own a whole CU each (k_mean_tail_mfma, gsage_tail_mfma.hip) -- few workgroups, many seeds
// each, so a hop is several trips of dependent (rowptr -> col) loads; the U samples of a trip have their loads issued
// together instead of one chain after the other.  Same Philox words, same ids as sample_hops_workgroup.
template <int THREADS, int U>
__device__ __forceinline__ void sample_hops_wide(const HopsParams &p, int wg, int64_t *frontier)
{
    const int seed0 = wg * p.spw;
    const int nseed = min(p.spw, p.B - seed0);
    if (nseed <= 0) return;
    int width = 1, widest = 1;
    for (int k = 1; k <= p.n_hops; ++k) { width *= p.fan[k]; widest = max(widest, width); }
    int64_t *cur = frontier, *nxt = frontier + (int64_t)p.spw * widest;
    const int32_t *sel = p.sel;
    if (p.seed_queue) {
        const int64_t b = (int64_t)((uint64_t)(*p.batch_idx + p.batch_base) % (uint64_t)p.n_batches);
        if (sel) sel += b * p.sel_stride;
        for (int t = threadIdx.x; t < nseed; t += THREADS) {
            const int64_t v = p.seed_queue[b * p.B + seed0 + t];
            cur[t] = v;
            p.ids[p.off[0] + seed0 + t] = v;
        }
    } else {
        for (int t = threadIdx.x; t < nseed; t += THREADS) cur[t] = p.ids[p.off[0] + seed0 + t];
    }
    lds_barrier();
    const uint64_t ctr = p.call_ctr ? *p.call_ctr : 0ull;
    int64_t per_seed = 1;
    for (int k = 1; k <= p.n_hops; ++k) {
        const uint32_t n = (uint32_t)p.fan[k];
        per_seed *= n;
        const int64_t count = per_seed * nseed;
        const uint64_t call = p.call_base + ctr + (uint64_t)(k - 1);
        const int64_t local0 = (int64_t)seed0 * per_seed;
        for (int64_t t0 = threadIdx.x; t0 < count; t0 += (int64_t)THREADS * U) {
            uint32_t s[U];
            int64_t parent[U], beg[U], deg[U], v[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t t = t0 + (int64_t)u * THREADS;
                ok[u] = t < count;
                const int64_t tt = ok[u] ? t : count - 1;
                if (sel) {
                    s[u] = (uint32_t)sel[p.off[k] - p.off[1] + local0 + tt];
                } else {
                    const uint64_t g = p.g0[k] + (uint64_t)(local0 + tt);
                    const uint64_t blk = g >> 2;
                    const philox4 r = philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)call,
                                                    (uint32_t)(call >> 32), p.seed_lo, p.seed_hi);
                    const uint32_t sel4 = (uint32_t)g & 3u;
                    const uint32_t w = sel4 == 0 ? r.v[0] : sel4 == 1 ? r.v[1] : sel4 == 2 ? r.v[2] : r.v[3];
                    s[u] = (uint32_t)(((uint64_t)w * (uint64_t)p.max_deg) >> 32);
                }
                parent[u] = cur[(uint32_t)tt / n];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool in = (uint64_t)parent[u] < (uint64_t)p.n_rows;
                if (!in && ok[u] && p.err_flag) *p.err_flag = 1;              // the reference raises IndexError here
                const int64_t id = in ? parent[u] : 0;
                beg[u] = p.rowptr[id];
                deg[u] = in ? p.rowptr[id + 1] - beg[u] : 0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint64_t o = (deg[u] <= 0xffffffffLL) ? (uint64_t)(s[u] % (uint32_t)(deg[u] > 0 ? deg[u] : 1))
                                                            : (uint64_t)s[u];
                v[u] = deg[u] > 0 ? (int64_t)p.col[beg[u] + (int64_t)o] : 0;      // (numpy: x % 0 == 0 -> column 0)
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t t = t0 + (int64_t)u * THREADS;
                if (ok[u]) {
                    nxt[t] = v[u];
                    p.ids[p.off[k] + local0 + t] = v[u];
                }
            }
        }
        lds_barrier();
        int64_t *tmp = cur; cur = nxt; nxt = tmp;
    }
}

// [host] validate a gsage_hops_desc and turn it into kernel parameters + dynamic LDS bytes
inline int fill_hops(HopsParams &p, size_t &lds, const gsage_hops_desc &d)
{
    GSAGE_REQUIRE(d.ids && ((d.rowptr && d.col) || d.dense_adj), "sample_hops_philox: null pointer");
    GSAGE_REQUIRE(!d.dense_adj || (d.sel && d.dense_ld > 0),
                  "sample_hops: a dense adjacency needs the kept columns in `sel` and its leading dimension");
    GSAGE_REQUIRE(!d.seed_queue || (d.batch_idx && d.n_batches > 0), "sample_hops_philox: bad seed queue");
    GSAGE_REQUIRE(d.n_hops >= 1 && d.n_hops <= 5, "sample_hops_philox: 1..5 hops");
    GSAGE_REQUIRE(d.B >= 0 && d.B < (1LL << 31) && d.max_deg > 0, "sample_hops_philox: bad sizes");
    p.rowptr = d.rowptr; p.col = d.col; p.ids = d.ids; p.call_ctr = d.call_ctr; p.err_flag = d.err_flag;
    p.seed_queue = d.seed_queue; p.batch_idx = d.batch_idx; p.n_batches = d.n_batches;
    p.batch_base = d.batch_base;
    p.sel = d.sel; p.sel_stride = d.sel ? d.sel_stride : 0;
    p.dense_adj = d.dense_adj; p.dense_ld = d.dense_adj ? d.dense_ld : 0;
    GSAGE_REQUIRE(!d.sel || d.sel_stride >= 0, "sample_hops: bad sel stride");
    p.n_rows = d.n_rows; p.call_base = d.call_base; p.n_hops = d.n_hops; p.B = (int32_t)d.B;
    p.max_deg = d.max_deg;
    p.seed_lo = (uint32_t)d.seed; p.seed_hi = (uint32_t)(d.seed >> 32);
    int64_t size = d.B, off = 0, widest = 1, width = 1;
    p.fan[0] = 1;
    for (int k = 0; k <= 5; ++k) {
        if (k >= 1 && k <= d.n_hops) {
            GSAGE_REQUIRE(d.fan[k - 1] > 0, "sample_hops_philox: n_samples must be > 0");
            p.fan[k] = d.fan[k - 1];
            size *= d.fan[k - 1];
            width *= d.fan[k - 1];
            if (width > widest) widest = width;
        } else if (k > d.n_hops) {
            p.fan[k] = 1;
        }
        p.off[k] = off;
        p.g0[k] = d.rank * (uint64_t)size;
        if (k <= d.n_hops) off += size;
    }
    p.spw = hops_spw();
    lds = sizeof(int64_t) * 2 * (size_t)p.spw * (size_t)widest;
    GSAGE_REQUIRE(lds <= 160 * 1024, "sample_hops_philox: fan-out product too large for the fused kernel");
    return GSAGE_OK;
}

}  // namespace gsage
