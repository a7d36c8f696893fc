// gsage_optim_dev.h -- device body of the clip + Adam update, shared by k_adam_clip
// (gsage_optim.hip) and k_gather_multi_adam (gsage_gather.hip), which applies the update of batch i
// side by side with the level-0 gathers of batch i+1.
#pragma once
#include "gsage_common.h"

namespace gsage {

__device__ __forceinline__ float block_sum_256(float v, float *red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    lds_barrier();              // LDS-only: callers may have global stores in flight that nobody here reads
    if (lane == 0) red[wave] = v;
    lds_barrier();
    return red[0] + red[1] + red[2] + red[3];
}

// ---- weight operand copies ------------------------------------------------------------------------
struct PrepDesc {
    const float *src;      // [rows, cols] fp32, contiguous
    uint16_t *dst;         // [rows, dst_ld] bf16 (may be null)
    uint16_t *dst_t;       // [cols, dst_t_ld] bf16 transposed copy (may be null)
    int32_t rows, cols, dst_ld, dst_t_ld;
    uint16_t *dst_p;       // MFMA-fragment-ordered copy for k_linear_nt_packed (may be null)
    int64_t kc_p;          // 16-wide k chunks per 32-column block of dst_p
    int32_t dst_f32;       // != 0: dst / dst_t are fp32 copies (exact-arithmetic parity mode; no dst_p)
    int32_t reserved;
};

// element (r, c) of a weight matrix inside its packed operand (include/gsage.h, gsage_linear_nt_packed)
__device__ __forceinline__ int64_t packed_offset(int r, int c, int64_t kc_total)
{
    const int64_t slot = ((int64_t)(r >> 5) * kc_total + (c >> 4)) * 64 + (r & 31) + 32 * ((c >> 3) & 1);
    return slot * 8 + (c & 7);
}

// store one (updated) weight into the operand copies of its descriptor
__device__ __forceinline__ void prep_store(const PrepDesc &q, int r, int c, float w)
{
    if (q.dst_f32) {
        if (q.dst) reinterpret_cast<float *>(q.dst)[(int64_t)r * q.dst_ld + c] = w;
        if (q.dst_t) reinterpret_cast<float *>(q.dst_t)[(int64_t)c * q.dst_t_ld + r] = w;
        return;
    }
    const uint16_t b = f32_to_bf16(w);
    if (q.dst) q.dst[(int64_t)r * q.dst_ld + c] = b;
    if (q.dst_t) q.dst_t[(int64_t)c * q.dst_t_ld + r] = b;
    if (q.dst_p) q.dst_p[packed_offset(r, c, q.kc_p)] = b;
}

// ---- gradient sources: partial buffers whose sum is a slice of the flat gradient bucket ---------------
// (gsage_reduce_desc: summed by gsage_finalize_grads -- or, ABI 5, by the update's own workgroups)
struct ReduceDesc {
    const float *src;       // S partial buffers, `stride` floats apart, each [rows, ld]
    int64_t stride;
    int64_t out_off;        // destination offset in the flat gradient bucket ([rows, cols] contiguous)
    int32_t S, rows, cols, ld;
};

struct AdamParams {
    float *p, *g, *m, *v;
    const float *partial;       // per-block squared-norm partials of g
    const float *lr;            // device scalar (a captured graph sees schedule changes)
    int64_t *step;              // device step counter; this launch uses *step + 1
    float *norm_out;            // optional: total gradient norm before clipping
    int64_t n;
    int32_t n_partial, step_off;
    float beta1, beta2, eps, weight_decay, max_norm;
    const struct PrepDesc *prep;   // optional: refresh the bf16 operand copies of the new weights
    int32_t n_prep;
    int64_t *tick1, *tick2;        // optional counters advanced at kernel start (not read here)
    int64_t inc1, inc2;
    int32_t discard_clipped;       // != 0: the clipped gradient is not written back (the caller zeroes g next)
    int32_t replay_math;           // != 0: the deferred-row arithmetic (adam_update<true>), for a table whose rows
                                   // may also be updated by gsage_rows_*: both must produce the same bits
    unsigned long long *norm_slots;   // != null: the workgroups form the squared norm themselves (slot = update << 32 | partial)
    const ReduceDesc *rdesc;          // != null (with norm_slots): g does not exist yet -- element i of the bucket is the
    int32_t n_rdesc;                  // sum of its descriptor's S partial buffers (what gsage_finalize_grads would store)
    int32_t stage_prep;               // != 0: kernels that can (adam_workgroup<.., STAGE>) stage the descriptors in LDS
};

// g[i] for four elements of the flat bucket (i[u] < 0: none) out of the partial buffers of their descriptors.
// Partials are added in buffer order 0 .. S-1 (gsage_finalize_grads' order: the same bits), four buffers of each of
// the four elements in flight together (16 loads per lane and round: the register budget of k_gather_multi_adam;
// R = 12 in k_gather_multi_adam_wide: 48 per round, K5b's 24 slabs in two rounds instead of six).
template <int R = 4>
__device__ __forceinline__ void reduce_partials4(const ReduceDesc *__restrict__ rd, int n_rd, const int64_t (&i)[4],
                                                 float (&g)[4])
{
    const float *src[4];
    int64_t stride[4];
    int S[4], Smax = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        src[u] = rd[0].src; stride[u] = 0; S[u] = 0; g[u] = 0.f;     // (no source: a valid address, nothing added)
        for (int d = 0; d < n_rd; ++d) {
            const int64_t o = i[u] - rd[d].out_off;
            if (i[u] >= 0 && o >= 0 && o < (int64_t)rd[d].rows * rd[d].cols) {
                const int64_t r = o / rd[d].cols;
                src[u] = rd[d].src + r * rd[d].ld + (o - r * rd[d].cols);
                stride[u] = rd[d].stride;
                S[u] = rd[d].S;
            }
        }
        Smax = S[u] > Smax ? S[u] : Smax;
    }
    for (int s0 = 0; s0 < Smax; s0 += R) {
        float v[4][R];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int sk = s0 + k;                       // (past the element's last buffer: its buffer 0 again,
                v[u][k] = src[u][(int64_t)(sk < S[u] ? sk : 0) * stride[u]];                  // dropped below:
                //                                                              the loads stay unconditional)
            }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < R; ++k)
                if (s0 + k < S[u]) g[u] += v[u][k];
    }
}

// The per-step constants and the per-element update of Adam (torch.optim.Adam's formulas), shared by the dense
// kernels and the deferred row updates of a trainable embedding table (gsage_rows_*): the latter replay the
// dense update row by row and must produce the SAME bits, so nothing here may be contracted differently
// from one call site to the next -- contraction is off and every rounding is the source's.
struct AdamConsts { float step_size, rsqrt_bc2; };

__device__ __forceinline__ AdamConsts adam_consts(float lr, float t, float beta1, float beta2)
{
#pragma clang fp contract(off)
    const float bc1 = 1.f - powf(beta1, t);
    const float bc2 = 1.f - powf(beta2, t);
    AdamConsts c;
    c.step_size = lr / bc1;
    c.rsqrt_bc2 = 1.f / sqrtf(bc2);
    return c;
}

// g: the (clipped) gradient; returns the new parameter, updates m and v in place.
// REPLAY = false (every dense optimizer path: FlatAdam, the engines' weight buckets): correctly rounded sqrtf and
// division, i.e. torch.optim.Adam's roundings.  REPLAY = true (gsage_rows_* and the dense update of a table that
// those kernels may also touch): v_sqrt_f32 / v_rcp_f32 (1 ulp) instead of the correctly rounded sequences
// (~25 instructions each) -- the deferred row updates replay this function once per row and skipped step, which
// makes them ALU-bound, and the two sides only have to agree WITH EACH OTHER bit for bit.
template <bool REPLAY>
__device__ __forceinline__ float adam_update(float g, float p, float &m, float &v, float beta1, float beta2,
                                             float eps, float weight_decay, float step_size, float rsqrt_bc2)
{
#pragma clang fp contract(off)
    if (weight_decay != 0.f) g = g + weight_decay * p;
    m = beta1 * m + (1.f - beta1) * g;
    v = beta2 * v + ((1.f - beta2) * g) * g;
    if (REPLAY) {
        const float denom = __builtin_amdgcn_sqrtf(v) * rsqrt_bc2 + eps;
        return p - step_size * (m * __builtin_amdgcn_rcpf(denom));
    }
    const float denom = sqrtf(v) * rsqrt_bc2 + eps;
    return p - step_size * (m / denom);
}

// one workgroup of the clip + Adam update: grid-stride slice bx of gx.  REPLAY_OK = false: the caller never sets
// a.replay_math (k_gather_multi_adam: only the exact arithmetic is compiled in, its registers are the gather role's)
// PR = partial buffers per element and round when the workgroup sums them itself (reduce_partials4)
// STAGE: the operand-copy descriptors travel through LDS (for kernels with registers to spare: k_gather_multi_adam_wide)
template <bool REPLAY_OK = true, int PR = 4, bool STAGE = false>
__device__ __forceinline__ void adam_workgroup(const AdamParams &a, int bx, int gx, float *red)
{
    float s = 0.f;
    // The operand-copy descriptors are read at the END of the update's chain, one scalar load after the other (each a
    // dependent round trip of its own): they are fetched into LDS here, beside the first operands, and read back from
    // there (the barriers of the norm's block sum order the two).
    static_assert(sizeof(PrepDesc) == 64, "PrepDesc is staged as four 16-byte words");
    __shared__ PrepDesc sh_prep[16];
    const bool staged = STAGE && a.stage_prep && a.n_prep > 0 && a.n_prep <= 16;
    if (staged && (int)threadIdx.x < 4 * a.n_prep)
        reinterpret_cast<vec16 *>(sh_prep)[threadIdx.x] = reinterpret_cast<const vec16 *>(a.prep)[threadIdx.x];
    // the first (with the in-launch norm: the only) trip's operands, requested before anything else
    const int64_t stride = (int64_t)gx * 256;
    const int64_t i_first = (int64_t)bx * 256 + threadIdx.x;
    float gv0[4], pv0[4], mv0[4], vv0[4];
    const bool meet = !REPLAY_OK && a.norm_slots != nullptr;
    if (meet) {
        // The norm of a gradient that exists only now (data-parallel: after the exchange), formed by the update's own
        // workgroups.  They are dispatched first and are few, so all of them are resident: each publishes the partial
        // of the elements it updates with ONE device-scope store tagged with the update number, every thread polls
        // one slot until it carries that number, and all workgroups add the same partials in the same order (the
        // same bits in every workgroup and on every rank).  RELAXED device-scope atomics only -- a release / acquire
        // FENCE at device scope writes back / invalidates the XCD's whole L2 beside a gather role that is filling
        // it (measured: 32 us per launch) -- and no counter: the tag makes a slot's value self-describing.
        float q = 0.f;
        if (a.rdesc) {
            // no finalisation launch ran: this workgroup sums the partial buffers of the elements it updates (K5b's
            // slabs, the head's per-workgroup partials) and leaves the result in the flat bucket for whoever reads
            // p.grad -- the clipped value, as always, when the clip is active (below)
            int64_t iv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) iv[u] = i_first + u * stride < a.n ? i_first + u * stride : -1;
            reduce_partials4<PR>(a.rdesc, a.n_rdesc, iv, gv0);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (iv[u] >= 0) a.g[iv[u]] = gv0[u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i_first + u * stride;
            const int64_t ic = i < a.n ? i : i_first < a.n ? i_first : 0;
            if (!a.rdesc) gv0[u] = a.g[ic];
            pv0[u] = a.p[ic]; mv0[u] = a.m[ic]; vv0[u] = a.v[ic];
            if (i < a.n) q += gv0[u] * gv0[u];
        }
        const float mine = block_sum_256(q, red);
        const unsigned long long tag = (unsigned long long)(uint32_t)(*a.step + a.step_off) << 32;
        if (threadIdx.x == 0)
            __hip_atomic_store(a.norm_slots + bx, tag | (unsigned long long)__float_as_uint(mine), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        for (int i = threadIdx.x; i < gx; i += 256) {
            unsigned long long v = __hip_atomic_load(a.norm_slots + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while ((v >> 32) != (tag >> 32)) {
                __builtin_amdgcn_s_sleep(1);
                v = __hip_atomic_load(a.norm_slots + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            s += __uint_as_float((uint32_t)v);
        }
    } else {
        // (the first trip's operands do not depend on the norm either: requested BEFORE its partials are summed, one
        // dependent round trip less in a chain that is the floor of the launch it rides in -- not for the big plain
        // buckets of the 16-byte path below, which read their operands themselves)
        if (a.n_prep > 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = i_first + u * stride;
                const int64_t ic = i < a.n ? i : i_first < a.n ? i_first : 0;
                gv0[u] = a.g[ic]; pv0[u] = a.p[ic]; mv0[u] = a.m[ic]; vv0[u] = a.v[ic];
            }
        }
        for (int i = threadIdx.x; i < a.n_partial; i += 256) s += a.partial[i];
    }
    const float sq = block_sum_256(s, red);
    const float total = sqrtf(sq);
    float coef = a.max_norm / (total + 1e-6f);          // torch.nn.utils.clip_grad_norm_
    coef = coef < 1.f ? coef : 1.f;
    const AdamConsts ac = adam_consts(*a.lr, (float)(*a.step + a.step_off), a.beta1, a.beta2);
    const float step_size = ac.step_size, rsqrt_bc2 = ac.rsqrt_bc2;
    if (bx == 0 && threadIdx.x == 0) {
        if (a.norm_out) *a.norm_out = total;
        if (a.tick1) *a.tick1 += a.inc1;
        if (a.tick2) *a.tick2 += a.inc2;
    }

    const bool clipped = coef < 1.f && !a.discard_clipped;   // (block-uniform) unclipped gradients are not rewritten
    if (a.n_prep == 0 && (a.n & 3) == 0 &&
        ((((uintptr_t)a.p | (uintptr_t)a.g | (uintptr_t)a.m | (uintptr_t)a.v) & 15) == 0)) {
        // big plain buckets (a trainable embedding table): 16-byte lanes, no operand copies to refresh
        typedef float v4 __attribute__((ext_vector_type(4)));
        const int64_t n4 = a.n >> 2;
        for (int64_t i = (int64_t)bx * 256 + threadIdx.x; i < n4; i += stride) {
            v4 g = reinterpret_cast<const v4 *>(a.g)[i] * coef;
            const v4 p = reinterpret_cast<const v4 *>(a.p)[i];
            v4 m = reinterpret_cast<const v4 *>(a.m)[i];
            v4 v = reinterpret_cast<const v4 *>(a.v)[i];
            if (clipped) reinterpret_cast<v4 *>(a.g)[i] = g;
            v4 pn;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float me = m[e], ve = v[e];
                pn[e] = (REPLAY_OK && a.replay_math)
                    ? adam_update<true>(g[e], p[e], me, ve, a.beta1, a.beta2, a.eps, a.weight_decay, step_size, rsqrt_bc2)
                    : adam_update<false>(g[e], p[e], me, ve, a.beta1, a.beta2, a.eps, a.weight_decay, step_size, rsqrt_bc2);
                m[e] = me; v[e] = ve;
            }
            reinterpret_cast<v4 *>(a.m)[i] = m;
            reinterpret_cast<v4 *>(a.v)[i] = v;
            reinterpret_cast<v4 *>(a.p)[i] = pn;
        }
        return;
    }
    // four elements per thread and trip, their 16 loads in flight together: a quarter of the workgroups
    // (each a chain of dependent round trips: partials -> norm -> loads -> stores) for the same update
    for (int64_t i0 = i_first; i0 < a.n; i0 += 4 * stride) {
        float gv[4], pv[4], mv[4], vv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (meet || (a.n_prep > 0 && i0 == i_first)) {      // (the in-launch norm is one trip: the host admits it
                gv[u] = gv0[u]; pv[u] = pv0[u]; mv[u] = mv0[u]; vv[u] = vv0[u];   // only when gx * 1 024 covers the bucket)
                continue;
            }
            const int64_t i = i0 + u * stride;
            const int64_t ic = i < a.n ? i : i0;
            gv[u] = a.g[ic]; pv[u] = a.p[ic]; mv[u] = a.m[ic]; vv[u] = a.v[ic];
        }
        float pnew[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + u * stride;
            pnew[u] = 0.f;
            if (i >= a.n) continue;
            float g = gv[u] * coef;
            if (clipped) a.g[i] = g;                    // clipped gradient stays visible (p.grad)
            float m = mv[u], v = vv[u];
            const float pn = (REPLAY_OK && a.replay_math)
                ? adam_update<true>(g, pv[u], m, v, a.beta1, a.beta2, a.eps, a.weight_decay, step_size, rsqrt_bc2)
                : adam_update<false>(g, pv[u], m, v, a.beta1, a.beta2, a.eps, a.weight_decay, step_size, rsqrt_bc2);
            a.m[i] = m;
            a.v[i] = v;
            a.p[i] = pn;
            pnew[u] = pn;
        }
        // operand copies for the next step's GEMMs (replaces a separate k_prep_weights launch).  Descriptor-major:
        // a descriptor is fetched ONCE per trip (a wave-uniform address in a uniform loop: scalar loads) and tested
        // against the trip's four elements -- element-major with an early exit it was up to 16 dependent descriptor
        // fetches per thread at the END of the update's chain, which is the floor of the launch it rides in
        for (int d = 0; d < a.n_prep; ++d) {
            const PrepDesc q = staged ? sh_prep[d] : a.prep[d];
            const int64_t base = q.src - a.p, span = (int64_t)q.rows * q.cols;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = i0 + u * stride;
                const int64_t o = i - base;
                if (i < a.n && o >= 0 && o < span) {    // (descriptors cover disjoint slices: at most one matches)
                    const int r = (int)(o / q.cols), c = (int)(o - (int64_t)r * q.cols);
                    prep_store(q, r, c, pnew[u]);
                }
            }
        }
    }
}


inline int adam_grid(int64_t items, int cap)
{
    int64_t b = ceil_div(items, 256);
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// [host] validate a gsage_adam_desc (norm partials must be ready) and turn it into kernel parameters
inline int fill_adam(AdamParams &a, const gsage_adam_desc &d)
{
    GSAGE_REQUIRE(d.p && d.g && d.m && d.v && d.partial && d.lr && d.step, "clip_adam_step: null pointer");
    GSAGE_REQUIRE(d.n > 0 && (d.n_partial_ready > 0 || d.norm_slots) && d.n_prep >= 0,
                  "clip_adam_step: bad sizes (norm partials must be ready, or norm_slots given)");
    a.prep = (const PrepDesc *)d.prep_descs; a.n_prep = d.prep_descs ? d.n_prep : 0;
    a.tick1 = d.tick1; a.inc1 = d.inc1; a.tick2 = d.tick2; a.inc2 = d.inc2;
    a.p = d.p; a.g = d.g; a.m = d.m; a.v = d.v; a.partial = d.partial; a.lr = d.lr; a.step = d.step;
    a.norm_out = d.norm_out; a.n = d.n; a.n_partial = d.n_partial_ready; a.beta1 = d.beta1;
    a.beta2 = d.beta2; a.eps = d.eps; a.weight_decay = d.weight_decay; a.max_norm = d.max_norm;
    a.step_off = d.step_is_current ? 0 : 1;
    a.discard_clipped = 0;
    a.replay_math = 0;
    const bool inside = d.n_partial_ready == 0 && d.norm_slots;
    a.norm_slots = inside ? (unsigned long long *)d.norm_slots : nullptr;
    GSAGE_REQUIRE(!d.reduce_descs || (inside && d.n_reduce > 0 && d.n_reduce <= 16),
                  "clip_adam_step: reduce_descs (the update sums the partial buffers itself) needs norm_slots, "
                  "n_partial_ready == 0 and 1..16 descriptors");
    a.rdesc = (const ReduceDesc *)d.reduce_descs;
    a.n_rdesc = d.reduce_descs ? d.n_reduce : 0;
    static const int stage = [] { const char *e = getenv("GSAGE_ADAM_STAGE"); return e ? atoi(e) : 1; }();
    a.stage_prep = stage;
    return GSAGE_OK;
}

}  // namespace gsage
