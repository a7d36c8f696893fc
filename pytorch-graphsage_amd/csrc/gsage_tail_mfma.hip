// gsage_tail_mfma.hip -- the seed level of a mean-aggregator step on the matrix cores: SIXTEEN seeds per workgroup.
//
// Same contract as k_mean_tail_ce (gsage_tail.hip; reference nn_modules.py:197-202, models.py:90-91,100,
// problem.py:34 and their autograd): segment mean of the n sampled neighbours, emb = [x Wx^T | agg Wn^T],
// normalize -> fc -> softmax cross-entropy, and every gradient down to the previous level's activations, in ONE
// launch.  What changed is who does the arithmetic and how many workgroups there are:
//
//   * k_mean_tail_ce owns 4 seeds per workgroup (B / 4 = 128 workgroups at B = 512) and runs its two
//     4 x 256 x 256 projections, the head and the input gradients on the VALU: ~22 us of dependent phases, each
//     workgroup streaming the same 256 KB of projection weights.  It was the longest launch of the headline step.
//   * here a workgroup of 512 threads owns 16 seeds = ONE MFMA row tile (B / 16 = 32 workgroups: the weights are
//     streamed 32 times instead of 128, and 224 CUs are free for the gather role).  Projections and input gradients:
//     v_mfma_f32_16x16x32_bf16 (A = the 16 seeds' rows from LDS, B = weight fragments straight from the row-major
//     operand copies: 16 contiguous bytes per lane).  Head (logits, d z, d fc.weight): v_mfma_f32_16x16x4_f32 --
//     exact fp32 products like the VALU kernel's, so predictions and head gradients keep their ~1e-7 agreement with
//     the oracle; bf16 enters only where the VALU kernel also rounds (agg, dE, the dH rows).
//
// Layouts.  "row layout" (global rows of H / dH / agg / dE, all 16 bytes per lane): wave w, half-wave h owns seed
// 2w + h, lane & 31 owns 8 consecutive columns; a lane sums its seed's n neighbour rows itself (no cross-lane
// traffic) and keeps their ReLU masks as bits.  "tile layout" (MFMA results): wave w owns output columns
// 32w .. 32w+31 as two 16 x 16 tiles; lane l holds column 16 ct + (l & 15), rows 4 (l >> 4) + r.  The same
// ownership is used for emb, z, d z and d emb, so z and the row norms stay in registers between the GEMMs.
// Everything that changes layout goes through LDS (padded so that the fragment reads are conflict-free).
#include "gsage_common.h"
#include "gsage_gather_dev.h"
#include "gsage_sample_dev.h"

namespace gsage {

typedef __attribute__((ext_vector_type(8))) __bf16 tm_bf16x8;
typedef float tm_f32x4 __attribute__((ext_vector_type(4)));

constexpr int TM_S = 16;            // seeds per workgroup
constexpr int TM_T = 512;           // threads per workgroup (8 waves, two per SIMD)
constexpr int TM_D = 256;           // width of the previous level's rows and of this level's output
constexpr int TM_CMAX = 64;
constexpr int TM_LDH = 264;         // bf16 elements per LDS row of xs / as / des (528 B: +4 banks per row)
constexpr int TM_LDF = 260;         // floats per LDS row of zs / Ws / gxs / gns (+4 banks per row)
constexpr int TM_LDP = 68;          // floats per row of the logit partials and of d logits (+4 banks per row,
                                    // 4 rows = +16 banks: the tile-layout stores are conflict-free)

struct TailMfmaParams {
    const uint16_t *H;       // previous level output, bf16 [B*(1+n), 256]: seeds first, then neighbours
    const uint16_t *w2;      // bf16 [2, 128, ldw2]:  fc_x | fc_neib           (rows = outputs)
    const uint16_t *w2t;     // bf16 [2, 256, ldw2t]: transposed copies        (rows = inputs)
    const float *Wfc;        // [C, 256]
    const float *bfc;        // [C]
    const int64_t *targets;
    const int64_t *batch_idx;
    int64_t n_batches;
    const int32_t *n_valid;
    uint16_t *agg;           // out: bf16 [B, 256] neighbour means (A operand of this level's K5b)
    uint16_t *dE;            // out: bf16 [B, 256] d loss / d emb   (dC operand of this level's K5b)
    float *preds;            // out: [B, C] logits
    uint16_t *dH;            // out: bf16 [B*(1+n), 256] gradient w.r.t. H (ReLU mask applied)
    float *partial;          // out: [grid, C*256 + C + 1] fc.weight / fc.bias / loss partials
    int64_t ldw2, ldw2t;
    int32_t B, n, C;
    int32_t stop;            // measurement only (GSAGE_TAIL_STOP = 1 .. 5): leave after that phase; 0 = the whole kernel
    int32_t n_smp;           // workgroups of the sampler role (behind the seed-level ones, before the gather role's)
};

constexpr size_t tm_lds_bytes()
{
    return sizeof(float) * ((size_t)TM_CMAX * TM_LDF            // Ws
                            + (size_t)TM_S * TM_LDF             // zs
                            + (size_t)8 * TM_S * TM_LDP         // part (later gxs | gns: 2 * 16 * 260 <= 8 * 16 * 68)
                            + (size_t)TM_S * TM_LDP             // dls
                            + 2 * 8 * TM_S + TM_S + 16)         // red, red2, lss
           + sizeof(uint16_t) * (size_t)3 * TM_S * TM_LDH;      // xs, as, des
}
static_assert(2 * TM_S * TM_LDF <= 8 * TM_S * TM_LDP, "gxs | gns must fit the logit partials they alias");
static_assert(tm_lds_bytes() <= 160 * 1024, "seed level: LDS budget");

// f32 -> bf16 bits, round-to-nearest-even, quiet NaN: f32_to_bf16's results without its early return (a branch per
// converted element splits the unrolled row loops into basic blocks the scheduler cannot order)
__device__ __forceinline__ uint32_t tm_bf16(float f)
{
    const uint32_t u = __float_as_uint(f);
    const uint32_t r = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    return (u & 0x7fffffffu) > 0x7f800000u ? ((u >> 16) | 0x0040u) : r;
}
__device__ __forceinline__ uint32_t tm_pack(float lo, float hi) { return tm_bf16(lo) | (tm_bf16(hi) << 16); }

__device__ __forceinline__ float tm_wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float tm_wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// sum over the 16 lanes that share l >> 4 (the columns of one tile row)
__device__ __forceinline__ float tm_sum16(float v)
{
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Buffer loads (a 128-bit resource in SGPRs + one 32-bit per-lane offset + an immediate): the 26 row requests and 16
// fragment requests a lane has in flight share THREE address registers instead of 42 pointer pairs (flat loads made
// the kernel spill), and a request past the end of its buffer returns zeros without a branch.
typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int tm_u32x4;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tm_rsrc(const void *base, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ vec16 tm_bload(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
    return __builtin_bit_cast(vec16, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

__device__ __forceinline__ tm_f32x4 tm_mma_bf16(const vec16 a, const vec16 b, const tm_f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(tm_bf16x8, a), __builtin_bit_cast(tm_bf16x8, b),
                                                   c, 0, 0, 0);
}
__device__ __forceinline__ tm_f32x4 tm_mma_f32(const float a, const float b, const tm_f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// N = the fan-out when the host knows a specialisation for it (every neighbour row a load slot of its own: nothing
// clamped, nothing predicated), 0 = any n <= 32 (32 slots, the index clamped to n - 1; the weight fragments are then
// requested after the rows have been consumed: the registers do not hold both).  GN = fan-out of the gather role.
#ifndef GSAGE_TM_ROLE_U
#define GSAGE_TM_ROLE_U 2
#endif
// Beside the gather role the launch can carry a SAMPLER role (gsage_hops_role_next): p.n_smp workgroups right behind
// the seed-level ones walk the sub-trees of a later batch's seeds (K1: dependent rowptr -> col loads that move almost
// nothing).  In the launch that carries the update K1's chain was the longer of that launch's two; here it runs
// under a launch that the gather role keeps busy for ~30 us anyway.
template <int N, int GN>
__global__ void __launch_bounds__(TM_T)
k_mean_tail_mfma(const TailMfmaParams p, const TailGather tg, const HopsParams hp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if (GN > 0) {
        const int n_tail = (p.B + TM_S - 1) / TM_S;
        if ((int)blockIdx.x >= n_tail) {
            const int r = (int)blockIdx.x - n_tail;
            if (r < p.n_smp) {
                sample_hops_wide<TM_T, 4>(hp, r, reinterpret_cast<int64_t *>(lds_raw));
                return;
            }
            gather_role<(GN > 0 ? GN : 1), (GN > 10 ? 2 : GSAGE_TM_ROLE_U), TM_T>(tg, r - p.n_smp);
            return;
        }
    }
    constexpr int NB = N > 0 ? N : 32;
    float *Ws = reinterpret_cast<float *>(lds_raw);               // [64][260] fc.weight, rows >= C zero
    float *zs = Ws + TM_CMAX * TM_LDF;                             // [16][260] emb, then z
    float *part = zs + TM_S * TM_LDF;                              // [8][16][68] logit partials per wave
    float *gxs = part, *gns = part + TM_S * TM_LDF;                // (later) [16][260] each: dX | dAgg
    float *dls = part + 8 * TM_S * TM_LDP;                         // [16][68] d logits
    float *red = dls + TM_S * TM_LDP;                              // [8][16]
    float *red2 = red + 8 * TM_S;                                  // [8][16]
    float *lss = red2 + 8 * TM_S;                                  // [16] (+16 spare)
    uint16_t *xs = reinterpret_cast<uint16_t *>(lss + TM_S + 16);  // [16][264] seed rows
    uint16_t *as = xs + TM_S * TM_LDH;                             // [16][264] neighbour means
    uint16_t *des = as + TM_S * TM_LDH;                            // [16][264] d emb

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int half = lane >> 5, cg = lane & 31;                    // row layout: seed 2 wave + half, columns 8 cg ..
    const int li = lane & 15, lg = lane >> 4;                      // tile layout: column li of a tile, rows 4 lg ..
    const int sr = 2 * wave + half;
    const int C = p.C, n = p.n;
    const int CT = (C + 15) >> 4;                                  // 16-class tiles
    const int64_t B = p.B;
    const int row0 = blockIdx.x * TM_S;

    const int64_t bq = p.batch_idx ? (int64_t)((uint64_t)*p.batch_idx % (uint64_t)p.n_batches) : 0;
    const int64_t *tgt = p.targets + bq * B;
    const int64_t Bv = p.n_valid ? (int64_t)min(max(p.n_valid[bq], 1), (int32_t)B) : B;

    // ---- 0. the rows of this half-wave's seed: all in flight at once -----------------------------------------
    const int64_t iw = row0 + sr;
    const bool live = iw < B;
    const int64_t iwc = live ? iw : B - 1;
    // fc.weight first (rows >= C: past the buffer's end = zeros): it lands before the rows do and is parked in LDS
    // while they are in flight (C <= 64 rows x 64 float4 = 8 per thread)
    vec16 wf[8];
    {
        const __amdgpu_buffer_rsrc_t rF = tm_rsrc(p.Wfc, (uint32_t)(C * TM_D * 4));
#pragma unroll
        for (int u = 0; u < 8; ++u) wf[u] = tm_bload(rF, (uint32_t)t * 16u, (uint32_t)u * TM_T * 16u);
    }
    const __amdgpu_buffer_rsrc_t rH = tm_rsrc(p.H, (uint32_t)(B * (1 + n) * TM_D * 2));
    vec16 xraw = tm_bload(rH, (uint32_t)(iwc * TM_D + cg * 8) * 2u, 0u);
    vec16 nb[NB];
    {
        const uint32_t base = (uint32_t)((B + iwc * n) * TM_D + cg * 8) * 2u;
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            if (N > 0) nb[u] = tm_bload(rH, base, (uint32_t)u * TM_D * 2u);
            else nb[u] = tm_bload(rH, base + (uint32_t)(u < n ? u : n - 1) * TM_D * 2u, 0u);
        }
    }
    // forward weight fragments of this wave's 32 output columns: B[k][j] = W_g[j][k], 8 consecutive k per lane
    const int g2 = wave >> 2;                                      // 0: x Wx^T (columns 0..127), 1: agg Wn^T
    vec16 bw[2][8];
    const __amdgpu_buffer_rsrc_t rW = tm_rsrc(p.w2, (uint32_t)(2 * 128 * p.ldw2 * 2));
    auto load_fwd_frags = [&]() {
        const uint32_t wr = (uint32_t)(((int64_t)g2 * 128 + (wave & 3) * 32 + li) * p.ldw2 + 8 * lg) * 2u;
        const uint32_t ct1 = (uint32_t)(16 * p.ldw2) * 2u;          // (wave-uniform: an SGPR offset)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            bw[0][ks] = tm_bload(rW, wr, 64u * ks);
            bw[1][ks] = tm_bload(rW, wr + ct1, 64u * ks);
        }
    };
    if (N > 0) load_fwd_frags();
    // (scheduling barriers: every request above is issued before the first row is consumed, and the rows are consumed
    // one at a time in arrival order -- left alone, the scheduler unpacks all 8 x n bf16 of a lane to fp32 as they
    // land and sums afterwards: 200 live registers on top of the 42 requests' targets, i.e. spills)
    __builtin_amdgcn_sched_barrier(0);
    // fc.weight -> LDS, all 64 rows (zeros from row C on)
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int q = t + u * TM_T;
        *reinterpret_cast<vec16 *>(Ws + (q >> 6) * TM_LDF + (q & 63) * 4) = wf[u];
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- 1. neighbour mean + ReLU masks of the rows this lane loaded ------------------------------------------
    uint32_t mbits[(NB + 3) / 4];
    uint32_t xbits = 0;
    {
        float s[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = 0.f;
#pragma unroll
        for (int w = 0; w < (NB + 3) / 4; ++w) mbits[w] = 0;
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const bool valid = N > 0 || u < n;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = valid ? tail_elem(nb[u], e) : 0.f;
                s[e] += f;
                mbits[u >> 2] |= (f > 0.f ? 1u : 0u) << ((u & 3) * 8 + e);
            }
            // (row u is done with here: sums and mask word pinned, nothing of the row's unpacked values lives on)
            asm volatile("" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]),
                         "+v"(s[7]), "+v"(mbits[u >> 2]));
            __builtin_amdgcn_sched_barrier(0);
        }
        vec16 am;
        const float fn = (float)n;
#pragma unroll
        for (int e = 0; e < 8; e += 2) am[e >> 1] = tm_pack(s[e] / fn, s[e + 1] / fn);
#pragma unroll
        for (int e = 0; e < 8; ++e) xbits |= (tail_elem(xraw, e) > 0.f ? 1u : 0u) << e;
        // (pinned: left alone the compiler sinks the mask computation to the stores at the end and keeps the loaded
        // rows alive through the whole kernel)
#pragma unroll
        for (int w = 0; w < (NB + 3) / 4; ++w) asm volatile("" : "+v"(mbits[w]));
        asm volatile("" : "+v"(xbits));
        if (live) *reinterpret_cast<vec16 *>(p.agg + iw * TM_D + cg * 8) = am;
        const vec16 zero = {0u, 0u, 0u, 0u};
        *reinterpret_cast<vec16 *>(xs + sr * TM_LDH + cg * 8) = live ? xraw : zero;
        *reinterpret_cast<vec16 *>(as + sr * TM_LDH + cg * 8) = live ? am : zero;
    }
    if (N == 0) load_fwd_frags();
    const float my_bias = (lane < C) ? p.bfc[lane] : 0.f;
    const int64_t tg0 = (row0 + 2 * wave < B) ? tgt[row0 + 2 * wave] : -1;          // softmax phase: this wave's
    const int64_t tg1 = (row0 + 2 * wave + 1 < B) ? tgt[row0 + 2 * wave + 1] : -1;  // two seeds
    lds_barrier();
    if (p.stop == 1) return;

    // ---- 2. emb = [x Wx^T | agg Wn^T] on the matrix cores; row norms; z ----------------------------------------
    tm_f32x4 acc[2];
    acc[0] = tm_f32x4{0.f, 0.f, 0.f, 0.f};
    acc[1] = acc[0];
    {
        const uint16_t *arow = (g2 ? as : xs) + li * TM_LDH + 8 * lg;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const vec16 a = *reinterpret_cast<const vec16 *>(arow + 32 * ks);
            acc[0] = tm_mma_bf16(a, bw[0][ks], acc[0]);
            acc[1] = tm_mma_bf16(a, bw[1][ks], acc[1]);
        }
    }
    const int col0 = 32 * wave + li;                               // this lane's columns: col0, col0 + 16
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float q = tm_sum16(acc[0][r] * acc[0][r] + acc[1][r] * acc[1][r]);
        if (li == 0) red[wave * TM_S + 4 * lg + r] = q;
    }
    // backward weight fragments (both groups, this wave's 32 INPUT columns): B[k = j][i] = W_g[j][i] = w2t[g][i][j]
    vec16 bb[2][2][4];
    {
        const __amdgpu_buffer_rsrc_t rT = tm_rsrc(p.w2t, (uint32_t)(2 * TM_D * p.ldw2t * 2));
        const uint32_t wr = (uint32_t)((32 * wave + li) * p.ldw2t + 8 * lg) * 2u;
        const uint32_t ct1 = (uint32_t)(16 * p.ldw2t) * 2u, g1 = (uint32_t)(TM_D * p.ldw2t) * 2u;
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) bb[g][ct][ks] = tm_bload(rT, wr + g * g1 + ct * ct1, 64u * ks);
    }
    lds_barrier();
    float nrm[4], z[2][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float ss = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) ss += red[w * TM_S + 4 * lg + r];
        nrm[r] = fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            z[ct][r] = acc[ct][r] / nrm[r];
            zs[(4 * lg + r) * TM_LDF + col0 + 16 * ct] = z[ct][r];
        }
    }
    lds_barrier();
    if (p.stop == 2) return;

    // ---- 3. logits = z Wfc^T (fp32 matrix cores; this wave: 32 of the 256 k), softmax cross-entropy ------------
    {
        const float *za = zs + li * TM_LDF + 32 * wave + lg;
        float a[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) a[ks] = za[4 * ks];
        for (int tt = 0; tt < CT; ++tt) {                          // (one class tile at a time: no per-tile branches)
            const float *wb = Ws + (16 * tt + li) * TM_LDF + 32 * wave + lg;
            tm_f32x4 lacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) lacc = tm_mma_f32(a[ks], wb[4 * ks], lacc);
#pragma unroll
            for (int r = 0; r < 4; ++r) part[(wave * TM_S + 4 * lg + r) * TM_LDP + 16 * tt + li] = lacc[r];
        }
    }
    lds_barrier();
    const float invB = 1.f / (float)Bv;
#pragma unroll
    for (int q = 0; q < 2; ++q) {                                  // this wave's two seeds, lane = class
        const int s = 2 * wave + q;
        const int64_t i = row0 + s;
        const int64_t my_target = q ? tg1 : tg0;
        float logit = -INFINITY;
        if (lane < C) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += part[(w * TM_S + s) * TM_LDP + lane];
            logit = v + my_bias;
        }
        const float mx = tm_wave_max(logit);
        const float ex = (lane < C) ? expf(logit - mx) : 0.f;
        const float den = tm_wave_sum(ex);
        const bool ok = lane < C && i < Bv;
        dls[s * TM_LDP + lane] = ok ? (ex / den - ((int64_t)lane == my_target ? 1.f : 0.f)) * invB : 0.f;
        if (lane < C && i < B) p.preds[i * C + lane] = logit;
        const int tl = (my_target >= 0 && my_target < C) ? (int)my_target : 0;
        const float lt = __shfl(logit, tl, 64);
        if (lane == 0) lss[s] = (i < Bv && my_target >= 0 && my_target < C) ? -(lt - mx - logf(den)) : 0.f;
    }
    lds_barrier();
    if (p.stop == 3) return;

    // ---- 4. d z = d logits Wfc (this wave's 32 columns), d emb; d fc.weight = d logits^T z ----------------------
    {
        tm_f32x4 dz[2];
        dz[0] = tm_f32x4{0.f, 0.f, 0.f, 0.f};
        dz[1] = dz[0];
        const float *da = dls + li * TM_LDP + lg;
        const float *wb = Ws + lg * TM_LDF + col0;
        for (int ks = 0; ks < 4 * CT; ++ks) {                      // k = class 4 ks + lg (rows >= C are zero)
            const float a = da[4 * ks];
            dz[0] = tm_mma_f32(a, wb[4 * ks * TM_LDF], dz[0]);
            dz[1] = tm_mma_f32(a, wb[4 * ks * TM_LDF + 16], dz[1]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float q = tm_sum16(z[0][r] * dz[0][r] + z[1][r] * dz[1][r]);
            if (li == 0) red2[wave * TM_S + 4 * lg + r] = q;
        }
        lds_barrier();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float zdz = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) zdz += red2[w * TM_S + 4 * lg + r];
            const bool in = row0 + 4 * lg + r < B;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const float gb = (dz[ct][r] - z[ct][r] * zdz) / nrm[r];
                des[(4 * lg + r) * TM_LDH + col0 + 16 * ct] = in ? (uint16_t)tm_bf16(gb) : (uint16_t)0;
            }
        }
    }
    {
        float *out = p.partial + (int64_t)blockIdx.x * ((int64_t)C * TM_D + C + 1);
        for (int rt = 0; rt < CT; ++rt) {                          // d W[c][col] = sum over the 16 seeds (k = seed)
            tm_f32x4 wacc[2];
            wacc[0] = tm_f32x4{0.f, 0.f, 0.f, 0.f};
            wacc[1] = wacc[0];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const float a = dls[(4 * ks + lg) * TM_LDP + 16 * rt + li];
                const float *zb = zs + (4 * ks + lg) * TM_LDF + col0;
                wacc[0] = tm_mma_f32(a, zb[0], wacc[0]);
                wacc[1] = tm_mma_f32(a, zb[16], wacc[1]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * rt + 4 * lg + r;
                if (c < C) {
                    out[c * TM_D + col0] = wacc[0][r];
                    out[c * TM_D + col0 + 16] = wacc[1][r];
                }
            }
        }
        if (wave == 0) {                                           // fc.bias / loss partials (fixed order)
            if (lane < C) {
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < TM_S; ++q) s += dls[q * TM_LDP + lane];
                out[C * TM_D + lane] = s;
            }
            if (lane == 0) {
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < TM_S; ++q) s += lss[q];
                out[C * TM_D + C] = s;
            }
        }
    }
    lds_barrier();                                                 // des complete; part (-> gxs | gns) free
    if (p.stop == 4) return;

    // ---- 5. d emb rows to HBM; input gradients dX = dE[:, :128] Wx, dA = dE[:, 128:] Wn --------------------------
    if (live) *reinterpret_cast<vec16 *>(p.dE + iw * TM_D + cg * 8) = *reinterpret_cast<const vec16 *>(des + sr * TM_LDH + cg * 8);
    {
        tm_f32x4 ga[2][2];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) ga[g][ct] = tm_f32x4{0.f, 0.f, 0.f, 0.f};
        const uint16_t *drow = des + li * TM_LDH + 8 * lg;
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const vec16 a = *reinterpret_cast<const vec16 *>(drow + 128 * g + 32 * ks);
                ga[g][0] = tm_mma_bf16(a, bb[g][0][ks], ga[g][0]);
                ga[g][1] = tm_mma_bf16(a, bb[g][1][ks], ga[g][1]);
            }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                gxs[(4 * lg + r) * TM_LDF + col0 + 16 * ct] = ga[0][ct][r];
                gns[(4 * lg + r) * TM_LDF + col0 + 16 * ct] = ga[1][ct][r];
            }
    }
    lds_barrier();
    if (p.stop == 5) return;

    // ---- 6. previous level's gradient rows of this half-wave's seed, ReLU masks applied ------------------------
    if (live) {
        const float inv_n = 1.f / (float)n;
        const float4 x0 = *reinterpret_cast<const float4 *>(gxs + sr * TM_LDF + cg * 8);
        const float4 x1 = *reinterpret_cast<const float4 *>(gxs + sr * TM_LDF + cg * 8 + 4);
        const float4 a0 = *reinterpret_cast<const float4 *>(gns + sr * TM_LDF + cg * 8);
        const float4 a1 = *reinterpret_cast<const float4 *>(gns + sr * TM_LDF + cg * 8 + 4);
        const float gx[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        const float gz[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        uint32_t zb[8];                                            // bf16 bits of the neighbours' common gradient
#pragma unroll
        for (int q = 0; q < 8; ++q) zb[q] = tm_bf16(gz[q] * inv_n);
        {
            vec16 o;
#pragma unroll
            for (int q = 0; q < 8; q += 2)
                o[q >> 1] = tm_pack(((xbits >> q) & 1u) ? gx[q] : 0.f, ((xbits >> (q + 1)) & 1u) ? gx[q + 1] : 0.f);
            *reinterpret_cast<vec16 *>(p.dH + iw * TM_D + cg * 8) = o;
        }
        uint16_t *base = p.dH + (B + iw * n) * TM_D + cg * 8;
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            if (N > 0 || u < n) {
                const uint32_t m = mbits[u >> 2] >> ((u & 3) * 8);
                vec16 o;
#pragma unroll
                for (int q = 0; q < 8; q += 2)
                    o[q >> 1] = (((m >> q) & 1u) ? zb[q] : 0u) | ((((m >> (q + 1)) & 1u) ? zb[q + 1] : 0u) << 16);
                *reinterpret_cast<vec16 *>(base + (int64_t)u * TM_D) = o;
            }
        }
    }
}

}  // namespace gsage

using namespace gsage;

extern "C" {

int64_t gsage_mean_tail_mfma_scratch(int32_t B, int32_t C)
{
    const int64_t n_wg = (B + TM_S - 1) / TM_S;
    return n_wg * ((int64_t)C * TM_D + C + 1);
}

// Workgroups the sampler role adds to the launch for a batch of B_hops seeds whose widest hop holds `widest` ids per
// seed (GSAGE_TAIL_SMP_WGS; default 32: sixteen seeds each at B = 512): every workgroup of this launch owns a CU, so
// they come out of the gather role's.  0: a workgroup's two frontier buffers would not fit the launch's LDS.
int32_t gsage_mean_tail_mfma_sampler_wgs(int64_t B_hops, int64_t widest)
{
    const char *e = getenv("GSAGE_TAIL_SMP_WGS");
    const int x = e ? atoi(e) : 32;
    const int want = x >= 1 && x <= 128 ? x : 32;
    if (B_hops <= 0 || widest <= 0) return 0;
    const int64_t spw = ceil_div(B_hops, (int64_t)want);
    if (sizeof(int64_t) * 2 * (size_t)spw * (size_t)widest > tm_lds_bytes()) return 0;
    return (int32_t)ceil_div(B_hops, spw);
}

int gsage_mean_tail_mfma(const void *H, int32_t B, int32_t n, const void *w2, int64_t ldw2,
                         const void *w2t, int64_t ldw2t, const float *Wfc, const float *bfc, int32_t C,
                         const int64_t *targets, const int64_t *batch_idx, int64_t n_batches, void *agg,
                         void *dE, float *preds, void *dH, float *partial, const gsage_tail_gather_desc *gather,
                         void *stream)
{
    const int32_t *n_valid = take_head_n_valid();     // (consumed before any return path: never left for a later launch)
    const gsage_hops_desc *hd = t_hops_role;          // gsage_hops_role_next(): likewise
    t_hops_role = nullptr;
    GSAGE_REQUIRE(H && w2 && w2t && Wfc && bfc && targets && agg && dE && preds && dH && partial,
                  "mean_tail_mfma: null pointer");
    GSAGE_REQUIRE(B > 0 && n >= 1 && n <= 32 && C >= 1 && C <= TM_CMAX,
                  "mean_tail_mfma: needs fan-out <= 32 and n_classes <= %d", TM_CMAX);
    GSAGE_REQUIRE(ldw2 >= TM_D && ldw2t >= 128 && ldw2 % 8 == 0 && ldw2t % 8 == 0,
                  "mean_tail_mfma: operand copies too narrow or rows not 16-byte multiples");
    GSAGE_REQUIRE((((uintptr_t)H | (uintptr_t)w2 | (uintptr_t)w2t | (uintptr_t)Wfc | (uintptr_t)agg | (uintptr_t)dE |
                    (uintptr_t)dH) & 15) == 0, "mean_tail_mfma: buffers must be 16-byte aligned");
    GSAGE_REQUIRE(!batch_idx || n_batches > 0, "mean_tail_mfma: bad target queue");
    GSAGE_REQUIRE((int64_t)B * (1 + n) * TM_D * 2 < ((int64_t)1 << 31) && 2 * 128 * ldw2 * 2 < ((int64_t)1 << 31) &&
                  2 * TM_D * ldw2t * 2 < ((int64_t)1 << 31),
                  "mean_tail_mfma: H and the operand copies are addressed with 32-bit buffer offsets (< 2 GiB each)");
    TailMfmaParams p;
    p.H = (const uint16_t *)H; p.w2 = (const uint16_t *)w2; p.w2t = (const uint16_t *)w2t;
    p.Wfc = Wfc; p.bfc = bfc; p.targets = targets; p.batch_idx = batch_idx; p.n_batches = n_batches;
    p.n_valid = n_valid;
    p.agg = (uint16_t *)agg; p.dE = (uint16_t *)dE; p.preds = preds; p.dH = (uint16_t *)dH;
    p.partial = partial; p.ldw2 = ldw2; p.ldw2t = ldw2t; p.B = B; p.n = n; p.C = C;
    {
        const char *e = getenv("GSAGE_TAIL_STOP");        // (phase timing, tools/kbench.py tailm: results are partial)
        p.stop = e ? atoi(e) : 0;
    }
    TailGather tg = {};
    const bool fused = gather && gather->rows > 0;
    if (fused) {
        const int rc = fill_gather_role(tg, *gather, "mean_tail_mfma");
        if (rc != GSAGE_OK) return rc;
    }
    HopsParams hp = {};
    p.n_smp = 0;
    if (hd) {
        // gsage_hops_role_next(): workgroups behind the seed-level ones sample a later batch's frontier
        GSAGE_REQUIRE(fused && !hd->dense_adj, "mean_tail_mfma: the sampler role rides beside the gather role and walks a CSR");
        size_t lds = 0;
        const int rc = fill_hops(hp, lds, *hd);
        if (rc != GSAGE_OK) return rc;
        int64_t widest = 1, width = 1;
        for (int k = 1; k <= hp.n_hops; ++k) { width *= hp.fan[k]; widest = width > widest ? width : widest; }
        p.n_smp = gsage_mean_tail_mfma_sampler_wgs(hd->B, widest);
        GSAGE_REQUIRE(p.n_smp > 0, "mean_tail_mfma: the sampler role's frontier (%lld ids per seed, twice) does not fit the "
                      "launch's LDS (gsage_mean_tail_mfma_sampler_wgs)", (long long)widest);
        hp.spw = (int32_t)ceil_div(hd->B, (int64_t)p.n_smp);
    }
    const int gn = fused ? gather->n : 0;
    // fan-outs with a specialisation of their own (one load slot per neighbour row): BASELINE's 25 (configs[1]) and
    // 15 (configs[4]'s first hop), 10; everything else takes the 32-slot kernel
    const int ex = n == 25 ? 1 : n == 15 ? 2 : n == 10 ? 3 : 0;
    typedef void (*kern_t)(const TailMfmaParams, const TailGather, const HopsParams);
    static const kern_t table[4][4] = {
        {k_mean_tail_mfma<0, 0>, k_mean_tail_mfma<0, 10>, k_mean_tail_mfma<0, 5>, k_mean_tail_mfma<0, 15>},
        {k_mean_tail_mfma<25, 0>, k_mean_tail_mfma<25, 10>, k_mean_tail_mfma<25, 5>, k_mean_tail_mfma<25, 15>},
        {k_mean_tail_mfma<15, 0>, k_mean_tail_mfma<15, 10>, k_mean_tail_mfma<15, 5>, k_mean_tail_mfma<15, 15>},
        {k_mean_tail_mfma<10, 0>, k_mean_tail_mfma<10, 10>, k_mean_tail_mfma<10, 5>, k_mean_tail_mfma<10, 15>}};
    const int gi = gn == 10 ? 1 : gn == 5 ? 2 : gn == 15 ? 3 : 0;
    kern_t kern = table[ex][gi];
    {   // more than the default 64 KB of dynamic LDS (gfx950 has 160 KB per CU)
        static bool raised[16] = {};
        const int slot = ex * 4 + gi;
        if (!raised[slot]) {
            if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)tm_lds_bytes()) != hipSuccess) {
                (void)hipGetLastError();
                set_error("mean_tail_mfma: cannot raise the dynamic LDS limit");
                return GSAGE_ELAUNCH;
            }
            raised[slot] = true;
        }
    }
    const unsigned n_tail = (unsigned)((B + TM_S - 1) / TM_S);
    launch(kern, dim3(n_tail + (unsigned)p.n_smp + (fused ? (unsigned)tg.n_wg : 0u)), dim3(TM_T), tm_lds_bytes(),
           (hipStream_t)stream, p, tg, hp);
    return check_launch("mean_tail_mfma");
}

}  // extern "C"
