// gsage_tail.hip -- the whole "seed level" of a mean-aggregator GraphSAGE step in ONE launch.
//
// At the last SAGE level only the B seed rows are live (B = 512 by default): the reference runs
//     agg   = neibs.view(B, n, D).mean(1)                         nn_modules.py:197-198
//     emb   = cat[fc_x(x), fc_neib(agg)]        (identity act)    nn_modules.py:200-202
//     preds = fc(F.normalize(emb));  loss = CE(preds, y)          models.py:90-91, problem.py:34
// and autograd walks it back to the previous level's activations (models.py:100).  As separate
// kernels that is 5 dependent launches (segment mean, GEMM, head, input-gradient GEMM, mask/route),
// each paying a ~5 us launch + first-load + store-drain floor for a few MFLOP.  Here a workgroup
// owns R = 4 seeds end to end.
//
// Every global access is 16 bytes per lane.  A wave-wide load costs the texture addresser ~16 clocks
// whatever its width, so 2-byte-per-lane loads (one column per thread, the obvious layout) moved
// 128 B per instruction and the first version of this kernel spent 16 + 22 us just streaming the
// 256 KB of projection weights per workgroup (ablation: tools/kbench.py tail).  Layouts used:
//   * rows of H / dH:   lane&31 owns 8 consecutive columns, wave w owns seed w, the two half-waves
//                       take alternate neighbours (all <= 16 row requests of a lane are in flight
//                       at once; ReLU masks of the rows stay in registers as bit masks),
//   * projections:      lane&31 owns 8 output columns, the 8 half-waves split the reduction
//                       dimension; partial sums meet in LDS (W^T copies for the forward, W for
//                       the backward: both row-contiguous, so every weight load is a full 512 B row
//                       segment per half-wave),
//   * head:             thread t owns column t (the k_head_ce body on registers).
// Only the weight gradient of this level (K5b, reads agg + dE written here) stays a launch of its
// own.  Fixed shape: previous level width 256 (= 2 x 128), this level 2h = 256, n <= 32, C <= 64.
#include "gsage_common.h"
#include "gsage_gather_dev.h"

namespace gsage {

constexpr int TAIL_D = 256;        // width of the previous level's rows and of this level's output
constexpr int TAIL_R = 4;          // seeds per workgroup (= waves per workgroup)
constexpr int TAIL_CMAX = 64;

// Storage type T of the activations and of the weight operand copies: uint16_t = bf16 (the
// production path) or float (the exact-arithmetic parity mode: the SAME kernel source replayed on the
// reference-generated golden fixtures at fp32 tolerance; every bf16 rounding point becomes a no-op).
struct TailParams {
    const void *H;           // previous level output, T [B*(1+n), 256]: seeds first, then neighbours
    const void *w2;          // T [2, 128, ldw2]:  fc_x | fc_neib           (rows = outputs)
    const void *w2t;         // T [2, 256, ldw2t]: transposed copies        (rows = inputs)
    const float *Wfc;        // [C, 256]
    const float *bfc;        // [C]
    const int64_t *targets;
    const int64_t *batch_idx;
    int64_t n_batches;
    const int32_t *n_valid;  // optional: live seeds of the batch (one word, or [n_batches] with batch_idx)
    void *agg;               // out: T [B, 256] neighbour means (A operand of this level's K5b)
    void *dE;                // out: T [B, 256] d loss / d emb   (dC operand of this level's K5b)
    float *preds;            // out: [B, C] logits
    void *dH;                // out: T [B*(1+n), 256] gradient w.r.t. H (ReLU mask applied)
    float *partial;          // out: [grid, C*256 + C + 1] fc.weight / fc.bias / loss partials
    int64_t ldw2, ldw2t;
    int32_t B, n, C;
};

__device__ __forceinline__ float tail_wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float tail_wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// eight consecutive elements of a row, as one lane holds them
template <typename T> struct row8;
template <> struct row8<uint16_t> {
    vec16 v;
    __device__ __forceinline__ void load(const uint16_t *p) { v = *(const vec16 *)p; }
    __device__ __forceinline__ float get(int e) const { return tail_elem(v, e); }
};
template <> struct row8<float> {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 a, b;
    __device__ __forceinline__ void load(const float *p) { a = *(const f4 *)p; b = *(const f4 *)(p + 4); }
    __device__ __forceinline__ float get(int e) const { return e < 4 ? a[e & 3] : b[e & 3]; }
};
// the value a store of x in type T holds
template <typename T> __device__ __forceinline__ float tail_round(float x);
template <> __device__ __forceinline__ float tail_round<uint16_t>(float x) { return bf16_to_f32(f32_to_bf16(x)); }
template <> __device__ __forceinline__ float tail_round<float>(float x) { return x; }
__device__ __forceinline__ void tail_store1(uint16_t *p, float x) { *p = f32_to_bf16(x); }
__device__ __forceinline__ void tail_store1(float *p, float x) { *p = x; }
__device__ __forceinline__ void tail_store8(uint16_t *p, const float (&x)[8])
{
    vec16 o;
#pragma unroll
    for (int q = 0; q < 8; q += 2) o[q >> 1] = pack_bf16x2(x[q], x[q + 1]);
    *(vec16 *)p = o;
}
__device__ __forceinline__ void tail_store8(float *p, const float (&x)[8])
{
    *(float4 *)p = make_float4(x[0], x[1], x[2], x[3]);
    *(float4 *)(p + 4) = make_float4(x[4], x[5], x[6], x[7]);
}
// the same for values that tail_round<T> produced (or zero): bf16 is then the upper half of the word,
// no rounding arithmetic per stored element (a lane stores up to 17 rows)
__device__ __forceinline__ void tail_store8_exact(uint16_t *p, const float (&x)[8])
{
    vec16 o;
#pragma unroll
    for (int q = 0; q < 8; q += 2) o[q >> 1] = (__float_as_uint(x[q]) >> 16) | (__float_as_uint(x[q + 1]) & 0xffff0000u);
    *(vec16 *)p = o;
}
__device__ __forceinline__ void tail_store8_exact(float *p, const float (&x)[8]) { tail_store8(p, x); }

// fc.weight rows in LDS are padded to a multiple of 8 classes (zero rows): 8-class chunks need no
// per-class guards, and every buffer behind them stays 16-byte aligned for ds_read_b128
constexpr int tail_cpad(int C) { return (C + 7) & ~7; }

constexpr size_t tail_lds_floats(int C)
{
    return (size_t)tail_cpad(C) * (TAIL_D + 1) + 4 * TAIL_R * TAIL_D + 4 * TAIL_R * TAIL_CMAX + TAIL_R * TAIL_CMAX +
           4 * TAIL_R + 4 + 2 * 8 * TAIL_R * TAIL_D;
}

// (the gather role some workgroups of this launch play: gsage_gather_dev.h)
// NBH = neighbour rows per half-wave lane (n <= 2 * NBH); GN = fan-out of the gather role (0: none)
template <typename T, int NBH, int GN>
__global__ void __launch_bounds__(256)
k_mean_tail_ce(const TailParams p, const TailGather tg)
{
    const T *const pH = (const T *)p.H, *const pw2 = (const T *)p.w2, *const pw2t = (const T *)p.w2t;
    T *const pagg = (T *)p.agg, *const pdE = (T *)p.dE, *const pdH = (T *)p.dH;
    if (GN > 0) {
        const int n_tail = (p.B + TAIL_R - 1) / TAIL_R;
        if ((int)blockIdx.x >= n_tail) {
            gather_role<(GN > 0 ? GN : 1), 4>(tg, (int)blockIdx.x - n_tail);
            return;
        }
    }
    constexpr int R = TAIL_R, D = TAIL_D;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int C = p.C;
    const int ldw = D + 1;
    const int Cp = tail_cpad(C);
    float *Ws = lds;                              // [Cp][257] fc.weight, rows >= C zero
    float *xs = Ws + Cp * ldw;                    // [R][256] seed rows
    float *as = xs + R * D;                       // [R][256] neighbour means
    float *zs = as + R * D;                       // [R][256] normalised embeddings
    float *des = zs + R * D;                      // [R][256] d emb (bf16-rounded)
    float *part = des + R * D;                    // [4][R][64]
    float *dls = part + 4 * R * TAIL_CMAX;        // [R][64]
    float *red = dls + R * TAIL_CMAX;             // [4*R]
    float *lss = red + 4 * R;                     // [R] per-seed loss terms
    float *big = lss + 4;                         // [2][8][R][256] split-reduction partial sums
    big = (float *)(((uintptr_t)big + 15) & ~(uintptr_t)15);

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int cg = t & 31;                        // this lane's 8 columns: cg*8 .. cg*8+7
    const int slot = t >> 5;                      // half-wave index 0..7
    const int half = slot & 1;
    const int row0 = blockIdx.x * R;
    const int n = p.n;
    const int64_t B = p.B;

    // ---- 0. every request that depends on nothing, all in flight together -------------------------
    const int64_t bq = p.batch_idx ? (int64_t)((uint64_t)*p.batch_idx % (uint64_t)p.n_batches) : 0;
    const int64_t *tgt = p.targets + bq * B;
    // seeds past Bv are padding (the reference's chunks are not all of one size): no loss, no gradient
    const int64_t Bv = p.n_valid ? (int64_t)min(max(p.n_valid[bq], 1), (int32_t)B) : B;
    const int64_t iw = row0 + wave;               // this wave's seed
    const bool live = iw < B;
    const int64_t iwc = live ? iw : B - 1;        // clamped: loads stay unconditional
    const int64_t my_target = live ? tgt[iwc] : -1;
    const float my_bias = (lane < C) ? p.bfc[lane] : 0.f;
    row8<T> xraw;
    xraw.load(pH + iwc * D + cg * 8);
    row8<T> nb[NBH];
    {
        const T *base = pH + (B + iwc * n) * D + cg * 8;
#pragma unroll
        for (int u = 0; u < NBH; ++u) {
            const int j = half + 2 * u;
            nb[u].load(base + (int64_t)(j < n ? j : n - 1) * D);
        }
    }
    {
        const int nq = C * (D / 4);               // C <= 64 rows x 64 float4 = at most 16 per thread
#pragma unroll
        for (int ub = 0; ub < 16; ub += 8) {
            float4 wf[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = t + (ub + u) * 256;
                wf[u] = *(const float4 *)(p.Wfc + (int64_t)(q < nq ? q : nq - 1) * 4);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = t + (ub + u) * 256;
                if (q < nq) {
                    float *d = Ws + (q >> 6) * ldw + (q & 63) * 4;
                    d[0] = wf[u].x; d[1] = wf[u].y; d[2] = wf[u].z; d[3] = wf[u].w;
                }
            }
        }
    }

    for (int q = t; q < (Cp - C) * ldw; q += 256) Ws[C * ldw + q] = 0.f;

    // ---- 1. neighbour mean of this wave's seed + ReLU masks of the rows this lane loaded -----------
    uint32_t mbits[(NBH + 3) / 4];
    uint32_t xbits = 0;
    {
        float s[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = 0.f;
#pragma unroll
        for (int w = 0; w < (NBH + 3) / 4; ++w) mbits[w] = 0;
#pragma unroll
        for (int u = 0; u < NBH; ++u) {
            const bool valid = half + 2 * u < n;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = valid ? nb[u].get(e) : 0.f;
                s[e] += f;
                mbits[u >> 2] |= (f > 0.f ? 1u : 0u) << ((u & 3) * 8 + e);
            }
        }
        float af[8], am[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s[e] += __shfl_xor(s[e], 32, 64);
            am[e] = tail_round<T>(s[e] / (float)n);                   // the GEMM operand has type T
            af[e] = live ? am[e] : 0.f;
            const float xf = xraw.get(e);
            xbits |= (xf > 0.f ? 1u : 0u) << e;
        }
        // pin the masks here: left alone the compiler sinks their computation to the stores at the
        // end and keeps the 16 loaded rows alive (64 VGPRs) through the whole kernel
#pragma unroll
        for (int w = 0; w < (NBH + 3) / 4; ++w) asm volatile("" : "+v"(mbits[w]));
        asm volatile("" : "+v"(xbits));
        if (half == 0) {
            if (live) tail_store8_exact(pagg + iw * D + cg * 8, am);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                xs[wave * D + cg * 8 + e] = live ? xraw.get(e) : 0.f;
                as[wave * D + cg * 8 + e] = af[e];
            }
        }
    }
    lds_barrier();

    // ---- 2. emb = [x Wx^T | agg Wn^T]: lane = 8 output columns, half-wave `slot` = 32 of the 256 k ---
    {
        float acc[R][8];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[r][e] = 0.f;
        const int g = cg >> 4;
        const T *wt = pw2t + (int64_t)g * D * p.ldw2t + (cg & 15) * 8 + (int64_t)slot * 32 * p.ldw2t;
        const float *in = (g ? as : xs) + slot * 32;
        // weight rows in flight per lane: 16 x 16 B (bf16); the fp32 instantiation takes half as many
        // rows per batch so that it needs no more registers (it must not spill, see the note on dls)
        constexpr int WB = sizeof(T) == 2 ? 16 : 8;
#pragma unroll
        for (int kb = 0; kb < 32; kb += WB) {
            row8<T> w[WB];
#pragma unroll
            for (int u = 0; u < WB; ++u) w[u].load(wt + (int64_t)(kb + u) * p.ldw2t);
#pragma unroll
            for (int u4 = 0; u4 < WB; u4 += 4) {
                float4 iv[R];
#pragma unroll
                for (int r = 0; r < R; ++r) iv[r] = *(const float4 *)(in + r * D + kb + u4);
#pragma unroll
                for (int uu = 0; uu < 4; ++uu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float wv = w[u4 + uu].get(e);
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const float a = uu == 0 ? iv[r].x : uu == 1 ? iv[r].y : uu == 2 ? iv[r].z : iv[r].w;
                            acc[r][e] += a * wv;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float *d = big + (slot * R + r) * D + cg * 8;
            *(float4 *)d = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
            *(float4 *)(d + 4) = make_float4(acc[r][4], acc[r][5], acc[r][6], acc[r][7]);
        }
    }
    lds_barrier();
    float e[R];                                   // from here to the end of the head: thread t = column t
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float v = 0.f;
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) v += big[(sl * R + r) * D + t];
        e[r] = v;
    }

    auto block_sum_rows = [&](float (&v)[R]) {
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = tail_wave_sum(v[r]);
        lds_barrier();
        if (lane == 0)
#pragma unroll
            for (int r = 0; r < R; ++r) red[wave * R + r] = v[r];
        lds_barrier();
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = (red[r] + red[R + r]) + (red[2 * R + r] + red[3 * R + r]);
    };

    // ---- 3. head: normalise, fc, softmax cross-entropy, gradients (k_head_ce body, KPT = 1) --------
    const float invB = 1.f / (float)Bv;
    float z[R], ss[R], nrm[R];
#pragma unroll
    for (int r = 0; r < R; ++r) ss[r] = e[r] * e[r];
    block_sum_rows(ss);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        nrm[r] = fmaxf(sqrtf(ss[r]), 1e-12f);
        z[r] = e[r] / nrm[r];
        zs[r * D + t] = z[r];
    }
    lds_barrier();
    {
        float s[R];
#pragma unroll
        for (int r = 0; r < R; ++r) s[r] = 0.f;
        if (lane < C) {
            const int k0 = wave * (D / 4), k1 = k0 + D / 4;
            const float *wr = Ws + lane * ldw;
            for (int k = k0; k < k1; k += 8) {            // 8 weights + 8 float4 of z per round trip
                float w[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) w[u] = wr[k + u];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float4 a = *(const float4 *)(zs + r * D + k);
                    const float4 b = *(const float4 *)(zs + r * D + k + 4);
                    s[r] += a.x * w[0] + a.y * w[1] + a.z * w[2] + a.w * w[3] +
                            b.x * w[4] + b.y * w[5] + b.z * w[6] + b.w * w[7];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) part[(wave * R + r) * TAIL_CMAX + lane] = s[r];
    }
    lds_barrier();
    // (d logits and the loss terms go to LDS at once and stay there until the partials are written at
    // the end: as per-lane accumulators they lived across the whole input-gradient phase, and in the fp32
    // instantiation hipcc spilled them to AGPRs INSIDE a divergent region -- lanes outside it lost them)
    {
        const int r = wave;                                   // R == 4 waves: wave r <-> row r
        const int64_t i = row0 + r;
        const bool ok = lane < C && i < Bv;
        float logit = -INFINITY;
        if (lane < C)
            logit = part[(0 * R + r) * TAIL_CMAX + lane] + part[(1 * R + r) * TAIL_CMAX + lane] +
                    part[(2 * R + r) * TAIL_CMAX + lane] + part[(3 * R + r) * TAIL_CMAX + lane] + my_bias;
        const float mx = tail_wave_max(logit);
        const float ex = (lane < C) ? expf(logit - mx) : 0.f;
        const float den = tail_wave_sum(ex);
        const float dl = ok ? (ex / den - ((int64_t)lane == my_target ? 1.f : 0.f)) * invB : 0.f;
        dls[r * TAIL_CMAX + lane] = dl;
        if (lane < C && i < B) p.preds[i * C + lane] = logit;
        const int tl = (my_target >= 0 && my_target < C) ? (int)my_target : 0;
        const float lt = __shfl(logit, tl, 64);                   // the target's logit (wave-uniform index)
        if (lane == 0) lss[r] = (i < Bv && my_target >= 0 && my_target < C) ? -(lt - mx - logf(den)) : 0.f;
    }
    lds_barrier();
    float dz[R], zdz[R];
#pragma unroll
    for (int r = 0; r < R; ++r) dz[r] = 0.f;
    float accW[TAIL_CMAX];
    {
        // d logits of the 4 rows: lane c keeps class c, v_readlane broadcasts them as scalars (the
        // obvious dls[r][c] LDS reads are a dependent ~100-clock round trip each with one wave per SIMD)
        int dlv[R];
#pragma unroll
        for (int r = 0; r < R; ++r) dlv[r] = __float_as_int(dls[r * TAIL_CMAX + lane]);
#pragma unroll
        for (int c0 = 0; c0 < TAIL_CMAX; c0 += 8) {
            if (c0 < C) {                                     // uniform; rows up to Cp exist (zeros)
                float w[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) w[u] = Ws[(c0 + u) * ldw + t];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    float acc = 0.f;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const float dl = __int_as_float(__builtin_amdgcn_readlane(dlv[r], c0 + u));
                        dz[r] += dl * w[u];
                        acc += dl * z[r];
                    }
                    accW[c0 + u] = acc;
                }
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) accW[c0 + u] = 0.f;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) zdz[r] = z[r] * dz[r];
    block_sum_rows(zdz);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t i = row0 + r;
        const float gb = tail_round<T>((dz[r] - z[r] * zdz[r]) / nrm[r]);
        if (i < B) tail_store1(pdE + i * D + t, gb);
        des[r * D + t] = (i < B) ? gb : 0.f;                   // K5b and the products below see type T
    }
    {
        float *out = p.partial + (int64_t)blockIdx.x * ((int64_t)C * D + C + 1);
#pragma unroll
        for (int c = 0; c < TAIL_CMAX; ++c)
            if (c < C) out[c * D + t] = accW[c];
    }
    lds_barrier();                                            // des complete; part / red free again

    // ---- 4. input gradients dX = dE[:, :128] Wx, dA = dE[:, 128:] Wn: lane = 8 columns of H,
    //         half-wave `slot` = 16 of the 128 rows of each weight ------------------------------------
    {
        float ax[R][8], aa[R][8];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int q = 0; q < 8; ++q) { ax[r][q] = 0.f; aa[r][q] = 0.f; }
        const T *wx = pw2 + (int64_t)slot * 16 * p.ldw2 + cg * 8;
        const T *wn = wx + (int64_t)128 * p.ldw2;
        const float *dx = des + slot * 16, *dn = des + 128 + slot * 16;
        constexpr int CB = sizeof(T) == 2 ? 8 : 4;
#pragma unroll
        for (int cb = 0; cb < 16; cb += CB) {
            row8<T> a[CB], b[CB];
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                a[u].load(wx + (int64_t)(cb + u) * p.ldw2);
                b[u].load(wn + (int64_t)(cb + u) * p.ldw2);
            }
#pragma unroll
            for (int u4 = 0; u4 < CB; u4 += 4) {
                float4 vx[R], vn[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    vx[r] = *(const float4 *)(dx + r * D + cb + u4);
                    vn[r] = *(const float4 *)(dn + r * D + cb + u4);
                }
#pragma unroll
                for (int uu = 0; uu < 4; ++uu) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float fa = a[u4 + uu].get(q), fb = b[u4 + uu].get(q);
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const float gx = uu == 0 ? vx[r].x : uu == 1 ? vx[r].y : uu == 2 ? vx[r].z : vx[r].w;
                            const float gn = uu == 0 ? vn[r].x : uu == 1 ? vn[r].y : uu == 2 ? vn[r].z : vn[r].w;
                            ax[r][q] += gx * fa;
                            aa[r][q] += gn * fb;
                        }
                    }
                }
            }
        }
        float *bx = big, *ba = big + 8 * R * D;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float *d = bx + (slot * R + r) * D + cg * 8;
            *(float4 *)d = make_float4(ax[r][0], ax[r][1], ax[r][2], ax[r][3]);
            *(float4 *)(d + 4) = make_float4(ax[r][4], ax[r][5], ax[r][6], ax[r][7]);
            d = ba + (slot * R + r) * D + cg * 8;
            *(float4 *)d = make_float4(aa[r][0], aa[r][1], aa[r][2], aa[r][3]);
            *(float4 *)(d + 4) = make_float4(aa[r][4], aa[r][5], aa[r][6], aa[r][7]);
        }
    }
    lds_barrier();
    // ---- 5. previous level's gradient rows of this wave's seed, ReLU mask applied -------------------
    if (live) {
        const float inv_n = 1.f / (float)n;
        float gx[8], gz[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { gx[q] = 0.f; gz[q] = 0.f; }
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) {
            const float *sx = big + (sl * R + wave) * D + cg * 8;
            const float *sa = sx + 8 * R * D;
            const float4 x0 = *(const float4 *)sx, x1 = *(const float4 *)(sx + 4);
            const float4 a0 = *(const float4 *)sa, a1 = *(const float4 *)(sa + 4);
            gx[0] += x0.x; gx[1] += x0.y; gx[2] += x0.z; gx[3] += x0.w;
            gx[4] += x1.x; gx[5] += x1.y; gx[6] += x1.z; gx[7] += x1.w;
            gz[0] += a0.x; gz[1] += a0.y; gz[2] += a0.z; gz[3] += a0.w;
            gz[4] += a1.x; gz[5] += a1.y; gz[6] += a1.z; gz[7] += a1.w;
        }
        float zb[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) zb[q] = tail_round<T>(gz[q] * inv_n);
        if (half == 0) {
            float o[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) o[q] = ((xbits >> q) & 1u) ? gx[q] : 0.f;
            tail_store8(pdH + iw * D + cg * 8, o);
        }
        T *base = pdH + (B + iw * n) * D + cg * 8;
#pragma unroll
        for (int u = 0; u < NBH; ++u) {
            const int j = half + 2 * u;
            if (j < n) {
                const uint32_t m = mbits[u >> 2] >> ((u & 3) * 8);
                float o[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) o[q] = ((m >> q) & 1u) ? zb[q] : 0.f;
                tail_store8_exact(base + (int64_t)j * D, o);
            }
        }
    }
    // ---- 6. fc.bias / loss partials ----------------------------------------------------------------
    if (wave == 0) {              // dls / lss: written before the barriers of the phases above, never since
        float *out = p.partial + (int64_t)blockIdx.x * ((int64_t)C * D + C + 1);
        if (lane < C)
            out[C * D + lane] = (dls[lane] + dls[TAIL_CMAX + lane]) + (dls[2 * TAIL_CMAX + lane] + dls[3 * TAIL_CMAX + lane]);
        if (lane == 0) out[C * D + C] = (lss[0] + lss[1]) + (lss[2] + lss[3]);
    }
}

}  // namespace gsage

using namespace gsage;

extern "C" {

int64_t gsage_mean_tail_ce_scratch(int32_t B, int32_t C)
{
    const int64_t n_wg = (B + TAIL_R - 1) / TAIL_R;
    return n_wg * ((int64_t)C * TAIL_D + C + 1);
}

int gsage_mean_tail_ce(const void *H, int32_t B, int32_t n, const void *w2, int64_t ldw2,
                       const void *w2t, int64_t ldw2t, const float *Wfc, const float *bfc, int32_t C,
                       const int64_t *targets, const int64_t *batch_idx, int64_t n_batches, void *agg,
                       void *dE, float *preds, void *dH, float *partial, const gsage_tail_gather_desc *gather,
                       int dtype, void *stream)
{
    const int32_t *n_valid = take_head_n_valid();     // (consumed before any return path: never left for a later launch)
    GSAGE_REQUIRE(dtype == GSAGE_BF16 || dtype == GSAGE_F32, "mean_tail_ce: bad dtype");
    GSAGE_REQUIRE(dtype == GSAGE_BF16 || !(gather && gather->rows > 0),
                  "mean_tail_ce: the gather role exists for bf16 tables only");
    GSAGE_REQUIRE(H && w2 && w2t && Wfc && bfc && targets && agg && dE && preds && dH && partial,
                  "mean_tail_ce: null pointer");
    GSAGE_REQUIRE(B > 0 && n >= 1 && n <= 32 && C >= 1 && C <= TAIL_CMAX,
                  "mean_tail_ce: needs fan-out <= 32 and n_classes <= %d", TAIL_CMAX);
    GSAGE_REQUIRE(ldw2 >= TAIL_D && ldw2t >= 128 && ldw2 % 8 == 0 && ldw2t % 8 == 0,
                  "mean_tail_ce: operand copies too narrow or rows not 16-byte multiples");
    GSAGE_REQUIRE((((uintptr_t)H | (uintptr_t)w2 | (uintptr_t)w2t | (uintptr_t)Wfc | (uintptr_t)agg |
                    (uintptr_t)dH) & 15) == 0, "mean_tail_ce: buffers must be 16-byte aligned");
    GSAGE_REQUIRE(!batch_idx || n_batches > 0, "mean_tail_ce: bad target queue");
    TailParams p;
    p.H = H; p.w2 = w2; p.w2t = w2t;
    p.Wfc = Wfc; p.bfc = bfc; p.targets = targets; p.batch_idx = batch_idx; p.n_batches = n_batches;
    p.n_valid = n_valid;
    p.agg = agg; p.dE = dE; p.preds = preds; p.dH = dH;
    p.partial = partial; p.ldw2 = ldw2; p.ldw2t = ldw2t; p.B = B; p.n = n; p.C = C;
    TailGather tg = {};
    const bool fused = gather && gather->rows > 0;
    if (fused) {
        const int rc = fill_gather_role(tg, *gather, "mean_tail_ce");
        if (rc != GSAGE_OK) return rc;
    }
    const size_t lds = sizeof(float) * tail_lds_floats(C) + 16;
    const int small = n <= 16;
    const int gn = fused ? gather->n : 0;
    void (*kern)(const TailParams, const TailGather) =
        dtype == GSAGE_F32 ? (small ? k_mean_tail_ce<float, 8, 0> : k_mean_tail_ce<float, 16, 0>)
        : gn == 10 ? (small ? k_mean_tail_ce<uint16_t, 8, 10> : k_mean_tail_ce<uint16_t, 16, 10>)
        : gn == 5 ? (small ? k_mean_tail_ce<uint16_t, 8, 5> : k_mean_tail_ce<uint16_t, 16, 5>)
        : gn == 15 ? (small ? k_mean_tail_ce<uint16_t, 8, 15> : k_mean_tail_ce<uint16_t, 16, 15>)
                   : (small ? k_mean_tail_ce<uint16_t, 8, 0> : k_mean_tail_ce<uint16_t, 16, 0>);
    {   // more than the default 64 KB of dynamic LDS (gfx950 has 160 KB per CU)
        static bool raised[10] = {false, false, false, false, false, false, false, false, false, false};
        const int slot = dtype == GSAGE_F32 ? 8 + small : small + 2 * (gn == 10 ? 1 : gn == 5 ? 2 : gn == 15 ? 3 : 0);
        if (!raised[slot]) {
            if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(sizeof(float) * tail_lds_floats(TAIL_CMAX) + 16)) != hipSuccess) {
                set_error("mean_tail_ce: cannot raise the dynamic LDS limit");
                return GSAGE_ELAUNCH;
            }
            raised[slot] = true;
        }
    }
    const unsigned n_tail = (unsigned)((B + TAIL_R - 1) / TAIL_R);
    launch(kern, dim3(n_tail + (fused ? (unsigned)tg.n_wg : 0u)), dim3(256), lds, (hipStream_t)stream, p, tg);
    return check_launch("mean_tail_ce");
}

}  // extern "C"
