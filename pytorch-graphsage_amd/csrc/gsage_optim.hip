// gsage_optim.hip -- the elementwise tail of train_step fused into three launches (gfx950):
//   k_grad_sqnorm   sum of squares of the flat gradient bucket (per-block partials, deterministic)
//   k_adam_clip     clip_grad_norm(5) + Adam update over the flat parameter bucket
//   k_prep_weights  fp32 parameters -> bf16 operand copies ([N, ld] padded, and transposed) for K5
//   k_bwd_merge     ReLU mask + routing of a layer's input gradient back to the hop rows
//
// Replaces, for the fused engine, what the reference does with ~60 tiny framework kernels per
// step: torch.nn.utils.clip_grad_norm(params, 5) and optimizer.step() (reference models.py:101-102;
// Adam betas (0.9, 0.999), eps 1e-8, L2 weight decay as at models.py:69) and autograd's
// relu / mean / index backward between layers (models.py:85-86).  All HBM-bound, 16-byte lanes.
#include "gsage_common.h"
#include "gsage_sample_dev.h"
#include "gsage_optim_dev.h"

namespace gsage {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// 4 consecutive columns of an activation row in storage type T (uint16_t = bf16, float = the
// exact-arithmetic parity mode): load as fp32, store from fp32
template <typename T> __device__ __forceinline__ f32x4 load4(const T *p);
template <> __device__ __forceinline__ f32x4 load4<uint16_t>(const uint16_t *p)
{
    const uint2 h = *reinterpret_cast<const uint2 *>(p);
    f32x4 r;
    r[0] = bf16_to_f32((uint16_t)(h.x & 0xffff)); r[1] = bf16_to_f32((uint16_t)(h.x >> 16));
    r[2] = bf16_to_f32((uint16_t)(h.y & 0xffff)); r[3] = bf16_to_f32((uint16_t)(h.y >> 16));
    return r;
}
template <> __device__ __forceinline__ f32x4 load4<float>(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
template <typename T> __device__ __forceinline__ void store4(T *p, const f32x4 v);
template <> __device__ __forceinline__ void store4<uint16_t>(uint16_t *p, const f32x4 v)
{
    uint2 o;
    o.x = pack_bf16x2(v[0], v[1]);
    o.y = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<uint2 *>(p) = o;
}
template <> __device__ __forceinline__ void store4<float>(float *p, const f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }

// partial[b] = sum over this block's grid-stride slice of g[i]^2
__global__ void __launch_bounds__(256)
k_grad_sqnorm(const float *__restrict__ g, int64_t n, float *__restrict__ partial)
{
    __shared__ float red[4];
    float s = 0.f;
    const int64_t stride = (int64_t)gridDim.x * 256;
    if ((n & 3) == 0 && ((uintptr_t)g & 15) == 0) {                     // 16-byte lanes
        typedef float v4 __attribute__((ext_vector_type(4)));
        const int64_t n4 = n >> 2;
        // four 16-byte loads in flight per lane: with one, the 418 MB gradient of a trainable embedding
        // table (Pokec) streamed at 2.2 TB/s
        int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
        for (; i + 3 * stride < n4; i += 4 * stride) {
            v4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = reinterpret_cast<const v4 *>(g)[i + u * stride];
#pragma unroll
            for (int u = 0; u < 4; ++u) s += (v[u][0] * v[u][0] + v[u][1] * v[u][1]) + (v[u][2] * v[u][2] + v[u][3] * v[u][3]);
        }
        for (; i < n4; i += stride) {
            const v4 v = reinterpret_cast<const v4 *>(g)[i];
            s += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
            const float v = g[i];
            s += v * v;
        }
    }
    const float tot = block_sum_256(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(256)
k_adam_clip(const AdamParams a)
{
    __shared__ float red[4];
    adam_workgroup(a, blockIdx.x, gridDim.x, red);
}

// the update with the norm formed by its own workgroups (and, given reduce descriptors, the gradient summed by them):
// the body that rides in k_gather_multi_adam, as a launch of its own -- same slots, same partial order, same bits
__global__ void __launch_bounds__(256)
k_adam_meet(const AdamParams a)
{
    __shared__ float red[4];
    adam_workgroup<false>(a, blockIdx.x, gridDim.x, red);
}

// ---- gradient finalisation: sum partial buffers into the flat bucket + squared-norm partials ------
// (struct ReduceDesc: gsage_optim_dev.h)

// one workgroup of the finalisation: descriptor `by`, grid-stride slice bx of gx
__device__ __forceinline__ void finalize_workgroup(const ReduceDesc *__restrict__ descs,
                                                   float *__restrict__ flat_g,
                                                   float *__restrict__ partial_sq, int bx, int by, int gx,
                                                   float *red, float *red4, float *red16f)
{
    const ReduceDesc d = descs[by];
    const int64_t gstride = (int64_t)gx * 256;
    float sq = 0.f;
    // 16-byte loads whenever the partial buffers allow it (the K5b slabs do: ld, stride % 4 == 0);
    // a wave-wide load costs the same issue slot whatever its width
    // -- and the descriptor has enough 4-column chunks to keep every thread of its grid row busy
    // (small descriptors with many partials want all the threads they can get instead)
    const bool vec = d.ld % 4 == 0 && d.stride % 4 == 0 && ((uintptr_t)d.src & 15) == 0 &&
                     (int64_t)d.rows * ((d.cols + 3) / 4) >= gstride / 2;
    if (vec) {
        const int cpr = (d.cols + 3) / 4;                  // 4-column chunks per row (last may be ragged)
        const int64_t total = (int64_t)d.rows * cpr;
        for (int64_t t = (int64_t)bx * 256 + threadIdx.x; t < total; t += gstride) {
            const int64_t r = t / cpr;
            const int c = (int)(t - r * cpr) * 4;           // c + 3 < ld because ld % 4 == 0 and c < cols <= ld
            const float *src = d.src + r * d.ld + c;
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            int i = 0;
            for (; i + 8 <= d.S; i += 8) {                  // 8 independent 16-byte loads in flight
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4 *>(src + (int64_t)(i + u) * d.stride);
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            for (; i < d.S; ++i) s += *reinterpret_cast<const f32x4 *>(src + (int64_t)i * d.stride);
            float *dst = flat_g + d.out_off + r * d.cols + c;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c + e < d.cols) {
                    dst[e] = s[e];
                    sq += s[e] * s[e];
                }
        }
    } else if (d.S >= 64 && d.ld % 4 == 0 && d.stride % 4 == 0 && ((uintptr_t)d.src & 15) == 0 &&
               (int64_t)d.rows * ((d.cols + 3) / 4) * 2 <= gstride) {
        // MANY partial tiles of a matrix too small to fill its grid row with 4-column chunks (configs[4]'s level 0: K5b's
        // 120 slices of a 256 x 128 gradient -- one thread per element walked 15 dependent rounds of 4-byte loads, the
        // launch's longest chain: 9.6 us).  16-byte loads as above, and SL threads to a chunk, a share of the partials each
        // (two rounds at S = 120, SL = 8), met in LDS in slice order.
        const int cpr = (d.cols + 3) / 4;
        const int64_t total = (int64_t)d.rows * cpr;
        int SL = 2;
        while (SL < 16 && total * (SL * 2) <= gstride) SL *= 2;
        const int epw = 256 / SL;                                   // chunks per workgroup and trip
        const int el = threadIdx.x % epw, sl = threadIdx.x / epw;
        const int s0 = (d.S * sl) / SL, s1 = (d.S * (sl + 1)) / SL;
        f32x4 *red16 = reinterpret_cast<f32x4 *>(red16f);
        for (int64_t t0 = (int64_t)bx * epw; t0 < total; t0 += (int64_t)gx * epw) {
            const int64_t t = t0 + el;
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            int64_t r = 0;
            int c = 0;
            if (t < total) {
                r = t / cpr;
                c = (int)(t - r * cpr) * 4;
                const float *src = d.src + r * d.ld + c;
                int i = s0;
                for (; i + 8 <= s1; i += 8) {
                    f32x4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4 *>(src + (int64_t)(i + u) * d.stride);
#pragma unroll
                    for (int u = 0; u < 8; ++u) s += v[u];
                }
                for (; i < s1; ++i) s += *reinterpret_cast<const f32x4 *>(src + (int64_t)i * d.stride);
            }
            lds_barrier();
            red16[threadIdx.x] = s;
            lds_barrier();
            if (sl == 0 && t < total) {
                f32x4 tot = red16[el];
                for (int k = 1; k < SL; ++k) tot += red16[k * epw + el];
                float *dst = flat_g + d.out_off + r * d.cols + c;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c + e < d.cols) {
                        dst[e] = tot[e];
                        sq += tot[e] * tot[e];
                    }
            }
        }
    } else if (d.S >= 128 && (int64_t)d.rows * d.cols * 16 <= gstride) {
        // very many partials of very few elements (the max pool's MLP bias: 512 channels x 512 partial rows): four waves
        // to an element would still walk 128 partials each -- 16 dependent rounds, the longest chain of the max-pool
        // step's finalisation (17 us).  Sixteen threads to an element, a sixteenth of the partials each (four rounds at
        // S = 512), met in LDS in slice order.
        const int64_t total = (int64_t)d.rows * d.cols;
        const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;
        const int s0 = (d.S * sl) / 16, s1 = (d.S * (sl + 1)) / 16;
        for (int64_t t0 = (int64_t)bx * 16; t0 < total; t0 += (int64_t)gx * 16) {
            const int64_t t = t0 + el;
            float s = 0.f;
            if (t < total) {
                const int64_t r = t / d.cols;
                const float *src = d.src + r * d.ld + (t - r * d.cols);
                int i = s0;
                for (; i + 8 <= s1; i += 8) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(i + u) * d.stride];
#pragma unroll
                    for (int u = 0; u < 8; ++u) s += v[u];
                }
                for (; i < s1; ++i) s += src[(int64_t)i * d.stride];
            }
            lds_barrier();
            red4[threadIdx.x] = s;
            lds_barrier();
            if (sl == 0 && t < total) {
                float tot = red4[el];
#pragma unroll
                for (int k = 1; k < 16; ++k) tot += red4[16 * k + el];
                flat_g[d.out_off + t] = tot;
                sq += tot * tot;
            }
        }
    } else if (d.S >= 32 && (int64_t)d.rows * d.cols * 4 <= gstride) {
        // few elements, many partials (the seed-level kernel's per-workgroup head gradients: 10.5 k
        // elements x 128 partials): one thread per element would walk all S partials alone -- 16
        // rounds of 8 loads, the longest dependent chain of the launch.  The four waves of a
        // workgroup take a quarter of the partials each for the same 64 elements and meet in LDS.
        const int64_t total = (int64_t)d.rows * d.cols;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int s0 = (d.S * wave) / 4, s1 = (d.S * (wave + 1)) / 4;
        for (int64_t t0 = (int64_t)bx * 64; t0 < total; t0 += (int64_t)gx * 64) {
            const int64_t t = t0 + lane;
            float s = 0.f;
            if (t < total) {
                const int64_t r = t / d.cols;
                const float *src = d.src + r * d.ld + (t - r * d.cols);
                int i = s0;
                for (; i + 8 <= s1; i += 8) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(i + u) * d.stride];
#pragma unroll
                    for (int u = 0; u < 8; ++u) s += v[u];
                }
                for (; i < s1; ++i) s += src[(int64_t)i * d.stride];
            }
            lds_barrier();
            red4[threadIdx.x] = s;
            lds_barrier();
            if (wave == 0 && t < total) {
                const float tot = (red4[lane] + red4[64 + lane]) + (red4[128 + lane] + red4[192 + lane]);
                flat_g[d.out_off + t] = tot;
                sq += tot * tot;
            }
        }
    } else {
        const int64_t total = (int64_t)d.rows * d.cols;
        for (int64_t t = (int64_t)bx * 256 + threadIdx.x; t < total; t += gstride) {
            const int64_t r = t / d.cols;
            const float *src = d.src + r * d.ld + (t - r * d.cols);
            float s = 0.f;
            int i = 0;
            for (; i + 8 <= d.S; i += 8) {                 // 8 independent loads in flight
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(i + u) * d.stride];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            for (; i < d.S; ++i) s += src[(int64_t)i * d.stride];
            flat_g[d.out_off + t] = s;
            sq += s * s;
        }
    }
    const float tot = block_sum_256(sq, red);
    if (threadIdx.x == 0) partial_sq[by * gx + bx] = tot;
}

__global__ void __launch_bounds__(256)
k_finalize_grads(const ReduceDesc *__restrict__ descs, float *__restrict__ flat_g,
                 float *__restrict__ partial_sq, int64_t *tick, int64_t *tick1, int64_t inc1,
                 int64_t *tick2, int64_t inc2)
{
    __shared__ float red[4];
    __shared__ float red4[256];
    __shared__ __attribute__((aligned(16))) float red16f[1024];
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        if (tick) *tick += 1;
        if (tick1) *tick1 += inc1;
        if (tick2) *tick2 += inc2;
    }
    finalize_workgroup(descs, flat_g, partial_sq, blockIdx.x, blockIdx.y, gridDim.x, red, red4, red16f);
}

__global__ void k_step_inc(int64_t *step) { *step += 1; }

// table[ids[r], 0:D] = 0: undoes a scatter-add (gsage_scatter_add_rows) once the optimizer has consumed it, so
// that a dense gradient table is zeroed by touching the rows that were written instead of all of it
__global__ void __launch_bounds__(256)
k_zero_rows(float *__restrict__ table, int64_t ld, const int64_t *__restrict__ ids, int64_t M, int32_t chunks)
{
    const int64_t total = M * chunks, stride = (int64_t)gridDim.x * 256;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
        const int64_t r = t / chunks;
        const int c = (int)(t - r * chunks) * 4;
        *reinterpret_cast<f32x4 *>(table + ids[r] * ld + c) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// ---- deferred ("lazy") Adam over the rows of a trainable embedding table ------------------------------------------
// The reference's optimizer is dense: every row of the table moves every step, also rows whose gradient is
// zero (m and v decay, p -= step_size * m / (sqrt(v)..)).  That update of a zero-gradient row depends on the row
// alone, so it can be replayed later without changing a bit: `last[r]` holds the number of the last update applied
// to row r, `hist` the per-step constants (step_size, 1/sqrt(bc2)) of every update since, and a row is brought
// up to date (a) before the forward reads it (k_rows_catch_up on the step's frontier) and (b) when anything else
// looks at the table (k_rows_catch_up<true>: every row).  A step then touches ~10 % of a Pokec-sized table
// instead of streaming 2.9 GB.  Duplicate ids in a frontier are settled by atomicMax on the row's stamp.
struct RowAdam {
    float *p, *g, *m, *v;
    int32_t *last, *seen;
    float *hist;                 // [2 * hist_cap]
    int32_t hist_cap, E, lpr;    // lanes per row: power of two >= E / 4
    int32_t sorted;              // != 0: ids0 is sorted ascending (n1 == 0): runs settled by comparing neighbours
    int64_t n_rows;
    const float *lr;
    const int64_t *step;
    float beta1, beta2, eps, weight_decay, max_norm;
};

// VEC elements per lane (1: rows of <= 64 elements, a row per wave -- lanes of a wave never wait for another row's
// longer replay; 4: wider rows).  Replays are ALU work (one adam_update per element and skipped step); the step
// constants of the next iteration are fetched while this one computes.
template <int VEC> struct RowVec { float x[VEC]; };

template <int VEC>
__device__ __forceinline__ RowVec<VEC> row_load(const float *base)
{
    RowVec<VEC> r;
    if (VEC == 4) {
        const f32x4 t = *reinterpret_cast<const f32x4 *>(base);
        r.x[0] = t[0]; r.x[1 % VEC] = t[1]; r.x[2 % VEC] = t[2]; r.x[3 % VEC] = t[3];
    } else {
        r.x[0] = *base;
    }
    return r;
}

template <int VEC>
__device__ __forceinline__ void row_store(float *base, const RowVec<VEC> &r)
{
    if (VEC == 4) *reinterpret_cast<f32x4 *>(base) = f32x4{r.x[0], r.x[1 % VEC], r.x[2 % VEC], r.x[3 % VEC]};
    else *base = r.x[0];
}

typedef const __attribute__((address_space(4))) float *const_f32_ptr;     // scalar (SMEM) loads

// group g of the ring: the constants of 8 consecutive updates, 16 floats, one scalar load
struct HistGroup { float c[16]; };
__device__ __forceinline__ HistGroup hist_group(const RowAdam &a, int32_t t)
{
    const_f32_ptr hp = (const_f32_ptr)(uintptr_t)(a.hist + 2 * (int64_t)((t & (a.hist_cap - 1)) & ~7));
    HistGroup h;
#pragma unroll
    for (int k = 0; k < 16; ++k) h.c[k] = hp[k];
    return h;
}

// One replayed update (zero gradient).  NOWD: the caller saw weight_decay == 0 -- then g stays +0, (1 - beta) * g is +0,
// and adam_update<true>'s first four operations (the decay's multiply / add / select, the product with g) only ever
// produce that +0: the same bits from 9 instead of 14 vector instructions per element and skipped step (the replays
// are what bounds the deferred rows: ~10 skipped steps per touched row at Pokec's size).  The additions of +0 stay --
// they turn a -0 into +0 exactly as the dense update does.
template <bool NOWD>
__device__ __forceinline__ float replay_update(float p, float &m, float &v, float beta1, float beta2, float eps,
                                               float weight_decay, float step_size, float rsqrt_bc2)
{
#pragma clang fp contract(off)
    if (!NOWD) return adam_update<true>(0.f, p, m, v, beta1, beta2, eps, weight_decay, step_size, rsqrt_bc2);
    m = beta1 * m + 0.f;
    v = beta2 * v + 0.f;
    const float denom = __builtin_amdgcn_sqrtf(v) * rsqrt_bc2 + eps;
    return p - step_size * (m * __builtin_amdgcn_rcpf(denom));
}

// updates from .. to with a zero gradient (the constants come from `hist`)
template <int VEC, bool UNI, bool NOWD>
__device__ __forceinline__ void row_replay_impl(const RowAdam &a, int32_t from, int32_t to, RowVec<VEC> &p, RowVec<VEC> &m,
                                                RowVec<VEC> &v);

template <int VEC, bool UNI>
__device__ __forceinline__ void row_replay(const RowAdam &a, int32_t from, int32_t to, RowVec<VEC> &p, RowVec<VEC> &m,
                                           RowVec<VEC> &v)
{
    if (a.weight_decay == 0.f) row_replay_impl<VEC, UNI, true>(a, from, to, p, m, v);       // (uniform: a kernel argument)
    else row_replay_impl<VEC, UNI, false>(a, from, to, p, m, v);
}

template <int VEC, bool UNI, bool NOWD>
__device__ __forceinline__ void row_replay_impl(const RowAdam &a, int32_t from, int32_t to, RowVec<VEC> &p, RowVec<VEC> &m,
                                                RowVec<VEC> &v)
{
    if (from > to) return;
    if (UNI) {
        // a row per wave: the trip count is wave-uniform.  Eight updates per trip (aligned groups of the ring: never
        // wrap), the next group's constants requested before this group is worked through.
        int32_t t = __builtin_amdgcn_readfirstlane(from);
        const int32_t last = __builtin_amdgcn_readfirstlane(to);
        HistGroup cur = hist_group(a, t);
        while (t <= last) {
            const int32_t k0 = t & 7;
            const int32_t k1 = (last - t + k0) < 7 ? (last - t + k0) : 7;      // slots k0 .. k1 of this group
            const HistGroup nxt = hist_group(a, t + 8);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (k < k0 || k > k1) continue;
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    p.x[e] = replay_update<NOWD>(p.x[e], m.x[e], v.x[e], a.beta1, a.beta2, a.eps, a.weight_decay,
                                                 cur.c[2 * k], cur.c[2 * k + 1]);
            }
            t += k1 - k0 + 1;
            cur = nxt;
        }
        return;
    }
    const float2 *hist = reinterpret_cast<const float2 *>(a.hist);
    const int32_t mask = a.hist_cap - 1;                        // (a power of two)
    for (int32_t t = from; t <= to; ++t) {
        const float2 h = hist[t & mask];
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            p.x[e] = replay_update<NOWD>(p.x[e], m.x[e], v.x[e], a.beta1, a.beta2, a.eps, a.weight_decay, h.x, h.y);
    }
}

// A wave takes RB * (64 / lpr) list entries at a time: one atomicMax per entry (all in flight together) settles
// which occurrence of a row does the work, then the rows' loads are issued together (RB row groups in flight per
// wave), the replays run, the stores follow.
enum { ROWS_CATCH_UP = 0, ROWS_CATCH_UP_ALL = 1, ROWS_SQNORM = 2, ROWS_ADAM = 3, ROWS_RB = 4 };

template <int MODE, int VEC, int RB, bool UNI>
__device__ __forceinline__ float rows_pass(const RowAdam &a, const int64_t *__restrict__ ids0, int64_t n0,
                                           const int64_t *__restrict__ ids1, int64_t n1, int32_t target, float coef,
                                           const AdamConsts ac)
{
    const int lane = threadIdx.x & 63, lpr = a.lpr, rpw = 64 / lpr;
    const int sub = lane & (lpr - 1), grp = lane / lpr;
    const int ch = RB * rpw;                                  // entries per wave and trip (<= 64: RB <= lpr)
    const int64_t total = MODE == ROWS_CATCH_UP_ALL ? a.n_rows : n0 + n1;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = (int64_t)gridDim.x * 4;
    float acc = 0.f;
    for (int64_t base = wave * ch; base < total; base += n_waves * ch) {
        const int64_t e = base + lane;
        const bool valid = lane < ch && e < total;
        const int32_t r_l = !valid ? 0 : MODE == ROWS_CATCH_UP_ALL ? (int32_t)e : (int32_t)(e < n0 ? ids0[e] : ids1[e - n0]);
        int32_t old_l = target;
        if (valid) {
            if (MODE == ROWS_CATCH_UP_ALL) { old_l = a.last[r_l]; if (old_l < target) a.last[r_l] = target; }
            else if (a.sorted) {
                // sorted list: the first entry of a run of equal ids does the row's work -- no atomics, and which
                // lane (hence which norm partial) a row lands on depends on the list alone
                if (e == 0 || (int32_t)ids0[e - 1] != r_l) {
                    if (MODE == ROWS_SQNORM) old_l = target - 1;
                    else { old_l = a.last[r_l]; if (old_l < target) a.last[r_l] = target; }
                }
            }
            else old_l = atomicMax(MODE == ROWS_SQNORM ? &a.seen[r_l] : &a.last[r_l], target);
        }
        if (!__any(old_l < target)) continue;
        RowVec<VEC> p[RB], m[RB], v[RB], g[RB];
        int32_t old[RB];
        int64_t o[RB];
        bool act[RB];
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            const int src = j * rpw + grp;
            const int32_t r = __shfl(r_l, src);
            old[j] = __shfl(old_l, src);
            act[j] = old[j] < target && sub * VEC < a.E;
            o[j] = (int64_t)r * a.E + sub * VEC;
            if (act[j]) {
                if (MODE == ROWS_SQNORM || MODE == ROWS_ADAM) g[j] = row_load<VEC>(a.g + o[j]);
                if (MODE != ROWS_SQNORM) {
                    p[j] = row_load<VEC>(a.p + o[j]); m[j] = row_load<VEC>(a.m + o[j]); v[j] = row_load<VEC>(a.v + o[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            if (!act[j]) continue;
            if (MODE == ROWS_SQNORM) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc += g[j].x[k] * g[j].x[k];
            } else if (MODE == ROWS_ADAM) {
                row_replay<VEC, UNI>(a, old[j] + 1, target - 1, p[j], m[j], v[j]);
#pragma unroll
                for (int k = 0; k < VEC; ++k)
                    p[j].x[k] = adam_update<true>(g[j].x[k] * coef, p[j].x[k], m[j].x[k], v[j].x[k], a.beta1, a.beta2, a.eps,
                                            a.weight_decay, ac.step_size, ac.rsqrt_bc2);
            } else {
                row_replay<VEC, UNI>(a, old[j] + 1, target, p[j], m[j], v[j]);
            }
        }
        if (MODE == ROWS_SQNORM) continue;
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            if (!act[j]) continue;
            row_store<VEC>(a.p + o[j], p[j]);
            row_store<VEC>(a.m + o[j], m[j]);
            row_store<VEC>(a.v + o[j], v[j]);
            if (MODE == ROWS_ADAM) {
                RowVec<VEC> zero;
#pragma unroll
                for (int k = 0; k < VEC; ++k) zero.x[k] = 0.f;
                row_store<VEC>(a.g + o[j], zero);
            }
        }
    }
    return acc;
}

// rows ids0[0:n0] ++ ids1[0:n1] (ALL: every row) brought up to update number *step + step_off
template <bool ALL, int VEC, bool UNI>
__global__ void __launch_bounds__(256)
k_rows_catch_up(const RowAdam a, const int64_t *__restrict__ ids0, int64_t n0, const int64_t *__restrict__ ids1,
                int64_t n1, int32_t step_off)
{
    rows_pass<ALL ? ROWS_CATCH_UP_ALL : ROWS_CATCH_UP, VEC, ROWS_RB, UNI>(a, ids0, n0, ids1, n1, (int32_t)(*a.step + step_off),
                                                                     1.f, AdamConsts{0.f, 0.f});
}

// partial[bx] = sum of squares of the gradient rows in the lists, every row once (stamp `seen`)
template <int VEC>
__global__ void __launch_bounds__(256)
k_rows_sqnorm(const RowAdam a, const int64_t *__restrict__ ids0, int64_t n0, const int64_t *__restrict__ ids1,
              int64_t n1, int32_t step_off, float *__restrict__ partial)
{
    __shared__ float red[4];
    const float acc = rows_pass<ROWS_SQNORM, VEC, 2 * ROWS_RB, false>(a, ids0, n0, ids1, n1, (int32_t)(*a.step + step_off), 1.f,
                                                               AdamConsts{0.f, 0.f});
    const float s = block_sum_256(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// update number t = *step + step_off of the rows in the lists (every row once; rows behind t - 1 are caught up
// first), their gradient rows zeroed; records the step's constants for later replays
template <int VEC, bool UNI>
__global__ void __launch_bounds__(256)
k_rows_adam(const RowAdam a, const int64_t *__restrict__ ids0, int64_t n0, const int64_t *__restrict__ ids1,
            int64_t n1, int32_t step_off, const float *__restrict__ partial, int32_t n_partial)
{
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n_partial; i += 256) s += partial[i];
    const float total_norm = sqrtf(block_sum_256(s, red));
    float coef = a.max_norm / (total_norm + 1e-6f);          // torch.nn.utils.clip_grad_norm_
    coef = coef < 1.f ? coef : 1.f;
    const int64_t tl = *a.step + step_off;
    const int32_t t = (int32_t)tl;
    const AdamConsts ac = adam_consts(*a.lr, (float)tl, a.beta1, a.beta2);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.hist[(t & (a.hist_cap - 1)) * 2] = ac.step_size;
        a.hist[(t & (a.hist_cap - 1)) * 2 + 1] = ac.rsqrt_bc2;
    }
    rows_pass<ROWS_ADAM, VEC, ROWS_RB, UNI>(a, ids0, n0, ids1, n1, t, coef, ac);
}

// part[b, c] = sum over rows i = b, b + n_part, ... of src[i, c]   (bias gradient of a Linear: column sums)
// A workgroup = 256 / DW row lanes x DW columns (DW = D rounded up to a power of two <= 256): every thread
// streams rows, four loads in flight, the row lanes meet in LDS in a fixed order.
__global__ void __launch_bounds__(256)
k_colsum_partials(const float *__restrict__ src, int64_t ld, int64_t M, int32_t D, int32_t DW, float *__restrict__ part)
{
    __shared__ float red[256];
    const int lanes = 256 / DW;                       // row lanes per workgroup
    const int c = blockIdx.y * DW + (threadIdx.x % DW), rl = threadIdx.x / DW;
    const int64_t step = (int64_t)gridDim.x * lanes;
    float s = 0.f;
    if (c < D) {
        int64_t i = (int64_t)blockIdx.x + (int64_t)rl * gridDim.x;     // rows b, b + n_part, ... split over the lanes
        for (; i + 3 * step < M; i += 4 * step) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = src[(i + u * step) * ld + c];
            s += (v[0] + v[1]) + (v[2] + v[3]);
        }
        for (; i < M; i += step) s += src[i * ld + c];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0 && c < D) {
        float t = 0.f;
        for (int l = 0; l < lanes; ++l) t += red[l * DW + (threadIdx.x % DW)];
        part[(int64_t)blockIdx.x * D + c] = t;
    }
}

__global__ void __launch_bounds__(256)
k_prep_weights(const PrepDesc *__restrict__ descs, int64_t *tick0, int64_t inc0, int64_t *tick1,
               int64_t inc1)
{
    // first kernel of a step: also advances the step's device counters (Adam step, Philox call)
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        if (tick0) *tick0 += inc0;
        if (tick1) *tick1 += inc1;
    }
    const PrepDesc d = descs[blockIdx.y];
    const int64_t total = (int64_t)d.rows * d.cols;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
        const int r = (int)(t / d.cols), c = (int)(t - (int64_t)r * d.cols);
        prep_store(d, r, c, d.src[t]);
    }
}

// ---- inter-layer backward routing ---------------------------------------------------------------
// Rows of a level's output are the hops concatenated: [hop 0 | hop 1 | ... | hop nh-1].  Row m of
// hop k received the level above's x-gradient dX[m] if m < r_x, and -- when k >= 1 -- 1/fan[k] of
// its parent's aggregate gradient dAgg[off[k-1] + (m - off[k]) / fan[k]].  dH = (H > 0) * that.
struct MergeParams {
    const void *H;           // [R, ldh] bf16 (fp32 in parity mode) post-ReLU output of the level (for the mask)
    const float *DG;         // [r_x, ldg] fp32: cols [0, D) = dX, cols [dagg_off, dagg_off + D) = dAgg
    void *dH;                // [R, ldo] same type as H
    int64_t ldh, ldg, ldo, dagg_off;
    int64_t R, r_x;
    int32_t D, n_hops;
    int64_t off[6];
    int32_t fan[6];
};

template <typename T>
__global__ void __launch_bounds__(256)
k_bwd_merge(const MergeParams q)
{
    const int chunks = q.D / 4;
    const int64_t total = q.R * chunks;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
        const int64_t m = t / chunks;
        const int c = (int)(t - m * chunks) * 4;
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        if (m < q.r_x) g = *reinterpret_cast<const f32x4 *>(q.DG + m * q.ldg + c);
        int k = 0;
#pragma unroll
        for (int j = 1; j < 6; ++j)
            if (j < q.n_hops && m >= q.off[j]) k = j;
        if (k >= 1) {
            const int64_t parent = q.off[k - 1] + (m - q.off[k]) / q.fan[k];
            const f32x4 a = *reinterpret_cast<const f32x4 *>(q.DG + parent * q.ldg + q.dagg_off + c);
            const float inv = 1.f / (float)q.fan[k];
            g += a * inv;
        }
        const f32x4 h = load4<T>((const T *)q.H + m * q.ldh + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = h[e] > 0.f ? g[e] : 0.f;
        store4<T>((T *)q.dH + m * q.ldo + c, g);
    }
}

// bf16 rows of whole 16-byte chunks: EIGHT columns per thread (16-byte accesses on the bf16 side: a wave-wide request
// costs the address path the same whatever its width) and two items per trip, their six loads requested together.
// Same arithmetic per element as k_bwd_merge: the same bits.  (Round 6: the 3-layer configs[4] step spent 30 us here on
// 104 MB -- a dependent round trip per item and thread.)
__global__ void __launch_bounds__(256)
k_bwd_merge_v8(const MergeParams q)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const int chunks = q.D / 8;
    const int64_t total = q.R * chunks;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const uint16_t *H = (const uint16_t *)q.H;
    uint16_t *dH = (uint16_t *)q.dH;
    for (int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x; t0 < total; t0 += 2 * stride) {
        f32x4 gx[2][2], ga[2][2];
        u32x4 h[2];
        int64_t m[2];
        int c[2];
        bool live[2], child[2];
        float inv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t t = t0 + u * stride;
            live[u] = t < total;
            const int64_t tc = live[u] ? t : t0;
            m[u] = tc / chunks;
            c[u] = (int)(tc - m[u] * chunks) * 8;
            int k = 0;
#pragma unroll
            for (int j = 1; j < 6; ++j)
                if (j < q.n_hops && m[u] >= q.off[j]) k = j;
            child[u] = k >= 1;
            const int kc = child[u] ? k : 1;
            const int64_t parent = child[u] ? q.off[kc - 1] + (m[u] - q.off[kc]) / q.fan[kc] : 0;
            inv[u] = 1.f / (float)q.fan[kc];
            const int64_t mx = m[u] < q.r_x ? m[u] : 0;
            const float *px = q.DG + mx * q.ldg + c[u], *pa = q.DG + parent * q.ldg + q.dagg_off + c[u];
            gx[u][0] = *reinterpret_cast<const f32x4 *>(px);
            gx[u][1] = *reinterpret_cast<const f32x4 *>(px + 4);
            ga[u][0] = *reinterpret_cast<const f32x4 *>(pa);
            ga[u][1] = *reinterpret_cast<const f32x4 *>(pa + 4);
            h[u] = *reinterpret_cast<const u32x4 *>(H + m[u] * q.ldh + c[u]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            u32x4 o;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                float g0 = 0.f, g1 = 0.f;
                if (m[u] < q.r_x) { g0 = gx[u][w >> 1][2 * (w & 1)]; g1 = gx[u][w >> 1][2 * (w & 1) + 1]; }
                if (child[u]) { g0 += ga[u][w >> 1][2 * (w & 1)] * inv[u]; g1 += ga[u][w >> 1][2 * (w & 1) + 1] * inv[u]; }
                const float h0 = bf16_to_f32((uint16_t)(h[u][w] & 0xffff)), h1 = bf16_to_f32((uint16_t)(h[u][w] >> 16));
                o[w] = pack_bf16x2(h0 > 0.f ? g0 : 0.f, h1 > 0.f ? g1 : 0.f);
            }
            if (live[u]) *reinterpret_cast<u32x4 *>(dH + m[u] * q.ldo + c[u]) = o;
        }
    }
}

// ---- backward routing of the max pool (K3) ------------------------------------------------------
// hidden[i*n + j, c] receives g[i, c] iff row j won the max of (segment i, channel c) and the max
// is positive (ReLU), else 0 -- autograd of nn_modules.py:224-226,240.  One thread = 8 channels of
// one segment: reads 8 (g, pooled, argmax), writes n rows x 16 bytes of the bf16 operand of K5b.
// fp32 variant of the two routing kernels (parity mode): one thread = 4 channels of one segment
__global__ void __launch_bounds__(256)
k_pool_route_bwd_f32(const float *__restrict__ g, int64_t ldg, const float *__restrict__ pooled, int64_t ldp,
                     const int32_t *__restrict__ argmax, int64_t lda, int64_t M, int32_t n, int32_t H,
                     float *__restrict__ out, int64_t ldo)
{
    const int chunks = H / 4;
    const int64_t total = M * chunks;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
        const int64_t i = t / chunks;
        const int c0 = (int)(t - i * chunks) * 4;
        float gv[4];
        int32_t am[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gv[e] = pooled[i * ldp + c0 + e] > 0.f ? g[i * ldg + c0 + e] : 0.f;
            am[e] = argmax[i * lda + c0 + e];
        }
        for (int j = 0; j < n; ++j) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = am[e] == j ? gv[e] : 0.f;
            *reinterpret_cast<f32x4 *>(out + (i * n + j) * ldo + c0) = o;
        }
    }
}

__global__ void __launch_bounds__(256)
k_pool_route_mean_bwd_f32(const float *__restrict__ g, int64_t ldg, const uint32_t *__restrict__ mask, int64_t M,
                          int32_t n, int32_t H, float *__restrict__ out, int64_t ldo)
{
    const int chunks = H / 4;
    const int words = H / 32;
    const int64_t total = M * chunks;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const float inv = 1.f / (float)n;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
        const int64_t i = t / chunks;
        const int c0 = (int)(t - i * chunks) * 4;
        float gv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) gv[e] = g[i * ldg + c0 + e] * inv;
        for (int j = 0; j < n; ++j) {
            const uint32_t m = mask[(i * n + j) * words + (c0 >> 5)] >> (c0 & 31);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = ((m >> e) & 1u) ? gv[e] : 0.f;
            *reinterpret_cast<f32x4 *>(out + (i * n + j) * ldo + c0) = o;
        }
    }
}

__global__ void __launch_bounds__(256)
k_pool_route_bwd(const float *__restrict__ g, int64_t ldg, const float *__restrict__ pooled, int64_t ldp,
                 const int32_t *__restrict__ argmax, int64_t lda, int64_t M, int32_t n, int32_t H,
                 uint16_t *__restrict__ out, int64_t ldo)
{
    const int chunks = H / 8;
    const int64_t total = M * chunks;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
        const int64_t i = t / chunks;
        const int c0 = (int)(t - i * chunks) * 8;
        uint16_t gb[8];
        int32_t am[8];
        if (((ldg | ldp | lda) & 3) == 0 && ((((uintptr_t)g | (uintptr_t)pooled | (uintptr_t)argmax) & 15) == 0)) {
            // 6 x 16-byte loads instead of 24 x 4-byte ones
            typedef int i32x4 __attribute__((ext_vector_type(4)));
            f32x4 gv[2], pv[2];
            i32x4 av[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                gv[h] = *reinterpret_cast<const f32x4 *>(g + i * ldg + c0 + 4 * h);
                pv[h] = *reinterpret_cast<const f32x4 *>(pooled + i * ldp + c0 + 4 * h);
                av[h] = *reinterpret_cast<const i32x4 *>(argmax + i * lda + c0 + 4 * h);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                gb[e] = pv[e >> 2][e & 3] > 0.f ? f32_to_bf16(gv[e >> 2][e & 3]) : (uint16_t)0;
                am[e] = av[e >> 2][e & 3];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool on = pooled[i * ldp + c0 + e] > 0.f;
                gb[e] = on ? f32_to_bf16(g[i * ldg + c0 + e]) : (uint16_t)0;
                am[e] = argmax[i * lda + c0 + e];
            }
        }
        for (int j = 0; j < n; ++j) {
            vec16 o;
#pragma unroll
            for (int e = 0; e < 8; e += 2)
                o[e >> 1] = (am[e] == j ? (uint32_t)gb[e] : 0u) | ((am[e + 1] == j ? (uint32_t)gb[e + 1] : 0u) << 16);
            *reinterpret_cast<vec16 *>(out + (i * n + j) * ldo + c0) = o;
        }
    }
}

// mean pool: hidden[i*n + j, c] receives g[i, c] / n iff its ReLU was active (sign bit from K3)
__global__ void __launch_bounds__(256)
k_pool_route_mean_bwd(const float *__restrict__ g, int64_t ldg, const uint32_t *__restrict__ mask, int64_t M,
                      int32_t n, int32_t H, uint16_t *__restrict__ out, int64_t ldo)
{
    const int chunks = H / 8;
    const int words = H / 32;
    const int64_t total = M * chunks;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const float inv = 1.f / (float)n;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
        const int64_t i = t / chunks;
        const int c0 = (int)(t - i * chunks) * 8;
        uint16_t gb[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) gb[e] = f32_to_bf16(g[i * ldg + c0 + e] * inv);
        for (int j = 0; j < n; ++j) {
            const uint32_t m = mask[(i * n + j) * words + (c0 >> 5)] >> (c0 & 31);
            vec16 o;
#pragma unroll
            for (int e = 0; e < 8; e += 2)
                o[e >> 1] = (((m >> e) & 1u) ? (uint32_t)gb[e] : 0u) | ((((m >> (e + 1)) & 1u) ? (uint32_t)gb[e + 1] : 0u) << 16);
            *reinterpret_cast<vec16 *>(out + (i * n + j) * ldo + c0) = o;
        }
    }
}

// part[b, c] = sum over segments i = b, b + n_part, ... of g[i, c] / n * #(active rows of segment i at c)
__global__ void __launch_bounds__(256)
k_pool_bias_partials_mean(const float *__restrict__ g, int64_t ldg, const uint32_t *__restrict__ mask,
                          int64_t M, int32_t n, int32_t H, float *__restrict__ part)
{
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= H) return;
    const int words = H / 32;
    float s = 0.f;
    for (int64_t i = blockIdx.x; i < M; i += gridDim.x) {
        int cnt = 0;
        for (int j = 0; j < n; ++j) cnt += (mask[(i * n + j) * words + (c >> 5)] >> (c & 31)) & 1u;
        s += g[i * ldg + c] * (float)cnt;
    }
    part[(int64_t)blockIdx.x * H + c] = s / (float)n;
}

// part[b, c] = sum over rows i = b, b + n_part, ... of g[i, c] * (pooled[i, c] > 0)
// thread = 4 consecutive channels (16-byte loads), 4 rows in flight; blockIdx.y walks the channels
__global__ void __launch_bounds__(256)
k_pool_bias_partials(const float *__restrict__ g, int64_t ldg, const float *__restrict__ pooled, int64_t ldp,
                     int64_t M, int32_t H, float *__restrict__ part)
{
    const int c = (blockIdx.y * 256 + threadIdx.x) * 4;
    if (c >= H) return;
    const int64_t step = gridDim.x;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int64_t i = blockIdx.x;
    for (; i + 3 * step < M; i += 4 * step) {
        f32x4 gv[4], pv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            gv[u] = *reinterpret_cast<const f32x4 *>(g + (i + u * step) * ldg + c);
            pv[u] = *reinterpret_cast<const f32x4 *>(pooled + (i + u * step) * ldp + c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] += pv[u][e] > 0.f ? gv[u][e] : 0.f;
    }
    for (; i < M; i += step) {
        const f32x4 gv = *reinterpret_cast<const f32x4 *>(g + i * ldg + c);
        const f32x4 pv = *reinterpret_cast<const f32x4 *>(pooled + i * ldp + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += pv[e] > 0.f ? gv[e] : 0.f;
    }
    *reinterpret_cast<f32x4 *>(part + (int64_t)blockIdx.x * H + c) = s;
}

template <typename T>
__global__ void __launch_bounds__(256)
k_pool_merge_bwd(const T *__restrict__ Hp, int64_t ldh, const float *__restrict__ DX, int64_t ldx,
                 int64_t r_x, const float *__restrict__ DN, int64_t ldn, int64_t r0,
                 T *__restrict__ dH, int64_t ldo, int64_t R, int32_t D)
{
    const int chunks = D / 4;
    const int64_t total = R * chunks;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
        const int64_t m = t / chunks;
        const int c = (int)(t - m * chunks) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < r_x) v = *reinterpret_cast<const f32x4 *>(DX + m * ldx + c);
        if (m >= r0) v += *reinterpret_cast<const f32x4 *>(DN + (m - r0) * ldn + c);
        const f32x4 h = load4<T>(Hp + m * ldh + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = h[e] > 0.f ? v[e] : 0.f;
        store4<T>(dH + m * ldo + c, v);
    }
}

static inline int grid_for(int64_t items, int cap)
{
    int64_t b = ceil_div(items, 256);
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace gsage

using namespace gsage;

extern "C" {

int gsage_adam_partials(int64_t n)
{
    return grid_for(n, 1024);
}

int gsage_clip_adam_step(float *p, float *g, float *m, float *v, int64_t n, float *partial,
                         const float *lr, int64_t *step, float beta1, float beta2, float eps,
                         float weight_decay, float max_norm, float *norm_out, int step_is_current,
                         int32_t n_partial_ready, const void *prep_descs, int32_t n_prep,
                         int64_t *tick1, int64_t inc1, int64_t *tick2, int64_t inc2, void *stream)
{
    // step_is_current & 2: the caller consumes the gradient here (it is about to be zeroed): do not write the
    // clipped values back (a trainable embedding table: 418 MB of stores per clipped step at Pokec's size)
    // step_is_current & 4: the deferred-row arithmetic (1-ulp sqrt / reciprocal, gsage_optim_dev.h) instead of
    // torch.optim.Adam's correctly rounded ones -- for a table whose rows gsage_rows_* may update as well
    const int discard = (step_is_current & 2) != 0;
    const int replay = (step_is_current & 4) != 0;
    step_is_current &= 1;
    gsage_adam_desc d;
    d.p = p; d.g = g; d.m = m; d.v = v; d.n = n; d.partial = partial; d.lr = lr; d.step = step;
    d.beta1 = beta1; d.beta2 = beta2; d.eps = eps; d.weight_decay = weight_decay; d.max_norm = max_norm;
    d.norm_out = norm_out; d.step_is_current = step_is_current; d.n_partial_ready = n_partial_ready;
    d.prep_descs = prep_descs; d.n_prep = n_prep; d.tick1 = tick1; d.inc1 = inc1; d.tick2 = tick2;
    d.inc2 = inc2; d.norm_slots = nullptr; d.reduce_descs = nullptr; d.n_reduce = 0;
    GSAGE_REQUIRE(p && g && m && v && partial && lr && step, "clip_adam_step: null pointer");
    GSAGE_REQUIRE(n > 0 && n_partial_ready >= 0 && n_prep >= 0, "clip_adam_step: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    int rc = GSAGE_OK;
    if (n_partial_ready == 0) {   // no gsage_finalize_grads before us: compute the norm partials here
        d.n_partial_ready = adam_grid(n, 1024);
        launch(k_grad_sqnorm, dim3(d.n_partial_ready), dim3(256), 0, s, (const float *)g, n, partial);
        rc = check_launch("grad_sqnorm");
        if (rc != GSAGE_OK) return rc;
    }
    AdamParams a;
    rc = fill_adam(a, d);
    if (rc != GSAGE_OK) return rc;
    a.discard_clipped = discard;
    a.replay_math = replay;
    launch(k_adam_clip, dim3(adam_grid(a.n_prep > 0 ? ceil_div(n, 4) : n, 2048)), dim3(256), 0, s, a);
    rc = check_launch("adam_clip");
    if (rc != GSAGE_OK) return rc;
    if (!step_is_current) {
        launch(k_step_inc, dim3(1), dim3(1), 0, s, step);
        rc = check_launch("step_inc");
    }
    return rc;
}

int gsage_clip_adam_meet(const gsage_adam_desc *adam, void *stream)
{
    GSAGE_REQUIRE(adam && adam->norm_slots && adam->n_partial_ready == 0 && adam->step_is_current,
                  "clip_adam_meet: needs norm_slots, n_partial_ready == 0 and step_is_current");
    AdamParams a;
    int rc = fill_adam(a, *adam);
    if (rc != GSAGE_OK) return rc;
    const int n_adam = adam_grid(ceil_div(adam->n, 4), 2048);
    int dev = 0, per_cu = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_adam_meet, 256, 0) != hipSuccess) {
        (void)hipGetLastError();
        per_cu = cus = 0;
    }
    per_cu = per_cu > 8 ? 8 : per_cu;
    GSAGE_REQUIRE(a.n_prep > 0 && n_adam <= 1024 && (int64_t)n_adam * 1024 >= adam->n && n_adam <= (per_cu - 1) * cus,
                  "clip_adam_meet: the in-launch norm needs one resident workgroup per 1 024 elements (%d asked, "
                  "%d resident); pass norm partials to gsage_clip_adam_step instead", n_adam, (per_cu - 1) * cus);
    launch(k_adam_meet, dim3(n_adam), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("adam_meet");
}

int gsage_grad_sqnorm(const float *g, int64_t n, float *partial, int32_t n_partial, void *stream)
{
    GSAGE_REQUIRE(g && partial && n > 0 && n_partial >= 1 && n_partial <= 1024, "grad_sqnorm: bad arguments");
    launch(k_grad_sqnorm, dim3(n_partial), dim3(256), 0, (hipStream_t)stream, g, n, partial);
    return check_launch("grad_sqnorm");
}

int gsage_zero_rows(float *table, int64_t ld, const int64_t *ids, int64_t M, int64_t D, void *stream)
{
    GSAGE_REQUIRE(table && ids && M >= 0 && D > 0 && D % 4 == 0 && ld % 4 == 0 && ld >= D &&
                  ((uintptr_t)table % 16) == 0, "zero_rows: needs 16-byte rows (D, ld multiples of 4)");
    if (M == 0) return GSAGE_OK;
    launch(k_zero_rows, dim3(grid_for(M * (D / 4), 4096)), dim3(256), 0, (hipStream_t)stream, table, ld, ids, M,
           (int32_t)(D / 4));
    return check_launch("zero_rows");
}

static int fill_rows(RowAdam &a, const gsage_row_adam *d, const char *who)
{
    GSAGE_REQUIRE(d && d->p && d->g && d->m && d->v && d->last && d->seen && d->hist && d->lr && d->step, who);
    GSAGE_REQUIRE(d->n_rows > 0 && d->E > 0 && d->E % 4 == 0 && d->E <= 256 && d->hist_cap >= 2 &&
                  (d->hist_cap & (d->hist_cap - 1)) == 0, who);
    GSAGE_REQUIRE(((((uintptr_t)d->p | (uintptr_t)d->g | (uintptr_t)d->m | (uintptr_t)d->v)) & 15) == 0, who);
    a.p = d->p; a.g = d->g; a.m = d->m; a.v = d->v; a.last = d->last; a.seen = d->seen; a.hist = d->hist;
    a.hist_cap = d->hist_cap; a.E = d->E; a.n_rows = d->n_rows; a.lr = d->lr; a.step = d->step;
    a.beta1 = d->beta1; a.beta2 = d->beta2; a.eps = d->eps; a.weight_decay = d->weight_decay; a.max_norm = d->max_norm;
    a.sorted = d->sorted_ids;
    GSAGE_REQUIRE(d->n_rows < ((int64_t)1 << 31), who);
    const int per = d->E <= 64 ? 1 : 4;          // elements per lane (rows_vec)
    a.lpr = 1;
    while (a.lpr * per < d->E || a.lpr < 2 * ROWS_RB) a.lpr <<= 1;     // (a wave's trip: <= 64 entries)
    return GSAGE_OK;
}

static inline bool rows_vec4(const RowAdam &a) { return a.E > 64; }

static int rows_grid(const RowAdam &a, int64_t entries, int cap)
{
    return grid_for(ceil_div(entries, (int64_t)(ROWS_RB * (64 / a.lpr))) * 64, cap);      // a wave per RB row groups
}

int gsage_rows_catch_up(const gsage_row_adam *d, const int64_t *ids0, int64_t n0, const int64_t *ids1, int64_t n1,
                        int32_t step_off, void *stream)
{
    RowAdam a;
    int rc = fill_rows(a, d, "rows_catch_up: bad descriptor");
    if (rc != GSAGE_OK) return rc;
    GSAGE_REQUIRE(n0 >= 0 && n1 >= 0 && (n0 == 0 || ids0) && (n1 == 0 || ids1), "rows_catch_up: bad id lists");
    GSAGE_REQUIRE(!a.sorted || n1 == 0, "rows_catch_up: a sorted list is ONE list (n1 == 0)");
    if (n0 + n1 == 0) return GSAGE_OK;
    const dim3 grid(rows_grid(a, n0 + n1, 4096));
    hipStream_t s = (hipStream_t)stream;
    if (rows_vec4(a) && a.lpr == 64) launch(k_rows_catch_up<false, 4, true>, grid, dim3(256), 0, s, a, ids0, n0, ids1, n1, step_off);
    else if (rows_vec4(a)) launch(k_rows_catch_up<false, 4, false>, grid, dim3(256), 0, s, a, ids0, n0, ids1, n1, step_off);
    else if (a.lpr == 64) launch(k_rows_catch_up<false, 1, true>, grid, dim3(256), 0, s, a, ids0, n0, ids1, n1, step_off);
    else launch(k_rows_catch_up<false, 1, false>, grid, dim3(256), 0, s, a, ids0, n0, ids1, n1, step_off);
    return check_launch("rows_catch_up");
}

int gsage_rows_catch_up_all(const gsage_row_adam *d, int32_t step_off, void *stream)
{
    RowAdam a;
    int rc = fill_rows(a, d, "rows_catch_up_all: bad descriptor");
    if (rc != GSAGE_OK) return rc;
    const dim3 grid(rows_grid(a, a.n_rows, 8192));
    hipStream_t s = (hipStream_t)stream;
    const int64_t *none = nullptr;
    const int64_t zero = 0;
    if (rows_vec4(a) && a.lpr == 64) launch(k_rows_catch_up<true, 4, true>, grid, dim3(256), 0, s, a, none, zero, none, zero, step_off);
    else if (rows_vec4(a)) launch(k_rows_catch_up<true, 4, false>, grid, dim3(256), 0, s, a, none, zero, none, zero, step_off);
    else if (a.lpr == 64) launch(k_rows_catch_up<true, 1, true>, grid, dim3(256), 0, s, a, none, zero, none, zero, step_off);
    else launch(k_rows_catch_up<true, 1, false>, grid, dim3(256), 0, s, a, none, zero, none, zero, step_off);
    return check_launch("rows_catch_up_all");
}

int gsage_rows_sqnorm(const gsage_row_adam *d, const int64_t *ids0, int64_t n0, const int64_t *ids1, int64_t n1,
                      int32_t step_off, float *partial, int32_t n_partial, void *stream)
{
    RowAdam a;
    int rc = fill_rows(a, d, "rows_sqnorm: bad descriptor");
    if (rc != GSAGE_OK) return rc;
    GSAGE_REQUIRE(n0 >= 0 && n1 >= 0 && (n0 == 0 || ids0) && (n1 == 0 || ids1) && partial && n_partial >= 1 &&
                  n_partial <= 1024, "rows_sqnorm: bad arguments");
    GSAGE_REQUIRE(!a.sorted || n1 == 0, "rows_sqnorm: a sorted list is ONE list (n1 == 0)");
    if (rows_vec4(a))
        launch(k_rows_sqnorm<4>, dim3(n_partial), dim3(256), 0, (hipStream_t)stream, a, ids0, n0, ids1, n1, step_off, partial);
    else
        launch(k_rows_sqnorm<1>, dim3(n_partial), dim3(256), 0, (hipStream_t)stream, a, ids0, n0, ids1, n1, step_off, partial);
    return check_launch("rows_sqnorm");
}

int gsage_rows_adam(const gsage_row_adam *d, const int64_t *ids0, int64_t n0, const int64_t *ids1, int64_t n1,
                    int32_t step_off, const float *partial, int32_t n_partial_ready, void *stream)
{
    RowAdam a;
    int rc = fill_rows(a, d, "rows_adam: bad descriptor");
    if (rc != GSAGE_OK) return rc;
    GSAGE_REQUIRE(n0 >= 0 && n1 >= 0 && (n0 == 0 || ids0) && (n1 == 0 || ids1) && partial && n_partial_ready >= 1,
                  "rows_adam: bad arguments");
    GSAGE_REQUIRE(!a.sorted || n1 == 0, "rows_adam: a sorted list is ONE list (n1 == 0)");
    // (always launched, also for empty lists: the launch records the step's constants)
    const dim3 grid(rows_grid(a, n0 + n1 > 0 ? n0 + n1 : 1, 4096));
    hipStream_t s = (hipStream_t)stream;
    // (its replays -- rows that skipped their catch-up -- read the ring it writes: vector loads, UNI = false)
    if (rows_vec4(a)) launch(k_rows_adam<4, false>, grid, dim3(256), 0, s, a, ids0, n0, ids1, n1, step_off, partial, n_partial_ready);
    else launch(k_rows_adam<1, false>, grid, dim3(256), 0, s, a, ids0, n0, ids1, n1, step_off, partial, n_partial_ready);
    return check_launch("rows_adam");
}

int gsage_colsum_partials(const float *src, int64_t ld, int64_t M, int32_t D, float *part, int32_t n_part, void *stream)
{
    GSAGE_REQUIRE(src && part && M >= 0 && D > 0 && ld >= D && n_part >= 1 && n_part <= 1024, "colsum_partials: bad arguments");
    int32_t DW = 1;
    while (DW < D && DW < 256) DW <<= 1;
    launch(k_colsum_partials, dim3((unsigned)n_part, (unsigned)ceil_div(D, DW)), dim3(256), 0, (hipStream_t)stream, src,
           ld, M, D, DW, part);
    return check_launch("colsum_partials");
}

int gsage_prep_weights(const void *descs, int32_t n_desc, int64_t max_elems, int64_t *tick0,
                       int64_t inc0, int64_t *tick1, int64_t inc1, void *stream)
{
    GSAGE_REQUIRE(descs && n_desc > 0 && max_elems > 0, "prep_weights: bad arguments");
    launch(k_prep_weights, dim3(grid_for(max_elems, 256), n_desc), dim3(256), 0,
                       (hipStream_t)stream, (const PrepDesc *)descs, tick0, inc0, tick1, inc1);
    return check_launch("prep_weights");
}

int gsage_finalize_partials(int32_t n_desc, int64_t max_elems)
{
    return grid_for(max_elems, 256) * n_desc;
}

int gsage_finalize_grads(const void *descs, int32_t n_desc, int64_t max_elems, float *flat_g,
                         float *partial_sq, int64_t *tick, int64_t *tick1, int64_t inc1,
                         int64_t *tick2, int64_t inc2, void *stream)
{
    GSAGE_REQUIRE(descs && flat_g && partial_sq && n_desc > 0 && max_elems > 0, "finalize_grads: bad arguments");
    launch(k_finalize_grads, dim3(grid_for(max_elems, 256), n_desc), dim3(256), 0,
                       (hipStream_t)stream, (const ReduceDesc *)descs, flat_g, partial_sq, tick, tick1, inc1,
                       tick2, inc2);
    return check_launch("finalize_grads");
}

int gsage_pool_route_bwd(const float *g, int64_t ldg, const float *pooled, int64_t ldp,
                         const int32_t *argmax, int64_t lda, int64_t M, int32_t n, int32_t H, void *out,
                         int out_dtype, int64_t ldo, void *stream)
{
    GSAGE_REQUIRE(g && pooled && argmax && out, "pool_route_bwd: null pointer");
    GSAGE_REQUIRE(out_dtype == GSAGE_BF16 || out_dtype == GSAGE_F32, "pool_route_bwd: bad dtype");
    if (out_dtype == GSAGE_F32) {
        GSAGE_REQUIRE(M >= 0 && n > 0 && H > 0 && H % 4 == 0 && ldo % 4 == 0 && ldo >= H && ldg >= H && ldp >= H &&
                      lda >= H && ((uintptr_t)out % 16) == 0, "pool_route_bwd: fp32 output needs H, ldo % 4 == 0");
        if (M == 0) return GSAGE_OK;
        launch(k_pool_route_bwd_f32, dim3(grid_for(M * (H / 4), 8192)), dim3(256), 0, (hipStream_t)stream, g, ldg,
               pooled, ldp, argmax, lda, M, n, H, (float *)out, ldo);
        return check_launch("pool_route_bwd");
    }
    GSAGE_REQUIRE(M >= 0 && n > 0 && H > 0 && H % 8 == 0 && ldo % 8 == 0 && ldo >= H &&
                  ((uintptr_t)out % 16) == 0, "pool_route_bwd: H and ldo must be multiples of 8, out 16-byte aligned");
    GSAGE_REQUIRE(ldg >= H && ldp >= H && lda >= H, "pool_route_bwd: leading dimension smaller than H");
    if (M == 0) return GSAGE_OK;
    launch(k_pool_route_bwd, dim3(grid_for(M * (H / 8), 8192)), dim3(256), 0, (hipStream_t)stream, g, ldg,
           pooled, ldp, argmax, lda, M, n, H, (uint16_t *)out, ldo);
    return check_launch("pool_route_bwd");
}

int gsage_pool_route_mean_bwd(const float *g, int64_t ldg, const uint32_t *relu_mask, int64_t M, int32_t n,
                              int32_t H, void *out, int out_dtype, int64_t ldo, float *bias_part, int32_t n_part,
                              void *stream)
{
    GSAGE_REQUIRE(out_dtype == GSAGE_BF16 || out_dtype == GSAGE_F32, "pool_route_mean_bwd: bad dtype");
    GSAGE_REQUIRE(g && relu_mask && out, "pool_route_mean_bwd: null pointer");
    GSAGE_REQUIRE(M >= 0 && n > 0 && H > 0 && H % 32 == 0 && ldo % 8 == 0 && ldo >= H && ldg >= H &&
                  ((uintptr_t)out % 16) == 0, "pool_route_mean_bwd: H % 32, ldo % 8, out 16-byte aligned");
    GSAGE_REQUIRE(!bias_part || (n_part >= 1 && n_part <= 1024), "pool_route_mean_bwd: bad n_part");
    if (M == 0) return GSAGE_OK;
    if (out_dtype == GSAGE_F32)
        launch(k_pool_route_mean_bwd_f32, dim3(grid_for(M * (H / 4), 8192)), dim3(256), 0, (hipStream_t)stream, g, ldg,
               relu_mask, M, n, H, (float *)out, ldo);
    else
        launch(k_pool_route_mean_bwd, dim3(grid_for(M * (H / 8), 8192)), dim3(256), 0, (hipStream_t)stream, g, ldg,
               relu_mask, M, n, H, (uint16_t *)out, ldo);
    int rc = check_launch("pool_route_mean_bwd");
    if (rc != GSAGE_OK || !bias_part) return rc;
    launch(k_pool_bias_partials_mean, dim3((unsigned)n_part, (unsigned)ceil_div(H, 256)), dim3(256), 0,
           (hipStream_t)stream, g, ldg, relu_mask, M, n, H, bias_part);
    return check_launch("pool_bias_partials_mean");
}

int gsage_pool_bias_partials(const float *g, int64_t ldg, const float *pooled, int64_t ldp, int64_t M,
                             int32_t H, float *part, int32_t n_part, void *stream)
{
    GSAGE_REQUIRE(g && pooled && part, "pool_bias_partials: null pointer");
    GSAGE_REQUIRE(M >= 0 && H > 0 && H % 4 == 0 && ldg >= H && ldp >= H && ldg % 4 == 0 && ldp % 4 == 0 &&
                  n_part >= 1 && n_part <= 1024, "pool_bias_partials: bad sizes (H, ldg, ldp multiples of 4)");
    GSAGE_REQUIRE((((uintptr_t)g | (uintptr_t)pooled | (uintptr_t)part) & 15) == 0,
                  "pool_bias_partials: buffers must be 16-byte aligned");
    launch(k_pool_bias_partials, dim3((unsigned)n_part, (unsigned)ceil_div(H, 1024)), dim3(256), 0,
           (hipStream_t)stream, g, ldg, pooled, ldp, M, H, part);
    return check_launch("pool_bias_partials");
}

int gsage_pool_merge_bwd(const void *Hprev, int dtype, int64_t ldh, const float *DX, int64_t ldx, int64_t r_x,
                         const float *DN, int64_t ldn, int64_t r0, void *dH, int64_t ldo, int64_t R,
                         int32_t D, void *stream)
{
    GSAGE_REQUIRE(dtype == GSAGE_BF16 || dtype == GSAGE_F32, "pool_merge_bwd: bad dtype");
    GSAGE_REQUIRE(Hprev && DX && DN && dH, "pool_merge_bwd: null pointer");
    GSAGE_REQUIRE(D > 0 && D % 4 == 0 && ldh % 4 == 0 && ldx % 4 == 0 && ldn % 4 == 0 && ldo % 4 == 0,
                  "pool_merge_bwd: D and leading dimensions must be multiples of 4");
    GSAGE_REQUIRE(R >= 0 && r_x >= 0 && r_x <= R && r0 >= 0 && r0 <= R, "pool_merge_bwd: bad row ranges");
    if (R == 0) return GSAGE_OK;
    if (dtype == GSAGE_F32)
        launch(k_pool_merge_bwd<float>, dim3(grid_for(R * (D / 4), 4096)), dim3(256), 0, (hipStream_t)stream,
               (const float *)Hprev, ldh, DX, ldx, r_x, DN, ldn, r0, (float *)dH, ldo, R, D);
    else
        launch(k_pool_merge_bwd<uint16_t>, dim3(grid_for(R * (D / 4), 4096)), dim3(256), 0, (hipStream_t)stream,
               (const uint16_t *)Hprev, ldh, DX, ldx, r_x, DN, ldn, r0, (uint16_t *)dH, ldo, R, D);
    return check_launch("pool_merge_bwd");
}

int gsage_bwd_merge(const void *H, int dtype, int64_t ldh, const float *DG, int64_t ldg, int64_t dagg_off,
                    void *dH, int64_t ldo, int64_t R, int64_t r_x, int32_t D, int32_t n_hops,
                    const int64_t *off, const int32_t *fan, void *stream)
{
    GSAGE_REQUIRE(dtype == GSAGE_BF16 || dtype == GSAGE_F32, "bwd_merge: bad dtype");
    GSAGE_REQUIRE(H && DG && dH && off && fan, "bwd_merge: null pointer");
    GSAGE_REQUIRE(n_hops >= 1 && n_hops <= 6, "bwd_merge: 1..6 hops");
    GSAGE_REQUIRE(D > 0 && D % 4 == 0 && ldh % 4 == 0 && ldg % 4 == 0 && ldo % 4 == 0 && dagg_off % 4 == 0,
                  "bwd_merge: D and leading dimensions must be multiples of 4");
    if (R == 0) return GSAGE_OK;
    MergeParams q;
    q.H = H; q.DG = DG; q.dH = dH; q.ldh = ldh; q.ldg = ldg; q.ldo = ldo;
    q.dagg_off = dagg_off; q.R = R; q.r_x = r_x; q.D = D; q.n_hops = n_hops;
    for (int i = 0; i < 6; ++i) { q.off[i] = i < n_hops ? off[i] : 0; q.fan[i] = i < n_hops ? fan[i] : 1; }
    if (dtype == GSAGE_F32)
        launch(k_bwd_merge<float>, dim3(grid_for(R * (D / 4), 4096)), dim3(256), 0, (hipStream_t)stream, q);
    else if (!(getenv("GSAGE_MERGE_V8") && atoi(getenv("GSAGE_MERGE_V8")) == 0) &&      // (=0: the 4-column kernel, for A/B)
             D % 8 == 0 && ldh % 8 == 0 && ldo % 8 == 0 && ldg % 4 == 0 && dagg_off % 4 == 0 &&
             (((uintptr_t)H | (uintptr_t)dH | (uintptr_t)DG) & 15) == 0)
        launch(k_bwd_merge_v8, dim3(grid_for(ceil_div(R * (D / 8), 2), 4096)), dim3(256), 0, (hipStream_t)stream, q);
    else
        launch(k_bwd_merge<uint16_t>, dim3(grid_for(R * (D / 4), 4096)), dim3(256), 0, (hipStream_t)stream, q);
    return check_launch("bwd_merge");
}

}  // extern "C"
