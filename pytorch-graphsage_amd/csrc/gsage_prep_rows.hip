// gsage_prep_rows.hip -- the trainable node-embedding prep as two row pipelines (gfx950, bf16 operands, E = 64).
//
// Reference nn_modules.py:126-155 (NodeEmbeddingPrep: fc(embedding(ids)), seeds read the spare row n_nodes) and its
// autograd.  Every weight of the prep fits a wave's registers (prep.fc 64 x 64, att.0 32 x 64), so a frontier row
// goes through its whole chain inside one launch instead of making an HBM round trip per link:
//
//   forward   table row (fp32, through the frontier's ids) -> bf16 -> prep.fc + bias -> level-0 row (bf16)
//             replaces the embedding gather (k_gather_mean_multi) and the prep.fc GEMM launch
//   backward  d level-0 row = [d hid W0] + [d x] + [ws * d agg(parent)]   (gsage_attn_merge_bwd2 and the GEMM that fed it)
//             -> bf16 operand copy (the prep.fc weight gradient's operand), column sums (prep.fc.bias gradient)
//             -> d embedding row = d row Wp  -> fp32 atomic adds into the table's gradient (the dense nn.Embedding
//                gradient of the reference), the seeds' rows summed per 16-row tile first (they all read one spare row)
//             replaces the GEMM through att.0^T, the merge kernel, two column-sum launches, the GEMM through prep.fc^T
//             and the scatter-add
//
// A wave owns 16 rows at a time (one MFMA tile, v_mfma_f32_16x16x32_bf16).  Products are taken transposed
// (out^T = W x^T: A = the weight's rows, B = the rows' 16-byte pieces straight from registers), which leaves lane
// (row, q) with outputs 16 t + 4 q .. + 3 of ITS row -- the B fragment of the next product under a fixed permutation
// of the reduction index (applied to that product's weight fragments once per launch), so the chain needs no LDS
// between links; only the atomics go through LDS, to leave as whole 256-byte rows.
#include "gsage_common.h"
#include "gsage_mma_dev.h"

namespace gsage {

typedef float pr_f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t pr_u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(2))) __bf16 pr_bf16x2;
typedef __attribute__((ext_vector_type(2))) float pr_f32x2;

constexpr int PR_E = 64;

__device__ __forceinline__ pr_f32x4 pr_mfma(const vec16 &a, const vec16 &b, const pr_f32x4 &c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0,
                                                   0);
}

__device__ __forceinline__ uint32_t pr_pack2(float lo, float hi)       // round to nearest even, as f32_to_bf16
{
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(pr_f32x2{lo, hi}, pr_bf16x2));
}

template <int CTRL>
__device__ __forceinline__ float pr_dpp(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}

// sum over the 16 lanes of a row of lanes (every lane ends with the same bits)
__device__ __forceinline__ float pr_row_sum(float v)
{
    v += pr_dpp<0xB1>(v);      // i ^ 1
    v += pr_dpp<0x4E>(v);      // i ^ 2
    v += pr_dpp<0x141>(v);     // 7 - i within a half row
    return v + pr_dpp<0x140>(v);   // 15 - i
}

struct PrepRowsFwd {
    const float *table;        // embedding table [n_rows][ldt] fp32
    int64_t ldt;
    const int64_t *ids;        // frontier: position pos reads row (pos < n_seed ? spare : ids[pos])
    int64_t n_seed, spare;
    const uint16_t *W;         // prep.fc operand copy [64][ldw] bf16
    int64_t ldw;
    const float *bias;         // [64]
    int64_t M;
    uint16_t *eraw;            // out: the embedding rows in bf16 [M][lde] (operand of the prep.fc weight gradient)
    int64_t lde;
    uint16_t *out;             // out: prep output rows [M][ldo] (the caller's pointer is already at the prep's columns)
    int64_t ldo;
};

__global__ void __launch_bounds__(256)
k_prep_rows_fwd(const PrepRowsFwd p)
{
    const int lane = threadIdx.x & 63, r16 = lane & 15, q = lane >> 4;
    // prep.fc as the A operand: out column 16 t + r16, reduction 32 ks + 8 q .. + 7
    vec16 wf[4][2];
    pr_f32x4 b4[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            wf[t][ks] = *reinterpret_cast<const vec16 *>(p.W + (16 * t + r16) * p.ldw + 32 * ks + 8 * q);
        b4[t] = *reinterpret_cast<const pr_f32x4 *>(p.bias + 16 * t + 4 * q);
    }
    const int64_t n_tiles = (p.M + 15) >> 4;
    const int64_t wv = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = (int64_t)gridDim.x * 4;
    for (int64_t tile = wv; tile < n_tiles; tile += n_waves) {
        const int64_t pos = tile * 16 + r16;
        const bool live = pos < p.M;
        const int64_t pc = live ? pos : p.M - 1;
        const int64_t row = pc < p.n_seed ? p.spare : p.ids[pc];
        const float *src = p.table + row * p.ldt + 8 * q;
        pr_f32x4 x[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            x[ks][0] = *reinterpret_cast<const pr_f32x4 *>(src + 32 * ks);
            x[ks][1] = *reinterpret_cast<const pr_f32x4 *>(src + 32 * ks + 4);
        }
        vec16 xb[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            xb[ks][0] = pr_pack2(x[ks][0][0], x[ks][0][1]);
            xb[ks][1] = pr_pack2(x[ks][0][2], x[ks][0][3]);
            xb[ks][2] = pr_pack2(x[ks][1][0], x[ks][1][1]);
            xb[ks][3] = pr_pack2(x[ks][1][2], x[ks][1][3]);
            if (live) *reinterpret_cast<vec16 *>(p.eraw + pos * p.lde + 32 * ks + 8 * q) = xb[ks];
        }
        pr_f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[t] = b4[t];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) acc[t] = pr_mfma(wf[t][ks], xb[ks], acc[t]);
        }
        if (live) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                *reinterpret_cast<pr_u32x2 *>(p.out + pos * p.ldo + 16 * t + 4 * q) =
                    pr_u32x2{pr_pack2(acc[t][0], acc[t][1]), pr_pack2(acc[t][2], acc[t][3])};
        }
    }
}

struct PrepRowsBwd {
    const uint16_t *dhid;      // optional: d hid of the level's att.0 [R][lddh] bf16 (32 columns) ...
    int64_t lddh;
    const uint16_t *W0T;       // ... and att.0's transposed operand copy [64][ldw0t]: W0T[c][k] = W0[k][c]
    int64_t ldw0t;
    const float *DATT;         // or (dhid == NULL) the gradient through att(.) itself, fp32 [R][ldatt]; may be NULL too
    int64_t ldatt;
    const float *DX;           // through fc_x: rows < r_x; may be NULL
    int64_t ldx, r_x;
    const float *DAGG;         // of the parents [.][ldagg]
    int64_t ldagg;
    const float *ws;           // weight of every (parent, child) pair in hop order, NULL: 1 / fan-out
    int32_t n_hops;
    int64_t off[6];
    int32_t fan[6];
    int64_t R;
    uint16_t *din0;            // out: bf16 operand copy of d prep output [R][ldd]
    int64_t ldd;
    float *bias_part;          // out: [gridDim.x][64] partial column sums of d prep output
    const uint16_t *WpT;       // prep.fc's transposed operand copy [64][ldwpt]: WpT[e][c] = Wp[c][e]
    int64_t ldwpt;
    const int64_t *ids;
    int64_t n_seed, spare;
    float *g_table;            // the table's gradient [n_rows][ldg]: fp32 atomic adds
    int64_t ldg;
    float *deraw;              // or (not NULL): d embedding rows written here [R][ldde], no atomics
    int64_t ldde;
};

__global__ void __launch_bounds__(256)
k_prep_rows_bwd(const PrepRowsBwd p)
{
    __shared__ __attribute__((aligned(16))) float tile_s[4][16][PR_E];      // a wave's 16 d embedding rows
    __shared__ float bsum_s[4][PR_E];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, q = lane >> 4;
    // att.0^T as the A operand of datt^T = W0^T dhid^T: column c = 16 t + r16 of the level-0 row, reduction 8 q .. + 7
    vec16 w0f[4];
    if (p.dhid) {
#pragma unroll
        for (int t = 0; t < 4; ++t) w0f[t] = *reinterpret_cast<const vec16 *>(p.W0T + (16 * t + r16) * p.ldw0t + 8 * q);
    }
    // prep.fc^T as the A operand of deraw^T = Wp^T din0^T: output e = 16 t + r16; reduction slot (ks, q, j) is column
    // 16 (2 ks + (j >> 2)) + 4 q + (j & 3) -- the order in which lane (row, q) holds its row's d prep output
    vec16 wpf[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const uint16_t *wr = p.WpT + (16 * t + r16) * p.ldwpt;
            const pr_u32x2 lo = *reinterpret_cast<const pr_u32x2 *>(wr + 16 * (2 * ks) + 4 * q);
            const pr_u32x2 hi = *reinterpret_cast<const pr_u32x2 *>(wr + 16 * (2 * ks + 1) + 4 * q);
            wpf[t][ks] = vec16{lo[0], lo[1], hi[0], hi[1]};
        }
    pr_f32x4 bsum[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) bsum[t] = pr_f32x4{0.f, 0.f, 0.f, 0.f};

    const int64_t n_tiles = (p.R + 15) >> 4;
    const int64_t wv = (int64_t)blockIdx.x * 4 + wave, n_waves = (int64_t)gridDim.x * 4;
    for (int64_t tile = wv; tile < n_tiles; tile += n_waves) {
        const int64_t pos = tile * 16 + r16;
        const bool live = pos < p.R;
        const int64_t pc = live ? pos : p.R - 1;
        // hop and parent of the row
        int k = 0;
#pragma unroll
        for (int j = 1; j < 6; ++j)
            if (j < p.n_hops && pc >= p.off[j]) k = j;
        const int64_t parent = k >= 1 ? p.off[k - 1] + (pc - p.off[k]) / p.fan[k] : 0;
        const float wgt = k >= 1 ? (p.ws ? p.ws[pc - p.off[1]] : 1.f / (float)p.fan[k]) : 0.f;
        pr_f32x4 v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = pr_f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.dhid) {
            const vec16 dh = *reinterpret_cast<const vec16 *>(p.dhid + pc * p.lddh + 8 * q);
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = pr_mfma(w0f[t], dh, v[t]);
        } else if (p.DATT) {
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = *reinterpret_cast<const pr_f32x4 *>(p.DATT + pc * p.ldatt + 16 * t + 4 * q);
        }
        if (p.DX) {
            const int64_t rx = pc < p.r_x ? pc : 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const pr_f32x4 d = *reinterpret_cast<const pr_f32x4 *>(p.DX + rx * p.ldx + 16 * t + 4 * q);
                if (pc < p.r_x) v[t] += d;
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const pr_f32x4 d = *reinterpret_cast<const pr_f32x4 *>(p.DAGG + parent * p.ldagg + 16 * t + 4 * q);
            v[t] += wgt * d;
        }
        // operand copy, column sums, and the B operand of the product through prep.fc^T
        vec16 vb[2];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t lo = pr_pack2(v[t][0], v[t][1]), hi = pr_pack2(v[t][2], v[t][3]);
            if (live) {
                *reinterpret_cast<pr_u32x2 *>(p.din0 + pos * p.ldd + 16 * t + 4 * q) = pr_u32x2{lo, hi};
                bsum[t] += v[t];
            }
            vb[t >> 1][2 * (t & 1)] = lo;
            vb[t >> 1][2 * (t & 1) + 1] = hi;
        }
        pr_f32x4 d[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            d[t] = pr_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) d[t] = pr_mfma(wpf[t][ks], vb[ks], d[t]);
        }
        if (p.deraw) {
            if (live) {
#pragma unroll
                for (int t = 0; t < 4; ++t) *reinterpret_cast<pr_f32x4 *>(p.deraw + pos * p.ldde + 16 * t + 4 * q) = d[t];
            }
            continue;
        }
        // the tile's 16 d embedding rows through LDS, so that an atomic instruction covers one whole 256-byte row
        const bool all_seeds = tile * 16 + 16 <= p.n_seed;           // (wave-uniform) every row of the tile is a seed's:
        if (all_seeds) {                                             // one sum per tile onto the spare row
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) d[t][e] = pr_row_sum(d[t][e]);
            if (r16 == 0) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomicAdd(p.g_table + p.spare * p.ldg + 16 * t + 4 * q + e, d[t][e]);
            }
            continue;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) *reinterpret_cast<pr_f32x4 *>(&tile_s[wave][r16][16 * t + 4 * q]) = d[t];
        const int64_t row_l = pc < p.n_seed ? p.spare : p.ids[pc];   // (lane r16 of every q: the row of tile row r16)
        __builtin_amdgcn_wave_barrier();
        const int n_live = (int)(p.R - tile * 16 < 16 ? p.R - tile * 16 : 16);
        for (int j = 0; j < n_live; ++j) {
            const int64_t rj = __shfl(row_l, j, 64);
            atomicAdd(p.g_table + rj * p.ldg + lane, tile_s[wave][j][lane]);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // column sums of the workgroup: over the 16 row lanes, then over the four waves (fixed order)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) bsum[t][e] = pr_row_sum(bsum[t][e]);
    if (r16 == 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) *reinterpret_cast<pr_f32x4 *>(&bsum_s[wave][16 * t + 4 * q]) = bsum[t];
    }
    __syncthreads();
    if (threadIdx.x < PR_E)
        p.bias_part[(int64_t)blockIdx.x * PR_E + threadIdx.x] =
            (bsum_s[0][threadIdx.x] + bsum_s[1][threadIdx.x]) + (bsum_s[2][threadIdx.x] + bsum_s[3][threadIdx.x]);
}

}  // namespace gsage

using namespace gsage;

extern "C" int gsage_prep_rows_ok(int dtype, int64_t E)
{
    return dtype == GSAGE_BF16 && E == PR_E ? 1 : 0;
}

extern "C" int gsage_prep_rows_fwd(const float *table, int64_t ldt, const int64_t *ids, int64_t n_seed, int64_t spare,
                                   const void *W, int64_t ldw, const float *bias, int64_t M, int64_t E, void *eraw,
                                   int64_t lde, void *out, int64_t ldo, void *stream)
{
    GSAGE_REQUIRE(E == PR_E, "prep_rows_fwd: the row pipelines cover 64-wide embeddings");
    GSAGE_REQUIRE(M >= 0 && n_seed >= 0 && ldt >= E && ldt % 4 == 0 && ldw >= E && ldw % 8 == 0 && lde >= E && lde % 8 == 0 &&
                  ldo >= E && ldo % 4 == 0, "prep_rows_fwd: bad leading dimension");
    if (M == 0) return GSAGE_OK;
    GSAGE_REQUIRE(table && (ids || n_seed >= M) && W && bias && eraw && out, "prep_rows_fwd: null pointer");
    GSAGE_REQUIRE((((uintptr_t)table | (uintptr_t)W | (uintptr_t)bias | (uintptr_t)eraw) & 15) == 0 && ((uintptr_t)out & 7) == 0,
                  "prep_rows_fwd: misaligned pointer");
    PrepRowsFwd p;
    p.table = table; p.ldt = ldt; p.ids = ids; p.n_seed = n_seed; p.spare = spare; p.W = (const uint16_t *)W; p.ldw = ldw;
    p.bias = bias; p.M = M; p.eraw = (uint16_t *)eraw; p.lde = lde; p.out = (uint16_t *)out; p.ldo = ldo;
    int64_t g = ceil_div(ceil_div(M, 16), 4);
    if (g > 4096) g = 4096;
    launch(k_prep_rows_fwd, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("prep_rows_fwd");
}

extern "C" int gsage_prep_rows_bwd(const void *dhid, int64_t lddh, const void *W0T, int64_t ldw0t, const float *DATT,
                                   int64_t ldatt, const float *DX, int64_t ldx, int64_t r_x, const float *DAGG,
                                   int64_t ldagg, const float *ws, int32_t n_hops, const int64_t *off, const int32_t *fan,
                                   int64_t R, int64_t E, void *din0, int64_t ldd, float *bias_part, int32_t n_part,
                                   const void *WpT, int64_t ldwpt, const int64_t *ids, int64_t n_seed, int64_t spare,
                                   float *g_table, int64_t ldg, float *deraw, int64_t ldde, void *stream)
{
    GSAGE_REQUIRE(E == PR_E, "prep_rows_bwd: the row pipelines cover 64-wide embeddings");
    GSAGE_REQUIRE(n_hops >= 2 && n_hops <= 6 && R >= 0 && r_x >= 0 && r_x <= R && n_part >= 1 && n_part <= 1024,
                  "prep_rows_bwd: bad sizes");
    GSAGE_REQUIRE((!dhid || (W0T && lddh >= 32 && lddh % 8 == 0 && ldw0t >= 32 && ldw0t % 8 == 0)) &&
                  (!DATT || (ldatt >= E && ldatt % 4 == 0)) && (!DX || (ldx >= E && ldx % 4 == 0)) && ldagg >= E &&
                  ldagg % 4 == 0 && ldd >= E && ldd % 4 == 0 && ldwpt >= E && ldwpt % 4 == 0 &&
                  (deraw ? (ldde >= E && ldde % 4 == 0) : ldg >= E), "prep_rows_bwd: bad leading dimension");
    GSAGE_REQUIRE(DAGG && off && fan && din0 && bias_part && WpT && (deraw || g_table) && (ids || n_seed >= R),
                  "prep_rows_bwd: null pointer");
    GSAGE_REQUIRE((((uintptr_t)dhid | (uintptr_t)W0T | (uintptr_t)DATT | (uintptr_t)DX | (uintptr_t)DAGG | (uintptr_t)deraw |
                    (uintptr_t)bias_part) & 15) == 0 && (((uintptr_t)din0 | (uintptr_t)WpT) & 7) == 0,
                  "prep_rows_bwd: misaligned pointer");
    PrepRowsBwd p;
    p.dhid = (const uint16_t *)dhid; p.lddh = lddh; p.W0T = (const uint16_t *)W0T; p.ldw0t = ldw0t; p.DATT = DATT;
    p.ldatt = ldatt; p.DX = DX; p.ldx = ldx; p.r_x = r_x; p.DAGG = DAGG; p.ldagg = ldagg; p.ws = ws; p.n_hops = n_hops;
    for (int i = 0; i < 6; ++i) { p.off[i] = i < n_hops ? off[i] : 0; p.fan[i] = i < n_hops ? fan[i] : 1; }
    p.R = R; p.din0 = (uint16_t *)din0; p.ldd = ldd; p.bias_part = bias_part; p.WpT = (const uint16_t *)WpT; p.ldwpt = ldwpt;
    p.ids = ids; p.n_seed = n_seed; p.spare = spare; p.g_table = g_table; p.ldg = ldg; p.deraw = deraw; p.ldde = ldde;
    // (every workgroup writes its row of bias_part: zeros from those without a tile)
    launch(k_prep_rows_bwd, dim3((unsigned)n_part), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("prep_rows_bwd");
}
