// gsage_attn_fused.hip -- K4 / K4' with the attention MLP inside: one pass over a hop's child rows per direction.
//
// Replaces, for the LAST hop of an attention level (the bulk of its rows: Reddit's 128 000 of 141 312), what
// engine.FusedAttnTrainStep issued as separate launches over the same rows (reference nn_modules.py:305-321):
//
//   forward    hid = tanh(rows W0^T)   (a GEMM launch that streamed every row)      nn_modules.py:293-295
//              a   = hid W2^T          (k_attn_mlp2_fwd)                            nn_modules.py:296
//              s   = <a_child, a_parent>, w = softmax over the fan-out, agg = sum w * row   (K4, the rows again)
//   backward   dws = <row, d agg>, softmax backward, d a_parent, d a_child          (K4', the rows a third time)
//              da  = bf16(d a_child),  dhid = bf16((da W2) (1 - hid^2))              (k_attn_mlp2_bwd)
//
// A child row is fetched ONCE per direction: global -> LDS by LDS-DMA (`global_load_lds_dwordx4`, 16 bytes per lane,
// whole row pieces of up to 1 KiB per instruction), the hidden layer of the attention MLP comes off the matrix cores
// from that LDS tile (v_mfma_f32_16x16x32_bf16, W0 resident in LDS for the whole launch), and the weighted sum /
// the dot products with d agg read the same tile with 16-byte lane reads.
//
// Decomposition: a wavefront owns a whole parent (its n <= 16 children are one 16-row MFMA tile) and a private LDS
// tile; there is no workgroup barrier after W0 has been staged, so the waves of a CU drift apart and one wave's
// loads cover another's arithmetic.  hid^T = W0 X^T is computed (A = W0 rows, B = child rows: both operands are read
// as "16 contiguous bytes of one row", no transposition), which leaves lane (row, q) with eight hidden units of ITS
// row -- exactly the B fragment of a^T = W2 hid^T under a fixed permutation of the reduction index (applied to W2's
// fragment once per launch), so the second layer needs no LDS round trip either.
//
// LDS image of a row tile: row-major [n][CH] 16-byte chunks, chunk c of row r at slot r * CH + (c ^ swz(r)),
// swz(r) = (4 - (r >> 2)) & 3: the DMA writes LDS linearly (wave-uniform base + lane * 16), so the permutation is
// applied to the SOURCE chunk each lane fetches; with it the four 16-lane groups of a fragment's ds_read_b128
// (lanes = 16 rows x chunks 4 ks + {0..3}) touch 16 distinct 16-byte slots of the 256-byte bank row.
#include "gsage_common.h"
#include "gsage_mma_dev.h"

namespace gsage {

typedef float af_f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t af_u32x2 __attribute__((ext_vector_type(2)));

struct AttnFusedFwd {
    const uint16_t *table;      // child rows: table[(ids ? ids[pos] : row0 + pos) * ld], pos = parent * n + j
    int64_t ld;
    const int64_t *ids;
    int64_t row0;
    const uint16_t *W0;         // att.0 operand copy [32][ldw0], zero beyond D
    int64_t ldw0;
    const uint16_t *W2;         // att.2 operand copy [32][ldw2]
    int64_t ldw2;
    const float *xa;            // a of the parents [M][xa_ld]
    int64_t xa_ld;
    int64_t M;
    int32_t n, D;
    uint16_t *hid;              // out: hidden layer of the children [M n][hid_ld]
    int64_t hid_ld;
    float *a;                   // out: a of the children [M n][a_ld]
    int64_t a_ld;
    float *ws;                  // out: softmax weights [M n]
    uint16_t *agg_lp;           // out: bf16 operand copy of the aggregate [M][lp_ld]
    int64_t lp_ld;
};

struct AttnFusedBwd {
    const uint16_t *table;
    int64_t ld;
    const int64_t *ids;
    int64_t row0;
    const uint16_t *W2T;        // transposed operand copy of att.2: W2T[k][h] = W2[h][k], [32][ldw2t]
    int64_t ldw2t;
    const float *g;             // d agg of the parents [M][g_ld]
    int64_t g_ld;
    const float *ws;            // [M n]
    const float *na;            // a of the children [M n][na_ld]
    int64_t na_ld;
    const float *xa;            // a of the parents [M][xa_ld]
    int64_t xa_ld;
    const uint16_t *hid;        // hidden layer of the children [M n][hid_ld]
    int64_t hid_ld;
    int64_t M;
    int32_t n, D;
    uint16_t *da;               // out: bf16(d a) of the children [M n][da_ld]
    int64_t da_ld;
    uint16_t *dhid;             // out: bf16((da W2)(1 - hid^2)) [M n][dhid_ld]
    int64_t dhid_ld;
    float *dxa;                 // out: d a of the parents through this hop [M][dxa_ld]
    int64_t dxa_ld;
};

constexpr int AF_NMAX = 16;          // children per parent (one 16-row MFMA tile)
constexpr int AF_MAX_WAVES = 8;      // per workgroup (512 threads: the register allocator may use up to 256 VGPRs)
constexpr int AF_SCRATCH = 4;        // 16-byte slots of per-wave scratch behind a wave's row tile (16 floats)

typedef __attribute__((ext_vector_type(4))) short af_s16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 af_bf16x2;
typedef __attribute__((ext_vector_type(2))) float af_f32x2;
typedef __attribute__((address_space(3))) af_s16x4 af_lds_s16x4;

__device__ __forceinline__ int af_swz(int row) { return (4 - ((row >> 2) & 3)) & 3; }

__device__ __forceinline__ af_f32x4 af_mfma(const vec16 &a, const vec16 &b, const af_f32x4 &c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0,
                                                   0);
}

// two floats -> two bf16 in one dword, round to nearest even (v_cvt_pk_bf16_f32: the values f32_to_bf16 gives)
__device__ __forceinline__ uint32_t af_pack2(float lo, float hi)
{
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(af_f32x2{lo, hi}, af_bf16x2));
}

__device__ __forceinline__ float af_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float af_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

__device__ __forceinline__ void af_unpack(const vec16 &raw, float (&f)[8])
{
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f[2 * e] = af_lo(raw[e]);
        f[2 * e + 1] = af_hi(raw[e]);
    }
}

// tanh(v) = sign(v) (1 - 2 / (exp(2 |v|) + 1)) on v_exp_f32 / v_rcp_f32 (~1 ulp each): |error| ~ 2e-7, far below the
// bf16 rounding that follows (the GEMM epilogue of the separate launches uses expf and a true division: ~1e-7)
__device__ __forceinline__ float af_tanh(float v)
{
    const float e = __expf(2.f * fabsf(v));
    const float t = 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
    return copysignf(t, v);
}

// The n row ids of parent p: lane j < n keeps the (low dword of the) table row of child j.  In two halves so that the
// load is unconditional and its value is first looked at where the caller wants the wait: af_id_request returns what
// the load brings (without a row list it reads the table -- any mapped address -- and the value is dropped),
// af_id_value turns it into the row.  (A load inside a branch on `ids` makes the compiler wait for EVERY outstanding
// request at the join.)
__device__ __forceinline__ uint32_t af_id_request(const int64_t *ids, const void *table, int64_t p, int n, int lane)
{
    const int64_t pos = p * n + (lane < n ? lane : n - 1);
    const uint32_t *src = ids ? reinterpret_cast<const uint32_t *>(ids + pos) : reinterpret_cast<const uint32_t *>(table);
    return *src;                                                           // (row ids are < 2^31: the low dword)
}

__device__ __forceinline__ uint32_t af_id_value(uint32_t requested, const int64_t *ids, int64_t row0, int64_t p, int n, int lane)
{
    return ids ? requested : (uint32_t)(row0 + p * n + (lane < n ? lane : n - 1));
}

__device__ __forceinline__ float af_readlane(float v, int l)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// rows of one parent -> the wave's LDS tile: instruction i fills slots 64 i .. 64 i + 63 (1 KiB, row-major).
// Rows of >= 64 chunks: an instruction covers pieces of at most two rows, whose ids come from v_readlane (scalar
// index); shorter rows take the id of each lane's row from a ds_bpermute.
template <int KS>
__device__ __forceinline__ void af_issue_tile(const uint16_t *table, int64_t ld, uint32_t idreg, int n, vec16 *xb, int lane)
{
    constexpr int CH = 4 * KS;
    const int n_instr = (n * CH + 63) >> 6;                     // wave-uniform
    for (int i = 0; i < n_instr; ++i) {
        const int s = 64 * i + lane;
        const int row = s / CH;                                  // (constant divisor)
        const int cp = s - row * CH;
        const int c = cp ^ af_swz(row);
        uint32_t id;
        if constexpr (CH >= 64) {
            const int rf = (64 * i) / CH;                        // scalar: the row of the instruction's first slot
            const uint32_t ida = (uint32_t)__builtin_amdgcn_readlane((int)idreg, rf < n ? rf : n - 1);
            const uint32_t idb = (uint32_t)__builtin_amdgcn_readlane((int)idreg, rf + 1 < n ? rf + 1 : n - 1);
            id = row == rf ? ida : idb;
        } else {
            id = (uint32_t)__shfl((int)idreg, row < n ? row : n - 1, 64);
        }
        // (inline asm, not the builtin: the compiler would order every later LDS read behind ALL outstanding LDS-DMA,
        //  i.e. wait for the NEXT parent's tile in the middle of this one's arithmetic; the waits are the callers')
        const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void_t *)(xb + 64 * i));
        const uint16_t *src = table + (int64_t)id * ld + c * 8;
        if (row < n)
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(dst) : "memory", "m0");
    }
}

// Lane-crossing inside a 16-lane row without LDS traffic (v_*_dpp): partners i ^ 1, i ^ 2 (quad permutations), 7 - i
// within a half row, 15 - i within the row.  A sum / max over the row taken in that order pairs equal partial results at
// every step, so all 16 lanes end with the same bits (fp add / max commute).
template <int CTRL>
__device__ __forceinline__ float af_dpp(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
constexpr int AF_DPP_XOR1 = 0xB1, AF_DPP_XOR2 = 0x4E, AF_DPP_HALF_MIRROR = 0x141, AF_DPP_MIRROR = 0x140;

__device__ __forceinline__ float af_group16_max(float v)
{
    v = fmaxf(v, af_dpp<AF_DPP_XOR1>(v));
    v = fmaxf(v, af_dpp<AF_DPP_XOR2>(v));
    v = fmaxf(v, af_dpp<AF_DPP_HALF_MIRROR>(v));
    return fmaxf(v, af_dpp<AF_DPP_MIRROR>(v));
}

__device__ __forceinline__ float af_group16_sum(float v)
{
    v += af_dpp<AF_DPP_XOR1>(v);
    v += af_dpp<AF_DPP_XOR2>(v);
    v += af_dpp<AF_DPP_HALF_MIRROR>(v);
    return v + af_dpp<AF_DPP_MIRROR>(v);
}

// ds_read_b64_tr_b16 as inline asm with hand-counted waits: through the builtin the compiler orders the read behind
// EVERY outstanding vector-memory request (the intrinsic carries no memory operand), i.e. it would wait for the next
// parent's rows in the middle of this parent's arithmetic.  LDS requests of a wave complete in issue order.
template <int T, int NT>
__device__ __forceinline__ void af_tr_read(af_s16x4 &dst, uint32_t a_even, uint32_t a_odd)
{
    if constexpr (T < NT)
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"((T & 1) ? a_odd : a_even), "n"((T >> 1) * 64) : "memory");
}

template <int R, int NT>
__device__ __forceinline__ void af_tr_group(af_s16x4 (&xt)[8], uint32_t a_even, uint32_t a_odd)
{
    af_tr_read<8 * R + 0, NT>(xt[0], a_even, a_odd);
    af_tr_read<8 * R + 1, NT>(xt[1], a_even, a_odd);
    af_tr_read<8 * R + 2, NT>(xt[2], a_even, a_odd);
    af_tr_read<8 * R + 3, NT>(xt[3], a_even, a_odd);
    af_tr_read<8 * R + 4, NT>(xt[4], a_even, a_odd);
    af_tr_read<8 * R + 5, NT>(xt[5], a_even, a_odd);
    af_tr_read<8 * R + 6, NT>(xt[6], a_even, a_odd);
    af_tr_read<8 * R + 7, NT>(xt[7], a_even, a_odd);
}

// accumulator R of the weighted sum: its eight column tiles' fragments were requested one group ago; the next
// group's are requested before this group's MFMAs, which wait until only those (at most eight) are outstanding
template <int R, int NACC, int NT>
__device__ __forceinline__ void af_wsum_groups(af_f32x4 (&acc)[NACC], af_s16x4 (&xt)[2][8], const af_s16x4 (&bsel)[8],
                                               uint32_t a_even, uint32_t a_odd)
{
    if constexpr (R < NACC) {
        constexpr int next = R + 1 < NACC ? (NT - 8 * (R + 1) < 8 ? NT - 8 * (R + 1) : 8) : 0;
        if constexpr (R + 1 < NACC) af_tr_group<R + 1, NT>(xt[(R + 1) & 1], a_even, a_odd);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(next) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        // (a last group of fewer than eight tiles: its spare column pairs take the group's tiles again, so that every
        //  lane ends with a real tile's sums -- the stores behind are then the same for every lane)
        constexpr int cnt = NT - 8 * R < 8 ? NT - 8 * R : 8;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            acc[R] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(xt[R & 1][u % cnt], bsel[u], acc[R], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        af_wsum_groups<R + 1, NACC, NT>(acc, xt, bsel, a_even, a_odd);
    }
}

// Forward.  Nothing below the tile issue depends on the fan-out except clamped row numbers: rows n .. 15 of the
// 16-row tile are whatever lies behind the wave's tile in LDS (any bits), they only reach MFMA outputs of their own
// rows / carry zero weight, and every lane that owns such a row is masked where it matters.
template <int KS>
__global__ void __launch_bounds__(AF_MAX_WAVES * 64)
k_attn_fused_fwd(const AttnFusedFwd p)
{
    constexpr int CH = 4 * KS;
    constexpr int NT = CH / 2;                  // 16-column tiles of a row
    constexpr int NACC = (NT + 7) / 8;          // eight column tiles share an accumulator (one per pair of MFMA columns)
    extern __shared__ __attribute__((aligned(16))) char af_smem[];
    vec16 *w0s = reinterpret_cast<vec16 *>(af_smem);            // [32][CH], swizzled like a row tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), n_waves = blockDim.x >> 6;
    const int n = p.n;
    for (int s = tid; s < 32 * CH; s += blockDim.x) {
        const int row = s / CH, cp = s - row * CH;
        w0s[s] = *reinterpret_cast<const vec16 *>(p.W0 + row * p.ldw0 + (cp ^ af_swz(row)) * 8);
    }
    __syncthreads();
    // per wave: two row tiles (the next parent's rows land while this parent is computed) and the scratch words
    vec16 *xb0 = w0s + 32 * CH + wave * (2 * n * CH + AF_SCRATCH);
    float *scr = reinterpret_cast<float *>(xb0 + 2 * n * CH);
    const int r16 = lane & 15, q = lane >> 4;
    const bool valid = r16 < n;
    const int rc = valid ? r16 : n - 1;          // rows beyond the fan-out repeat the last child (masked where it matters)
    const int sw = af_swz(rc);

    // att.2 as the A operand of a^T = W2 hid^T: reduction slot (q, e) is hidden unit (e < 4 ? 4 q + e : 16 + 4 q + e - 4)
    vec16 w2a[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const uint16_t *wr = p.W2 + (16 * t + r16) * p.ldw2;
        const af_u32x2 lo = *reinterpret_cast<const af_u32x2 *>(wr + 4 * q);
        const af_u32x2 hi = *reinterpret_cast<const af_u32x2 *>(wr + 16 + 4 * q);
        w2a[t] = vec16{lo[0], lo[1], hi[0], hi[1]};
    }
    // the weighted sum on the matrix cores: agg^T [16 columns][.] = X^T [16 columns x 16 rows] w, the column tile's
    // X^T fragment read TRANSPOSED from the row-major tile (ds_read_b64_tr_b16: lane i of a 16-lane group supplies
    // the four columns 4 (i & 3).. of row i >> 2 and receives column i of the group's four rows).  Row (clamped to a
    // fetched one: its weight is zero) and byte address of this lane's piece for even / odd column tiles:
    const int trow = 4 * q + (r16 >> 2) < n ? 4 * q + (r16 >> 2) : n - 1;
    const int tsw = af_swz(trow), tb = (r16 & 3) >> 1;
    const int t_even = (trow * CH + (tb ^ tsw)) * 16 + (r16 & 1) * 8;
    const int t_odd = (trow * CH + ((2 + tb) ^ tsw)) * 16 + (r16 & 1) * 8;

    // Software pipeline over the wave's parents: the rows of parent i + 1 are requested (into the other tile) before
    // parent i is computed.  Requests return in issue order, so once the values requested BEHIND a tile's rows (the
    // next parent's xa, the ids of the parent after it) have arrived the tile has landed: they are touched at the END of
    // a trip, where the only younger requests are the trip's own stores -- every store below is executed by every lane
    // (lanes without a result of their own repeat another lane's store, same address, same bits) so that the code is
    // branch-free and the compiler can count them (s_waitcnt vmcnt(#stores): the stores are not waited for).  The asm
    // statements with a memory clobber pin the issue order.
    const int64_t stride = (int64_t)gridDim.x * n_waves;
    int64_t par = (int64_t)blockIdx.x * n_waves + wave;
    int buf = 0;
    uint32_t id_next = 0u;                                      // (as requested: af_id_value makes it a row)
    af_f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = x0;
    if (par < p.M) {
        uint32_t id0 = af_id_request(p.ids, p.table, par, n, lane);
        asm volatile("" : "+v"(id0));
        af_issue_tile<KS>(p.table, p.ld, af_id_value(id0, p.ids, p.row0, par, n, lane), n, xb0, lane);
        asm volatile("" ::: "memory");
        x0 = *reinterpret_cast<const af_f32x4 *>(p.xa + par * p.xa_ld + 4 * q);
        x1 = *reinterpret_cast<const af_f32x4 *>(p.xa + par * p.xa_ld + 16 + 4 * q);
        id_next = af_id_request(p.ids, p.table, par + stride < p.M ? par + stride : p.M - 1, n, lane);
        asm volatile("" : "+v"(x0), "+v"(x1), "+v"(id_next) : : "memory");      // arrived => the first tile has landed
    }
    while (par < p.M) {
        const vec16 *xb = xb0 + buf * (n * CH);
        const int64_t nxt = par + stride;
        if (nxt < p.M)
            af_issue_tile<KS>(p.table, p.ld, af_id_value(id_next, p.ids, p.row0, nxt, n, lane), n, xb0 + (buf ^ 1) * (n * CH), lane);
        asm volatile("" ::: "memory");
        const int64_t nx = nxt < p.M ? nxt : p.M - 1, nn = nxt + stride < p.M ? nxt + stride : p.M - 1;
        af_f32x4 xn0 = *reinterpret_cast<const af_f32x4 *>(p.xa + nx * p.xa_ld + 4 * q);
        af_f32x4 xn1 = *reinterpret_cast<const af_f32x4 *>(p.xa + nx * p.xa_ld + 16 + 4 * q);
        uint32_t id_nn = af_id_request(p.ids, p.table, nn, n, lane);
        asm volatile("" ::: "memory");

        // hid^T [32 x 16 rows] = W0 [32 x D] X^T; the fragments of step ks + 1 are requested before the MFMAs of step ks
        af_f32x4 h0 = {0.f, 0.f, 0.f, 0.f}, h1 = {0.f, 0.f, 0.f, 0.f};
        {
            constexpr int HB = 4, NB = (KS + HB - 1) / HB;      // k-steps per batch, batches
            vec16 fb[2][HB], fa0[2][HB], fa1[2][HB];
            auto load_batch = [&](int b, int slot) {
#pragma unroll
                for (int u = 0; u < HB; ++u)
                    if (b * HB + u < KS) {
                        const int c = (4 * (b * HB + u) + q) ^ af_swz(r16);
                        fb[slot][u] = xb[rc * CH + ((4 * (b * HB + u) + q) ^ sw)];
                        fa0[slot][u] = w0s[r16 * CH + c];
                        fa1[slot][u] = w0s[(16 + r16) * CH + c];
                    }
            };
            load_batch(0, 0);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (b + 1 < NB) load_batch(b + 1, (b + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);              // (or the scheduler sinks every read to just above its MFMA)
#pragma unroll
                for (int u = 0; u < HB; ++u)
                    if (b * HB + u < KS) {
                        h0 = af_mfma(fa0[b & 1][u], fb[b & 1][u], h0);
                        h1 = af_mfma(fa1[b & 1][u], fb[b & 1][u], h1);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        vec16 hb;
        hb[0] = af_pack2(af_tanh(h0[0]), af_tanh(h0[1]));
        hb[1] = af_pack2(af_tanh(h0[2]), af_tanh(h0[3]));
        hb[2] = af_pack2(af_tanh(h1[0]), af_tanh(h1[1]));
        hb[3] = af_pack2(af_tanh(h1[2]), af_tanh(h1[3]));
        const af_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        const af_f32x4 a0v = af_mfma(w2a[0], hb, zero);          // a[row][4 q + reg]
        const af_f32x4 a1v = af_mfma(w2a[1], hb, zero);          // a[row][16 + 4 q + reg]
        // (lanes of the rows beyond the fan-out computed the last child again: the same stores)
        const int64_t child = par * n + rc;
        {
            uint16_t *hr = p.hid + child * p.hid_ld;
            *reinterpret_cast<af_u32x2 *>(hr + 4 * q) = af_u32x2{hb[0], hb[1]};
            *reinterpret_cast<af_u32x2 *>(hr + 16 + 4 * q) = af_u32x2{hb[2], hb[3]};
            float *ar = p.a + child * p.a_ld;
            *reinterpret_cast<af_f32x4 *>(ar + 4 * q) = a0v;
            *reinterpret_cast<af_f32x4 *>(ar + 16 + 4 * q) = a1v;
        }
        // score of this lane's row, softmax over the rows of the 16-lane group
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) s += a0v[e] * x0[e];
#pragma unroll
        for (int e = 0; e < 4; ++e) s += a1v[e] * x1[e];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        const float mx = af_group16_max(valid ? s : -INFINITY);
        const float ex = valid ? expf(s - mx) : 0.f;
        const float w = ex / af_group16_sum(ex);                 // (0 for the rows beyond the fan-out)
        p.ws[child] = valid ? w : af_readlane(w, n - 1);

        // the weights as the B operand: rows 4 q .. 4 q + 3, split into bf16 high and low parts (w = hi + lo to 2^-17:
        // the sum keeps fp32-grade weights); even MFMA columns carry the high parts, odd ones the low parts
        if (q == 0) scr[r16] = w;
        const af_f32x4 w4 = *reinterpret_cast<const af_f32x4 *>(scr + 4 * q);
        const uint32_t wh0 = af_pack2(w4[0], w4[1]), wh1 = af_pack2(w4[2], w4[3]);
        const uint32_t wl0 = af_pack2(w4[0] - af_lo(wh0), w4[1] - af_hi(wh0));
        const uint32_t wl1 = af_pack2(w4[2] - af_lo(wh1), w4[3] - af_hi(wh1));
        const uint32_t ws0 = (r16 & 1) ? wl0 : wh0, ws1 = (r16 & 1) ? wl1 : wh1;
        af_f32x4 acc[NACC];
#pragma unroll
        for (int r = 0; r < NACC; ++r) acc[r] = zero;
        {
            af_s16x4 bsel[8];                                   // column pair u of an accumulator takes tile 8 r + u
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool mine = (r16 >> 1) == u;
                bsel[u] = __builtin_bit_cast(af_s16x4, af_u32x2{mine ? ws0 : 0u, mine ? ws1 : 0u});
            }
            af_s16x4 xt[2][8];
            const uint32_t lbase = (uint32_t)(uintptr_t)(lds_void_t *)xb;
            const uint32_t a_even = lbase + (uint32_t)t_even, a_odd = lbase + (uint32_t)t_odd;
            __builtin_amdgcn_sched_barrier(0);
            af_tr_group<0, NT>(xt[0], a_even, a_odd);
            af_wsum_groups<0, NACC, NT>(acc, xt, bsel, a_even, a_odd);
        }
        // lane (column pair, q) of accumulator r: columns 16 t + 4 q .. + 3 of tile t = 8 r + pair (a spare pair: the tile
        // it repeated); high + low parts meet in BOTH lanes of the pair (fp add commutes: the same bits, the same store)
#pragma unroll
        for (int r = 0; r < NACC; ++r) {
            af_f32x4 v = acc[r];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += af_dpp<AF_DPP_XOR1>(v[e]);
            constexpr int full = 8;
            const int cnt = NT - 8 * r < full ? NT - 8 * r : full;
            const int t = 8 * r + (r16 >> 1) % cnt;
            *reinterpret_cast<af_u32x2 *>(p.agg_lp + par * p.lp_ld + 16 * t + 4 * q) =
                af_u32x2{af_pack2(v[0], v[1]), af_pack2(v[2], v[3])};
        }
        __builtin_amdgcn_sched_barrier(0);
        // the next parent's values have arrived => its tile has landed (the stores above are younger: not waited for)
        asm volatile("" : "+v"(xn0), "+v"(xn1), "+v"(id_nn) : : "memory");
        par = nxt; buf ^= 1; x0 = xn0; x1 = xn1; id_next = id_nn;
    }
}

// Backward.  Per-wave LDS: two row tiles (as in the forward), d agg of the parent split into bf16 high / low parts
// (2 CH slots), scratch.
template <int ROUNDS>
struct AfBwdIn {                 // what a parent needs besides its rows: requested one parent ahead
    af_f32x4 gv[ROUNDS][2];      // d agg, this lane's column chunk(s)
    af_f32x4 xq0, xq1;           // a of the parent, hidden units 8 q .. 8 q + 7
    af_u32x2 hlo, hhi;           // hid of this lane's row, units 4 q .. + 3 and 16 + 4 q .. + 3
    float wgt;                   // softmax weight of this lane's row
    float nav[AF_NMAX / 2];      // a of children 2 i + half, hidden unit lane & 31
    uint32_t id;                 // ids of the NEXT parent (as requested: af_id_value)
};

template <int KS>
__global__ void __launch_bounds__(AF_MAX_WAVES * 64)
k_attn_fused_bwd(const AttnFusedBwd p)
{
    constexpr int CH = 4 * KS;
    constexpr int ROUNDS = (CH + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) char af_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), n_waves = blockDim.x >> 6;
    const int n = p.n;
    vec16 *xb0 = reinterpret_cast<vec16 *>(af_smem) + wave * (2 * n * CH + 2 * CH + AF_SCRATCH);
    vec16 *ghi = xb0 + 2 * n * CH, *glo = ghi + CH;
    float *scr = reinterpret_cast<float *>(glo + CH);
    const int r16 = lane & 15, q = lane >> 4;
    const bool valid = r16 < n;
    const int rc = valid ? r16 : n - 1;          // rows beyond the fan-out repeat the last child (masked where it matters)
    const int sw = af_swz(rc);
    const int h32 = lane & 31, half = lane >> 5;

    // att.2 as the A operand of dhg^T = W2^T da^T: A[m = k][h] = W2[h][k] = W2T[k][h]
    vec16 w2ta[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) w2ta[t] = *reinterpret_cast<const vec16 *>(p.W2T + (16 * t + r16) * p.ldw2t + 8 * q);

    const int64_t stride = (int64_t)gridDim.x * n_waves;
    int64_t par = (int64_t)blockIdx.x * n_waves + wave;
    auto request = [&](int64_t pr, int64_t pr_ids) {
        AfBwdIn<ROUNDS> in;
        const int64_t child = pr * n + rc;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int c = 64 * r + lane;
            const int cc = c < CH ? c : CH - 1;
            in.gv[r][0] = *reinterpret_cast<const af_f32x4 *>(p.g + pr * p.g_ld + 8 * cc);
            in.gv[r][1] = *reinterpret_cast<const af_f32x4 *>(p.g + pr * p.g_ld + 8 * cc + 4);
        }
        in.wgt = p.ws[child];
        in.xq0 = *reinterpret_cast<const af_f32x4 *>(p.xa + pr * p.xa_ld + 8 * q);
        in.xq1 = *reinterpret_cast<const af_f32x4 *>(p.xa + pr * p.xa_ld + 8 * q + 4);
        in.hlo = *reinterpret_cast<const af_u32x2 *>(p.hid + child * p.hid_ld + 4 * q);
        in.hhi = *reinterpret_cast<const af_u32x2 *>(p.hid + child * p.hid_ld + 16 + 4 * q);
#pragma unroll
        for (int i = 0; i < AF_NMAX / 2; ++i) {
            const int j = 2 * i + half;
            in.nav[i] = p.na[(pr * n + (j < n ? j : n - 1)) * p.na_ld + h32];
        }
        in.id = af_id_request(p.ids, p.table, pr_ids, n, lane);
        return in;
    };
    // the pipeline of the forward kernel: rows of parent i + 1 requested before parent i is computed; everything
    // requested behind a tile's rows is touched at the end of the trip before (requests return in issue order; every
    // store of a trip is executed by every lane, so the compiler can count the younger requests)
    auto touch = [&](AfBwdIn<ROUNDS> &in) {
        asm volatile("" : "+v"(in.gv[0][0]), "+v"(in.gv[0][1]), "+v"(in.xq0), "+v"(in.xq1), "+v"(in.hlo), "+v"(in.hhi),
                          "+v"(in.wgt), "+v"(in.id) : : "memory");
        if constexpr (ROUNDS > 1) asm volatile("" : "+v"(in.gv[ROUNDS - 1][0]), "+v"(in.gv[ROUNDS - 1][1]));
        asm volatile("" : "+v"(in.nav[0]), "+v"(in.nav[1]), "+v"(in.nav[2]), "+v"(in.nav[3]), "+v"(in.nav[4]),
                          "+v"(in.nav[5]), "+v"(in.nav[6]), "+v"(in.nav[7]));
    };
    int buf = 0;
    AfBwdIn<ROUNDS> cur;
    if (par < p.M) {
        uint32_t id0 = af_id_request(p.ids, p.table, par, n, lane);
        asm volatile("" : "+v"(id0));
        af_issue_tile<KS>(p.table, p.ld, af_id_value(id0, p.ids, p.row0, par, n, lane), n, xb0, lane);
        asm volatile("" ::: "memory");
        cur = request(par, par + stride < p.M ? par + stride : p.M - 1);
        touch(cur);
    }
    while (par < p.M) {
        const vec16 *xb = xb0 + buf * (n * CH);
        const int64_t child = par * n + rc;
        const int64_t nxt = par + stride;
        if (nxt < p.M)
            af_issue_tile<KS>(p.table, p.ld, af_id_value(cur.id, p.ids, p.row0, nxt, n, lane), n, xb0 + (buf ^ 1) * (n * CH), lane);
        asm volatile("" ::: "memory");
        AfBwdIn<ROUNDS> nxin = request(nxt < p.M ? nxt : p.M - 1, nxt + stride < p.M ? nxt + stride : p.M - 1);
        asm volatile("" ::: "memory");
        const af_f32x4 (&gv)[ROUNDS][2] = cur.gv;
        const af_f32x4 xq0 = cur.xq0, xq1 = cur.xq1;
        const af_u32x2 hlo = cur.hlo, hhi = cur.hhi;
        const float wgt_raw = cur.wgt;
        const float (&nav)[AF_NMAX / 2] = cur.nav;

        // d agg -> bf16 high and low parts in LDS (g = hi + lo to 2^-17 relative: the dot products below keep
        // fp32-grade factors), as the B operand of every reduction step
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int c = 64 * r + lane;
            if (c < CH) {
                vec16 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float f0 = gv[r][e >> 1][2 * (e & 1)], f1 = gv[r][e >> 1][2 * (e & 1) + 1];
                    hi[e] = af_pack2(f0, f1);
                    lo[e] = af_pack2(f0 - af_lo(hi[e]), f1 - af_hi(hi[e]));
                }
                ghi[c] = hi;
                glo[c] = lo;
            }
        }
        // dws[row] = <row, d agg> on the matrix cores: A = the tile's rows, B = d agg (the same for every MFMA column)
        // (two accumulators: the high and the low parts' products are independent MFMA chains)
        af_f32x4 dacc = {0.f, 0.f, 0.f, 0.f}, dacl = {0.f, 0.f, 0.f, 0.f};
        {
            vec16 fa[2], fh[2], fl[2];
            fa[0] = xb[rc * CH + (q ^ sw)];
            fh[0] = ghi[q];
            fl[0] = glo[q];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + 1 < KS) {
                    fa[(ks + 1) & 1] = xb[rc * CH + ((4 * (ks + 1) + q) ^ sw)];
                    fh[(ks + 1) & 1] = ghi[4 * (ks + 1) + q];
                    fl[(ks + 1) & 1] = glo[4 * (ks + 1) + q];
                }
                dacc = af_mfma(fa[ks & 1], fh[ks & 1], dacc);
                dacl = af_mfma(fa[ks & 1], fl[ks & 1], dacl);
            }
            dacc += dacl;
        }
        // lane (any column, q) holds rows 4 q + reg; through the scratch words to the lane that owns the row
        if (r16 == 0) *reinterpret_cast<af_f32x4 *>(scr + 4 * q) = dacc;
        // softmax backward (rows of the 16-lane group).  The lanes of the rows beyond the fan-out repeat the last child:
        // they stay out of the sums and of d a(parent), and issue the last child's stores again (same address, same bits)
        const float dws = scr[rc];
        const float dot = af_group16_sum(valid ? dws * wgt_raw : 0.f);
        const float ds = wgt_raw * (dws - dot);
        const float dsm = valid ? ds : 0.f;

        // d a of the parent through this hop: dxa[h] = sum_j ds_j a_child[j][h]
        float dx = 0.f;
#pragma unroll
        for (int i = 0; i < AF_NMAX / 2; ++i) {
            const float d0 = af_readlane(dsm, 2 * i), d1 = af_readlane(dsm, 2 * i + 1);
            dx += (half ? d1 : d0) * nav[i];
        }
        dx += __shfl_xor(dx, 32, 64);                           // (both halves of the wave: the same sum, the same store)
        p.dxa[par * p.dxa_ld + h32] = dx;

        // d a of this lane's row, its hidden units 8 q .. 8 q + 7 (no gradient reaches a last-hop row as a parent)
        vec16 dav;
        dav[0] = af_pack2(ds * xq0[0], ds * xq0[1]);
        dav[1] = af_pack2(ds * xq0[2], ds * xq0[3]);
        dav[2] = af_pack2(ds * xq1[0], ds * xq1[1]);
        dav[3] = af_pack2(ds * xq1[2], ds * xq1[3]);
        *reinterpret_cast<vec16 *>(p.da + child * p.da_ld + 8 * q) = dav;
        // dhg^T [k][row] = sum_h W2[h][k] da[row][h]; lane (row, q) gets k = 4 q + reg and 16 + 4 q + reg
        const af_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        const af_f32x4 g0 = af_mfma(w2ta[0], dav, zero);
        const af_f32x4 g1 = af_mfma(w2ta[1], dav, zero);
        float hl[4], hh[4];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            hl[2 * e] = af_lo(hlo[e]); hl[2 * e + 1] = af_hi(hlo[e]);
            hh[2 * e] = af_lo(hhi[e]); hh[2 * e + 1] = af_hi(hhi[e]);
        }
        {
            uint16_t *dr = p.dhid + child * p.dhid_ld;
            *reinterpret_cast<af_u32x2 *>(dr + 4 * q) =
                af_u32x2{af_pack2(g0[0] * (1.f - hl[0] * hl[0]), g0[1] * (1.f - hl[1] * hl[1])),
                         af_pack2(g0[2] * (1.f - hl[2] * hl[2]), g0[3] * (1.f - hl[3] * hl[3]))};
            *reinterpret_cast<af_u32x2 *>(dr + 16 + 4 * q) =
                af_u32x2{af_pack2(g1[0] * (1.f - hh[0] * hh[0]), g1[1] * (1.f - hh[1] * hh[1])),
                         af_pack2(g1[2] * (1.f - hh[2] * hh[2]), g1[3] * (1.f - hh[3] * hh[3]))};
        }
        __builtin_amdgcn_sched_barrier(0);
        touch(nxin);                     // arrived => the next tile has landed (the stores above are not waited for)
        par = nxt; buf ^= 1; cur = nxin;
    }
}

// ---- host side ----------------------------------------------------------------------------------------------
constexpr int AF_LDS_BYTES = 160 * 1024;

static int af_ksteps(int64_t D)
{
    const int64_t ks = ceil_div(D, 32);
    for (int k : {1, 2, 4, 8, 19, 20})
        if (ks <= k) return k;
    return 0;
}

// Waves per workgroup and workgroups per CU: one LDS tile per wave, W0 once per workgroup.  The kernels need 104-128
// registers (four waves per SIMD: at most 16 waves per CU), so the pair (waves, workgroups per CU) with the most
// waves per CU that the 160 KiB hold wins; ties go to fewer, larger workgroups (fewer copies of W0).
static bool af_geometry(int ks, int n, bool with_w0, int64_t M, int *waves, int *grid, size_t *lds)
{
    // per wave two row tiles (double-buffered) + scratch (+ the two halves of d agg in the backward); per workgroup W0
    // in the forward
    const int64_t ch = 4 * ks;
    const int64_t tile = (2 * (int64_t)n * ch + AF_SCRATCH + (with_w0 ? 0 : 2 * ch)) * 16;
    const int64_t fixed = with_w0 ? 32 * ch * 16 : 0;
    int best_nw = 0, best_pc = 0;
    for (int nw = 1; nw <= AF_MAX_WAVES; ++nw)
        for (int pc = 1; pc <= 16; ++pc) {
            if (pc * (fixed + nw * tile) > AF_LDS_BYTES || nw * pc > 16) break;
            // ties: fewer, larger workgroups in the forward (fewer copies of W0); more, smaller ones in the backward
            // (measured at Reddit's last hop: 2 x 3 waves 40.9 us, 1 x 6 waves 43.9 us)
            const bool tie = nw * pc == best_nw * best_pc && (with_w0 ? nw > best_nw : (nw < best_nw && nw >= 3));
            if (nw * pc > best_nw * best_pc || tie) { best_nw = nw; best_pc = pc; }
        }
    if (!best_nw) return false;
    if (const char *e = getenv("GSAGE_AF_WAVES")) {              // diagnostic: waves per workgroup / workgroups per CU
        const int nw = atoi(e), pc = getenv("GSAGE_AF_PER_CU") ? atoi(getenv("GSAGE_AF_PER_CU")) : 1;
        if (nw >= 1 && nw <= AF_MAX_WAVES && pc >= 1 && pc * (fixed + nw * tile) <= AF_LDS_BYTES) { best_nw = nw; best_pc = pc; }
    }
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        else (void)hipGetLastError();
    } else (void)hipGetLastError();
    int64_t g = ceil_div(M, best_nw);
    if (g > (int64_t)cus * best_pc) g = (int64_t)cus * best_pc;
    *waves = best_nw;
    *grid = (int)(g < 1 ? 1 : g);
    *lds = (size_t)(fixed + best_nw * tile);
    return true;
}

// more than the default 64 KiB of dynamic LDS (gfx950 has 160 KiB per CU): raised once per kernel
template <typename K>
static int af_raise_lds(K kernel, bool &done)
{
    if (done) return GSAGE_OK;
    if (hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, AF_LDS_BYTES) != hipSuccess) {
        (void)hipGetLastError();
        set_error("attn_fused: cannot raise the dynamic LDS limit");
        return GSAGE_ELAUNCH;
    }
    done = true;
    return GSAGE_OK;
}

}  // namespace gsage

using namespace gsage;

extern "C" int gsage_attn_fused_ok(int dtype, int64_t ld, int64_t D, int32_t n, int64_t Ha)
{
    if (dtype != GSAGE_BF16 || Ha != 32 || n < 2 || n > AF_NMAX || D < 1 || ld % 8 != 0) return 0;
    const int ks = af_ksteps(D);
    return ks > 0 && 32 * (int64_t)ks <= ld ? 1 : 0;
}

#define GSAGE_AF_DISPATCH(ks, KERNEL, ...)                                       \
    do {                                                                         \
        switch (ks) {                                                            \
        case 1: KERNEL(1, __VA_ARGS__); break;                                   \
        case 2: KERNEL(2, __VA_ARGS__); break;                                   \
        case 4: KERNEL(4, __VA_ARGS__); break;                                   \
        case 8: KERNEL(8, __VA_ARGS__); break;                                   \
        case 19: KERNEL(19, __VA_ARGS__); break;                                 \
        default: KERNEL(20, __VA_ARGS__); break;                                 \
        }                                                                        \
    } while (0)

extern "C" int gsage_attn_fused_fwd(const void *table, int dtype, int64_t ld, const int64_t *ids, int64_t row0,
                                    const void *W0, int64_t ldw0, const void *W2, int64_t ldw2, const float *xa,
                                    int64_t xa_ld, int64_t M, int32_t n, int64_t D, void *hid, int64_t hid_ld, float *a,
                                    int64_t a_ld, float *ws, void *agg_lp, int64_t lp_ld,
                                    void *stream)
{
    GSAGE_REQUIRE(gsage_attn_fused_ok(dtype, ld, D, n, 32), "attn_fused_fwd: shape not covered (bf16 rows of whole 16-byte "
                  "chunks, D <= 640, fan-out 2..16)");
    const int ks = af_ksteps(D);
    const int64_t cols = 32 * (int64_t)ks;
    GSAGE_REQUIRE(M >= 0 && ldw0 >= cols && ldw0 % 8 == 0 && ldw2 >= 32 && ldw2 % 4 == 0 && xa_ld >= 32 && xa_ld % 4 == 0 &&
                  hid_ld >= 32 && hid_ld % 4 == 0 && a_ld >= 32 && a_ld % 4 == 0, "attn_fused_fwd: bad leading dimension");
    GSAGE_REQUIRE(lp_ld >= cols && lp_ld % 4 == 0, "attn_fused_fwd: output rows must hold the columns up to the 32-column step");
    if (M == 0) return GSAGE_OK;
    GSAGE_REQUIRE(table && W0 && W2 && xa && hid && a && ws && agg_lp, "attn_fused_fwd: null pointer");
    GSAGE_REQUIRE((((uintptr_t)table | (uintptr_t)W0 | (uintptr_t)xa | (uintptr_t)a) & 15) == 0 &&
                  (((uintptr_t)W2 | (uintptr_t)hid | (uintptr_t)agg_lp) & 7) == 0, "attn_fused_fwd: misaligned pointer");
    int waves, grid;
    size_t lds;
    GSAGE_REQUIRE(af_geometry(ks, n, true, M, &waves, &grid, &lds), "attn_fused_fwd: a row tile does not fit the LDS");
    AttnFusedFwd p;
    p.table = (const uint16_t *)table; p.ld = ld; p.ids = ids; p.row0 = row0; p.W0 = (const uint16_t *)W0; p.ldw0 = ldw0;
    p.W2 = (const uint16_t *)W2; p.ldw2 = ldw2; p.xa = xa; p.xa_ld = xa_ld; p.M = M; p.n = n; p.D = (int32_t)D;
    p.hid = (uint16_t *)hid; p.hid_ld = hid_ld; p.a = a; p.a_ld = a_ld; p.ws = ws;
    p.agg_lp = (uint16_t *)agg_lp; p.lp_ld = lp_ld;
#define GSAGE_AF_FWD(KSV, P)                                                                                              \
    do {                                                                                                                  \
        static bool raised = false;                                                                                       \
        rc = af_raise_lds(k_attn_fused_fwd<KSV>, raised);                                                                 \
        if (rc == GSAGE_OK) launch(k_attn_fused_fwd<KSV>, dim3(grid), dim3(waves * 64), lds, (hipStream_t)stream, P);     \
    } while (0)
    int rc = GSAGE_OK;
    GSAGE_AF_DISPATCH(ks, GSAGE_AF_FWD, p);
#undef GSAGE_AF_FWD
    if (rc != GSAGE_OK) return rc;
    return check_launch("attn_fused_fwd");
}

extern "C" int gsage_attn_fused_bwd(const void *table, int dtype, int64_t ld, const int64_t *ids, int64_t row0,
                                    const void *W2T, int64_t ldw2t, const float *g, int64_t g_ld, const float *ws,
                                    const float *na, int64_t na_ld, const float *xa, int64_t xa_ld, const void *hid,
                                    int64_t hid_ld, int64_t M, int32_t n, int64_t D, void *da, int64_t da_ld, void *dhid,
                                    int64_t dhid_ld, float *dxa, int64_t dxa_ld, void *stream)
{
    GSAGE_REQUIRE(gsage_attn_fused_ok(dtype, ld, D, n, 32), "attn_fused_bwd: shape not covered (bf16 rows of whole 16-byte "
                  "chunks, D <= 640, fan-out 2..16)");
    const int ks = af_ksteps(D);
    const int64_t cols = 32 * (int64_t)ks;
    GSAGE_REQUIRE(M >= 0 && ldw2t >= 32 && ldw2t % 8 == 0 && g_ld >= cols && g_ld % 4 == 0 && na_ld >= 32 && xa_ld >= 32 &&
                  xa_ld % 4 == 0 && hid_ld >= 32 && hid_ld % 4 == 0 && da_ld >= 32 && da_ld % 8 == 0 && dhid_ld >= 32 &&
                  dhid_ld % 4 == 0 && dxa_ld >= 32, "attn_fused_bwd: bad leading dimension");
    if (M == 0) return GSAGE_OK;
    GSAGE_REQUIRE(table && W2T && g && ws && na && xa && hid && da && dhid && dxa, "attn_fused_bwd: null pointer");
    GSAGE_REQUIRE((((uintptr_t)table | (uintptr_t)W2T | (uintptr_t)g | (uintptr_t)xa | (uintptr_t)da) & 15) == 0 &&
                  (((uintptr_t)hid | (uintptr_t)dhid) & 7) == 0, "attn_fused_bwd: misaligned pointer");
    int waves, grid;
    size_t lds;
    GSAGE_REQUIRE(af_geometry(ks, n, false, M, &waves, &grid, &lds), "attn_fused_bwd: a row tile does not fit the LDS");
    AttnFusedBwd p;
    p.table = (const uint16_t *)table; p.ld = ld; p.ids = ids; p.row0 = row0; p.W2T = (const uint16_t *)W2T; p.ldw2t = ldw2t;
    p.g = g; p.g_ld = g_ld; p.ws = ws; p.na = na; p.na_ld = na_ld; p.xa = xa; p.xa_ld = xa_ld; p.hid = (const uint16_t *)hid;
    p.hid_ld = hid_ld; p.M = M; p.n = n; p.D = (int32_t)D; p.da = (uint16_t *)da; p.da_ld = da_ld; p.dhid = (uint16_t *)dhid;
    p.dhid_ld = dhid_ld; p.dxa = dxa; p.dxa_ld = dxa_ld;
#define GSAGE_AF_BWD(KSV, P)                                                                                              \
    do {                                                                                                                  \
        static bool raised = false;                                                                                       \
        rc = af_raise_lds(k_attn_fused_bwd<KSV>, raised);                                                                 \
        if (rc == GSAGE_OK) launch(k_attn_fused_bwd<KSV>, dim3(grid), dim3(waves * 64), lds, (hipStream_t)stream, P);     \
    } while (0)
    int rc = GSAGE_OK;
    GSAGE_AF_DISPATCH(ks, GSAGE_AF_BWD, p);
#undef GSAGE_AF_BWD
    if (rc != GSAGE_OK) return rc;
    return check_launch("attn_fused_bwd");
}
