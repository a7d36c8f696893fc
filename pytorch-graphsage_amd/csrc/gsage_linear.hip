// gsage_linear.hip -- K5 projection GEMM and K3 pooling MLP on the gfx950 matrix cores.
//
// Replaces every nn.Linear on the hot path: fc_x / fc_neib + cat + activation of the
// aggregators (reference nn_modules.py:200-202, :228-230, :317-319), the pooling MLP
// (:224-226 + the max/mean over the fanout at :240 / :252) and the attention MLP (:307-308).
//
//   C[m, j] = act( sum_k A[m, k] * W[j, k] + bias[j] )          "NT": both operands K-contiguous
//
// which is exactly the operand shape v_mfma wants: lane l of a wave holds, for row (l & 31), the
// 16 contiguous bytes of K-chunk (2*kk + (l >> 5)) -- 8 bf16 for v_mfma_f32_32x32x16_bf16, or
// 4 fp32 fed through four v_mfma_f32_32x32x2_f32 (exact fp32, for the tight-parity mode).
//
// Tiling (wave64, 4 waves / workgroup, MFMA-bound for K3, L2/LDS-bound for the skinny K5):
//   block tile 64(M) x 128(N) x 128 bytes of K; wave w owns rows 32*(w&1).. and columns
//   64*(w>>1)..: two 32x32 accumulators (32 VGPRs).  A and W tiles are staged global -> regs
//   -> LDS in full 128-byte lines (8 lanes per row), next tile's loads issued before the
//   current tile's MFMAs (register prefetch).  LDS image is [row][8 x 16 B] with
//       slot = chunk ^ (row & 7),  row' = row with bits 0 and 3 swapped
//   so the four 16-lane groups of a ds_read_b128 (microarch guide, LDS table) each touch 16
//   distinct 16-byte slots of the 256-byte bank row: conflict-free operand reads.
//   Grid: x = M tiles (>= 208 workgroups at the Reddit layer-0 shape so all 256 CUs get
//   work), y = N tiles, z = group (x|agg halves of the concat in one launch).
//   The A tile can be row-gathered (a_rows) so feats[ids] never exists in HBM.
//
// K3 reuses the same main loop; its epilogue keeps the bias+ReLU'd 64x128 tile in LDS and
// reduces max / mean over each group of `n` consecutive rows, so the [M*n, 512] hidden
// activations of the reference never reach HBM.
#include "gsage_common.h"
#include "gsage_mma_dev.h"

namespace gsage {

struct LinearParams {
    const void *A;
    const void *W;
    const float *bias;
    const int64_t *a_rows;
    void *C;
    int64_t lda, ldw, ldc;
    int64_t M, N, K;
    int64_t a_gstride, w_gstride, c_gstride;
    int32_t a_rows_group0_only;
    int32_t act;
    int32_t c_dtype;
    // pooling epilogue (K3)
    int32_t pool_n;        // rows per segment (0 = plain linear)
    int32_t pool_groups;   // segments per workgroup
    int32_t pool_mode;
    float *pooled;
    int64_t pooled_ld;
    int32_t *argmax;
    uint32_t *relu_mask;   // optional [M, N/32] sign bits of the bias+ReLU'd tile (mean-pool backward)
};

// Epilogue shared by both K5 kernels: bias + activation, then the 64x128 tile goes through LDS
// so that global memory sees full 16-byte row chunks (256 B contiguous per 16 lanes) instead of the
// MFMA layout's 2-byte column-per-lane scatter -- the scatter alone cost ~5.5 us per launch at the
// layer-0 shape (measured by ablating the stores).  Falls back to per-element
// stores when the output rows are not 16-byte chunk aligned or the tile is ragged in N.
// C/D layout of the 32x32 MFMA: lane l, register r -> column (l & 31),
// row (r & 3) + 8 * (r >> 2) + 4 * (l >> 5).
template <int ACT>
__device__ __forceinline__ void store_tile(const LinearParams &p, const f32x16_t &acc0,
                                           const f32x16_t &acc1, int g, int64_t m0, int64_t n0,
                                           int wm, int wn, int lane, int tid, void *lds_raw)
{
    const float *bias = p.bias ? p.bias + (int64_t)g * p.N : nullptr;
    const int esz = p.c_dtype == GSAGE_BF16 ? 2 : 4;
    const int epc = 16 / esz;                                    // output elements per 16-byte chunk
    const int64_t cbase = (int64_t)g * p.c_gstride + n0;         // first output column of the tile
    // columns of this tile that exist: whole 16-byte chunks of them leave through LDS as 16-byte lane stores
    // (also the narrow outputs of the att MLP / embedding prep: N = 32, 64)
    const int ncols = (int)(p.N - n0 < BN ? p.N - n0 : BN);
    const bool wide = ncols % epc == 0 && p.ldc % epc == 0 && cbase % epc == 0 && ((uintptr_t)p.C % 16) == 0;
    if (wide) {
        __syncthreads();                                         // operand buffers are free now
        const int ldt = BN + epc;                                // padded row, still 16-byte aligned
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x16_t &acc = t ? acc1 : acc0;
            const int jl = wn * 64 + t * 32 + (lane & 31);
            if (jl >= ncols) continue;
            const float bj = bias ? bias[n0 + jl] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float v = apply_act(acc[r] + bj, ACT);
                if (p.c_dtype == GSAGE_BF16)
                    ((uint16_t *)lds_raw)[i * ldt + jl] = f32_to_bf16(v);
                else
                    ((float *)lds_raw)[i * ldt + jl] = v;
            }
        }
        __syncthreads();
        const int cpr = ncols / epc;                             // chunks per tile row
        for (int q = tid; q < BM * cpr; q += 256) {
            const int row = q / cpr, ch = q - row * cpr;
            const int64_t m = m0 + row;
            if (m < p.M) {
                const vec16 v = *reinterpret_cast<const vec16 *>((const char *)lds_raw +
                                                                 ((size_t)row * ldt + ch * epc) * esz);
                *reinterpret_cast<vec16 *>((char *)p.C + ((size_t)m * p.ldc + cbase + ch * epc) * esz) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const f32x16_t &acc = t ? acc1 : acc0;
        const int64_t j = n0 + wn * 64 + t * 32 + (lane & 31);
        const float bj = (bias && j < p.N) ? bias[j] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int64_t m = m0 + wm * 32 + i;
            if (m < p.M && j < p.N) {
                const float v = apply_act(acc[r] + bj, ACT);
                const int64_t off = m * p.ldc + (int64_t)g * p.c_gstride + j;
                if (p.c_dtype == GSAGE_BF16)
                    ((uint16_t *)p.C)[off] = f32_to_bf16(v);
                else
                    ((float *)p.C)[off] = v;
            }
        }
    }
}

template <typename T, bool POOL, int ACT>
__global__ void __launch_bounds__(256)
k_linear_nt(const LinearParams p)
{
    constexpr int EPC = 16 / (int)sizeof(T);        // elements per 16-byte chunk
    // one raw LDS array (keeps the compiler from serialising waits across objects)
    __shared__ vec16 smem[POOL ? (BM * BN * 4 / 16) : (BM * (BN + 4) * 4 / 16)];   // >= operand tiles (24 KiB) and the fp32 output staging tile
    vec16 *sA = smem;
    vec16 *sW = smem + BM * CH;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int g = blockIdx.z;
    const int rows_per_wg = POOL ? p.pool_groups * p.pool_n : BM;
    const int64_t m0 = (int64_t)blockIdx.x * rows_per_wg;
    const int64_t n0 = (int64_t)blockIdx.y * BN;

    const T *A = (const T *)p.A + (int64_t)g * p.a_gstride;
    const T *W = (const T *)p.W + (int64_t)g * p.w_gstride;
    const int64_t *a_rows = (p.a_rows && (g == 0 || !p.a_rows_group0_only)) ? p.a_rows : nullptr;

    // ---- staging assignment: 8 consecutive lanes fetch one full 128-byte line ----------------
    // Out-of-range rows / K-chunks read a valid (clamped) address and are zeroed afterwards, so
    // every load is unconditional and the staging registers stay in VGPRs (no scratch).
    const int srow = tid >> 3;           // 0..31
    const int sch = tid & 7;             // chunk inside the tile row
    const int kchunks = (int)((p.K + EPC - 1) / EPC);
    const int nk = (kchunks + CH - 1) / CH;

    const T *a_ptr0, *a_ptr1, *w_ptr0, *w_ptr1, *w_ptr2, *w_ptr3;
    bool a_ok0, a_ok1, w_ok0, w_ok1, w_ok2, w_ok3;
    {
        auto a_row = [&](int row, bool &ok) -> const T * {
            const int64_t m = m0 + row;
            ok = row < rows_per_wg && m < p.M;
            const int64_t mm = ok ? m : 0;
            const int64_t r = a_rows ? a_rows[mm] : mm;
            return A + r * p.lda;
        };
        auto w_row = [&](int row, bool &ok) -> const T * {
            const int64_t j = n0 + row;
            ok = j < p.N;
            return W + (ok ? j : 0) * p.ldw;
        };
        a_ptr0 = a_row(srow, a_ok0);
        a_ptr1 = a_row(srow + 32, a_ok1);
        w_ptr0 = w_row(srow, w_ok0);
        w_ptr1 = w_row(srow + 32, w_ok1);
        w_ptr2 = w_row(srow + 64, w_ok2);
        w_ptr3 = w_row(srow + 96, w_ok3);
    }
    const vec16 zero = {0u, 0u, 0u, 0u};
    vec16 ra0, ra1, rw0, rw1, rw2, rw3;
    bool kin_held = false;               // does the tile held in registers lie inside K?

#define GSAGE_LOAD_TILE(kt)                                                                   \
    do {                                                                                      \
        const int kc_ = (kt) * CH + sch;                                                      \
        const bool kin_ = kc_ < kchunks;                                                      \
        const int64_t ko_ = (int64_t)(kin_ ? kc_ : 0) * EPC;                                  \
        ra0 = *reinterpret_cast<const vec16 *>(a_ptr0 + ko_);                                 \
        ra1 = *reinterpret_cast<const vec16 *>(a_ptr1 + ko_);                                 \
        rw0 = *reinterpret_cast<const vec16 *>(w_ptr0 + ko_);                                 \
        rw1 = *reinterpret_cast<const vec16 *>(w_ptr1 + ko_);                                 \
        rw2 = *reinterpret_cast<const vec16 *>(w_ptr2 + ko_);                                 \
        rw3 = *reinterpret_cast<const vec16 *>(w_ptr3 + ko_);                                 \
        kin_held = kin_;                                                                      \
    } while (0)

    const int wm = wave & 1;
    const int wn = wave >> 1;
    f32x16_t acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

    // LDS slots this thread writes (staging) and reads (fragments) are loop invariant
    const int wsa0 = lds_slot(srow, sch), wsa1 = lds_slot(srow + 32, sch);
    const int wsw0 = lds_slot(srow, sch), wsw1 = lds_slot(srow + 32, sch);
    const int wsw2 = lds_slot(srow + 64, sch), wsw3 = lds_slot(srow + 96, sch);
    const int arow = wm * 32 + (lane & 31);
    const int wrow0 = wn * 64 + (lane & 31), wrow1 = wrow0 + 32;

    GSAGE_LOAD_TILE(0);
    for (int kt = 0; kt < nk; ++kt) {
        // masking happens here, at the consumer, so the loads above stay in flight during the
        // previous tile's MFMAs instead of being waited for right after issue
        sA[wsa0] = (kin_held && a_ok0) ? ra0 : zero;
        sA[wsa1] = (kin_held && a_ok1) ? ra1 : zero;
        sW[wsw0] = (kin_held && w_ok0) ? rw0 : zero;
        sW[wsw1] = (kin_held && w_ok1) ? rw1 : zero;
        sW[wsw2] = (kin_held && w_ok2) ? rw2 : zero;
        sW[wsw3] = (kin_held && w_ok3) ? rw3 : zero;
        __syncthreads();
        if (kt + 1 < nk) GSAGE_LOAD_TILE(kt + 1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int ch = kk * 2 + (lane >> 5);
            const vec16 a = sA[lds_slot(arow, ch)];
            const vec16 b0 = sW[lds_slot(wrow0, ch)];
            const vec16 b1 = sW[lds_slot(wrow1, ch)];
            mma_chunk<T>::run(a, b0, acc0);
            mma_chunk<T>::run(a, b1, acc1);
        }
        __syncthreads();
    }
#undef GSAGE_LOAD_TILE

    // ---- epilogue ---------------------------------------------------------------------------------
    // C/D layout of the 32x32 MFMA: lane l, register r -> column (l & 31),
    // row (r & 3) + 8 * (r >> 2) + 4 * (l >> 5).
    if (!POOL) {
        store_tile<ACT>(p, acc0, acc1, g, m0, n0, wm, wn, lane, tid, smem);
    } else {
        // bias + ReLU'd tile -> LDS [64][128] fp32, then segment max / mean down the rows
        float *tile = reinterpret_cast<float *>(smem);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x16_t &acc = t ? acc1 : acc0;
            const int jl = wn * 64 + t * 32 + (lane & 31);
            const int64_t j = n0 + jl;
            const float bj = (p.bias && j < p.N) ? p.bias[j] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float v = acc[r] + bj;
                tile[i * BN + jl] = v > 0.f ? v : 0.f;
            }
        }
        __syncthreads();
        if (p.relu_mask) {
            // sign bits of the hidden activations: thread t packs 32 channels of tile row t >> 2
            const int i = tid >> 2, q = tid & 3;
            const int64_t row = (int64_t)blockIdx.x * rows_per_wg + i;
            if (i < rows_per_wg && row < p.M && n0 + q * 32 < p.N) {
                uint32_t bits = 0;
#pragma unroll
                for (int e0 = 0; e0 < 32; ++e0) {
                    const int e = (e0 + i) & 31;               // rotate per row: spreads the LDS banks
                    bits |= (tile[i * BN + q * 32 + e] > 0.f ? 1u : 0u) << e;
                }
                p.relu_mask[row * (p.N / 32) + (n0 >> 5) + q] = bits;
            }
        }
        const int jl = tid & (BN - 1);
        const int64_t j = n0 + jl;
        for (int sg = tid >> 7; sg < p.pool_groups; sg += 2) {
            const int64_t seg = (int64_t)blockIdx.x * p.pool_groups + sg;      // output row
            if (seg * p.pool_n >= p.M || j >= p.N) continue;
            const float *colp = tile + (sg * p.pool_n) * BN + jl;
            float best = colp[0];
            int arg = 0;
            float sum = best;
            for (int r = 1; r < p.pool_n; ++r) {
                const float v = colp[r * BN];
                sum += v;
                if (v > best) { best = v; arg = r; }
            }
            if (p.pool_mode == GSAGE_POOL_MAX) {
                p.pooled[seg * p.pooled_ld + j] = best;
                if (p.C) ((uint16_t *)p.C)[seg * p.ldc + j] = f32_to_bf16(best);   // operand copy for K5 / K5b
                if (p.argmax) p.argmax[seg * p.N + j] = arg;
            } else {
                p.pooled[seg * p.pooled_ld + j] = sum / (float)p.pool_n;
                if (p.C) ((uint16_t *)p.C)[seg * p.ldc + j] = f32_to_bf16(sum / (float)p.pool_n);
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------
// K5, LDS-DMA pipelined variant (the fast path whenever operand rows are whole 128-byte lines).
//
// Same tile, same LDS image and same MFMA loop as k_linear_nt, but the tiles travel global -> LDS
// with `global_load_lds_dwordx4` (no staging VGPRs, no ds_write pass) through a 3-buffer ring:
//     wait vmcnt(6) -> s_barrier -> issue tile kt+3 -> read fragments of tile kt+1 -> MFMAs on tile kt
// i.e. ONE barrier per K-tile and two tiles (48 KiB per workgroup) in flight across it.  hipcc
// cannot express "wait for the older of two in-flight tiles" across a barrier by itself (it drains
// to vmcnt(0)), hence the raw s_barrier + explicit counted waits (CDNA guide, "Pipelining across
// barriers").  The DMA writes LDS linearly (wave-uniform base + lane*16), so the conflict-free
// swizzle is applied on the SOURCE side: the lane that fills LDS slot (R, c') fetches global
// (row = swap03(R), chunk = c' ^ (row & 7)) -- still one full 128-byte line per 8 lanes.
// Out-of-range rows are clamped (their outputs are never stored); the K tail relies on the
// operands' zero padding, which is why rows must be whole lines (lda, ldw % (8*EPC) == 0).
// -------------------------------------------------------------------------------------------------

// NBUF: operand buffers.  3 = the ring described above (72 KiB: two workgroups per CU).  1 / 2 = reductions of one / two
// k-tiles (K <= 64 / 128 bf16: the 64-d embedding prep, the 32-wide att MLP, 128-wide hidden layers over a 164 k-row
// frontier): little or nothing to pipeline inside a workgroup, so the overlap has to come from more workgroups per
// CU -- 34 KiB (one operand buffer, or the output staging tile) lets four of them share a CU, 48 KiB three.
template <typename T, int ACT, int NBUF>
__global__ void __launch_bounds__(256)
k_linear_nt_dma(const LinearParams p)
{
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int TILE = (BM + BN) * CH;                 // vec16 slots per buffer (24 KiB)
    constexpr int STAGE = (BM * (BN + 4) * 4) / 16;      // output staging tile of store_tile
    __shared__ vec16 smem[NBUF * TILE > STAGE ? NBUF * TILE : STAGE];   // single LDS object: 72 KiB (NBUF = 3)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int g = blockIdx.z;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int64_t n0 = (int64_t)blockIdx.y * BN;
    const T *A = (const T *)p.A + (int64_t)g * p.a_gstride;
    const T *W = (const T *)p.W + (int64_t)g * p.w_gstride;
    const int64_t *a_rows = (p.a_rows && (g == 0 || !p.a_rows_group0_only)) ? p.a_rows : nullptr;

    const int kchunks = (int)((p.K + EPC - 1) / EPC);
    const int nk = (kchunks + CH - 1) / CH;

    // ---- DMA assignment: wave w, instruction s fills LDS rows R = 32*s + 8*w + (lane >> 3),
    //      slot c' = lane & 7 (1 KiB contiguous per instruction) from the swizzled source
    const int cdst = lane & 7;
    const T *a_src[2];
    const T *w_src[4];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const int R = 32 * s2 + 8 * wave + (lane >> 3);
        const int row = (R & ~9) | ((R & 1) << 3) | ((R >> 3) & 1);
        int64_t m = m0 + row;
        if (m >= p.M) m = p.M - 1;
        const int64_t r = a_rows ? a_rows[m] : m;
        a_src[s2] = A + r * p.lda + (cdst ^ (row & 7)) * EPC;
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const int R = 32 * s4 + 8 * wave + (lane >> 3);
        const int row = (R & ~9) | ((R & 1) << 3) | ((R >> 3) & 1);
        int64_t j = n0 + row;
        if (j >= p.N) j = p.N - 1;
        w_src[s4] = W + j * p.ldw + (cdst ^ (row & 7)) * EPC;
    }
    // wave-uniform LDS destinations (slot index of the instruction's first lane)
    const int a_dst0 = (8 * wave) * CH, a_dst1 = (32 + 8 * wave) * CH;
    const int w_dst0 = BM * CH + (8 * wave) * CH;

    // <= 64 output columns: the upper half of the W tile would be 64 copies of the last row -- not fetched (a tile is
    // then 4 DMA instructions per wave instead of 6: the counted waits below follow), and the two waves that own those
    // columns sit the MFMAs out
    const bool half_w = p.N - n0 <= 64;
    auto issue_tile = [&](int kt, int buf) {
        vec16 *base = smem + buf * TILE;
        const int64_t ko = (int64_t)kt * CH * EPC;
        __builtin_amdgcn_global_load_lds((global_void_t *)(a_src[0] + ko), (lds_void_t *)(base + a_dst0), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((global_void_t *)(a_src[1] + ko), (lds_void_t *)(base + a_dst1), 16, 0, 0);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
            if (s4 < 2 || !half_w)
                __builtin_amdgcn_global_load_lds((global_void_t *)(w_src[s4] + ko),
                                                 (lds_void_t *)(base + w_dst0 + 32 * s4 * CH), 16, 0, 0);
    };

    const int wm = wave & 1;
    const int wn = wave >> 1;
    f32x16_t acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const int arow = wm * 32 + (lane & 31);
    const int wrow0 = wn * 64 + (lane & 31), wrow1 = wrow0 + 32;

    // Fragment reads run ONE tile ahead of the MFMAs (two static register sets, loop unrolled x 2):
    // with the reads and the MFMAs of the same tile back to back, LDS bandwidth (96 KiB per k-tile
    // and CU) and the MFMA pipe took turns instead of overlapping.  Step kt:
    //     wait: my DMAs of tile kt+1 landed, my fragment reads of tile kt completed
    //     s_barrier                      -> everybody's did; tile kt's buffer is free
    //     issue tile kt+3 into it        -> still two tiles in flight across the barrier
    //     read fragments of tile kt+1    (no wait)
    //     MFMAs of tile kt               (fragments read during step kt-1)
    auto read_frags = [&](int b, vec16 (&fa)[4], vec16 (&fb0)[4], vec16 (&fb1)[4]) {
        const vec16 *sA = smem + b * TILE;
        const vec16 *sW = sA + BM * CH;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int ch = kk * 2 + (lane >> 5);
            fa[kk] = sA[lds_slot(arow, ch)];
            fb0[kk] = sW[lds_slot(wrow0, ch)];
            fb1[kk] = sW[lds_slot(wrow1, ch)];
        }
    };
    auto mma_tile = [&](const vec16 (&fa)[4], const vec16 (&fb0)[4], const vec16 (&fb1)[4]) {
        if (half_w && wn == 1) return;                   // (wave-uniform)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            mma_chunk<T>::run(fa[kk], fb0[kk], acc0);
            mma_chunk<T>::run(fa[kk], fb1[kk], acc1);
        }
    };
    vec16 xa[4], xb0[4], xb1[4], ya[4], yb0[4], yb1[4];
    auto step = [&](int kt, int b, vec16 (&ca)[4], vec16 (&cb0)[4], vec16 (&cb1)[4], vec16 (&na)[4],
                    vec16 (&nb0)[4], vec16 (&nb1)[4]) {
        // my fragment reads of tile kt are complete (the builtin, not inline asm, and on every path:
        // the compiler's own wait insertion then knows the current set is ready and does not drain
        // the reads issued below before the MFMAs)
        __builtin_amdgcn_s_waitcnt(0xC07F);                       // lgkmcnt(0)
        if (kt + 1 < nk) {
            if (kt + 2 < nk) {                                    // one later tile may stay in flight
                if (half_w) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + 3 < nk) issue_tile(kt + 3, b);               // tile kt's own buffer
            read_frags(b == 2 ? 0 : b + 1, na, nb0, nb1);
        }
        mma_tile(ca, cb0, cb1);
    };

    issue_tile(0, 0);
    if (nk > 1) issue_tile(1, 1);
    if (nk > 2) issue_tile(2, 2);
    if (nk > 2) {
        if (half_w) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    } else if (nk > 1) {
        if (half_w) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_frags(0, xa, xb0, xb1);
    int buf = 0;
    for (int kt = 0; kt < nk; kt += 2) {
        step(kt, buf, xa, xb0, xb1, ya, yb0, yb1);
        buf = buf == 2 ? 0 : buf + 1;
        if (kt + 1 < nk) {
            step(kt + 1, buf, ya, yb0, yb1, xa, xb0, xb1);
            buf = buf == 2 ? 0 : buf + 1;
        }
    }

    store_tile<ACT>(p, acc0, acc1, g, m0, n0, wm, wn, lane, tid, smem);
}

static int check_operands(const char *who, const void *A, int dtype, int64_t lda, const void *W,
                          int64_t ldw, int64_t M, int64_t N, int64_t K)
{
    const int64_t epc = dtype == GSAGE_BF16 ? 8 : 4;
    GSAGE_REQUIRE(dtype == GSAGE_BF16 || dtype == GSAGE_F32, "%s: bad dtype %d", who, dtype);
    GSAGE_REQUIRE(M >= 0 && N > 0 && K > 0, "%s: bad sizes", who);
    GSAGE_REQUIRE(A && W, "%s: null pointer", who);
    GSAGE_REQUIRE(lda % epc == 0 && ldw % epc == 0, "%s: lda/ldw must be multiples of %lld", who,
                  (long long)epc);
    GSAGE_REQUIRE(ceil_div(K, epc) * epc <= lda && ceil_div(K, epc) * epc <= ldw,
                  "%s: K rounded up to a 16-byte chunk must fit in lda and ldw", who);
    GSAGE_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0,
                  "%s: A and W must be 16-byte aligned", who);
    return GSAGE_OK;
}

}  // namespace gsage

using namespace gsage;

extern "C" {

int gsage_linear_nt(const void *A, int dtype, int64_t lda, const int64_t *a_rows,
                    int a_rows_group0_only, const void *W, int64_t ldw, const float *bias, void *C,
                    int c_dtype, int64_t ldc, int64_t M, int64_t N, int64_t K, int act, int groups,
                    int64_t a_gstride, int64_t w_gstride, int64_t c_gstride, void *stream)
{
    int rc = check_operands("linear_nt", A, dtype, lda, W, ldw, M, N, K);
    if (rc != GSAGE_OK) return rc;
    GSAGE_REQUIRE(C, "linear_nt: null output");
    GSAGE_REQUIRE(c_dtype == GSAGE_BF16 || c_dtype == GSAGE_F32, "linear_nt: bad c_dtype");
    GSAGE_REQUIRE(groups >= 1 && groups <= 65535, "linear_nt: bad group count");
    GSAGE_REQUIRE(act >= ACT_NONE && act <= ACT_TANH, "linear_nt: bad activation code");
    if (M == 0) return GSAGE_OK;
    LinearParams p;
    p.A = A; p.W = W; p.bias = bias; p.a_rows = a_rows; p.C = C;
    p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
    p.a_gstride = a_gstride; p.w_gstride = w_gstride; p.c_gstride = c_gstride;
    p.a_rows_group0_only = a_rows_group0_only; p.act = act; p.c_dtype = c_dtype;
    p.pool_n = 0; p.pool_groups = 0; p.pool_mode = 0; p.pooled = nullptr; p.pooled_ld = 0;
    p.argmax = nullptr; p.relu_mask = nullptr;
    dim3 grid((unsigned)ceil_div(M, BM), (unsigned)ceil_div(N, BN), (unsigned)groups);
    hipStream_t s = (hipStream_t)stream;
    // whole-line operand rows -> LDS-DMA pipelined kernel
    const int64_t epc = dtype == GSAGE_BF16 ? 8 : 4;
    const int64_t kpad = ceil_div(K, 8 * epc) * 8 * epc;
    const bool dma = lda % (8 * epc) == 0 && ldw % (8 * epc) == 0 && kpad <= lda && kpad <= ldw;
#define GSAGE_LAUNCH_DMA(T, NB)                                                                   \
    do {                                                                                          \
        if (act == ACT_RELU)                                                                      \
            launch(k_linear_nt_dma<T, ACT_RELU, NB>, grid, dim3(256), 0, s, p);     \
        else if (act == ACT_TANH)                                                                 \
            launch(k_linear_nt_dma<T, ACT_TANH, NB>, grid, dim3(256), 0, s, p);     \
        else                                                                                      \
            launch(k_linear_nt_dma<T, ACT_NONE, NB>, grid, dim3(256), 0, s, p);     \
    } while (0)
    if (dma) {
        const int64_t n_tiles = kpad / (8 * epc);         // one or two k-tiles: see NBUF
        if (dtype == GSAGE_BF16) {
            if (n_tiles <= 1) GSAGE_LAUNCH_DMA(uint16_t, 1);
            else if (n_tiles == 2) GSAGE_LAUNCH_DMA(uint16_t, 2);
            else GSAGE_LAUNCH_DMA(uint16_t, 3);
        } else {
            if (n_tiles <= 1) GSAGE_LAUNCH_DMA(float, 1);
            else if (n_tiles == 2) GSAGE_LAUNCH_DMA(float, 2);
            else GSAGE_LAUNCH_DMA(float, 3);
        }
        return check_launch("linear_nt_dma");
    }
#undef GSAGE_LAUNCH_DMA
#define GSAGE_LAUNCH_LINEAR(T)                                                                    \
    do {                                                                                          \
        if (act == ACT_RELU)                                                                      \
            launch(k_linear_nt<T, false, ACT_RELU>, grid, dim3(256), 0, s, p);      \
        else if (act == ACT_TANH)                                                                 \
            launch(k_linear_nt<T, false, ACT_TANH>, grid, dim3(256), 0, s, p);      \
        else                                                                                      \
            launch(k_linear_nt<T, false, ACT_NONE>, grid, dim3(256), 0, s, p);      \
    } while (0)
    if (dtype == GSAGE_BF16)
        GSAGE_LAUNCH_LINEAR(uint16_t);
    else
        GSAGE_LAUNCH_LINEAR(float);
#undef GSAGE_LAUNCH_LINEAR
    return check_launch("linear_nt");
}

int gsage_pool_mlp(const void *A, int dtype, int64_t lda, const int64_t *a_rows, const void *W,
                   int64_t ldw, const float *bias, int64_t M, int32_t n, int64_t H, int64_t K,
                   int pool, float *pooled, int64_t pooled_ld, int32_t *argmax, void *pooled_bf16,
                   int64_t pooled_bf16_ld, uint32_t *relu_mask, void *stream)
{
    GSAGE_REQUIRE(!relu_mask || H % 32 == 0, "pool_mlp: relu_mask needs H % 32 == 0");
    GSAGE_REQUIRE(n >= 1 && n <= BM, "pool_mlp: fanout must be in [1, %d]", BM);
    GSAGE_REQUIRE(!pooled_bf16 || pooled_bf16_ld >= H, "pool_mlp: bad bf16 output");
    int rc = check_operands("pool_mlp", A, dtype, lda, W, ldw, M * (int64_t)n, H, K);
    if (rc != GSAGE_OK) return rc;
    GSAGE_REQUIRE(pooled && pooled_ld >= H, "pool_mlp: bad output");
    GSAGE_REQUIRE(pool == GSAGE_POOL_MAX || pool == GSAGE_POOL_MEAN, "pool_mlp: bad pool mode");
    if (M == 0) return GSAGE_OK;
    LinearParams p;
    p.A = A; p.W = W; p.bias = bias; p.a_rows = a_rows; p.C = pooled_bf16;
    p.lda = lda; p.ldw = ldw; p.ldc = pooled_bf16_ld; p.M = M * (int64_t)n; p.N = H; p.K = K;
    p.a_gstride = 0; p.w_gstride = 0; p.c_gstride = 0;
    p.a_rows_group0_only = 0; p.act = ACT_RELU; p.c_dtype = GSAGE_F32;
    p.pool_n = n; p.pool_groups = BM / n; p.pool_mode = pool; p.pooled = pooled;
    p.pooled_ld = pooled_ld; p.argmax = argmax; p.relu_mask = relu_mask;
    dim3 grid((unsigned)ceil_div(M, p.pool_groups), (unsigned)ceil_div(H, BN), 1);
    if (dtype == GSAGE_BF16)
        launch(k_linear_nt<uint16_t, true, ACT_RELU>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else
        launch(k_linear_nt<float, true, ACT_RELU>, grid, dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("pool_mlp");
}

}  // extern "C"
