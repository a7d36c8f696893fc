// gsage_packed.hip -- K5 / K3 on a weight operand kept in MFMA fragment order.
//
// Weights change once per optimizer step, so the optimizer (gsage_prep_desc.dst_p) or
// gsage_pack_weight writes them in the order the matrix cores consume them; a wave's B fragment is
// then one coalesced 1 KiB load from L2 straight into registers and only the A operand goes
// through LDS.  Same arithmetic, same results as gsage_linear_nt / gsage_pool_mlp
// (reference nn_modules.py:200-202, :224-226).
#include "gsage_common.h"
#include "gsage_gather_dev.h"
#include "gsage_mma_dev.h"
#include "gsage_sample_dev.h"

namespace gsage {

// -------------------------------------------------------------------------------------------------
// K5 with a PACKED weight operand (bf16): W travels L2 -> registers directly, only A goes through LDS.
//
// The LDS-DMA kernel above is bound by bytes in flight: a tile needs ~1.3 us from issue to landing
// (even from L2) and LDS capacity caps what a CU can have outstanding (2 workgroups x 2 tiles x 24 KiB).
// Weights are written once per optimizer step, so the optimizer writes them in MFMA fragment order
//     Wp[g][jb][kc][lane][8] = W_g[jb*32 + (lane & 31)][kc*16 + (lane >> 5)*8 .. +7]      (zero padded)
// and a wave's B fragment of (32 columns, 16 k) is ONE fully coalesced 1 KiB load into registers:
// no LDS traffic for W at all (it was 2/3 of the DMA bytes and 2/3 of the fragment reads), and the
// register file holds the in-flight W tiles, so both rings can be deep (WP_R tiles in flight:
// 8 KiB of A per tile in LDS, 16 VGPRs of W per tile and lane).
// Wave w owns columns [32w, 32w+32) of the 64 x 128 tile and all 64 rows (two 32 x 32 accumulators);
// the four waves read the same A fragments.
// -------------------------------------------------------------------------------------------------
#ifndef GSAGE_WP_R
#define GSAGE_WP_R 4
#endif
constexpr int WP_R = GSAGE_WP_R;              // tiles in flight per workgroup (4, 6 or 8: WP_UNROLL is a multiple)
// s_waitcnt vmcnt(6 * (WP_R - 1)): "this tile landed" = at most the WP_R - 1 later tiles' requests outstanding
//     simm16 = vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 15 << 8 | vmcnt[5:4] << 14
constexpr int WP_WAIT = ((6 * (WP_R - 1)) & 15) | (7 << 4) | (15 << 8) | (((6 * (WP_R - 1)) >> 4) << 14);
static_assert(WP_R == 4 || WP_R == 6 || WP_R == 8, "WP_R");
static_assert(WP_R != 4 || WP_WAIT == 0x4F72, "vmcnt(18)");
constexpr int WP_NBUF = WP_R + 1;             // A ring: the tile being read + WP_R in flight
constexpr int WP_UNROLL = 24;                 // k-tiles of straight-line code (multiple of WP_R and PK_R)

struct PackedParams {
    const uint16_t *A;
    const uint16_t *Wp;
    const float *bias;
    const int64_t *a_rows;
    void *C;
    int64_t lda, ldc;
    int64_t M, N, K;
    int64_t a_gstride, wp_gstride, c_gstride;
    int32_t kc_total;        // 16-element k chunks per packed column block (= round_up(K, 64) / 16)
    int32_t a_rows_group0_only;
    int32_t c_dtype;
};

// GN > 0: gridDim.z is one more than the number of groups, and the workgroups of that last z-slice play the gather
// role (gsage_gather_dev.h) on fan-out GN: the projection fills 416 of the 768 workgroup slots of the chip at
// config 2's level 0 and streams its operands at half of what a CU can keep in flight.
// HOPS: gridDim.z is one more than the number of groups, and the workgroups of that last z-slice sample a LATER
// batch's frontier (the fused multi-hop sampler, gsage_sample_dev.h; gsage_hops_role_next): K1 is a chain of six
// dependent loads per seed (~9 us) that moves almost nothing -- inside the projection's launch, whose 416 workgroups
// leave 352 slots free and wait on their own operand stream, it costs nothing, where in the launch that carries the
// update it was the longer of that launch's two chains (round 5: 15.8 -> ~12 us for that launch).
constexpr int WP_SMEM_VEC = (BM * (BN + 4) * 4) / 16 > WP_NBUF * (BM * CH) ? (BM * (BN + 4) * 4) / 16 : WP_NBUF * (BM * CH);

template <int ACT, int GN, bool HOPS>
__global__ void __launch_bounds__(256)
k_linear_nt_packed(const PackedParams p, const TailGather tg, const HopsParams hp)
{
    constexpr int EPC = 8;
    constexpr int ATILE = BM * CH;                       // vec16 slots per A buffer (8 KiB)
    __shared__ vec16 smem[WP_SMEM_VEC];
    if (HOPS && blockIdx.z + 1 == gridDim.z) {
        sample_hops_workgroup<false>(hp, (int)(blockIdx.y * gridDim.x + blockIdx.x), reinterpret_cast<int64_t *>(smem));
        return;
    }
    if (GN > 0 && blockIdx.z + 1 == gridDim.z) {
        gather_role<(GN > 0 ? GN : 1), 2>(tg, (int)(blockIdx.y * gridDim.x + blockIdx.x));
        return;
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int g = blockIdx.z;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int64_t n0 = (int64_t)blockIdx.y * BN;
    const uint16_t *A = p.A + (int64_t)g * p.a_gstride;
    const int64_t *a_rows = (p.a_rows && (g == 0 || !p.a_rows_group0_only)) ? p.a_rows : nullptr;
    const int nk = (int)((p.K + 63) / 64);
    // ragged N: a wave whose 32 columns do not exist computes the last existing block again (its
    // results are never stored) -- no divergent region around the MFMAs
    const int64_t jb_last = (p.N - 1) >> 5;
    const int64_t jb = min((n0 >> 5) + wave, jb_last);
    const vec16 *wp = reinterpret_cast<const vec16 *>(p.Wp + (int64_t)g * p.wp_gstride) +
                      (jb * p.kc_total) * 64 + lane;

    // A DMA assignment as in k_linear_nt_dma: instruction s fills rows 32*s + 8*wave + (lane >> 3)
    const int cdst = lane & 7;
    const uint16_t *a_src[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const int R = 32 * s2 + 8 * wave + (lane >> 3);
        const int row = (R & ~9) | ((R & 1) << 3) | ((R >> 3) & 1);
        int64_t m = m0 + row;
        if (m >= p.M) m = p.M - 1;
        const int64_t r = a_rows ? a_rows[m] : m;
        a_src[s2] = A + r * p.lda + (cdst ^ (row & 7)) * EPC;
    }
    const int a_dst0 = (8 * wave) * CH, a_dst1 = (32 + 8 * wave) * CH;

    vec16 wr[WP_R][4];
    // (scheduling barriers: the wait counts below assume that the six memory instructions of a tile
    //  are not interleaved with another tile's -- left alone, the scheduler shuffles them)
    auto issue_tile = [&](int kt, int buf, vec16 (&w)[4]) {
        vec16 *base = smem + buf * ATILE;
        const int64_t ko = (int64_t)kt * 64;
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_global_load_lds((global_void_t *)(a_src[0] + ko), (lds_void_t *)(base + a_dst0), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((global_void_t *)(a_src[1] + ko), (lds_void_t *)(base + a_dst1), 16, 0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) w[kk] = wp[(int64_t)(kt * 4 + kk) * 64];
        __builtin_amdgcn_sched_barrier(0);
    };

    f32x16_t acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const int arow0 = lane & 31, arow1 = arow0 + 32;

    // Every tile is 6 vector-memory instructions per wave (2 DMA + 4 loads), issued in tile order and
    // returning in order: "tile kt landed" == at most 6 * (tiles issued after it) still outstanding.
    auto compute_tile = [&](int kt, const vec16 (&w)[4]) {
        const vec16 *sA = smem + (kt % WP_NBUF) * ATILE;
        vec16 fa0[4], fa1[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int ch = kk * 2 + (lane >> 5);
            fa0[kk] = sA[lds_slot(arow0, ch)];
            fa1[kk] = sA[lds_slot(arow1, ch)];
        }
        // all eight fragment reads in flight together, then the MFMAs back to back
        __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            mma_chunk<uint16_t>::run(fa0[kk], w[kk], acc0);
            mma_chunk<uint16_t>::run(fa1[kk], w[kk], acc1);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    int kt = 0, phase = 0;
    if (nk > WP_R) {
#pragma unroll
        for (int t = 0; t < WP_R; ++t) issue_tile(t, t, wr[t]);
        // steady state: every step waits for ITS tile only (three later tiles stay in flight), computes
        // it and refills its registers / the A buffer of the tile before it.  No conditional issue in
        // here, and the waits are builtins (not inline asm): the compiler's own wait insertion can then
        // prove that wr[s] is ready and adds no vmcnt(0) in front of the MFMAs.
        //     simm16 = vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 15 << 8 | vmcnt[5:4] << 14
        // The first WP_UNROLL tiles run as straight-line code: at a loop header the compiler's wait
        // insertion merges the prologue's and the back edge's pending-load state conservatively and
        // drains everything (vmcnt(0)) in front of the first MFMAs of every trip -- one full pipeline
        // stall per WP_R tiles.  Longer reductions continue in the rolled loop.
#pragma unroll
        for (int u = 0; u < WP_UNROLL; ++u) {
            if (kt + WP_R >= nk) { phase = u % WP_R; goto drain; }
            __builtin_amdgcn_s_waitcnt(WP_WAIT);         // vmcnt(6 (WP_R - 1))
            __builtin_amdgcn_s_barrier();                // everybody's part of A(kt) landed; A(kt-1) is free
            compute_tile(kt, wr[u % WP_R]);
            issue_tile(kt + WP_R, (kt + WP_R) % WP_NBUF, wr[u % WP_R]);
            ++kt;
        }
        for (;;) {
#pragma unroll
            for (int s = 0; s < WP_R; ++s) {
                if (kt + WP_R >= nk) { phase = s; goto drain; }
                __builtin_amdgcn_s_waitcnt(WP_WAIT);
                __builtin_amdgcn_s_barrier();
                compute_tile(kt, wr[s]);
                issue_tile(kt + WP_R, (kt + WP_R) % WP_NBUF, wr[s]);
                ++kt;
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < WP_R; ++t)
            if (t < nk) issue_tile(t, t, wr[t]);
    }
drain:
    // the last (up to) WP_R tiles are all in flight and nothing is issued any more: one full wait
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int t = 0; t < WP_R; ++t) {
        if (kt + t < nk) {
            const int idx = (phase + t) % WP_R;
#pragma unroll
            for (int s = 0; s < WP_R; ++s)
                if (idx == s) compute_tile(kt + t, wr[s]);
        }
    }

    // ---- epilogue: bias + activation, tile through LDS, 16-byte row chunks to global ----------------
    const float *bias = p.bias ? p.bias + (int64_t)g * p.N : nullptr;
    const int esz = p.c_dtype == GSAGE_BF16 ? 2 : 4;
    const int epc = 16 / esz;
    const int64_t cbase = (int64_t)g * p.c_gstride + n0;
    const bool wide = n0 + BN <= p.N && p.ldc % epc == 0 && cbase % epc == 0 && ((uintptr_t)p.C % 16) == 0;
    const int jl = wave * 32 + (lane & 31);
    const float bj = (bias && n0 + jl < p.N) ? bias[n0 + jl] : 0.f;
    if (wide) {
        __syncthreads();
        const int ldt = BN + epc;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const f32x16_t &acc = rb ? acc1 : acc0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float v = apply_act(acc[r] + bj, ACT);
                if (p.c_dtype == GSAGE_BF16)
                    ((uint16_t *)smem)[i * ldt + jl] = f32_to_bf16(v);
                else
                    ((float *)smem)[i * ldt + jl] = v;
            }
        }
        __syncthreads();
        const int cpr = BN / epc;
        for (int q = tid; q < BM * cpr; q += 256) {
            const int row = q / cpr, ch = q - row * cpr;
            const int64_t m = m0 + row;
            if (m < p.M) {
                const vec16 v = *reinterpret_cast<const vec16 *>((const char *)smem +
                                                                 ((size_t)row * ldt + ch * epc) * esz);
                *reinterpret_cast<vec16 *>((char *)p.C + ((size_t)m * p.ldc + cbase + ch * epc) * esz) = v;
            }
        }
        return;
    }
    const int64_t j = n0 + jl;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const f32x16_t &acc = rb ? acc1 : acc0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
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

// -------------------------------------------------------------------------------------------------
// K5 for SHORT reductions (K <= 256: papers100M's 128-d features, round 6): weight-stationary, persistent.
//
// With two k-tiles per 64-row tile the kernel above never reaches its steady state: a workgroup requests its A tile
// (16 KB) AND its whole weight block (32 KB of fragments at K = 128: two thirds of what the CU ingests), waits, runs
// 16 MFMAs per wave and leaves -- 2 656 workgroups at configs[4]'s level 0, 127 MB through the CUs' vector-memory paths
// for 87 MB of HBM traffic.  Here a workgroup loads its 128-column block of W ONCE (a wave's fragments of all NK
// k-tiles: 16 NK registers) and walks `tpw` consecutive row tiles: only A moves (16 KB in, 16 KB out per tile), the
// LDS-DMA ring runs across tile boundaries (R k-tiles = a whole number of row tiles in flight), and a tile's epilogue
// (bias + activation -> bf16 tile in LDS -> 16-byte row chunks) is the only other traffic.
//   * LDS-DMA and its waits are inline asm (the compiler would drain every outstanding DMA at the loop's back edge);
//     vector-memory operations complete in issue order, so "tile j landed" is a count: the R - 1 tiles requested
//     behind it (2 instructions per wave each) + the 4 stores of each of the TR epilogues issued since.  Every
//     instruction of that count is issued unconditionally: row tiles past the end repeat the last tile, rows past M the
//     last row -- the same values stored to the same addresses again.
//   * the rows' table ids (a_rows) of all the workgroup's tiles are read once into LDS.
// -------------------------------------------------------------------------------------------------
constexpr int WS_TPW = 8;                      // row tiles per workgroup (at most)
constexpr int WS_LDT = BN + 8;                 // bf16 elements per row of the output staging tile

template <int NK> constexpr int ws_ring() { return NK == 3 ? 3 : 4; }           // k-tiles in flight (a multiple of NK)
template <int NK> constexpr size_t ws_lds_bytes()
{
    return (size_t)(ws_ring<NK>() + 1) * BM * CH * 16 + (size_t)BM * WS_LDT * 2 + (size_t)WS_TPW * 2 * 256 * 4;
}

template <int N>
__device__ __forceinline__ void ws_wait_vm()
{
    static_assert(N >= 0 && N < 64, "vmcnt is six bits");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// BASE + STEP t for the (unrolled, hence constant) t = 0 .. 3
template <int BASE, int STEP>
__device__ __forceinline__ void ws_wait_vm_sel(int t)
{
    if (t == 0) ws_wait_vm<BASE>();
    else if (t == 1) ws_wait_vm<BASE + STEP>();
    else if (t == 2) ws_wait_vm<BASE + 2 * STEP>();
    else ws_wait_vm<BASE + 3 * STEP>();
}

template <int ACT, int NK>
__global__ void __launch_bounds__(256)
k_linear_nt_packed_ws(const PackedParams p, const int tpw)
{
    constexpr int R = ws_ring<NK>(), TR = R / NK, NBUF = R + 1, ATILE = BM * CH, NST = 4;
    constexpr int STEADY = 2 * (R - 1) + NST * TR;        // vector-memory instructions younger than a tile about to be read
    extern __shared__ __attribute__((aligned(16))) char ws_smem[];
    vec16 *ring = reinterpret_cast<vec16 *>(ws_smem);
    uint16_t *stage = reinterpret_cast<uint16_t *>(ws_smem + (size_t)NBUF * ATILE * 16);
    uint32_t *rid_s = reinterpret_cast<uint32_t *>(ws_smem + (size_t)NBUF * ATILE * 16 + (size_t)BM * WS_LDT * 2);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.z;
    const int64_t n0 = (int64_t)blockIdx.y * BN;
    const int64_t n_tiles = (p.M + BM - 1) / BM;
    const int64_t tile0 = (int64_t)blockIdx.x * tpw;
    const uint16_t *A = p.A + (int64_t)g * p.a_gstride;
    const int64_t *a_rows = (p.a_rows && (g == 0 || !p.a_rows_group0_only)) ? p.a_rows : nullptr;

    // the two rows this lane requests per k-tile (as in k_linear_nt_packed) and their chunk, swizzled on the source side
    const int cdst = lane & 7;
    int rowof[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const int Rr = 32 * s2 + 8 * wave + (lane >> 3);
        rowof[s2] = (Rr & ~9) | ((Rr & 1) << 3) | ((Rr >> 3) & 1);
    }
    for (int t = 0; t < tpw; ++t) {
        int64_t tile = tile0 + t;
        if (tile >= n_tiles) tile = n_tiles - 1;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            int64_t m = tile * BM + rowof[s2];
            if (m >= p.M) m = p.M - 1;
            rid_s[(t * 2 + s2) * 256 + tid] = (uint32_t)(a_rows ? a_rows[m] : m);
        }
    }
    // this wave's 32 columns of W, every k-tile: registers for the whole launch
    const vec16 *wp = reinterpret_cast<const vec16 *>(p.Wp + (int64_t)g * p.wp_gstride) +
                      (((n0 >> 5) + wave) * p.kc_total) * 64 + lane;
    vec16 wr[NK][4];
#pragma unroll
    for (int kt = 0; kt < NK; ++kt)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) wr[kt][kk] = wp[(int64_t)(kt * 4 + kk) * 64];
    const int jl = wave * 32 + (lane & 31);
    float bj = p.bias ? p.bias[(int64_t)g * p.N + n0 + jl] : 0.f;
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(lds_void_t *)ring;
    const uint32_t dst0 = (uint32_t)((8 * wave) * CH * 16), dst1 = (uint32_t)((32 + 8 * wave) * CH * 16);

    // k-tile kt of local row tile t (past the workgroup's last tile: that tile again) -> ring buffer `buf`
    auto issue = [&](int t, int kt, int buf) {
        const int tc = t < tpw ? t : tpw - 1;
        const uint32_t r0 = rid_s[(tc * 2 + 0) * 256 + tid], r1 = rid_s[(tc * 2 + 1) * 256 + tid];
        const uint16_t *s0 = A + (int64_t)r0 * p.lda + (cdst ^ (rowof[0] & 7)) * 8 + kt * 64;
        const uint16_t *s1 = A + (int64_t)r1 * p.lda + (cdst ^ (rowof[1] & 7)) * 8 + kt * 64;
        const uint32_t base = ring_lds + (uint32_t)(buf * ATILE * 16);
        const uint32_t m0a = __builtin_amdgcn_readfirstlane(base + dst0), m0b = __builtin_amdgcn_readfirstlane(base + dst1);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(s0), "s"(m0a) : "memory", "m0");
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(s1), "s"(m0b) : "memory", "m0");
    };
    f32x16_t acc0, acc1;
    const int arow0 = lane & 31, arow1 = arow0 + 32;
    auto compute = [&](int buf, const vec16 (&w)[4]) {
        const vec16 *sA = ring + buf * ATILE;
        vec16 fa0[4], fa1[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int ch = kk * 2 + (lane >> 5);
            fa0[kk] = sA[lds_slot(arow0, ch)];
            fa1[kk] = sA[lds_slot(arow1, ch)];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            mma_chunk<uint16_t>::run(fa0[kk], w[kk], acc0);
            mma_chunk<uint16_t>::run(fa1[kk], w[kk], acc1);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // bias + activation -> bf16 tile in LDS -> 16-byte row chunks; FOUR stores per thread, every one of them issued
    auto epilogue = [&](int t) {
        int64_t tile = tile0 + t;
        if (tile >= n_tiles) tile = n_tiles - 1;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const f32x16_t &acc = rb ? acc1 : acc0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                stage[i * WS_LDT + jl] = f32_to_bf16(apply_act(acc[r] + bj, ACT));
            }
        }
        __syncthreads();
        const int64_t cbase = (int64_t)g * p.c_gstride + n0;
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int qd = tid + 256 * u, row = qd >> 4, ch = qd & 15;
            int64_t m = tile * BM + row;
            if (m >= p.M) m = p.M - 1;                    // (that row's values: rows past M were read from row M - 1)
            const vec16 v = *reinterpret_cast<const vec16 *>(stage + row * WS_LDT + ch * 8);
            *reinterpret_cast<vec16 *>((uint16_t *)p.C + m * p.ldc + cbase + ch * 8) = v;
        }
    };
    auto zero = [&]() {
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    };

    // the first R k-tiles (TR row tiles) go out before anything is waited for
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // (the ids are in LDS, W and the bias in registers)
    // (W and the bias count as read HERE: the compiler's own waits for them land before the first LDS-DMA request, where
    //  they cost nothing -- it does not see those requests and would otherwise wait for all of them at W's first use)
#pragma unroll
    for (int kt = 0; kt < NK; ++kt)
        asm volatile("" : "+v"(wr[kt][0]), "+v"(wr[kt][1]), "+v"(wr[kt][2]), "+v"(wr[kt][3]));
    asm volatile("" : "+v"(bj));
#pragma unroll
    for (int j = 0; j < R; ++j) issue(j / NK, j % NK, j);
    int buf = 0;
    // row tiles 0 .. TR - 1: the epilogues issued so far are counted one by one
#pragma unroll
    for (int t = 0; t < TR; ++t) {
        zero();
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) {
            ws_wait_vm_sel<2 * (R - 1), NST>(t);
            __builtin_amdgcn_s_barrier();
            compute(buf, wr[kt]);
            issue(t + TR, kt, (buf + R) % NBUF);
            buf = (buf + 1) % NBUF;
        }
        epilogue(t);
    }
    for (int t = TR; t < tpw; ++t) {
        zero();
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) {
            ws_wait_vm<STEADY>();
            __builtin_amdgcn_s_barrier();
            compute(buf, wr[kt]);
            issue(t + TR, kt, buf == 0 ? NBUF - 1 : buf - 1);          // (buf + R) % NBUF with NBUF = R + 1
            buf = buf + 1 == NBUF ? 0 : buf + 1;
        }
        epilogue(t);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (requests past the end must not land in the next workgroup's LDS)
}

// -------------------------------------------------------------------------------------------------
// K3 with the packed weight operand: pooling MLP, 64 rows x (NW x 64) hidden columns per workgroup.
//
// The 64 x 128-tile K3 above reads every A tile once per 128 hidden columns (4 x 164 MB on the hop-2
// frontier of the Reddit shape) and moves W through LDS.  Here wave w owns columns [64w, 64w + 64) and
// all 64 rows (four 32 x 32 accumulators) and takes its B fragments from the packed operand straight
// into registers (8 loads per k-tile); A (8 KiB per k-tile) goes through a four-buffer LDS-DMA ring.
// Epilogue as in K3: bias + ReLU'd tile -> LDS (fp32), optional sign bits, segment max / mean down
// the rows.  Measured on the hop-2 frontier (128 000 rows x 602 -> 512, tools/kbench.py pool):
//     64 x 128 tiles, W through LDS (k_linear_nt POOL)            180 us   437 TF/s
//     NW = 8: 512 columns, A read once, ONE workgroup per CU      156 us   506 TF/s
//     NW = 4: 256 columns, A read twice, two workgroups per CU    130 us   606 TF/s   <- launched
//     weight-stationary persistent variant (a wave keeps its 32 columns x 640 k in 160 registers,
//     only A moves, column blocks of a row tile on one XCD)       133 us   595 TF/s   (removed)
// One workgroup per CU loses to two even at half the A traffic: all eight waves stall at the same
// barrier.  The weight-stationary variant moved no W at all and was no faster: its A stream (the
// same tile requested by four workgroups in lock step) ran at 5.2 TB/s, i.e. every tile still paid
// the full miss latency with only 4 x 8 KiB in flight per workgroup.  Round 3's diagnostics (DESIGN.md
// section 5): with every operand made L1 / L2 resident and no epilogue the main loop still takes 90 of
// its 130 us -- it is bound by the bytes a CU's vector-memory path moves per MFMA (40 KiB per k-tile
// and workgroup, 32 of them W fragments), and a 128-row tile per W fragment loses more by running one
// workgroup per CU (182 us) than it saves.
// -------------------------------------------------------------------------------------------------
constexpr int PK_R = 3;                       // tiles in flight
constexpr int PK_NBUF = PK_R + 1;

struct PoolPackedParams {
    const uint16_t *A;
    const uint16_t *Wp;
    const float *bias;
    const int64_t *a_rows;
    int64_t lda;
    int64_t M, N, K;          // M = rows of A (= segments * pool_n)
    int32_t kc_total;
    int32_t pool_n, pool_groups, pool_mode;
    float *pooled;
    int64_t pooled_ld;
    uint16_t *pooled_b;
    int64_t pooled_b_ld;
    int32_t *argmax;
    uint32_t *relu_mask;
};

// PN > 0 (round 6): segments of exactly PN rows, max pooling, pooled IN REGISTERS -- a wave holds all 64 rows of its
// columns (row i of the tile sits in accumulator register (i & 3) + 4 ((i & 31) >> 3) of row block i >> 5, in the
// lane half (i >> 2) & 1), so a segment's maximum is a chain of compares over compile-time register sets in the two
// halves and ONE exchange between them; no fp32 tile in LDS (67 KiB per workgroup: the LDS then holds the A ring only),
// no barrier after the main loop.  PN == 0: any fan-out, mean pooling, sign bits -- through the LDS tile as before.
template <int NW, int PN>
__global__ void __launch_bounds__(NW * 64)
k_pool_mlp_packed(const PoolPackedParams p)
{
    constexpr int EPC = 8;
    constexpr int PK_BN = NW * 64;
    constexpr int PK_LDT = PK_BN + 8;                    // fp32 tile row: rows 4 apart land 32 banks apart
    constexpr int NDMA = 8 / NW;                         // A DMA instructions per wave and tile
    constexpr int WAIT_STEADY = NW == 8 ? 0x4F72 : 0x4F74;   // vmcnt((PK_R - 1) * (NDMA + 8)): 18 / 20
    constexpr int ATILE = BM * CH;                       // vec16 slots per A buffer (8 KiB)
    __shared__ vec16 smem[PN > 0 ? PK_NBUF * ATILE : (BM * PK_LDT * 4) / 16];   // A ring (32 KiB) / + fp32 output tile

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int rows_per_wg = p.pool_groups * p.pool_n;
    const int64_t m0 = (int64_t)blockIdx.x * rows_per_wg;
    const int64_t n0 = (int64_t)blockIdx.y * PK_BN;
    const int nk = (int)((p.K + 63) / 64);
    const int64_t jb_last = (p.N - 1) >> 5;
    const int64_t jb0 = min((n0 >> 5) + wave * 2, jb_last), jb1 = min((n0 >> 5) + wave * 2 + 1, jb_last);
    const vec16 *wp0 = reinterpret_cast<const vec16 *>(p.Wp) + (jb0 * p.kc_total) * 64 + lane;
    const vec16 *wp1 = reinterpret_cast<const vec16 *>(p.Wp) + (jb1 * p.kc_total) * 64 + lane;

    // A DMA: instruction d of wave w fills LDS rows 8 (NW d + w) + (lane >> 3), slot lane & 7, from the
    // swizzled source
    const uint16_t *a_src[NDMA];
#pragma unroll
    for (int d = 0; d < NDMA; ++d) {
        const int R = 8 * (NW * d + wave) + (lane >> 3);
        const int row = (R & ~9) | ((R & 1) << 3) | ((R >> 3) & 1);
        int64_t m = m0 + min(row, rows_per_wg - 1);
        if (m >= p.M) m = p.M - 1;
        const int64_t r = p.a_rows ? p.a_rows[m] : m;
        a_src[d] = p.A + r * p.lda + ((lane & 7) ^ (row & 7)) * EPC;
    }
    const int a_dst = (8 * wave) * CH;

    vec16 wr[PK_R][2][4];
    auto issue_tile = [&](int kt, int buf, vec16 (&w)[2][4]) {
        vec16 *base = smem + buf * ATILE;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int d = 0; d < NDMA; ++d)
            __builtin_amdgcn_global_load_lds((global_void_t *)(a_src[d] + (int64_t)kt * 64),
                                             (lds_void_t *)(base + a_dst + 8 * NW * d * CH), 16, 0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            w[0][kk] = wp0[(int64_t)(kt * 4 + kk) * 64];
            w[1][kk] = wp1[(int64_t)(kt * 4 + kk) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    f32x16_t acc[2][2];                                  // [row block][column block]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int arow0 = lane & 31, arow1 = arow0 + 32;

    // Fragment reads run TWO 16-deep steps ahead of the MFMAs inside a k-tile (round 6): the four MFMAs of step kk
    // (128 clocks of the matrix pipe) cover the LDS round trip of step kk + 2's fragments; read up front and waited
    // for together (lgkmcnt(0) before the first MFMA) the eight reads' latency was exposed once per k-tile and wave.
    // GSAGE_PK_READ_AHEAD=0 at build time: the old order.
#ifndef GSAGE_PK_READ_AHEAD
#define GSAGE_PK_READ_AHEAD 1
#endif
    auto compute_tile = [&](int kt, const vec16 (&w)[2][4]) {
        const vec16 *sA = smem + (kt % PK_NBUF) * ATILE;
        vec16 fa0[4], fa1[4];
#if GSAGE_PK_READ_AHEAD
        auto rd = [&](int kk) {
            const int ch = kk * 2 + (lane >> 5);
            fa0[kk] = sA[lds_slot(arow0, ch)];
            fa1[kk] = sA[lds_slot(arow1, ch)];
        };
        rd(0);
        rd(1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            // simm16 = vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt << 8 | vmcnt[5:4] << 14: vmcnt / expcnt left alone
            if (kk < 3) __builtin_amdgcn_s_waitcnt(0xC27F);          // lgkmcnt(2): step kk's two reads have landed
            else __builtin_amdgcn_s_waitcnt(0xC07F);                 // lgkmcnt(0)
            __builtin_amdgcn_sched_barrier(0);
            mma_chunk<uint16_t>::run(fa0[kk], w[0][kk], acc[0][0]);
            mma_chunk<uint16_t>::run(fa0[kk], w[1][kk], acc[0][1]);
            mma_chunk<uint16_t>::run(fa1[kk], w[0][kk], acc[1][0]);
            mma_chunk<uint16_t>::run(fa1[kk], w[1][kk], acc[1][1]);
            if (kk + 2 < 4) rd(kk + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
#else
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int ch = kk * 2 + (lane >> 5);
            fa0[kk] = sA[lds_slot(arow0, ch)];
            fa1[kk] = sA[lds_slot(arow1, ch)];
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            mma_chunk<uint16_t>::run(fa0[kk], w[0][kk], acc[0][0]);
            mma_chunk<uint16_t>::run(fa0[kk], w[1][kk], acc[0][1]);
            mma_chunk<uint16_t>::run(fa1[kk], w[0][kk], acc[1][0]);
            mma_chunk<uint16_t>::run(fa1[kk], w[1][kk], acc[1][1]);
        }
        __builtin_amdgcn_sched_barrier(0);
#endif
    };

    // NDMA + 8 vector-memory instructions per tile and wave; see k_linear_nt_packed
    int kt = 0, phase = 0;
    if (nk > PK_R) {
#pragma unroll
        for (int t = 0; t < PK_R; ++t) issue_tile(t, t, wr[t]);
        // straight-line for the first WP_UNROLL tiles (see k_linear_nt_packed), rolled beyond
#pragma unroll
        for (int u = 0; u < WP_UNROLL; ++u) {
            if (kt + PK_R >= nk) { phase = u % PK_R; goto drain; }
            __builtin_amdgcn_s_waitcnt(WAIT_STEADY);     // two later tiles stay in flight
            __builtin_amdgcn_s_barrier();
            compute_tile(kt, wr[u % PK_R]);
            issue_tile(kt + PK_R, (kt + PK_R) % PK_NBUF, wr[u % PK_R]);
            ++kt;
        }
        for (;;) {
#pragma unroll
            for (int s = 0; s < PK_R; ++s) {
                if (kt + PK_R >= nk) { phase = s; goto drain; }
                __builtin_amdgcn_s_waitcnt(WAIT_STEADY);
                __builtin_amdgcn_s_barrier();
                compute_tile(kt, wr[s]);
                issue_tile(kt + PK_R, (kt + PK_R) % PK_NBUF, wr[s]);
                ++kt;
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < PK_R; ++t)
            if (t < nk) issue_tile(t, t, wr[t]);
    }
drain:
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int t = 0; t < PK_R; ++t) {
        if (kt + t < nk) {
            const int idx = (phase + t) % PK_R;
#pragma unroll
            for (int s = 0; s < PK_R; ++s)
                if (idx == s) compute_tile(kt + t, wr[s]);
        }
    }

    if constexpr (PN > 0) {
        // ---- epilogue in registers: bias + ReLU, segment max + argmax (the first maximum of a segment wins, as in
        //      the sequential scan of the LDS epilogue and of torch.max) ------------------------------------------
        constexpr int G = BM / PN;                       // segments per tile
        const int h = lane >> 5;
        float best[2][G];
        int arg[2][G];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int64_t jc = n0 + wave * 64 + cb * 32 + (lane & 31);
            const float bj = (p.bias && jc < p.N) ? p.bias[jc] : 0.f;
#pragma unroll
            for (int sg = 0; sg < G; ++sg) { best[cb][sg] = -1.f; arg[cb][sg] = 0; }      // (ReLU'd values are >= 0)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i0 = 32 * rb + (r & 3) + 8 * (r >> 2);         // this register's row in half 0; half 1: + 4
                    float v = acc[rb][cb][r] + bj;
                    v = v > 0.f ? v : 0.f;
#pragma unroll
                    for (int sg = 0; sg < G; ++sg) {
                        const bool in0 = i0 / PN == sg, in1 = (i0 + 4) / PN == sg;       // (compile-time)
                        if (!in0 && !in1) continue;
                        // (selects, not branches: a divergent `if` here costs an exec save / restore per register)
                        const bool mine = (in0 && in1) || (in0 ? h == 0 : h == 1);
                        const bool take = mine && v > best[cb][sg];
                        best[cb][sg] = take ? v : best[cb][sg];
                        arg[cb][sg] = take ? i0 - sg * PN + 4 * h : arg[cb][sg];
                    }
                }
#pragma unroll
            for (int sg = 0; sg < G; ++sg) {             // the other half's rows of the segment
                const float ob = __shfl_xor(best[cb][sg], 32, 64);
                const int oa = __shfl_xor(arg[cb][sg], 32, 64);
                const bool take = ob > best[cb][sg] || (ob == best[cb][sg] && oa < arg[cb][sg]);
                best[cb][sg] = take ? ob : best[cb][sg];
                arg[cb][sg] = take ? oa : arg[cb][sg];
            }
        }
        // both halves hold every result: half 0 stores column block 0, half 1 column block 1 -- the wave's 64 columns
        // leave as ONE store per segment and array
        const int64_t j = n0 + wave * 64 + lane;
#pragma unroll
        for (int sg = 0; sg < G; ++sg) {
            const float bv = h ? best[1][sg] : best[0][sg];
            const int av = h ? arg[1][sg] : arg[0][sg];
            const int64_t seg = (int64_t)blockIdx.x * G + sg;
            if (seg * PN < p.M && j < p.N) {
                p.pooled[seg * p.pooled_ld + j] = bv;
                if (p.pooled_b) p.pooled_b[seg * p.pooled_b_ld + j] = f32_to_bf16(bv);
                if (p.argmax) p.argmax[seg * p.N + j] = av;
            }
        }
        return;
    }
    // ---- epilogue: bias + ReLU'd tile -> LDS, sign bits, segment max / mean down the rows ----------
    float *tile = reinterpret_cast<float *>(smem);
    __syncthreads();                                     // the A ring is free
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int jl = wave * 64 + cb * 32 + (lane & 31);
        const int64_t j = n0 + jl;
        const float bj = (p.bias && j < p.N) ? p.bias[j] : 0.f;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float v = acc[rb][cb][r] + bj;
                tile[i * PK_LDT + jl] = v > 0.f ? v : 0.f;
            }
    }
    __syncthreads();
    if (p.relu_mask) {
        // sign bits of the hidden activations: one task = 32 channels of one tile row
        for (int task = tid; task < BM * (PK_BN / 32); task += NW * 64) {
            const int i = task / (PK_BN / 32), q = task % (PK_BN / 32);
            const int64_t row = m0 + i;
            if (i < rows_per_wg && row < p.M && n0 + q * 32 < p.N) {
                uint32_t bits = 0;
#pragma unroll
                for (int e0 = 0; e0 < 32; ++e0) {
                    const int e = (e0 + i) & 31;           // rotate per row: spreads the LDS banks
                    bits |= (tile[i * PK_LDT + q * 32 + e] > 0.f ? 1u : 0u) << e;
                }
                p.relu_mask[row * (p.N / 32) + (n0 >> 5) + q] = bits;
            }
        }
    }
    const int64_t j = n0 + tid;                          // thread t owns hidden column t of the tile
    if (j < p.N) {
        for (int sg = 0; sg < p.pool_groups; ++sg) {
            const int64_t seg = (int64_t)blockIdx.x * p.pool_groups + sg;
            if (seg * p.pool_n >= p.M) break;
            const float *colp = tile + (sg * p.pool_n) * PK_LDT + tid;
            float best = colp[0];
            int arg = 0;
            float sum = best;
            for (int r = 1; r < p.pool_n; ++r) {
                const float v = colp[r * PK_LDT];
                sum += v;
                if (v > best) { best = v; arg = r; }
            }
            if (p.pool_mode == GSAGE_POOL_MAX) {
                p.pooled[seg * p.pooled_ld + j] = best;
                if (p.pooled_b) p.pooled_b[seg * p.pooled_b_ld + j] = f32_to_bf16(best);
                if (p.argmax) p.argmax[seg * p.N + j] = arg;
            } else {
                p.pooled[seg * p.pooled_ld + j] = sum / (float)p.pool_n;
                if (p.pooled_b) p.pooled_b[seg * p.pooled_b_ld + j] = f32_to_bf16(sum / (float)p.pool_n);
            }
        }
    }
}


// W [groups][N][ldw] (fp32 or bf16) -> packed operand; one thread per 16-byte lane slot
template <typename TW>
__global__ void __launch_bounds__(256)
k_pack_weight(const TW *__restrict__ W, int64_t ldw, int64_t w_gstride, int64_t N, int64_t K,
              int32_t kc_total, int32_t jb_total, int32_t groups, uint16_t *__restrict__ Wp)
{
    const int64_t total = (int64_t)groups * jb_total * kc_total * 64;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
        const int lane = (int)(t & 63);
        int64_t u = t >> 6;
        const int kc = (int)(u % kc_total);
        u /= kc_total;
        const int jb = (int)(u % jb_total);
        const int g = (int)(u / jb_total);
        const int64_t j = (int64_t)jb * 32 + (lane & 31);
        const int64_t k0 = (int64_t)kc * 16 + (lane >> 5) * 8;
        uint16_t out[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = 0.f;
            if (j < N && k0 + e < K) {
                const TW w = W[(int64_t)g * w_gstride + j * ldw + k0 + e];
                v = sizeof(TW) == 2 ? bf16_to_f32((uint16_t)w) : (float)w;
            }
            out[e] = f32_to_bf16(v);
        }
        vec16 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (uint32_t)out[2 * e] | ((uint32_t)out[2 * e + 1] << 16);
        reinterpret_cast<vec16 *>(Wp)[t] = o;
    }
}

}  // namespace gsage

using namespace gsage;

extern "C" {

int64_t gsage_packed_weight_elems(int64_t N, int64_t K, int32_t groups)
{
    return (int64_t)groups * ceil_div(N, 32) * (ceil_div(K, 64) * 4) * 64 * 8;
}

int gsage_pack_weight(const void *W, int dtype, int64_t ldw, int64_t w_gstride, int64_t N, int64_t K,
                      int32_t groups, void *Wp, void *stream)
{
    GSAGE_REQUIRE(W && Wp && N > 0 && K > 0 && groups >= 1 && ldw >= K, "pack_weight: bad arguments");
    GSAGE_REQUIRE(dtype == GSAGE_BF16 || dtype == GSAGE_F32, "pack_weight: bad dtype %d", dtype);
    GSAGE_REQUIRE(((uintptr_t)Wp % 16) == 0, "pack_weight: output must be 16-byte aligned");
    const int32_t kc = (int32_t)(ceil_div(K, 64) * 4), jb = (int32_t)ceil_div(N, 32);
    const int64_t slots = (int64_t)groups * jb * kc * 64;
    int64_t blocks = ceil_div(slots, 256);
    if (blocks > 4096) blocks = 4096;
    if (dtype == GSAGE_BF16)
        launch(k_pack_weight<uint16_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
               (const uint16_t *)W, ldw, w_gstride, N, K, kc, jb, groups, (uint16_t *)Wp);
    else
        launch(k_pack_weight<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
               (const float *)W, ldw, w_gstride, N, K, kc, jb, groups, (uint16_t *)Wp);
    return check_launch("pack_weight");
}

int gsage_linear_nt_packed(const void *A, int64_t lda, const int64_t *a_rows, int a_rows_group0_only,
                           const void *Wp, const float *bias, void *C, int c_dtype, int64_t ldc,
                           int64_t M, int64_t N, int64_t K, int act, int groups, int64_t a_gstride,
                           int64_t c_gstride, void *stream)
{
    // gsage_gather_role_next() / gsage_hops_role_next(): consumed before any return path, never left for a later launch
    const gsage_tail_gather_desc *gd = take_gather_role();
    const gsage_hops_desc *hd = t_hops_role;
    t_hops_role = nullptr;
    GSAGE_REQUIRE(A && Wp && C, "linear_nt_packed: null pointer");
    GSAGE_REQUIRE(M >= 0 && N > 0 && K > 0, "linear_nt_packed: bad sizes");
    GSAGE_REQUIRE(c_dtype == GSAGE_BF16 || c_dtype == GSAGE_F32, "linear_nt_packed: bad c_dtype");
    GSAGE_REQUIRE(groups >= 1 && groups <= 65535, "linear_nt_packed: bad group count");
    GSAGE_REQUIRE(act >= ACT_NONE && act <= ACT_TANH, "linear_nt_packed: bad activation code");
    // A rows must be whole 128-byte lines (zero padded up to round_up(K, 64)): the DMA streams lines
    GSAGE_REQUIRE(lda % 64 == 0 && ceil_div(K, 64) * 64 <= lda && ((uintptr_t)A % 16) == 0 &&
                  ((uintptr_t)Wp % 16) == 0, "linear_nt_packed: A rows must be whole zero-padded 128-byte lines");
    if (M == 0) return GSAGE_OK;
    PackedParams p;
    p.A = (const uint16_t *)A; p.Wp = (const uint16_t *)Wp; p.bias = bias; p.a_rows = a_rows; p.C = C;
    p.lda = lda; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
    p.a_gstride = a_gstride; p.c_gstride = c_gstride;
    p.kc_total = (int32_t)(ceil_div(K, 64) * 4);
    p.wp_gstride = ceil_div(N, 32) * (int64_t)p.kc_total * 64 * 8;
    p.a_rows_group0_only = a_rows_group0_only; p.c_dtype = c_dtype;
    dim3 grid((unsigned)ceil_div(M, BM), (unsigned)ceil_div(N, BN), (unsigned)groups);
    hipStream_t s = (hipStream_t)stream;
    // gsage_gather_role_next(): one more z-slice of workgroups gathers part of the next batch's level-0 rows
    TailGather tg = {};
    HopsParams hp = {};
    if (hd) {
        // gsage_hops_role_next(): one more z-slice of workgroups samples a later batch's frontier
        GSAGE_REQUIRE(!(gd && gd->rows > 0) && act == ACT_RELU && !hd->dense_adj,
                      "linear_nt_packed: the sampler role rides with the ReLU projection, walks a CSR, and excludes the "
                      "gather role");
        size_t lds = 0;
        const int rc = fill_hops(hp, lds, *hd);
        if (rc != GSAGE_OK) return rc;
        const int64_t slots = (int64_t)grid.x * grid.y;
        hp.spw = (int32_t)ceil_div(hd->B > 0 ? hd->B : 1, slots);     // seeds per workgroup: the slice serves the batch
        int64_t widest = 1, width = 1;
        for (int k = 1; k <= hp.n_hops; ++k) { width *= hp.fan[k]; widest = width > widest ? width : widest; }
        GSAGE_REQUIRE(sizeof(int64_t) * 2 * (size_t)hp.spw * (size_t)widest <= sizeof(vec16) * WP_SMEM_VEC,
                      "linear_nt_packed: the sampler role's frontier (%d seeds x %lld ids, twice) does not fit the "
                      "projection's LDS", hp.spw, (long long)widest);
        grid.z += 1;
        launch(k_linear_nt_packed<ACT_RELU, 0, true>, grid, dim3(256), 0, s, p, tg, hp);
        return check_launch("linear_nt_packed");
    }
    if (gd && gd->rows > 0) {
        const int rc = fill_gather_role(tg, *gd, "linear_nt_packed (gather role)");
        if (rc != GSAGE_OK) return rc;
        GSAGE_REQUIRE(act == ACT_RELU && (gd->n == 10 || gd->n == 5),
                      "linear_nt_packed: the gather role rides with the ReLU projection, fan-out 5 or 10");
        tg.n_wg = (int32_t)(grid.x * grid.y);
        grid.z += 1;
        if (gd->n == 10) launch(k_linear_nt_packed<ACT_RELU, 10, false>, grid, dim3(256), 0, s, p, tg, hp);
        else launch(k_linear_nt_packed<ACT_RELU, 5, false>, grid, dim3(256), 0, s, p, tg, hp);
        return check_launch("linear_nt_packed");
    }
    {
        // short reductions into bf16 rows: the weight-stationary persistent kernel (GSAGE_K5_WS=0: never)
        const char *ws_env = getenv("GSAGE_K5_WS");
        const bool ws_on = !ws_env || atoi(ws_env) != 0;
        const int nk = (int)ceil_div(K, 64);
        const int tr = nk == 1 ? 4 : nk == 2 ? 2 : 1;
        const int64_t n_tiles = ceil_div(M, BM);
        if (ws_on && c_dtype == GSAGE_BF16 && nk <= 4 && N % BN == 0 && ldc % 8 == 0 && c_gstride % 8 == 0 &&
            ((uintptr_t)C % 16) == 0 && n_tiles >= 4 * tr && (act == ACT_RELU || act == ACT_NONE)) {
            int dev = 0, cus = 256;
            if (hipGetDevice(&dev) == hipSuccess) {
                int v = 0;
                if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
                else (void)hipGetLastError();
            } else (void)hipGetLastError();
            // two workgroups fit a CU: aim at one round of 2 x CUs workgroups over all column blocks and groups
            const int64_t slices = (int64_t)grid.y * groups;
            int64_t per = ceil_div(2 * (int64_t)cus, slices);
            if (per < 1) per = 1;
            int64_t tpw = ceil_div(n_tiles, per);
            if (const char *e = getenv("GSAGE_K5_WS_TPW")) tpw = atoi(e);          // (a sweep knob)
            if (tpw < tr) tpw = tr;
            if (tpw > WS_TPW) tpw = WS_TPW;
            grid.x = (unsigned)ceil_div(n_tiles, tpw);
#define GSAGE_WS_LAUNCH(ACTV, NKV)                                                                                        \
            do {                                                                                                      \
                static bool raised = false;                                                                           \
                if (!raised) {                                                                                        \
                    if (hipFuncSetAttribute((const void *)k_linear_nt_packed_ws<ACTV, NKV>,                           \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)ws_lds_bytes<NKV>()) != hipSuccess) { \
                        (void)hipGetLastError();                                                                      \
                        set_error("linear_nt_packed: cannot raise the dynamic LDS limit");                          \
                        return GSAGE_ELAUNCH;                                                                         \
                    }                                                                                                 \
                    raised = true;                                                                                    \
                }                                                                                                     \
                launch(k_linear_nt_packed_ws<ACTV, NKV>, grid, dim3(256), ws_lds_bytes<NKV>(), s, p, (int)tpw);       \
            } while (0)
#define GSAGE_WS_NK(ACTV)                                                                                                 \
            do {                                                                                                      \
                if (nk == 1) GSAGE_WS_LAUNCH(ACTV, 1);                                                                \
                else if (nk == 2) GSAGE_WS_LAUNCH(ACTV, 2);                                                           \
                else if (nk == 3) GSAGE_WS_LAUNCH(ACTV, 3);                                                           \
                else GSAGE_WS_LAUNCH(ACTV, 4);                                                                        \
            } while (0)
            if (act == ACT_RELU) GSAGE_WS_NK(ACT_RELU);
            else GSAGE_WS_NK(ACT_NONE);
#undef GSAGE_WS_NK
#undef GSAGE_WS_LAUNCH
            return check_launch("linear_nt_packed");
        }
    }
    if (act == ACT_RELU)
        launch(k_linear_nt_packed<ACT_RELU, 0, false>, grid, dim3(256), 0, s, p, tg, hp);
    else if (act == ACT_TANH)
        launch(k_linear_nt_packed<ACT_TANH, 0, false>, grid, dim3(256), 0, s, p, tg, hp);
    else
        launch(k_linear_nt_packed<ACT_NONE, 0, false>, grid, dim3(256), 0, s, p, tg, hp);
    return check_launch("linear_nt_packed");
}

int gsage_pool_mlp_packed(const void *A, int64_t lda, const int64_t *a_rows, const void *Wp,
                          const float *bias, int64_t M, int32_t n, int64_t H, int64_t K, int pool,
                          float *pooled, int64_t pooled_ld, int32_t *argmax, void *pooled_bf16,
                          int64_t pooled_bf16_ld, uint32_t *relu_mask, void *stream)
{
    GSAGE_REQUIRE(A && Wp && pooled && pooled_ld >= H, "pool_mlp_packed: bad pointers / output");
    GSAGE_REQUIRE(M >= 0 && H > 0 && K > 0, "pool_mlp_packed: bad sizes");
    GSAGE_REQUIRE(n >= 1 && n <= BM, "pool_mlp_packed: fanout must be in [1, %d]", BM);
    GSAGE_REQUIRE(!relu_mask || H % 32 == 0, "pool_mlp_packed: relu_mask needs H % 32 == 0");
    GSAGE_REQUIRE(!pooled_bf16 || pooled_bf16_ld >= H, "pool_mlp_packed: bad bf16 output");
    GSAGE_REQUIRE(pool == GSAGE_POOL_MAX || pool == GSAGE_POOL_MEAN, "pool_mlp_packed: bad pool mode");
    GSAGE_REQUIRE(lda % 64 == 0 && ceil_div(K, 64) * 64 <= lda && ((uintptr_t)A % 16) == 0 &&
                  ((uintptr_t)Wp % 16) == 0, "pool_mlp_packed: A rows must be whole zero-padded 128-byte lines");
    if (M == 0) return GSAGE_OK;
    PoolPackedParams p;
    p.A = (const uint16_t *)A; p.Wp = (const uint16_t *)Wp; p.bias = bias; p.a_rows = a_rows; p.lda = lda;
    p.M = M * (int64_t)n; p.N = H; p.K = K; p.kc_total = (int32_t)(ceil_div(K, 64) * 4);
    p.pool_n = n; p.pool_groups = BM / n; p.pool_mode = pool; p.pooled = pooled; p.pooled_ld = pooled_ld;
    p.pooled_b = (uint16_t *)pooled_bf16; p.pooled_b_ld = pooled_bf16_ld; p.argmax = argmax;
    p.relu_mask = relu_mask;
    const int64_t subs = ceil_div(M, p.pool_groups);
    dim3 grid((unsigned)subs, (unsigned)ceil_div(H, 256), 1);
    // max pooling over the usual fan-outs: pooled in registers (GSAGE_POOL_REGS=0: the LDS epilogue)
    static const bool regs = [] { const char *e = getenv("GSAGE_POOL_REGS"); return !e || atoi(e) != 0; }();
    const bool rp = regs && pool == GSAGE_POOL_MAX && !relu_mask;
    hipStream_t s = (hipStream_t)stream;
    if (rp && n == 10) launch(k_pool_mlp_packed<4, 10>, grid, dim3(256), 0, s, p);
    else if (rp && n == 25) launch(k_pool_mlp_packed<4, 25>, grid, dim3(256), 0, s, p);
    else if (rp && n == 5) launch(k_pool_mlp_packed<4, 5>, grid, dim3(256), 0, s, p);
    else if (rp && n == 15) launch(k_pool_mlp_packed<4, 15>, grid, dim3(256), 0, s, p);
    else if (rp && n == 20) launch(k_pool_mlp_packed<4, 20>, grid, dim3(256), 0, s, p);
    else launch(k_pool_mlp_packed<4, 0>, grid, dim3(256), 0, s, p);
    return check_launch("pool_mlp_packed");
}

}  // extern "C"
