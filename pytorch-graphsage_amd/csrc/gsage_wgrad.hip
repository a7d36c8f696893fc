// gsage_wgrad.hip -- weight-gradient contraction on the gfx950 matrix cores.
//
// Replaces what autograd runs for the backward of nn.Linear's weight on the hot path
// (reference: loss.backward() at models.py:100, through fc_x / fc_neib of nn_modules.py:189-190):
//
//     dW_g[n, k] = sum_m dC[m, g*n_per_group + n] * A_g[m, k]          fp32 result
//
// Both operands are "M-major" (the reduction index m is the SLOW dimension of both matrices),
// the opposite of what the MFMA wants (8 consecutive reduction elements per lane).  Instead of
// transposing through LDS, each lane loads, for 8 consecutive rows m, the 16 bytes holding 8 adjacent
// columns, and re-packs them in registers with v_perm_b32 into eight operands -- one per column
// residue e -- so MFMA (e, f) owns output rows n = 8*i + e and columns k = 8*j + f (a fixed
// interleave of the output, undone for free in the epilogue's addressing).  Same trick on the A
// side.  The MFMA is v_mfma_f32_16x16x32_bf16: 16 lanes x 8 columns span the 128-wide tile and the
// four 16-lane groups take four 8-row groups, so one step is 32 rows and every global load is a
// 16-byte lane load (256 B contiguous per 16 lanes).  (The first version used 32x32x16 with 8-byte
// lane loads: twice the vector-memory instructions per row, and the texture addresser -- ~16 clocks per
// wave-wide load whatever its width -- was what bound it.)
//
// Decomposition: workgroup tile = wave tile = 128 (n) x 128 (k) = 64 accumulators of 4 (256 AGPRs, one
// wave per SIMD, one workgroup per CU).  The reduction over M is split across grid.x into
// `rows_per_split` slices; inside a workgroup the four waves quarter the slice, then sum their tiles
// through LDS (every wave parks its tile, then sums and stores a quarter of the output), so one fp32
// partial tile per workgroup goes to the slabs (deterministic; summed by k_reduce_slabs /
// gsage_finalize_grads -- no atomics).
#include "gsage_common.h"

namespace gsage {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct WgradParams {
    const uint16_t *dC;
    const uint16_t *A;
    float *slabs;
    int64_t ldc, lda, a_gstride;
    int64_t M, Ntot, K, n_per_group, ldk, rows_per_split;
    const int64_t *a_rows;      // optional: row m of the A operand is A[a_rows[m]] (a frontier read in place)
};

// operand for column residue E (0..7) out of eight 16-byte row segments: element E of rows 0..7
template <int E>
__device__ __forceinline__ u32x4 pack_column(const u32x4 (&v)[8])
{
    constexpr uint32_t sel = (E & 1) ? 0x07060302u : 0x05040100u;     // hi|hi : lo|lo halves
    u32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        o[q] = __builtin_amdgcn_perm(v[2 * q + 1][E >> 1], v[2 * q][E >> 1], sel);
    return o;
}

template <int E>
__device__ __forceinline__ void mma_row(const u32x4 (&c_cur)[8], const u32x4 (&opb)[8], f32x4 (&acc)[8][8])
{
    const u32x4 opa = pack_column<E>(c_cur);
#pragma unroll
    for (int f = 0; f < 8; ++f)
        acc[E][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, opa),
                                                            __builtin_bit_cast(bf16x8_t, opb[f]), acc[E][f], 0, 0, 0);
}

// one 32-row step: lane l holds rows 8 (l >> 4) .. +7, columns 8 (l & 15) .. +7 of both operands
__device__ __forceinline__ void mma_step(const u32x4 (&c_cur)[8], const u32x4 (&a_cur)[8], f32x4 (&acc)[8][8])
{
    u32x4 opb[8];
    opb[0] = pack_column<0>(a_cur);
    opb[1] = pack_column<1>(a_cur);
    opb[2] = pack_column<2>(a_cur);
    opb[3] = pack_column<3>(a_cur);
    opb[4] = pack_column<4>(a_cur);
    opb[5] = pack_column<5>(a_cur);
    opb[6] = pack_column<6>(a_cur);
    opb[7] = pack_column<7>(a_cur);
    mma_row<0>(c_cur, opb, acc);
    mma_row<1>(c_cur, opb, acc);
    mma_row<2>(c_cur, opb, acc);
    mma_row<3>(c_cur, opb, acc);
    mma_row<4>(c_cur, opb, acc);
    mma_row<5>(c_cur, opb, acc);
    mma_row<6>(c_cur, opb, acc);
    mma_row<7>(c_cur, opb, acc);
}

// Software pipeline (one wave per SIMD has nothing else to hide latency with): phase s issues the
// 16 loads of step s+1, then runs the 64 MFMAs of step s -- one step (16 KiB per wave) stays in
// flight.  Two register slots with static numbering (loop unrolled x2).
// The loads are inline asm and the waits explicit: with ordinary loads the compiler's own wait
// insertion has to merge the "more steps follow" and "last step" paths at every join and ends up
// draining the loads it has just issued (vmcnt(13) ... vmcnt(0) where vmcnt(16) is right) -- the
// pipeline then degenerates to load, wait, compute, one step at a time.  What the asm hides from the
// compiler is replaced by hand: a wait before a slot is consumed, scheduling barriers so that nothing
// that reads a slot moves above its wait and no load moves above the MFMAs still reading its target.
__device__ __forceinline__ void load16_async(u32x4 &dst, const uint16_t *ptr)
{
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory");
}

__device__ __forceinline__ void wgrad_mainloop(const WgradParams &p, const uint16_t *A,
                                               int64_t m_begin, int64_t m_end, int64_t n_off,
                                               int64_t k_off, int rg, f32x4 (&acc)[8][8])
{
    u32x4 cb[2][8], ab[2][8];
    const int64_t nfull = (m_end - m_begin) / 32;            // steps with 32 valid rows
    const uint16_t *c_base = p.dC + (m_begin + 8 * rg) * p.ldc + n_off;
    const uint16_t *a_base = A + (m_begin + 8 * rg) * p.lda + k_off;
    auto load_data = [&](u32x4 (&cdst)[8], u32x4 (&adst)[8], int64_t step) {
        const uint16_t *cp = c_base + step * 32 * p.ldc;
        const uint16_t *ap = a_base + step * 32 * p.lda;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            load16_async(cdst[r], cp + r * p.ldc);
            load16_async(adst[r], ap + r * p.lda);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // simm16 = vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 15 << 8 | vmcnt[5:4] << 14
    auto consume = [&](u32x4 (&c)[8], u32x4 (&a)[8], bool newer_in_flight) {
        if (newer_in_flight) __builtin_amdgcn_s_waitcnt(0x4F70);     // vmcnt(16): the step issued after this one
        else __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
        mma_step(c, a, acc);
        __builtin_amdgcn_sched_barrier(0);
    };
    if (nfull > 0) load_data(cb[0], ab[0], 0);
    for (int64_t s0 = 0; s0 < nfull; s0 += 2) {
        const bool more1 = s0 + 1 < nfull, more2 = s0 + 2 < nfull;   // wave-uniform
        if (more1) load_data(cb[1], ab[1], s0 + 1);
        consume(cb[0], ab[0], more1);
        if (more1) {
            if (more2) load_data(cb[0], ab[0], s0 + 2);
            consume(cb[1], ab[1], more2);
        }
    }
    // ragged tail (< 32 rows, last slice only): rows past the end contribute zeros
    const int64_t m_tail = m_begin + nfull * 32;
    if (m_tail < m_end) {
        u32x4 ct[8], at[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int64_t m = m_tail + 8 * rg + r;
            const bool ok = m < m_end;
            const int64_t mm = ok ? m : m_begin;
            u32x4 c = *reinterpret_cast<const u32x4 *>(p.dC + mm * p.ldc + n_off);
            u32x4 a = *reinterpret_cast<const u32x4 *>(A + mm * p.lda + k_off);
            ct[r] = ok ? c : u32x4{0u, 0u, 0u, 0u};
            at[r] = ok ? a : u32x4{0u, 0u, 0u, 0u};
        }
        mma_step(ct, at, acc);
    }
}

// Same pipeline with the A operand read IN PLACE through a row list (a_rows[m] = table row of reduction index m:
// the frontier's level-0 rows, no gathered copy).  A lane needs the eight ids of its row group per step: 64
// contiguous bytes, four more 16-byte loads per step and lane, issued TWO steps ahead of the data they address so
// that they have landed when the addresses are formed.  In-order completion makes the waits countable:
//     ... ids(s+1) | data(s) | ids(s+2) | data(s+1) ...
//   addresses of data(s+1) need ids(s+1):  younger = data(s) [16] + ids(s+2) [4]      -> vmcnt(20)
//   MFMAs of step s need data(s):          younger = ids(s+2) [4] + data(s+1) [16]    -> vmcnt(20)
// (16 / 4 / 0 where the younger loads were not issued: the last two steps).
__device__ __forceinline__ void wgrad_mainloop_rows(const WgradParams &p, const uint16_t *A, int64_t m_begin,
                                                    int64_t m_end, int64_t n_off, int64_t k_off, int rg,
                                                    f32x4 (&acc)[8][8])
{
    u32x4 cb[2][8], ab[2][8], ib[2][4];
    const int64_t nfull = (m_end - m_begin) / 32;
    const uint16_t *c_base = p.dC + (m_begin + 8 * rg) * p.ldc + n_off;
    const int64_t *i_base = p.a_rows + m_begin + 8 * rg;
    const uint16_t *a_col = A + k_off;
    auto load_ids = [&](u32x4 (&dst)[4], int64_t step) {
        const uint16_t *ip = reinterpret_cast<const uint16_t *>(i_base + step * 32);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) load16_async(dst[q], ip + 8 * q);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto wait = [&](int n) {      // simm16 = vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 15 << 8 | vmcnt[5:4] << 14
        if (n == 20) __builtin_amdgcn_s_waitcnt(0x4F74);
        else if (n == 16) __builtin_amdgcn_s_waitcnt(0x4F70);
        else if (n == 4) __builtin_amdgcn_s_waitcnt(0x0F74);
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto load_data = [&](u32x4 (&cdst)[8], u32x4 (&adst)[8], const u32x4 (&ids)[4], int64_t step) {
        const uint16_t *cp = c_base + step * 32 * p.ldc;
        const uint16_t *ap[8];
        // Every dword of the id registers counts as read HERE: the high dwords are never used, and a register the
        // compiler considers dead while its load is still in flight can be handed out as scratch and is then
        // overwritten when the load lands.
        asm volatile("" ::"v"(ids[0]), "v"(ids[1]), "v"(ids[2]), "v"(ids[3]));
#pragma unroll
        for (int r = 0; r < 8; ++r)          // (row ids are < 2^32: the low dword)
            ap[r] = a_col + (int64_t)ids[r >> 1][(r & 1) * 2] * p.lda;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            load16_async(cdst[r], cp + r * p.ldc);
            load16_async(adst[r], ap[r]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto consume = [&](u32x4 (&c)[8], u32x4 (&a)[8], int younger) {
        wait(younger);
        mma_step(c, a, acc);
        __builtin_amdgcn_sched_barrier(0);
    };
    if (nfull > 0) {
        load_ids(ib[0], 0);
        if (nfull > 1) load_ids(ib[1], 1);
        wait(nfull > 1 ? 4 : 0);
        load_data(cb[0], ab[0], ib[0], 0);
    }
    for (int64_t s0 = 0; s0 < nfull; s0 += 2) {
        const bool more1 = s0 + 1 < nfull, more2 = s0 + 2 < nfull, more3 = s0 + 3 < nfull;   // wave-uniform
        if (more2) load_ids(ib[0], s0 + 2);
        if (more1) {
            wait(more2 ? 20 : 16);
            load_data(cb[1], ab[1], ib[1], s0 + 1);
        }
        consume(cb[0], ab[0], (more2 ? 4 : 0) + (more1 ? 16 : 0));
        if (more1) {
            if (more3) load_ids(ib[1], s0 + 3);
            if (more2) {
                wait(more3 ? 20 : 16);
                load_data(cb[0], ab[0], ib[0], s0 + 2);
            }
            consume(cb[1], ab[1], (more3 ? 4 : 0) + (more2 ? 16 : 0));
        }
    }
    const int64_t m_tail = m_begin + nfull * 32;
    if (m_tail < m_end) {
        u32x4 ct[8], at[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int64_t m = m_tail + 8 * rg + r;
            const bool ok = m < m_end;
            const int64_t mm = ok ? m : m_begin;
            u32x4 c = *reinterpret_cast<const u32x4 *>(p.dC + mm * p.ldc + n_off);
            u32x4 a = *reinterpret_cast<const u32x4 *>(A + p.a_rows[mm] * p.lda + k_off);
            ct[r] = ok ? c : u32x4{0u, 0u, 0u, 0u};
            at[r] = ok ? a : u32x4{0u, 0u, 0u, 0u};
        }
        mma_step(ct, at, acc);
    }
}

// ---- in-workgroup reduction of the four waves' partial tiles ---------------------------------------
// D[i][j] of MFMA (E, f) is dW[n_base + 8i + E][k_base + 8j + f]; lane l holds j = l & 15 and
// i = 4 (l >> 4) + reg, so (f = 0..7) of one reg are eight consecutive floats of one output row.
// Two rounds of four residue blocks; in a round every wave parks all four blocks (slot
// (E & 3) * 4 + wave, laid out [reg][half][lane] float4 -- consecutive lanes, consecutive 16-byte
// words), then wave w sums the four parked copies of block E = 4 * round + w in wave order
// (deterministic) and stores it.  Branch-free on purpose: with per-wave `if (wave == E)` special
// cases hipcc spilled accumulators to scratch.
constexpr int WGRAD_SLOT = 8 * 64 * 4;                      // floats per parked block (8 KiB)
constexpr size_t WGRAD_LDS_BYTES = 16 * WGRAD_SLOT * sizeof(float);

template <int E>
__device__ __forceinline__ void park_block(const f32x4 (&acc)[8][8], float *lds, int wave, int lane)
{
    float *slot = lds + ((E & 3) * 4 + wave) * WGRAD_SLOT;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const f32x4 v = {acc[E][4 * hh][reg], acc[E][4 * hh + 1][reg], acc[E][4 * hh + 2][reg],
                             acc[E][4 * hh + 3][reg]};
            *reinterpret_cast<f32x4 *>(slot + (((reg * 2 + hh) * 64 + lane) << 2)) = v;
        }
}

// residue block E of one output tile: COPIES parked copies, `step` slots apart from `slot` on, summed in that order
template <int COPIES>
__device__ __forceinline__ void reduce_store_block(const float *slot, int step, int E, const WgradParams &p,
                                                   float *slab, int64_t n_base, int64_t k0, int rg, int lane)
{
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int64_t n = n_base + 8 * (4 * rg + reg) + E;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int off = ((reg * 2 + hh) * 64 + lane) << 2;
            f32x4 v = *reinterpret_cast<const f32x4 *>(slot + off);
#pragma unroll
            for (int q = 1; q < COPIES; ++q) v += *reinterpret_cast<const f32x4 *>(slot + q * step * WGRAD_SLOT + off);
            const int64_t k = k0 + 4 * hh;
            if (k + 3 < p.ldk && n < p.Ntot) *reinterpret_cast<f32x4 *>(slab + n * p.ldk + k) = v;
        }
    }
}

__device__ __forceinline__ void reduce_store(const float *lds, int round, int wave, const WgradParams &p,
                                             float *slab, int64_t n_base, int64_t k0, int rg, int lane)
{
    reduce_store_block<4>(lds + (wave * 4) * WGRAD_SLOT, 1, 4 * round + wave, p, slab, n_base, k0, rg, lane);
}

// PAIR mode (below): the workgroup owns TWO output tiles, waves t and t + 2 hold the two halves of tile t's slice.  Of
// a round's four parked blocks per tile wave w sums blocks 2 (w >> 1) and 2 (w >> 1) + 1 of its own tile w & 1
// (slots j * 4 + t and j * 4 + t + 2, in that order) and stores them.
__device__ __forceinline__ void reduce_store_pair(const float *lds, int round, int wave, const WgradParams &p,
                                                  float *slab, int64_t n_base, int64_t k0, int rg, int lane)
{
    const int t = wave & 1;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int j = 2 * (wave >> 1) + u;
        reduce_store_block<2>(lds + (j * 4 + t) * WGRAD_SLOT, 2, 4 * round + j, p, slab, n_base, k0, rg, lane);
    }
}

// One workgroup's share of problem p: M-slice bx, 128 x 128 output tile (by, bz).  The four waves
// take a quarter of the slice each (whole 32-row steps), then meet in LDS.  Compared with one wave
// per tile and slice this quarters the number of partial tiles that travel through HBM to
// gsage_finalize_grads for the same number of busy SIMDs.
// PAIR (round 6; problems with an even number of 128-row output tiles per group and a long reduction): the workgroup owns
// the two tiles (2 by, bz) and (2 by + 1, bz); waves 0 / 1 walk the first half of the slice for tile 0 / 1, waves 2 / 3 the
// second half.  The two waves of a half read the SAME rows of A at the same time (the second request finds the line in
// the CU's L1 or in flight), so a K tile's rows are wanted by half as many workgroups: at the max-pool shape every A
// line by two instead of four -- the re-reads that miss the L2 (a slice's readers drift apart by more than the 4 MB
// hold) were most of the launch's HBM traffic.  Same number of steps per wave, twice the slices (partial tiles).
__device__ __forceinline__ void wgrad_workgroup(const WgradParams &p, int64_t bx, int64_t by, int64_t bz,
                                                float *lds, bool pair = false)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c16 = lane & 15;                                           // this lane's 8 columns
    const int rg = lane >> 4;                                            // this lane's 8-row group
    const int64_t n_base = (pair ? 2 * by + (wave & 1) : by) * 128;
    const int64_t k_base = bz * 128;
    const int g = (int)(n_base / p.n_per_group);
    const uint16_t *A = p.A + (int64_t)g * p.a_gstride;
    const bool by_rows = p.a_rows != nullptr && g == 0;     // (the row list belongs to group 0's operand)

    const int64_t m_begin = bx * p.rows_per_split;
    const int64_t m_end = (m_begin + p.rows_per_split < p.M) ? m_begin + p.rows_per_split : p.M;
    const int parts = pair ? 2 : 4;
    const int64_t quarter = ((m_end - m_begin + 32 * parts - 1) / (32 * parts)) * 32;   // rows per wave, whole steps
    int64_t w_begin = m_begin + (pair ? wave >> 1 : wave) * quarter, w_end = w_begin + quarter;
    if (w_begin > m_end) w_begin = m_end;
    if (w_end > m_end) w_end = m_end;

    // Column validity is lane invariant: a lane whose columns fall outside the matrices reads
    // column 0 instead and its (garbage) accumulators are simply never stored -- output (n, k)
    // depends on column n of dC and column k of A only.  So the steady-state loop has no selects
    // on loaded data, and hipcc keeps the loads in flight instead of waiting right after issue.
    const bool n_ok = n_base + 8 * c16 + 7 < p.ldc && n_base + 8 * c16 < p.Ntot;
    const bool k_ok = k_base + 8 * c16 + 7 < p.lda;
    const int64_t n_off = n_ok ? n_base + 8 * c16 : 0;
    const int64_t k_off = k_ok ? k_base + 8 * c16 : 0;

    f32x4 acc[8][8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int f = 0; f < 8; ++f) acc[e][f] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (by_rows) wgrad_mainloop_rows(p, A, w_begin, w_end, n_off, k_off, rg, acc);
    else wgrad_mainloop(p, A, w_begin, w_end, n_off, k_off, rg, acc);

    // two rounds of four residue blocks (16 x 8 KiB of LDS)
    float *slab = p.slabs + bx * p.Ntot * p.ldk;
    const int64_t k0 = k_base + 8 * c16;
    park_block<0>(acc, lds, wave, lane);
    park_block<1>(acc, lds, wave, lane);
    park_block<2>(acc, lds, wave, lane);
    park_block<3>(acc, lds, wave, lane);
    lds_barrier();
    if (pair) reduce_store_pair(lds, 0, wave, p, slab, n_base, k0, rg, lane);
    else reduce_store(lds, 0, wave, p, slab, n_base, k0, rg, lane);
    lds_barrier();
    park_block<4>(acc, lds, wave, lane);
    park_block<5>(acc, lds, wave, lane);
    park_block<6>(acc, lds, wave, lane);
    park_block<7>(acc, lds, wave, lane);
    lds_barrier();
    if (pair) reduce_store_pair(lds, 1, wave, p, slab, n_base, k0, rg, lane);
    else reduce_store(lds, 1, wave, p, slab, n_base, k0, rg, lane);
}

__global__ void __launch_bounds__(256, 1)
k_wgrad_bf16(const WgradParams p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    wgrad_workgroup(p, blockIdx.x, blockIdx.y, blockIdx.z, lds);
}

// Several weight-gradient problems in ONE launch (all levels of a backward pass).  Each problem
// alone occupies a fraction of the chip for the length of its M-slice (Reddit shapes: 196 and 16
// workgroups on 256 CUs), so side by side they cost the longest one instead of the sum.
constexpr int WGRAD_MAX_PROBLEMS = 8;
struct WgradMulti {
    WgradParams p[WGRAD_MAX_PROBLEMS];
    int32_t first[WGRAD_MAX_PROBLEMS + 1];     // workgroups [first[s], first[s+1]) belong to problem s
    int32_t S[WGRAD_MAX_PROBLEMS], ny[WGRAD_MAX_PROBLEMS];    // (ny: tile rows -- or PAIRS of tile rows -- of the grid)
    int32_t pair[WGRAD_MAX_PROBLEMS];
    int32_t n_prob;
    // gsage_wgrad_ticks_next: counters workgroup 0 advances when it starts (nothing in this launch reads them)
    int64_t *tick, *tick1, *tick2;
    int64_t inc1, inc2;
};

__device__ __forceinline__ void wgrad_ticks(const WgradMulti &q)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (q.tick) *q.tick += 1;
        if (q.tick1) *q.tick1 += q.inc1;
        if (q.tick2) *q.tick2 += q.inc2;
    }
}

__global__ void __launch_bounds__(256, 1)
k_wgrad_multi(const WgradMulti q)
{
    wgrad_ticks(q);
    int s = 0;
#pragma unroll
    for (int j = 1; j < WGRAD_MAX_PROBLEMS; ++j)
        if (j < q.n_prob && (int)blockIdx.x >= q.first[j]) s = j;
    const int local = (int)blockIdx.x - q.first[s];
    const int bx = local % q.S[s];
    const int rest = local / q.S[s];
    extern __shared__ __attribute__((aligned(16))) float lds[];
    wgrad_workgroup(q.p[s], bx, rest % q.ny[s], rest / q.ny[s], lds, q.pair[s] != 0);
}

// ---- fp32 operands (exact-arithmetic parity mode) ---------------------------------------------------
// Same decomposition and slab layout as the bf16 kernel (M-slice bx, 128 x 128 output tile (by, bz)),
// plain fp32 FMAs: thread (ty, tx) owns an 8 x 8 block of the tile, eight rows of both operands are
// staged in LDS per step.  This is the checker's path (golden fixtures replayed at 2e-4 through the
// same engine that runs the bf16 kernels), not a throughput kernel.
__device__ __forceinline__ void wgrad_workgroup_f32(const WgradParams &p, int64_t bx, int64_t by, int64_t bz,
                                                    float *lds)
{
    const float *dC = reinterpret_cast<const float *>(p.dC);
    const int64_t n_base = by * 128, k_base = bz * 128;
    const int g = (int)(n_base / p.n_per_group);
    const float *A = reinterpret_cast<const float *>(p.A) + (int64_t)g * p.a_gstride;
    const int64_t m_begin = bx * p.rows_per_split;
    const int64_t m_end = (m_begin + p.rows_per_split < p.M) ? m_begin + p.rows_per_split : p.M;
    float *cs = lds, *as = lds + 8 * 128;
    const int t = threadIdx.x, ty = t >> 4, tx = t & 15;
    const int lr = t >> 5, lc = (t & 31) * 4;
    float acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
    for (int64_t m0 = m_begin; m0 < m_end; m0 += 8) {
        const int64_t m = m0 + lr;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t n = n_base + lc + e, k = k_base + lc + e;
            cs[lr * 128 + lc + e] = (m < m_end && n < p.Ntot) ? dC[m * p.ldc + n] : 0.f;
            as[lr * 128 + lc + e] = (m < m_end && k < p.lda) ? A[(p.a_rows && g == 0 ? p.a_rows[m] : m) * p.lda + k] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float cv[8], av[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { cv[i] = cs[r * 128 + ty * 8 + i]; av[i] = as[r * 128 + tx * 8 + i]; }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] += cv[i] * av[j];
        }
        __syncthreads();
    }
    float *slab = p.slabs + bx * p.Ntot * p.ldk;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int64_t n = n_base + ty * 8 + i;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t k = k_base + tx * 8 + j;
            if (n < p.Ntot && k < p.ldk) slab[n * p.ldk + k] = acc[i][j];
        }
    }
}

__global__ void __launch_bounds__(256)
k_wgrad_f32(const WgradParams p)
{
    __shared__ float lds[2 * 8 * 128];
    wgrad_workgroup_f32(p, blockIdx.x, blockIdx.y, blockIdx.z, lds);
}

__global__ void __launch_bounds__(256)
k_wgrad_multi_f32(const WgradMulti q)
{
    wgrad_ticks(q);
    int s = 0;
#pragma unroll
    for (int j = 1; j < WGRAD_MAX_PROBLEMS; ++j)
        if (j < q.n_prob && (int)blockIdx.x >= q.first[j]) s = j;
    const int local = (int)blockIdx.x - q.first[s];
    const int bx = local % q.S[s];
    const int rest = local / q.S[s];
    __shared__ float lds[2 * 8 * 128];
    wgrad_workgroup_f32(q.p[s], bx, rest % q.ny[s], rest / q.ny[s], lds);
}

// out_g[n_local * K + k] = sum_s slabs[s][g * n_per_group + n_local][k]
__global__ void __launch_bounds__(256)
k_reduce_slabs(const float *__restrict__ slabs, int32_t S, int64_t Ntot, int64_t K, int64_t ldk,
               int64_t n_per_group, float *__restrict__ out, int64_t out_gstride)
{
    const int64_t total = Ntot * K;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
        const int64_t n = t / K;
        const int64_t k = t - n * K;
        const float *src = slabs + n * ldk + k;
        float s = 0.f;
        const int64_t sstride = Ntot * ldk;
        int i = 0;
        for (; i + 8 <= S; i += 8) {                       // 8 independent loads in flight (same order of sums)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(i + u) * sstride];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; i < S; ++i) s += src[(int64_t)i * sstride];
        const int64_t g = n / n_per_group;
        out[g * out_gstride + (n - g * n_per_group) * K + k] = s;
    }
}

}  // namespace gsage

using namespace gsage;

extern "C" int gsage_wgrad_slabs(int64_t M, int64_t rows_per_split)
{
    return (int)ceil_div(M, rows_per_split);
}

// 128 KiB of dynamic LDS: above the 64 KiB a kernel gets by default (gfx950 has 160 KiB per CU)
template <typename K>
static int wgrad_raise_lds(K kernel, bool &done)
{
    if (done) return GSAGE_OK;
    if (hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)WGRAD_LDS_BYTES) != hipSuccess) {
        (void)hipGetLastError();
        set_error("wgrad: cannot raise the dynamic LDS limit");
        return GSAGE_ELAUNCH;
    }
    done = true;
    return GSAGE_OK;
}

static int wgrad_fill(WgradParams &p, int dtype, const void *dC, int64_t ldc, const void *A, int64_t lda,
                      int64_t a_gstride, int64_t M, int64_t Ntot, int64_t K, int64_t n_per_group,
                      int64_t rows_per_split, float *slabs, int64_t ldk, const int64_t *a_rows = nullptr)
{
    GSAGE_REQUIRE(((uintptr_t)a_rows % 16) == 0, "wgrad: a_rows must be 16-byte aligned");
    p.a_rows = a_rows;
    GSAGE_REQUIRE(dC && A && slabs, "wgrad: null pointer");
    GSAGE_REQUIRE(dtype == GSAGE_BF16 || dtype == GSAGE_F32, "wgrad: bad dtype");
    GSAGE_REQUIRE(M > 0 && Ntot > 0 && K > 0, "wgrad: bad sizes");
    GSAGE_REQUIRE(dtype == GSAGE_F32 || (ldc % 8 == 0 && lda % 8 == 0), 
                  "wgrad: ldc, lda must be multiples of 8 (16-byte row chunks)");
    GSAGE_REQUIRE(ldk % 4 == 0, "wgrad: ldk must be a multiple of 4");
    GSAGE_REQUIRE(Ntot % 4 == 0 && Ntot <= ldc, "wgrad: Ntot must be a multiple of 4 and <= ldc");
    GSAGE_REQUIRE(ldk >= K && ldk <= lda + 3, "wgrad: need K <= ldk <= lda");
    GSAGE_REQUIRE(n_per_group > 0 && (n_per_group % 128 == 0 || n_per_group >= Ntot),
                  "wgrad: n_per_group must be a multiple of 128 (or a single group)");
    GSAGE_REQUIRE(rows_per_split >= 16 && rows_per_split % 16 == 0, "wgrad: rows_per_split must be a multiple of 16");
    GSAGE_REQUIRE(((uintptr_t)dC % 16) == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)slabs % 16) == 0 &&
                  (a_gstride % 8) == 0, "wgrad: misaligned pointer (16-byte lane loads)");
    p.dC = (const uint16_t *)dC; p.A = (const uint16_t *)A; p.slabs = slabs;
    p.ldc = ldc; p.lda = lda; p.a_gstride = a_gstride; p.M = M; p.Ntot = Ntot; p.K = K;
    p.n_per_group = n_per_group; p.ldk = ldk; p.rows_per_split = rows_per_split;
    return GSAGE_OK;
}

namespace gsage {
struct WgradTicks { int64_t *tick, *tick1, *tick2; int64_t inc1, inc2; };
static thread_local WgradTicks t_wgrad_ticks = {nullptr, nullptr, nullptr, 0, 0};
}

extern "C" {

int gsage_wgrad_ticks_next(int64_t *tick, int64_t *tick1, int64_t inc1, int64_t *tick2, int64_t inc2)
{
    t_wgrad_ticks = WgradTicks{tick, tick1, tick2, inc1, inc2};
    return GSAGE_OK;
}

int gsage_wgrad_pair_ok(int dtype, int64_t M, int64_t Ntot, int64_t n_per_group, int64_t rows_per_split)
{
    // two output tiles per workgroup (k_wgrad_multi, PAIR): bf16, whole pairs of 128-row tiles that share their A
    // operand (the same group), a reduction long enough that the operands' re-reads are what the launch waits for
    const char *e = getenv("GSAGE_WGRAD_PAIR");
    const int mode = e ? atoi(e) : 0;
    return mode != 0 && dtype == GSAGE_BF16 && Ntot % 256 == 0 && (n_per_group % 256 == 0 || n_per_group >= Ntot) &&
           M >= 65536 && rows_per_split >= 256 ? 1 : 0;
}

int gsage_wgrad_multi(int32_t n_prob, const gsage_wgrad_desc *probs, int dtype, void *stream)
{
    const WgradTicks ticks = t_wgrad_ticks;           // (consumed before any return path: never left for a later launch)
    t_wgrad_ticks = WgradTicks{nullptr, nullptr, nullptr, 0, 0};
    GSAGE_REQUIRE(probs && n_prob >= 1 && n_prob <= WGRAD_MAX_PROBLEMS, "wgrad_multi: 1..%d problems",
                  WGRAD_MAX_PROBLEMS);
    WgradMulti q;
    q.n_prob = n_prob;
    q.tick = ticks.tick; q.tick1 = ticks.tick1; q.tick2 = ticks.tick2; q.inc1 = ticks.inc1; q.inc2 = ticks.inc2;
    q.first[0] = 0;
    for (int s = 0; s < WGRAD_MAX_PROBLEMS; ++s) {
        if (s < n_prob) {
            const gsage_wgrad_desc &d = probs[s];
            int rc = wgrad_fill(q.p[s], dtype, d.dC, d.ldc, d.A, d.lda, d.a_gstride, d.M, d.Ntot, d.K,
                                d.n_per_group, d.rows_per_split, d.slabs, d.ldk, d.a_rows);
            if (rc != GSAGE_OK) return rc;
            q.S[s] = (int32_t)ceil_div(d.M, d.rows_per_split);
            q.pair[s] = gsage_wgrad_pair_ok(dtype, d.M, d.Ntot, d.n_per_group, d.rows_per_split);
            q.ny[s] = (int32_t)ceil_div(d.Ntot, 128) / (q.pair[s] ? 2 : 1);
            q.first[s + 1] = q.first[s] + q.S[s] * q.ny[s] * (int32_t)ceil_div(d.ldk, 128);
        } else {
            q.p[s] = q.p[0];
            q.S[s] = q.ny[s] = 1;
            q.pair[s] = 0;
            q.first[s + 1] = q.first[s];
        }
    }
    if (dtype == GSAGE_F32) {
        launch(k_wgrad_multi_f32, dim3((unsigned)q.first[n_prob]), dim3(256), 0, (hipStream_t)stream, q);
        return check_launch("wgrad_multi");
    }
    static bool raised = false;
    int rc2 = wgrad_raise_lds(k_wgrad_multi, raised);
    if (rc2 != GSAGE_OK) return rc2;
    launch(k_wgrad_multi, dim3((unsigned)q.first[n_prob]), dim3(256), WGRAD_LDS_BYTES, (hipStream_t)stream, q);
    return check_launch("wgrad_multi");
}

int gsage_wgrad(const void *dC, int dtype, int64_t ldc, const void *A, int64_t lda, int64_t a_gstride, int64_t M,
                int64_t Ntot, int64_t K, int64_t n_per_group, int64_t rows_per_split, float *slabs, int64_t ldk,
                float *out, int64_t out_gstride, void *stream)
{
    WgradParams p;
    int rc0 = wgrad_fill(p, dtype, dC, ldc, A, lda, a_gstride, M, Ntot, K, n_per_group, rows_per_split, slabs, ldk);
    if (rc0 != GSAGE_OK) return rc0;
    const int S = (int)ceil_div(M, rows_per_split);
    dim3 grid((unsigned)S, (unsigned)ceil_div(Ntot, 128), (unsigned)ceil_div(ldk, 128));
    if (dtype == GSAGE_F32) {
        launch(k_wgrad_f32, grid, dim3(256), 0, (hipStream_t)stream, p);
    } else {
        static bool raised = false;
        int rc1 = wgrad_raise_lds(k_wgrad_bf16, raised);
        if (rc1 != GSAGE_OK) return rc1;
        launch(k_wgrad_bf16, grid, dim3(256), WGRAD_LDS_BYTES, (hipStream_t)stream, p);
    }
    int rc = check_launch("wgrad");
    if (rc != GSAGE_OK || out == nullptr) return rc;      // out == NULL: caller reduces the slabs
    int64_t blocks = ceil_div(Ntot * K, 256);
    if (blocks > 4096) blocks = 4096;
    launch(k_reduce_slabs, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const float *)slabs, S, Ntot, K, ldk, n_per_group, out, out_gstride);
    return check_launch("wgrad_reduce");
}

}  // extern "C"
