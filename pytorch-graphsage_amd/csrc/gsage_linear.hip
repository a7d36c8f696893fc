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

namespace gsage {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2 };

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
};

constexpr int BM = 64;
constexpr int BN = 128;
constexpr int CH = 8;          // 16-byte chunks per tile row (128 bytes of K)

__device__ __forceinline__ int lds_slot(int row, int ch)
{
    const int rp = (row & ~9) | ((row & 1) << 3) | ((row >> 3) & 1);     // swap bits 0 and 3
    return rp * CH + (ch ^ (row & 7));
}

template <typename T>
struct mma_chunk;

template <>
struct mma_chunk<uint16_t> {
    __device__ static __forceinline__ void run(const vec16 &a, const vec16 &b, f32x16_t &acc)
    {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                      __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    }
};

template <>
struct mma_chunk<float> {
    __device__ static __forceinline__ void run(const vec16 &a, const vec16 &b, f32x16_t &acc)
    {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w[e]),
                                                       __uint_as_float(b.w[e]), acc, 0, 0, 0);
    }
};

__device__ __forceinline__ float apply_act(float v, int act)
{
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == ACT_TANH) return tanhf(v);
    return v;
}

template <typename T, bool POOL>
__global__ void __launch_bounds__(256)
k_linear_nt(const LinearParams p)
{
    constexpr int EPC = 16 / (int)sizeof(T);        // elements per 16-byte chunk
    // one raw LDS array (keeps the compiler from serialising waits across objects)
    __shared__ vec16 smem[POOL ? (BM * BN * 4 / 16) : ((BM + BN) * CH)];
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
    const int srow = tid >> 3;           // 0..31
    const int sch = tid & 7;             // chunk inside the tile row
    const T *a_ptr[2];
    const T *w_ptr[4];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int row = srow + 32 * s;
        const int64_t m = m0 + row;
        a_ptr[s] = nullptr;
        if (row < rows_per_wg && m < p.M) {
            const int64_t r = a_rows ? a_rows[m] : m;
            a_ptr[s] = A + r * p.lda + sch * EPC;
        }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int64_t j = n0 + srow + 32 * s;
        w_ptr[s] = (j < p.N) ? W + j * p.ldw + sch * EPC : nullptr;
    }
    const int kchunks = (int)((p.K + EPC - 1) / EPC);
    const int nk = (kchunks + CH - 1) / CH;

    vec16 ra[2], rw[4];
    const vec16 zero = {{0u, 0u, 0u, 0u}};
    auto load_tile = [&](int kt) {
        const bool kin = (kt * CH + sch) < kchunks;
        const int64_t koff = (int64_t)kt * CH * EPC;
#pragma unroll
        for (int s = 0; s < 2; ++s)
            ra[s] = (a_ptr[s] && kin) ? *reinterpret_cast<const vec16 *>(a_ptr[s] + koff) : zero;
#pragma unroll
        for (int s = 0; s < 4; ++s)
            rw[s] = (w_ptr[s] && kin) ? *reinterpret_cast<const vec16 *>(w_ptr[s] + koff) : zero;
    };

    const int wm = wave & 1;
    const int wn = wave >> 1;
    f32x16_t acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    load_tile(0);
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int s = 0; s < 2; ++s) sA[lds_slot(srow + 32 * s, sch)] = ra[s];
#pragma unroll
        for (int s = 0; s < 4; ++s) sW[lds_slot(srow + 32 * s, sch)] = rw[s];
        __syncthreads();
        if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int ch = kk * 2 + (lane >> 5);
            const vec16 a = sA[lds_slot(wm * 32 + (lane & 31), ch)];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const vec16 b = sW[lds_slot(wn * 64 + t * 32 + (lane & 31), ch)];
                mma_chunk<T>::run(a, b, acc[t]);
            }
        }
        __syncthreads();
    }

    // ---- epilogue ---------------------------------------------------------------------------------
    // C/D layout of the 32x32 MFMA: lane l, register r -> column (l & 31),
    // row (r & 3) + 8 * (r >> 2) + 4 * (l >> 5).
    if (!POOL) {
        const float *bias = p.bias ? p.bias + (int64_t)g * p.N : nullptr;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int64_t j = n0 + wn * 64 + t * 32 + (lane & 31);
            const float bj = (bias && j < p.N) ? bias[j] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int64_t m = m0 + wm * 32 + i;
                if (m < p.M && j < p.N) {
                    const float v = apply_act(acc[t][r] + bj, p.act);
                    const int64_t off = m * p.ldc + (int64_t)g * p.c_gstride + j;
                    if (p.c_dtype == GSAGE_BF16)
                        ((uint16_t *)p.C)[off] = f32_to_bf16(v);
                    else
                        ((float *)p.C)[off] = v;
                }
            }
        }
    } else {
        // bias + ReLU'd tile -> LDS [64][128] fp32, then segment max / mean down the rows
        float *tile = reinterpret_cast<float *>(smem);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int jl = wn * 64 + t * 32 + (lane & 31);
            const int64_t j = n0 + jl;
            const float bj = (p.bias && j < p.N) ? p.bias[j] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float v = acc[t][r] + bj;
                tile[i * BN + jl] = v > 0.f ? v : 0.f;
            }
        }
        __syncthreads();
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
                if (p.argmax) p.argmax[seg * p.N + j] = arg;
            } else {
                p.pooled[seg * p.pooled_ld + j] = sum / (float)p.pool_n;
            }
        }
    }
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
    p.argmax = nullptr;
    dim3 grid((unsigned)ceil_div(M, BM), (unsigned)ceil_div(N, BN), (unsigned)groups);
    if (dtype == GSAGE_BF16)
        hipLaunchKernelGGL((k_linear_nt<uint16_t, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((k_linear_nt<float, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("linear_nt");
}

int gsage_pool_mlp(const void *A, int dtype, int64_t lda, const int64_t *a_rows, const void *W,
                   int64_t ldw, const float *bias, int64_t M, int32_t n, int64_t H, int64_t K,
                   int pool, float *pooled, int64_t pooled_ld, int32_t *argmax, void *stream)
{
    GSAGE_REQUIRE(n >= 1 && n <= BM, "pool_mlp: fanout must be in [1, %d]", BM);
    int rc = check_operands("pool_mlp", A, dtype, lda, W, ldw, M * (int64_t)n, H, K);
    if (rc != GSAGE_OK) return rc;
    GSAGE_REQUIRE(pooled && pooled_ld >= H, "pool_mlp: bad output");
    GSAGE_REQUIRE(pool == GSAGE_POOL_MAX || pool == GSAGE_POOL_MEAN, "pool_mlp: bad pool mode");
    if (M == 0) return GSAGE_OK;
    LinearParams p;
    p.A = A; p.W = W; p.bias = bias; p.a_rows = a_rows; p.C = nullptr;
    p.lda = lda; p.ldw = ldw; p.ldc = 0; p.M = M * (int64_t)n; p.N = H; p.K = K;
    p.a_gstride = 0; p.w_gstride = 0; p.c_gstride = 0;
    p.a_rows_group0_only = 0; p.act = ACT_RELU; p.c_dtype = GSAGE_F32;
    p.pool_n = n; p.pool_groups = BM / n; p.pool_mode = pool; p.pooled = pooled;
    p.pooled_ld = pooled_ld; p.argmax = argmax;
    dim3 grid((unsigned)ceil_div(M, p.pool_groups), (unsigned)ceil_div(H, BN), 1);
    if (dtype == GSAGE_BF16)
        hipLaunchKernelGGL((k_linear_nt<uint16_t, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((k_linear_nt<float, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("pool_mlp");
}

}  // extern "C"
