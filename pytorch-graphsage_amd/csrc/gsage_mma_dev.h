// gsage_mma_dev.h -- pieces shared by the MFMA GEMM kernels (gsage_linear.hip, gsage_packed.hip):
// tile geometry, the conflict-free LDS image of an operand tile, the bf16 / fp32 MFMA step and the
// fused activations.
#pragma once
#include "gsage_common.h"

namespace gsage {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2 };

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
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[e]), __uint_as_float(b[e]),
                                                       acc, 0, 0, 0);
    }
};

__device__ __forceinline__ float apply_act(float v, int act)
{
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == ACT_TANH) {
        // tanh(v) = sign(v) * (1 - 2 / (exp(2|v|) + 1)): full-precision expf, |err| ~ 1e-7
        const float e = expf(2.f * fabsf(v));
        const float t = 1.f - 2.f / (e + 1.f);
        return v < 0.f ? -t : t;
    }
    return v;
}

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void global_void_t;

}  // namespace gsage
