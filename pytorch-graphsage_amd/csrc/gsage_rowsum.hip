// gsage_rowsum.hip -- deterministic gradient of a trainable embedding table (include/gsage.h, "Deterministic
// gradient of a trainable table").
//
// Reference: nn_modules.py:131-155 under autograd -- `nn.Embedding`'s dense gradient is the sum of the gradient rows
// of every occurrence of a node in the frontier (index_add into a [n_nodes + 1, 64] tensor).  K6
// (gsage_scatter_add_rows) forms that sum with fp32 atomics, i.e. in an order that differs from run to run and, in a
// data-parallel run, from rank to rank: replicas that apply "the same" update would drift apart bit by bit.  Here
// the frontier's ids are sorted once (vendor radix sort over the bits a node id needs: a plain library primitive, as
// hipBLASLt would be for a plain GEMM) and every run of equal ids is summed IN LIST ORDER by one group of lanes and
// stored -- no atomics, no zero-fill, the same bits on every rank.  HBM-bound integer / byte work: 8 + 4 bytes per
// entry through the sort passes, each gradient row read once, each distinct row written once.
#include "gsage_common.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

namespace gsage {

// keys of the sort: a frontier's ids followed by n_tail entries of one spare row
struct KeyAt {
    const int64_t *ids0;
    int64_t n0, tail_id;
    __host__ __device__ int64_t operator()(int64_t i) const { return i < n0 ? ids0[i] : tail_id; }
};
typedef rocprim::transform_iterator<rocprim::counting_iterator<int64_t>, KeyAt, int64_t> KeyIter;

static hipError_t sort_pairs(void *temp, size_t &bytes, const KeyAt &k, int64_t n, int key_bits, int64_t *keys_out,
                             int32_t *vals_out, hipStream_t s)
{
    KeyIter keys_in(rocprim::counting_iterator<int64_t>(0), k);
    rocprim::counting_iterator<int32_t> vals_in(0);
    return rocprim::radix_sort_pairs(temp, bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u,
                                     (unsigned)key_bits, s, false);
}

// One group of E / 4 lanes (16 for 64-wide rows) per list entry; groups whose entry is not the first of its run
// leave at once.  A run of length 1 (the common case) is one 16-byte load and one 16-byte store per lane; longer
// runs are walked in list order, four rows in flight.
__global__ void __launch_bounds__(256)
k_segment_sum_rows(const int64_t *__restrict__ ids, const int32_t *__restrict__ pos, int64_t n,
                   const float *__restrict__ rows0, int64_t ld0, int64_t n0, const float *__restrict__ rows1,
                   int64_t ld1, int32_t lpr, int32_t chunks, float scale, float *__restrict__ table, int64_t ldt)
{
    typedef float v4 __attribute__((ext_vector_type(4)));
    const int64_t gpb = 256 / lpr;                                    // groups per workgroup
    const int64_t g = (int64_t)blockIdx.x * gpb + threadIdx.x / lpr;
    const int sub = threadIdx.x % lpr;
    if (g >= n || sub >= chunks) return;
    const int64_t id = ids[g];
    if (g > 0 && ids[g - 1] == id) return;
    auto row = [&](int64_t j) -> const float * {
        const int64_t p = pos[j];
        return p < n0 ? rows0 + p * ld0 : rows1 + (p - n0) * ld1;
    };
    v4 acc = *reinterpret_cast<const v4 *>(row(g) + sub * 4);
    int64_t j = g + 1;
    while (j < n && ids[j] == id) {
        // up to four more rows of the run requested together, added in list order
        v4 t[4];
        int m = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (j + u < n && ids[j + u] == id && m == u) { t[u] = *reinterpret_cast<const v4 *>(row(j + u) + sub * 4); m = u + 1; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (u < m) acc += t[u];
        j += m;
    }
    *reinterpret_cast<v4 *>(table + id * ldt + sub * 4) = acc * scale;
}

}  // namespace gsage

using namespace gsage;

extern "C" {

int64_t gsage_sort_rows_temp_bytes(int64_t n, int32_t key_bits)
{
    if (n <= 0 || key_bits <= 0 || key_bits > 63) return -1;
    size_t bytes = 0;
    KeyAt k{nullptr, 0, 0};
    if (sort_pairs(nullptr, bytes, k, n, key_bits, nullptr, nullptr, nullptr) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return (int64_t)((bytes + 255) / 256 * 256);
}

int gsage_sort_rows(const int64_t *ids0, int64_t n0, int64_t tail_id, int64_t n_tail, int32_t key_bits,
                    int64_t *ids_sorted, int32_t *pos_sorted, void *temp, int64_t temp_bytes, void *stream)
{
    const int64_t n = n0 + n_tail;
    GSAGE_REQUIRE(n0 >= 0 && n_tail >= 0 && n > 0 && n < ((int64_t)1 << 31), "sort_rows: bad sizes");
    GSAGE_REQUIRE((ids0 || n0 == 0) && ids_sorted && pos_sorted && temp, "sort_rows: null pointer");
    GSAGE_REQUIRE(key_bits > 0 && key_bits <= 63 && (tail_id >> key_bits) == 0, "sort_rows: ids must fit key_bits");
    GSAGE_REQUIRE(temp_bytes >= gsage_sort_rows_temp_bytes(n, key_bits), "sort_rows: temp storage too small");
    const KeyAt k{ids0, n0, tail_id};
    auto run = [=](hipStream_t s) -> int {
        size_t bytes = (size_t)temp_bytes;
        return sort_pairs(temp, bytes, k, n, key_bits, ids_sorted, pos_sorted, s) == hipSuccess ? 0 : 1;
    };
    if (t_recording) {
        // the vendor's launches are issued when the list is replayed (a host-call node)
        t_recording->target().emplace_back([run](hipStream_t s) {
            if (run(s) != 0) {
                (void)hipGetLastError();
                set_error("sort_rows: rocprim::radix_sort_pairs failed during a replay");
                t_node_error = 1;
            }
        });
        t_recording->n_marks += 1;
        return GSAGE_OK;
    }
    if (run((hipStream_t)stream) != 0) {
        set_error("sort_rows: rocprim::radix_sort_pairs: %s", hipGetErrorString(hipGetLastError()));
        return GSAGE_ELAUNCH;
    }
    g_launches.fetch_add(1, std::memory_order_relaxed);
    return GSAGE_OK;
}

int gsage_segment_sum_rows(const int64_t *ids_sorted, const int32_t *pos_sorted, int64_t n, const float *rows0,
                           int64_t ld0, int64_t n0, const float *rows1, int64_t ld1, int32_t E, float scale,
                           float *table, int64_t ldt, void *stream)
{
    GSAGE_REQUIRE(ids_sorted && pos_sorted && table && (rows0 || n0 == 0) && (rows1 || n0 >= n),
                  "segment_sum_rows: null pointer");
    GSAGE_REQUIRE(n > 0 && n0 >= 0 && n0 <= n, "segment_sum_rows: bad sizes");
    GSAGE_REQUIRE(E > 0 && E % 4 == 0 && E <= 256 && ld0 % 4 == 0 && ld1 % 4 == 0 && ldt % 4 == 0 && ldt >= E,
                  "segment_sum_rows: rows of whole 16-byte chunks, E <= 256");
    const int chunks = E / 4;
    int lpr = 1;
    while (lpr < chunks) lpr *= 2;
    const int64_t gpb = 256 / lpr;
    launch(k_segment_sum_rows, dim3((unsigned)ceil_div(n, gpb)), dim3(256), 0, (hipStream_t)stream, ids_sorted,
           pos_sorted, n, rows0, ld0, n0, rows1 ? rows1 : rows0, ld1, (int32_t)lpr, (int32_t)chunks, scale, table, ldt);
    return check_launch("segment_sum_rows");
}

}  // extern "C"
