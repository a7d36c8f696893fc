// gsage_common.h -- shared device/host helpers for the gfx950 GraphSAGE kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <functional>
#include <tuple>
#include <vector>

#include "../../include/gsage.h"

namespace gsage {

// ---- error plumbing -------------------------------------------------------------------------
void set_error(const char *fmt, ...);
extern std::atomic<uint64_t> g_launches;

#define GSAGE_REQUIRE(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            ::gsage::set_error(__VA_ARGS__);     \
            return GSAGE_EINVAL;                 \
        }                                        \
    } while (0)

// GSAGE_DEBUG_SYNC=1: wait for every launch and report the kernel that failed; =2: also name every launch on
// stderr BEFORE waiting for it (a memory fault kills the process inside the wait: the last name is the culprit).
// Debugging aid only -- launches that are being recorded into a command list, or captured into a hipGraph, are
// not affected (the stream of the thread's last launch() is asked whether it is capturing).
inline int debug_sync_level()
{
    static const int level = [] { const char *e = getenv("GSAGE_DEBUG_SYNC"); return e ? atoi(e) : 0; }();
    return level;
}

extern thread_local struct CmdList *t_recording;
extern thread_local hipStream_t t_last_stream;      // stream of this thread's last launch() (debug sync only)

inline int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && debug_sync_level() > 0 && !t_recording) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(t_last_stream, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusActive; }
        if (cap == hipStreamCaptureStatusNone) {
            if (debug_sync_level() > 1) { fprintf(stderr, "[gsage] %s\n", what); fflush(stderr); }
            e = hipStreamSynchronize(t_last_stream);
        }
    }
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return GSAGE_ELAUNCH;
    }
    g_launches.fetch_add(1, std::memory_order_relaxed);
    return GSAGE_OK;
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- command lists ------------------------------------------------------------------------------
// Every kernel of this library is launched through launch().  While a command list is being
// recorded on the calling thread (gsage_cmdlist_begin .. gsage_cmdlist_end) the launch is not
// issued: the kernel, its geometry and a by-value copy of its arguments (already converted to the
// kernel's parameter types) become a node of the list, and gsage_cmdlist_replay() later issues the
// nodes back to back on a stream of the caller's choice.  Compared with a hipGraph of the same
// kernels a replay costs the host one hipLaunchKernel per node (~2-3 us) instead of ~10-16 us per
// graph launch PLUS an 8-15 us start-up gap on the device at every graph boundary, which matters
// when a step has to be cut into several pieces around a collective (engine.py, data-parallel).
constexpr int CMDLIST_MARKS = 16;
struct CmdList {
    typedef std::function<void(hipStream_t)> Node;
    std::vector<Node> nodes;
    hipEvent_t marks[CMDLIST_MARKS] = {};      // gsage_cmdlist_mark: events recorded between kernels
    int64_t n_marks = 0;                       // mark nodes in `nodes` (not counted as kernel launches)
    int64_t n_launches = 0;                    // kernel launches recorded (main stream + side sections)
    int time_a = -1, time_b = -1;              // gsage_cmdlist_time_next: events for the next recorded kernel
    // side sections (gsage_cmdlist_side_begin / _end / _join): launches that replay on a second stream of the
    // list's own, forked from the main stream where the section was recorded and joined where asked
    std::vector<Node> *side_open = nullptr;    // != null while a side section is being recorded
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::vector<Node> &target() { return side_open ? *side_open : nodes; }
    // stream of the last replay: the destructor waits for it (and for the side stream) before it lets go of the
    // list's events and side stream -- a list may be dropped by Python's cyclic collector at ANY allocation, e.g.
    // in the middle of another engine's warm-up, while its last replay is still queued on the device
    mutable hipStream_t last_stream = nullptr;
    mutable bool replayed = false;
    ~CmdList()
    {
        if (replayed) {
            if (hipStreamSynchronize(last_stream) != hipSuccess) (void)hipGetLastError();
            if (side_stream && hipStreamSynchronize(side_stream) != hipSuccess) (void)hipGetLastError();
        }
        for (int i = 0; i < CMDLIST_MARKS; ++i)
            if (marks[i]) (void)hipEventDestroy(marks[i]);
        delete side_open;
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (side_stream) (void)hipStreamDestroy(side_stream);
    }
};
// set (to 1) by a node that is not a kernel launch and failed during a replay: gsage_cmdlist_replay then fails
extern thread_local int t_node_error;
// gsage_head_n_valid_next(): live-row count(s) for the NEXT head launch of this thread (consumed by it)
extern thread_local const int32_t *t_head_n_valid;
// gsage_gather_role_next(): gather-role descriptor for the NEXT gsage_linear_nt_packed launch of this thread
extern thread_local const gsage_tail_gather_desc *t_gather_role;
// gsage_hops_role_next(): sampler descriptor for the NEXT gsage_linear_nt_packed launch of this thread
extern thread_local const gsage_hops_desc *t_hops_role;
inline const gsage_tail_gather_desc *take_gather_role()
{
    const gsage_tail_gather_desc *p = t_gather_role;
    t_gather_role = nullptr;
    return p;
}
inline const int32_t *take_head_n_valid()
{
    const int32_t *p = t_head_n_valid;
    t_head_n_valid = nullptr;
    return p;
}

template <typename... KArgs, typename... Args>
inline void launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t stream,
                   Args... args)
{
    static_assert(sizeof...(KArgs) == sizeof...(Args), "launch: argument count mismatch");
    if (t_recording) {
        std::tuple<KArgs...> packed(static_cast<KArgs>(args)...);
        if (t_recording->time_a >= 0) {
            // gsage_cmdlist_time_next: this kernel's dispatch carries a start and a stop event (the
            // timestamps of the dispatch itself, what a kernel trace reports), nothing else changes
            hipEvent_t ea = t_recording->marks[t_recording->time_a], eb = t_recording->marks[t_recording->time_b];
            t_recording->time_a = t_recording->time_b = -1;
            t_recording->n_launches += 1;
            t_recording->target().emplace_back([kernel, grid, block, lds, packed, ea, eb](hipStream_t s) {
                std::apply([&](const KArgs &...a) {
                    hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, s, ea, eb, 0u, a...);
                }, packed);
            });
            return;
        }
        t_recording->n_launches += 1;
        t_recording->target().emplace_back([kernel, grid, block, lds, packed](hipStream_t s) {
            std::apply([&](const KArgs &...a) { hipLaunchKernelGGL(kernel, grid, block, lds, s, a...); },
                       packed);
        });
        return;
    }
    t_last_stream = stream;
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, static_cast<KArgs>(args)...);
}

// ---- bf16 <-> fp32 (bf16 = upper half of an IEEE fp32, round-to-nearest-even) ------------------
__device__ __forceinline__ float bf16_to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

__device__ __forceinline__ uint16_t f32_to_bf16(float f)
{
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);   // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi)
{
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. waits
// for every outstanding GLOBAL store to be acknowledged (1-2 us each time on gfx950) -- wasted when
// the stores are results no wave of this launch reads back.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// 16-byte vector of raw bits; the unit every kernel moves per lane.  A first-class vector type
// (not a struct with an array member): hipcc keeps these in VGPRs, whereas the struct form was
// observed to live in scratch memory when selected/zeroed conditionally.
typedef uint32_t vec16 __attribute__((ext_vector_type(4)));

// ---- Philox4x32-10 (Salmon et al., SC'11) -----------------------------------------------------
struct philox4 { uint32_t v[4]; };

__host__ __device__ __forceinline__ philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2,
                                                          uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    philox4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}

}  // namespace gsage
