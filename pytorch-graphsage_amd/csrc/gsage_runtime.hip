// gsage_runtime.hip -- library-wide host state of libgsage_hip.so: error text, launch counter,
// device probe, and the command lists (recorded kernel launches replayed with one call).
//
// The reference has no counterpart (it is eager PyTorch: one Python-dispatched op per tensor
// expression, models.py:71-104).  A train_step here is ~9 kernels of 5-40 us each, so how they are
// issued decides the step time as much as the kernels do: see gsage_common.h launch().
#include "gsage_common.h"
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <cstring>

#include <stdarg.h>
#include <string.h>
#include <memory>

namespace gsage {

static thread_local char t_err[512] = "";
std::atomic<uint64_t> g_launches{0};
thread_local CmdList *t_recording = nullptr;
thread_local hipStream_t t_last_stream = nullptr;
thread_local const int32_t *t_head_n_valid = nullptr;
thread_local const gsage_tail_gather_desc *t_gather_role = nullptr;
thread_local const gsage_hops_desc *t_hops_role = nullptr;
thread_local int t_node_error = 0;      // set by a host-call node that failed during a replay

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_err, sizeof(t_err), fmt, ap);
    va_end(ap);
}

}  // namespace gsage

using namespace gsage;

extern "C" {

int gsage_abi_version(void) { return GSAGE_ABI_VERSION; }

int gsage_head_n_valid_next(const int32_t *n_valid)
{
    t_head_n_valid = n_valid;
    return GSAGE_OK;
}
int gsage_gather_role_next(const gsage_tail_gather_desc *gather)
{
    t_gather_role = gather;
    return GSAGE_OK;
}
int gsage_hops_role_next(const gsage_hops_desc *hops)
{
    t_hops_role = hops;
    return GSAGE_OK;
}
const char *gsage_last_error(void) { return t_err; }
uint64_t gsage_launch_count(void) { return g_launches.load(); }

static struct sigaction g_prev_abrt;
static int g_abort_fd = 2;
static void on_abort(int sig, siginfo_t *info, void *uc)
{
    static const char head[] = "\n[gsage] SIGABRT -- native backtrace of the aborting thread:\n";
    (void)!write(g_abort_fd, head, sizeof(head) - 1);
    void *frames[48];
    const int n = backtrace(frames, 48);
    backtrace_symbols_fd(frames, n, g_abort_fd);
    if (g_prev_abrt.sa_flags & SA_SIGINFO) {
        if (g_prev_abrt.sa_sigaction) g_prev_abrt.sa_sigaction(sig, info, uc);
    } else if (g_prev_abrt.sa_handler != SIG_DFL && g_prev_abrt.sa_handler != SIG_IGN) {
        g_prev_abrt.sa_handler(sig);
    } else {
        signal(SIGABRT, SIG_DFL);
        raise(SIGABRT);
    }
}

int gsage_debug_abort_trace(int fd)
{
    static bool installed = false;
    g_abort_fd = fd >= 0 ? fd : 2;
    if (installed) return GSAGE_OK;
    void *warm[2];
    (void)backtrace(warm, 2);                  // (loads libgcc now, not inside the handler)
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = on_abort;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
    sigemptyset(&sa.sa_mask);
    GSAGE_REQUIRE(sigaction(SIGABRT, &sa, &g_prev_abrt) == 0, "debug_abort_trace: sigaction failed");
    installed = true;
    return GSAGE_OK;
}

int gsage_device_info(char *arch, int arch_len, int *cu_count, int *wave_size)
{
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
        (void)hipGetLastError();
        set_error("no HIP device visible");
        return GSAGE_ENODEV;
    }
    if (arch && arch_len > 0) {
        strncpy(arch, p.gcnArchName, (size_t)arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (wave_size) *wave_size = p.warpSize;
    return GSAGE_OK;
}

int gsage_cmdlist_begin(void)
{
    GSAGE_REQUIRE(!t_recording, "cmdlist_begin: this thread is already recording");
    t_recording = new CmdList();
    return GSAGE_OK;
}

int gsage_cmdlist_end(void **list)
{
    GSAGE_REQUIRE(t_recording, "cmdlist_end: no recording in progress on this thread");
    CmdList *l = t_recording;
    t_recording = nullptr;
    if (!list) {
        delete l;
        set_error("cmdlist_end: null output pointer");
        return GSAGE_EINVAL;
    }
    *list = l;
    return GSAGE_OK;
}

int64_t gsage_cmdlist_size(const void *list)
{
    const CmdList *l = (const CmdList *)list;
    return l ? l->n_launches : -1;
}

int gsage_cmdlist_replay(const void *list, void *stream)
{
    GSAGE_REQUIRE(list, "cmdlist_replay: null list");
    GSAGE_REQUIRE(!t_recording, "cmdlist_replay: cannot replay while recording");
    const CmdList *l = (const CmdList *)list;
    if (l->nodes.empty()) return GSAGE_OK;
    t_node_error = 0;
    l->last_stream = (hipStream_t)stream;
    l->replayed = true;
    for (const auto &node : l->nodes) node((hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("cmdlist_replay: %s", hipGetErrorString(e));
        return GSAGE_ELAUNCH;
    }
    if (t_node_error) {                  // (the node that failed has left its message in gsage_last_error)
        t_node_error = 0;
        return GSAGE_ELAUNCH;
    }
    g_launches.fetch_add((uint64_t)l->n_launches, std::memory_order_relaxed);
    return GSAGE_OK;
}

int gsage_cmdlist_side_begin(void)
{
    GSAGE_REQUIRE(t_recording, "cmdlist_side_begin: no recording in progress on this thread");
    CmdList *l = t_recording;
    GSAGE_REQUIRE(!l->side_open, "cmdlist_side_begin: a side section is already open");
    if (!l->side_stream) {
        hipError_t e = hipStreamCreateWithFlags(&l->side_stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&l->ev_fork, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&l->ev_join, hipEventDisableTiming);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            set_error("cmdlist_side_begin: %s", hipGetErrorString(e));
            return GSAGE_ELAUNCH;
        }
    }
    l->side_open = new std::vector<CmdList::Node>();
    return GSAGE_OK;
}

int gsage_cmdlist_side_end(void)
{
    GSAGE_REQUIRE(t_recording && t_recording->side_open, "cmdlist_side_end: no side section is open");
    CmdList *l = t_recording;
    std::shared_ptr<std::vector<CmdList::Node>> sub(l->side_open);
    l->side_open = nullptr;
    hipStream_t side = l->side_stream;
    hipEvent_t ef = l->ev_fork, ej = l->ev_join;
    // at replay: the side stream picks up from this point of the main stream, runs the section, marks its end
    l->nodes.emplace_back([sub, side, ef, ej](hipStream_t s) {
        (void)hipEventRecord(ef, s);
        (void)hipStreamWaitEvent(side, ef, 0);
        for (const auto &node : *sub) node(side);
        (void)hipEventRecord(ej, side);
    });
    l->n_marks += 1;
    return GSAGE_OK;
}

int gsage_cmdlist_join(void)
{
    GSAGE_REQUIRE(t_recording && !t_recording->side_open && t_recording->ev_join,
                  "cmdlist_join: needs a closed side section recorded before it");
    hipEvent_t ej = t_recording->ev_join;
    t_recording->nodes.emplace_back([ej](hipStream_t s) { (void)hipStreamWaitEvent(s, ej, 0); });
    t_recording->n_marks += 1;
    return GSAGE_OK;
}

int gsage_host_call(gsage_host_fn fn, void *ctx, void *stream)
{
    GSAGE_REQUIRE(fn, "host_call: null function");
    if (t_recording) {
        t_recording->target().emplace_back([fn, ctx](hipStream_t s) {
            if (fn(ctx, (void *)s) != 0) {
                if (!t_node_error) set_error("host call failed during a command-list replay");
                t_node_error = 1;
            }
        });
        t_recording->n_marks += 1;
        return GSAGE_OK;
    }
    if (fn(ctx, stream) != 0) {
        set_error("host_call: the callee reported a failure");
        return GSAGE_ELAUNCH;
    }
    return GSAGE_OK;
}

int gsage_cmdlist_mark(int slot)
{
    GSAGE_REQUIRE(t_recording, "cmdlist_mark: no recording in progress on this thread");
    GSAGE_REQUIRE(slot >= 0 && slot < CMDLIST_MARKS, "cmdlist_mark: slot must be in [0, %d)", CMDLIST_MARKS);
    CmdList *l = t_recording;
    if (!l->marks[slot] && hipEventCreate(&l->marks[slot]) != hipSuccess) {
        (void)hipGetLastError();
        l->marks[slot] = nullptr;
        set_error("cmdlist_mark: hipEventCreate failed");
        return GSAGE_ELAUNCH;
    }
    hipEvent_t ev = l->marks[slot];
    l->nodes.emplace_back([ev](hipStream_t s) { (void)hipEventRecord(ev, s); });
    l->n_marks += 1;
    return GSAGE_OK;
}

int gsage_cmdlist_time_next(int slot_a, int slot_b)
{
    GSAGE_REQUIRE(t_recording, "cmdlist_time_next: no recording in progress on this thread");
    GSAGE_REQUIRE(slot_a >= 0 && slot_a < CMDLIST_MARKS && slot_b >= 0 && slot_b < CMDLIST_MARKS && slot_a != slot_b,
                  "cmdlist_time_next: two different slots in [0, %d)", CMDLIST_MARKS);
    CmdList *l = t_recording;
    for (int slot : {slot_a, slot_b})
        if (!l->marks[slot] && hipEventCreate(&l->marks[slot]) != hipSuccess) {
            (void)hipGetLastError();
            l->marks[slot] = nullptr;
            set_error("cmdlist_time_next: hipEventCreate failed");
            return GSAGE_ELAUNCH;
        }
    l->time_a = slot_a;
    l->time_b = slot_b;
    return GSAGE_OK;
}

int gsage_cmdlist_elapsed(const void *list, int slot_a, int slot_b, float *ms)
{
    GSAGE_REQUIRE(list && ms, "cmdlist_elapsed: null pointer");
    GSAGE_REQUIRE(slot_a >= 0 && slot_a < CMDLIST_MARKS && slot_b >= 0 && slot_b < CMDLIST_MARKS,
                  "cmdlist_elapsed: bad slot");
    const CmdList *l = (const CmdList *)list;
    GSAGE_REQUIRE(l->marks[slot_a] && l->marks[slot_b], "cmdlist_elapsed: mark not recorded in this list");
    hipError_t e = hipEventSynchronize(l->marks[slot_b]);
    if (e == hipSuccess) e = hipEventElapsedTime(ms, l->marks[slot_a], l->marks[slot_b]);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("cmdlist_elapsed: %s", hipGetErrorString(e));
        return GSAGE_ELAUNCH;
    }
    return GSAGE_OK;
}

void gsage_cmdlist_destroy(void *list)
{
    delete (CmdList *)list;
}

int gsage_stream_create_masked(const uint32_t *cu_mask, int32_t words, void **stream)
{
    GSAGE_REQUIRE(cu_mask && words > 0 && stream, "stream_create_masked: bad arguments");
    hipStream_t s = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, cu_mask);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("stream_create_masked: %s", hipGetErrorString(e));
        return GSAGE_ELAUNCH;
    }
    *stream = (void *)s;
    return GSAGE_OK;
}

int gsage_event_create(void **event)
{
    GSAGE_REQUIRE(event, "event_create: null pointer");
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        set_error("event_create failed");
        return GSAGE_ELAUNCH;
    }
    *event = (void *)e;
    return GSAGE_OK;
}

void gsage_event_destroy(void *event)
{
    if (event) (void)hipEventDestroy((hipEvent_t)event);
}

int gsage_cmdlist_replay_pair(const void *list_a, void *stream_a, void *wait_a, void *record_a,
                              const void *list_b, void *stream_b, void *wait_b, void *record_b,
                              void *then_wait_stream, int then_wait_on_b)
{
    GSAGE_REQUIRE(!t_recording, "cmdlist_replay_pair: cannot replay while recording");
    GSAGE_REQUIRE(stream_a && stream_b && stream_a != stream_b, "cmdlist_replay_pair: needs two different streams");
    hipError_t e = hipSuccess;
    int rc = GSAGE_OK;
    // stream b first: it carries the longer, latency-bound chain
    if (wait_b) e = hipStreamWaitEvent((hipStream_t)stream_b, (hipEvent_t)wait_b, 0);
    if (e == hipSuccess && list_b) rc = gsage_cmdlist_replay(list_b, stream_b);
    if (rc != GSAGE_OK) return rc;
    if (e == hipSuccess && record_b) e = hipEventRecord((hipEvent_t)record_b, (hipStream_t)stream_b);
    if (e == hipSuccess && wait_a) e = hipStreamWaitEvent((hipStream_t)stream_a, (hipEvent_t)wait_a, 0);
    if (e == hipSuccess && list_a) rc = gsage_cmdlist_replay(list_a, stream_a);
    if (rc != GSAGE_OK) return rc;
    if (e == hipSuccess && record_a) e = hipEventRecord((hipEvent_t)record_a, (hipStream_t)stream_a);
    if (e == hipSuccess && then_wait_stream) {
        hipEvent_t ev = (hipEvent_t)(then_wait_on_b ? record_b : record_a);
        if (ev) e = hipStreamWaitEvent((hipStream_t)then_wait_stream, ev, 0);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("cmdlist_replay_pair: %s", hipGetErrorString(e));
        return GSAGE_ELAUNCH;
    }
    return GSAGE_OK;
}

int gsage_stream_destroy(void *stream)
{
    if (stream && hipStreamDestroy((hipStream_t)stream) != hipSuccess) {
        (void)hipGetLastError();
        set_error("stream_destroy failed");
        return GSAGE_ELAUNCH;
    }
    return GSAGE_OK;
}

}  // extern "C"
