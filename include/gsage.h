/*
 * gsage.h -- C ABI of libgsage_hip.so, the MI355X (gfx950) GraphSAGE hot path.
 *
 * This is the drop-in boundary.  The reference (bkj/pytorch-graphsage) is pure Python with no
 * FFI of its own; its plugin surface is three string->class tables
 * (nn_modules.py:104-107, :169-173, :324-330) plus GSSupervised (models.py:21-104).  The
 * entry points below are what a ctypes binding inside those classes would call -- one per
 * tensor-level operation of the hot path -- and INTEGRATION.md shows that binding.  Each
 * declaration cites the reference lines it replaces.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / C++ types.
 *   - Unless marked [host], every pointer is a DEVICE pointer (HBM) owned by the caller;
 *     nothing is allocated or freed behind the caller's back, outputs are caller-allocated.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Every call only
 *     enqueues work on that stream and returns; no call synchronises.  All kernels are
 *     hipGraph-capturable (no allocation, no sync, no host-side state besides launch counters).
 *   - Return value: 0 on success, a negative GSAGE_E* code otherwise; gsage_last_error()
 *     returns a thread-local message.  Asynchronous data errors (an id outside the graph) are
 *     reported through the caller-supplied `err_flag` device word (0 = ok).
 *   - dtype codes: GSAGE_F32 / GSAGE_BF16 (bf16 = upper 16 bits of an IEEE fp32, RNE).
 *   - Graph layout ("device CSR", built once from the reference's scipy csr_matrix in the
 *     (v,r,c) convention of problem.py:70-72 / utils/convert.py:100-126):
 *         rowptr int64 [n_rows+1]   rowptr[i+1]-rowptr[i] = degree of node i
 *         col    int32 [nnz]        neighbour ids (1-based; 0 is the dummy node), stored in
 *                                   column order 0..deg-1 of the reference matrix
 *   - Feature-table layout: row-major [n_rows, ld] with ld >= D; vectorised paths need
 *     ld % 8 == 0 (bf16) / ld % 4 == 0 (fp32) and columns [D, ld) equal to zero.
 */
#ifndef GSAGE_H
#define GSAGE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSAGE_ABI_VERSION 6

enum { GSAGE_F32 = 0, GSAGE_BF16 = 1 };
enum {
    GSAGE_OK = 0,
    GSAGE_EINVAL = -1,   /* bad argument (null pointer, bad size / alignment / dtype) */
    GSAGE_ELAUNCH = -2,  /* hipLaunchKernel / runtime error (message in gsage_last_error) */
    GSAGE_ENODEV = -3    /* no gfx950 device visible */
};
enum { GSAGE_POOL_MAX = 0, GSAGE_POOL_MEAN = 1 };
enum { GSAGE_ACT_NONE = 0, GSAGE_ACT_RELU = 1, GSAGE_ACT_TANH = 2 };

int gsage_abi_version(void);
const char *gsage_last_error(void);
/* Number of kernels this library has launched since load (tests use it to prove the HIP path
 * ran instead of some fallback). */
uint64_t gsage_launch_count(void);
/* [host] debugging aid: from now on a SIGABRT first writes the native backtrace of the aborting thread to the
 * file descriptor `fd` (< 0: stderr), then runs whatever handler was installed before.  The HSA runtime reports a
 * GPU memory fault by calling abort() on a thread of its own, with a message that a test runner capturing fd 2
 * swallows: the backtrace says who aborted (tests/conftest.py installs it for -m gpu sessions). */
int gsage_debug_abort_trace(int fd);
/* [host] device name / CU count of the current device; returns GSAGE_ENODEV without a GPU. */
int gsage_device_info(char *arch, int arch_len, int *cu_count, int *wave_size);

/* ------------------------------------------------------------------------------------------
 * Command lists.  No reference counterpart: the reference dispatches one PyTorch op per tensor
 * expression from Python (models.py:71-104); here a train_step is ~9 kernels of 5-40 us and the
 * way they are issued matters as much as the kernels.
 *
 *   gsage_cmdlist_begin()            [host] every gsage_* kernel call made by THIS thread from now
 *                                    on is validated and recorded (kernel, geometry, by-value
 *                                    arguments) instead of launched; `stream` arguments are ignored.
 *   gsage_cmdlist_end(&list)         [host] stops recording, returns the list.
 *   gsage_cmdlist_replay(list, s)    [host] issues the recorded kernels back to back on stream s
 *                                    (one hipLaunchKernel each, no validation, no Python).  The
 *                                    pointers recorded must still be valid: same rule as a hipGraph.
 *   gsage_cmdlist_size / _destroy
 * Unlike a hipGraph a list has no start-up gap on the device and can be cut anywhere (e.g.
 * around an RCCL collective on another stream) at no cost.  Host-side calls (gsage_mt_*) and the
 * one-time LDS-limit setup of a kernel are not recordable: run one un-recorded step first.
 * ---------------------------------------------------------------------------------------- */
/*   gsage_cmdlist_mark(slot)        [host] while recording: the list records HIP event `slot` (0..15) at
 *                                    this point of every replay (on the replay's stream).
 *   gsage_cmdlist_elapsed(l,a,b,&ms) [host] waits for mark b of the last replay and returns the time
 *                                    between marks a and b: how bench.py times ONE kernel of a step in
 *                                    place, on the stream it runs on (measurement only: a list with
 *                                    marks pays an event record per mark). */
/*   gsage_stream_create_masked      [host] a HIP stream whose kernels run on the compute units set in
 *                                    cu_mask only (bit n = CU n of the device, `words` x 32 bits): how
 *                                    the engines give the weight-independent, HBM-bound gathers of the
 *                                    NEXT batch and the latency-bound forward / backward chain of the
 *                                    current batch disjoint halves of the chip, so that the two really
 *                                    run side by side.  gsage_stream_destroy frees it. */
int gsage_stream_create_masked(const uint32_t *cu_mask, int32_t words, void **stream);
int gsage_stream_destroy(void *stream);
/*   gsage_event_create / _destroy   [host] a HIP event without timing, for ordering between streams.
 *   gsage_cmdlist_replay_pair       [host] one step of a two-stream pipeline in one call: on stream b
 *                                    {wait for event wait_b; replay list_b; record record_b}, then on
 *                                    stream a {wait for wait_a; replay list_a; record record_a}; finally
 *                                    then_wait_stream (may be NULL) is made to wait for record_b
 *                                    (then_wait_on_b != 0) or record_a.  Events / lists may be NULL. */
int gsage_event_create(void **event);
void gsage_event_destroy(void *event);
int gsage_cmdlist_replay_pair(const void *list_a, void *stream_a, void *wait_a, void *record_a,
                              const void *list_b, void *stream_b, void *wait_b, void *record_b,
                              void *then_wait_stream, int then_wait_on_b);
int gsage_cmdlist_begin(void);
int gsage_cmdlist_end(void **list);
int gsage_cmdlist_mark(int slot);
/*   gsage_cmdlist_time_next(a, b)    [host] while recording: the NEXT recorded kernel is dispatched with
 *                                    start event a and stop event b attached to the dispatch itself
 *                                    (hipExtLaunchKernel): gsage_cmdlist_elapsed(l, a, b) is then that
 *                                    kernel's own duration -- no event packets before or after it. */
int gsage_cmdlist_time_next(int slot_a, int slot_b);
int gsage_cmdlist_elapsed(const void *list, int slot_a, int slot_b, float *ms);
int64_t gsage_cmdlist_size(const void *list);
int gsage_cmdlist_replay(const void *list, void *stream);
/* Side sections: launches recorded between _side_begin and _side_end replay on a second stream that belongs to
 * the list, forked from the main stream at the point of the section (the side stream waits for everything the
 * main stream was given before it); _join makes the main stream wait for the end of the last side section.
 * What it is for: a bandwidth-bound job that does not depend on the neighbouring launches (the NEXT batch's
 * level-0 gathers) beside a latency-bound launch whose workgroups leave registers and the memory pipes of
 * their CUs idle (the seed-level kernel: one 384-register wave per SIMD) -- two kernels can share a CU where one
 * kernel's uniform LDS / register footprint cannot.  One section open at a time; the main-stream launches that
 * follow _side_end and precede _join run concurrently with the section. */
int gsage_cmdlist_side_begin(void);
int gsage_cmdlist_side_end(void);
int gsage_cmdlist_join(void);
void gsage_cmdlist_destroy(void *list);
/* Host calls inside a list (ABI 4).  A data-parallel step has ONE exchange and a deterministic
 * row reduction (a vendor sort until round 6): what is not a kernel of this library still belongs INTO the step's list so that a
 * step stays one C call (no Python between the pieces, no second list around the collective).
 *   gsage_host_call(fn, ctx, s)      [host] recording: a node that calls fn(ctx, stream) at this point of every replay
 *                                    (inside a side section: with the side stream); not recording: calls fn(ctx, s)
 *                                    now.  fn enqueues work on the stream it is given and returns 0, or non-zero to
 *                                    make the replay fail (GSAGE_ELAUNCH, message "host call failed").  The engines
 *                                    use it for collectives of backends that have no C entry point (gloo in the
 *                                    CPU-side tests: a ctypes callback); RCCL goes through gsage_comm_* below. */
typedef int (*gsage_host_fn)(void *ctx, void *stream);
int gsage_host_call(gsage_host_fn fn, void *ctx, void *stream);

/* ------------------------------------------------------------------------------------------
 * The step's collectives, issued by the library itself (ABI 4).  No reference counterpart: the reference is one
 * process.  One RCCL communicator per process, created from an id that rank 0 makes and the host
 * language distributes (torch.distributed's store, MPI, a file -- not this library's business):
 *   gsage_comm_load(path)            [host] dlopen of librccl (path NULL: the loader's search order).  The library has
 *                                    no link-time dependency on RCCL: single-GPU processes never load it.
 *   gsage_comm_unique_id(id)         [host] 128 bytes, rank 0 only
 *   gsage_comm_create(id, r, w, &c)  [host] collective over the w ranks (current HIP device = this rank's GPU)
 *   gsage_comm_all_reduce_f32        in place over n floats; average != 0: ncclAvg, else ncclSum
 *   gsage_comm_all_gather            recv[r * bytes : (r + 1) * bytes] = rank r's send[0:bytes]  (bytes % 4 == 0)
 *   gsage_comm_group(begin)          ncclGroupStart (begin != 0) / ncclGroupEnd: several collectives as one
 * The three collective calls follow the library's convention: recorded as a node while a command list is being
 * recorded (side sections included: that is how the exchange overlaps the next batch's gathers), issued on `stream`
 * otherwise.  Every rank must issue the same sequence. */
int gsage_comm_load(const char *path);
int gsage_comm_unique_id(void *id128);
int gsage_comm_create(const void *id128, int32_t rank, int32_t world, void **comm);
int gsage_comm_destroy(void *comm);
int gsage_comm_all_reduce_f32(void *comm, float *buf, int64_t n, int32_t average, void *stream);
int gsage_comm_all_gather(void *comm, const void *send, void *recv, int64_t bytes, void *stream);
int gsage_comm_group(void *comm, int32_t begin, void *stream);

/* ------------------------------------------------------------------------------------------
 * K1  neighbour sampler     replaces SparseUniformNeighborSampler.__call__, nn_modules.py:80-101
 *     (and __init__, :72-78: degrees are rowptr differences, nothing is precomputed)
 *
 *     out[i*n+j] = col[rowptr[ids[i]] + sel[i*n+j] % deg_i]   (deg_i > 0)      nn_modules.py:89-93
 *                = 0 (the dummy node)                          (deg_i == 0)
 * ---------------------------------------------------------------------------------------- */

/* `sel` supplied by the caller (int32 [M*n], each in [0, max_deg)): parity level 1, and the
 * device half of compat mode where sel comes from the legacy MT19937 stream (gsage_mt_*). */
int gsage_sample_csr_sel(const int64_t *rowptr, const int32_t *col, int64_t n_rows,
                         const int64_t *ids, int64_t M, int32_t n, const int32_t *sel,
                         int64_t *out, int32_t *err_flag, void *stream);

/* UniformNeighborSampler.__call__ (reference nn_modules.py:43-49): `adj[ids][:, perm][:, :n]` over the dense
 * int64 [n_rows, ld] adjacency (every row pre-sampled to exactly ld neighbours by the converter), in one launch:
 *     out[i*n + j] = adj[ids[i]*ld + keep[j]]
 * keep: DEVICE int64 [n] = the first n entries of the torch.randperm(ld) the caller drew from torch's global
 * CPU generator (one permutation per call, shared by the whole batch).  An id outside
 * [0, n_rows) or a column outside [0, ld) yields 0 and raises *err_flag (the reference: IndexError). */
int gsage_sample_dense(const int64_t *adj, int64_t ld, int64_t n_rows, const int64_t *ids, int64_t M,
                       const int64_t *keep, int32_t n, int64_t *out, int32_t *err_flag, void *stream);

/* Counter mode (throughput): sel is generated in-kernel by Philox4x32-10,
 *     g    = g0 + i*n + j                      global sample index of the whole job
 *     call = call_base + (call_ctr ? *call_ctr : 0)        (call_ctr: device word, may be NULL;
 *            lets a captured hipGraph advance the stream between replays)
 *     word = philox4x32_10({lo(g>>2), hi(g>>2), lo(call), hi(call)}, {lo(seed), hi(seed)})[g&3]
 *     sel  = (word * max_deg) >> 32
 * `sel_out` (int32 [M*n], may be NULL) receives sel for verification. */
int gsage_sample_csr_philox(const int64_t *rowptr, const int32_t *col, int64_t n_rows,
                            const int64_t *ids, int64_t M, int32_t n, uint32_t max_deg,
                            uint64_t seed, const uint64_t *call_ctr, uint64_t call_base,
                            uint64_t g0, int64_t *out, int32_t *sel_out, int32_t *err_flag,
                            void *stream);

/* Every hop of a frontier in one launch (counter mode).  `ids` is the concatenated frontier
 * [hop 0 (B seeds, filled by the caller) | hop 1 (B*fan[0]) | hop 2 (B*fan[0]*fan[1]) | ...];
 * hop k is sampled exactly as gsage_sample_csr_philox would with call index call_base + k - 1 and
 * g0 = rank * |hop k| -- results are bit-identical to n_hops separate launches.  fan: HOST array
 * of n_hops (<= 5) fan-outs.  seed_queue (may be NULL): device-resident [n_batches, B] seed
 * batches; the batch *batch_idx % n_batches is copied into hop 0 by the kernel itself, so a
 * captured graph can walk an epoch without per-step host copies. */
int gsage_sample_hops_philox(const int64_t *rowptr, const int32_t *col, int64_t n_rows, int64_t *ids,
                             int64_t B, int32_t n_hops, const int32_t *fan, uint32_t max_deg,
                             uint64_t seed, const uint64_t *call_ctr, uint64_t call_base,
                             uint64_t rank, const int64_t *seed_queue, const int64_t *batch_idx,
                             int64_t n_batches, int32_t *err_flag, void *stream);
/* The same arguments as a struct (HOST memory).  batch_base is added to *batch_idx before the modulo:
 * with call_base it addresses a batch AHEAD of the device counters without touching them -- how a
 * later batch's frontier is sampled from inside another launch (gsage_gather_mean_multi_adam). */
typedef struct gsage_hops_desc {
    const int64_t *rowptr;
    const int32_t *col;
    int64_t n_rows;
    int64_t *ids;
    int64_t B;
    int32_t n_hops;
    int32_t fan[5];
    uint32_t max_deg;
    uint64_t seed;
    const uint64_t *call_ctr;
    uint64_t call_base;
    uint64_t rank;
    const int64_t *seed_queue;
    const int64_t *batch_idx;
    int64_t batch_base;
    int64_t n_batches;
    int32_t *err_flag;
    /* sel (may be NULL): caller-supplied draws instead of Philox -- int32, this rank's samples of
     * [hop 1 (B*fan[0]) | hop 2 | ...] back to back, each in [0, max_deg): the fused sampler then
     * computes exactly gsage_sample_csr_sel per hop (parity level 1: replays the `sel` the reference
     * drew at nn_modules.py:88).  With a seed queue the draws of batch b start at sel + b*sel_stride. */
    const int32_t *sel;
    int64_t sel_stride;
    /* dense_adj (may be NULL; ABI 3): the frontier of the reference's DENSE sampler (UniformNeighborSampler,
     * nn_modules.py:19-49) instead of the CSR walk -- int64 [n_rows, dense_ld], every row pre-sampled to exactly
     * dense_ld neighbours.  `sel` is then mandatory and holds, per batch, the columns the sampler keeps:
     * [fan[0] columns for hop 1 | fan[1] columns for hop 2 | ...] (sel_stride = their sum), i.e. the head of the
     * torch.randperm(K) the reference draws ONCE per sampler call and shares between all parents:
     *     ids[hop k][i*fan + j] = dense_adj[ids[hop k-1][i], keep_k[j]].
     * rowptr / col / max_deg / seed / call_* are ignored. */
    const int64_t *dense_adj;
    int64_t dense_ld;
} gsage_hops_desc;
/* gsage_sample_hops_philox from a descriptor (the only way to pass batch_base). */
int gsage_sample_hops(const gsage_hops_desc *hops, void *stream);

/* *ctr += inc on the stream (advances the Philox call counter inside a captured graph). */
int gsage_counter_add(uint64_t *ctr, uint64_t inc, void *stream);
/* Two small device-to-device copies in one launch (4-byte granularity): a step's seed ids and targets into the
 * static buffers a recorded / captured step reads (train.py:141-148 hands over fresh tensors per batch). */
int gsage_copy_pair(void *dst0, const void *src0, int64_t bytes0, void *dst1, const void *src1, int64_t bytes1,
                    void *stream);

/* [host] the numpy legacy MT19937 stream the reference draws from (helpers.py:15,
 * nn_modules.py:88, problem.py:146), for compat mode.  All pointers here are HOST pointers. */
void *gsage_mt_create(uint32_t seed);
void gsage_mt_destroy(void *mt);
void gsage_mt_seed(void *mt, uint32_t seed);
/* np.random.choice(high, count) -> int32 out[count]; returns 32-bit words consumed. */
int64_t gsage_mt_choice_i32(void *mt, int64_t high, int64_t count, int32_t *out);
/* The same np.random.choice(high, count) with the stream ON THE DEVICE: `state` = 625 device words (the
 * 624 MT19937 state words + the position, numpy's np.random.get_state()[1], [2]); the launch consumes
 * exactly the words numpy would, leaves state / position where numpy would, and writes int32 out[count]
 * (device).  One workgroup (the recurrence is sequential; refill, tempering and the order-preserving
 * rejection are parallel inside it).  Not recordable into a graph-free command list restriction: it IS an
 * ordinary kernel launch and can be recorded like any other. */
int gsage_mt_choice_device(uint32_t *state, int64_t high, int64_t count, int32_t *out, void *stream);
/* n_seg requests served back to back from the same device-resident stream in ONE launch: request q writes
 * seg_cnt[q] values to out + seg_off[q] (seg_off / seg_cnt: DEVICE int64 arrays).  Consecutive
 * np.random.choice(high, .) calls consume numpy's stream exactly like this, so a whole training epoch's sampler
 * draws (per batch: hop 1's M * n_1 values, hop 2's M * n_1 * n_2, ... -- nn_modules.py:88 called from
 * models.py:78-80 batch after batch) are produced up front and the fused engines' device-side batch queue replays
 * them (gsage_hops_desc.sel): samples bit-identical to the reference's for the same seed.  high >= 2 (numpy draws
 * nothing for a range of one value). */
int gsage_mt_choice_segments(uint32_t *state, int64_t high, int64_t n_seg, const int64_t *seg_off,
                             const int64_t *seg_cnt, int32_t *out, void *stream);

/* The same requests served by MANY workgroups (ABI 5; csrc/gsage_mtjump.hip): MT19937 is linear over GF(2), so the
 * state n refills ahead is a convolution of a seed-independent polynomial's bits with the stream's raw words -- a
 * workgroup jumps to its chunk with two table look-ups, pass A counts every chunk's accepted words, a scan turns the
 * counts into offsets, pass B regenerates the chunks and stores value #n of the request at its slot; state and
 * position are left exactly where numpy would leave them (same contract as gsage_mt_choice_segments).
 *   seg_cum [n_seg + 1] (DEVICE): prefix sums of the requests' counts; seg_off [n_seg] (DEVICE); n_total = seg_cum[n_seg]
 *   table (DEVICE): gsage_mt_jump_table()'s 128 x 312 words;  scratch: gsage_mt_choice_par_scratch(n_wg) bytes, 16-byte aligned
 *   n_wg chunks of units_per_wg x 64 refills each (n_wg x units_per_wg <= 4096): choose them to cover
 *   n_total / (624 x acceptance rate) refills with a margin; a serial finisher serves what an unlucky stretch left. */
int gsage_mt_choice_par(uint32_t *state, int64_t high, int64_t n_seg, const int64_t *seg_cum, const int64_t *seg_off,
                        int64_t n_total, int32_t *out, const uint64_t *table, void *scratch, int64_t scratch_bytes,
                        int32_t n_wg, int32_t units_per_wg, void *stream);
int64_t gsage_mt_choice_par_scratch(int32_t n_wg);
/* HOST: the jump polynomials x^(b * 64 * 624) and x^(a * 64 * 64 * 624) mod phi(x), a, b < 64 (phi: MT19937's
 * characteristic polynomial, found by Berlekamp-Massey on the generator's own output bits; ~1 s the first time in a
 * process, callers cache the words).  out: gsage_mt_jump_table_words() x 8 bytes: [b = 0..63 | a = 0..63][312]. */
int gsage_mt_jump_table(uint64_t *out, int64_t words);
int64_t gsage_mt_jump_table_words(void);
/* HOST (tests): out[624] = the state `poly` steps ahead of state[624] (word 0's low 31 bits are not part of the state) */
int gsage_mt_jump_host(const uint32_t *state, const uint64_t *poly, uint32_t *out);
/* np.random.permutation(n) -> int64 out[n]. */
void gsage_mt_permutation(void *mt, int64_t n, int64_t *out);

/* ------------------------------------------------------------------------------------------
 * K2  gather + mean         replaces feats[ids] (models.py:76,80) followed by
 *                           neibs.view(M,-1,D).mean(dim=1) (nn_modules.py:197-198)
 *
 *     out[i, c] = (1/n) * sum_j table[rows(i,j), c],   rows(i,j) = ids ? ids[i*n+j] : i*n+j
 *     n == 1 is the plain row gather feats[ids].  fp32 accumulation; `out` has `out_dtype`.
 *     Columns [D, round_up(D, vec)) of out are written as zero.
 *     PRECONDITION: 0 <= ids[.] < 2^31 (the kernels read the low 32-bit word of each int64 id: node ids of a table
 *     with < 2^31 rows; store.FeatureStore / DeviceCSR refuse larger tables) -- also for gsage_gather_mean_multi(_adam).
 * ---------------------------------------------------------------------------------------- */
int gsage_gather_mean(const void *table, int dtype, int64_t ld, const int64_t *ids, int64_t M,
                      int32_t n, int64_t D, void *out, int out_dtype, int64_t out_ld,
                      void *stream);

/* Up to 8 gather+mean problems (e.g. all hops of one level) in one launch; bf16 in / bf16 out (or
 * fp32 in / fp32 out: the exact-arithmetic parity mode).
 * tables / ids / outs / M / n are HOST arrays of n_seg entries holding DEVICE pointers and sizes;
 * segment s computes outs[s][i] = mean_j tables[s][ rows(i, j) ] exactly like gsage_gather_mean
 * (ids[s] == NULL: rows i*n+j of tables[s]).  All segments share ld, D, out_ld. */
int gsage_gather_mean_multi(int32_t n_seg, const void *const *tables, const int64_t *const *ids,
                            void *const *outs, const int64_t *M, const int32_t *n, int dtype,
                            int64_t ld, int64_t D, int out_dtype, int64_t out_ld, void *stream);

/* Backward of the segment mean w.r.t. contiguous neighbour rows (autograd of nn_modules.py:198):
 *     dneibs[i*n+j, c] = dagg[i, c] / n          (fp32 in, fp32 out) */
int gsage_segment_mean_bwd(const float *dagg, int64_t ld, int64_t M, int32_t n, int64_t D,
                           float *dneibs, int64_t out_ld, void *stream);

/* K6  table_grad[ids[i], c] += scale * rows[i / n, c]   for i in [0, M*n)  (atomic fp32).
 *     Backward of gather_mean w.r.t. a trainable table (the dense nn.Embedding gradient of
 *     NodeEmbeddingPrep, nn_modules.py:134,146-149); n = 1, scale = 1 for a plain gather. */
int gsage_scatter_add_rows(const float *rows, int64_t ld, const int64_t *ids, int64_t M,
                           int32_t n, int64_t D, float scale, float *table_grad,
                           int64_t table_ld, void *stream);

/* ------------------------------------------------------------------------------------------
 * K5  projection GEMM (MFMA) replaces fc_x(x), fc_neib(agg), cat, activation
 *                           (nn_modules.py:200-202) and every other nn.Linear on the path
 *
 *     act in GSAGE_ACT_{NONE,RELU,TANH}.
 *     for g in [0, groups):   C[m, g*c_gstride + j] = act( sum_k A_g[m,k] * W_g[j,k] + bias_g[j] )
 *         A_g row m  = (a_rows ? A + a_rows[m]*lda : A + m*lda) + g*a_gstride      [M, K]
 *         W_g        = W + g*w_gstride                                             [N, K] row-major
 *         bias_g     = bias ? bias + g*N : none                                    fp32
 *     groups = 2 writes both halves of the concat in one launch (x | agg against Wx | Wn).
 *     `a_rows` (int64 [M], may be NULL, group 0 only when `a_rows_group0_only`) fuses the row
 *     gather feats[ids] into the A-operand load.
 *     dtype = type of A and W (bf16: v_mfma_f32_32x32x16_bf16; fp32: v_mfma_f32_32x32x2_f32),
 *     fp32 accumulation, C stored as c_dtype.  Needs lda, ldw % (16/sizeof) == 0 and zero
 *     padding of A and W in columns [K, round_up(K, 16/sizeof)).
 * ---------------------------------------------------------------------------------------- */
int gsage_linear_nt(const void *A, int dtype, int64_t lda, const int64_t *a_rows,
                    int a_rows_group0_only, const void *W, int64_t ldw, const float *bias,
                    void *C, int c_dtype, int64_t ldc, int64_t M, int64_t N, int64_t K, int act,
                    int groups, int64_t a_gstride, int64_t w_gstride, int64_t c_gstride,
                    void *stream);

/* K5 with the weight operand in MFMA fragment order (bf16 only): a wave's B fragment is one coalesced
 * 1 KiB load from L2 straight into registers, so only A goes through LDS and both operand rings run
 * deeper than LDS capacity allows the plain kernel.  Same arithmetic and
 * arguments as gsage_linear_nt with dtype = bf16; Wp from gsage_pack_weight (or kept current by the
 * optimizer through gsage_prep_desc.dst_p), groups laid out back to back.  A rows must be whole
 * 128-byte lines, zero padded up to round_up(K, 64).
 *     Wp[g][jb][kc][lane][e] = W_g[jb*32 + (lane & 31)][kc*16 + (lane >> 5)*8 + e]   (0 outside N x K)
 *     jb < ceil(N/32), kc < 4*ceil(K/64), lane < 64, e < 8;  gsage_packed_weight_elems() bf16 elements. */
int64_t gsage_packed_weight_elems(int64_t N, int64_t K, int32_t groups);
int gsage_pack_weight(const void *W, int dtype, int64_t ldw, int64_t w_gstride, int64_t N, int64_t K,
                      int32_t groups, void *Wp, void *stream);
int gsage_linear_nt_packed(const void *A, int64_t lda, const int64_t *a_rows, int a_rows_group0_only,
                           const void *Wp, const float *bias, void *C, int c_dtype, int64_t ldc,
                           int64_t M, int64_t N, int64_t K, int act, int groups, int64_t a_gstride,
                           int64_t c_gstride, void *stream);

/* K5b weight gradient (MFMA, bf16 in / fp32 out): the backward of the projection above w.r.t. its
 *     weights (autograd of nn_modules.py:189-190,200 under loss.backward(), models.py:100)
 *
 *     out[g*out_gstride + n*K + k] = sum_m dC[m, g*n_per_group + n] * A_g[m, k]
 *         A_g = A + g*a_gstride  ([M, lda] row-major)      dC, A: bf16;  out: fp32 [groups, n_per_group, K]
 *     The reduction over M is split into ceil(M / rows_per_split) slices whose partial tiles are
 *     written to `slabs` (fp32 [slices, Ntot, ldk], caller-allocated) and summed deterministically
 *     into `out` (out == NULL: the slabs are left for gsage_finalize_grads).
 *     Needs ldc, lda % 8 == 0 and 16-byte aligned dC, A (16-byte lane loads), ldk % 4 == 0,
 *     Ntot % 4 == 0, K <= ldk <= lda, rows_per_split % 16 == 0, n_per_group % 128 == 0 unless there
 *     is a single group. */
int gsage_wgrad(const void *dC, int dtype, int64_t ldc, const void *A, int64_t lda, int64_t a_gstride, int64_t M,
                int64_t Ntot, int64_t K,
                int64_t n_per_group, int64_t rows_per_split, float *slabs, int64_t ldk, float *out,
                int64_t out_gstride, void *stream);
/* [host] number of slabs gsage_wgrad writes for (M, rows_per_split). */
int gsage_wgrad_slabs(int64_t M, int64_t rows_per_split);
/* Up to 8 weight-gradient problems (all the levels of one backward pass) in ONE launch; the partial
 * tiles stay in each problem's slabs for gsage_finalize_grads.  Each problem alone fills a fraction
 * of the chip for the length of its M-slice, so side by side they cost the longest, not the sum.
 * `probs` is a HOST array (copied into the kernel arguments).  Field meaning as in gsage_wgrad. */
typedef struct gsage_wgrad_desc {
    const void *dC;
    const void *A;
    float *slabs;
    int64_t ldc, lda, a_gstride;
    int64_t M, Ntot, K, n_per_group, ldk, rows_per_split;
    const int64_t *a_rows;      /* optional (NULL = A's own rows): reduction index m reads row a_rows[m] of A -- a
                                 * frontier's table rows in place, no gathered copy; 16-byte aligned, ids < 2^32.
                                 * With several groups the list belongs to group 0's operand only (the other groups
                                 * read A + g * a_gstride row by row, like gsage_linear_nt's a_rows_group0_only) */
} gsage_wgrad_desc;
int gsage_wgrad_multi(int32_t n_prob, const gsage_wgrad_desc *probs, int dtype, void *stream);
/* [host] 1 when gsage_wgrad_multi gives a problem's workgroups TWO 128-row output tiles each (its waves pair up on the
 * same rows of A: half the readers per A line; long bf16 reductions with whole tile pairs per group) -- the caller then
 * wants slices half as long for the same number of workgroups (ops.wgrad_plan). */
int gsage_wgrad_pair_ok(int dtype, int64_t M, int64_t Ntot, int64_t n_per_group, int64_t rows_per_split);
/* Device counters for the NEXT gsage_wgrad_multi launch of the calling thread to advance when it starts (consumed by
 * it, also while a command list is being recorded): *tick += 1, *tick1 += inc1, *tick2 += inc2 (any may be NULL).
 * For steps without a finalisation launch (gsage_adam_desc.reduce_descs): the Adam step count, the Philox call
 * counter and the batch-queue index must move AFTER the seed level read them and BEFORE the launch that carries the
 * update and the next frontier's sampling -- K5b sits exactly there and reads none of them. */
int gsage_wgrad_ticks_next(int64_t *tick, int64_t *tick1, int64_t inc1, int64_t *tick2, int64_t inc2);
/* dtype (both entry points) = type of dC and A: GSAGE_BF16 (the MFMA kernel described above) or
 * GSAGE_F32 (plain fp32 FMAs, same decomposition and slab layout: the exact-arithmetic parity mode in
 * which the golden fixtures generated from the reference are replayed through the fused engines). */

/* ------------------------------------------------------------------------------------------
 * K3  pooling MLP           replaces mlp(neibs) -> view(M,-1,H) -> max/mean over the fanout
 *                           (nn_modules.py:224-226 with pool_fn of :240 / :252)
 *
 *     pooled[i, j] = pool_{r in [0,n)} relu( sum_k A[i*n+r, k] * W[j,k] + bias[j] )
 *     A rows are gathered through a_rows when given (A row = A + a_rows[i*n+r]*lda).
 *     argmax (int32 [M, H], may be NULL; max pool only) receives the winning r for backward.
 *     The [M*n, H] hidden activations are never written to HBM.
 * ---------------------------------------------------------------------------------------- */
int gsage_pool_mlp(const void *A, int dtype, int64_t lda, const int64_t *a_rows, const void *W,
                   int64_t ldw, const float *bias, int64_t M, int32_t n, int64_t H, int64_t K,
                   int pool, float *pooled, int64_t pooled_ld, int32_t *argmax, void *pooled_bf16,
                   int64_t pooled_bf16_ld, uint32_t *relu_mask, void *stream);
/* pooled_bf16 (may be NULL): the same result rounded to bf16, [M, pooled_bf16_ld] -- the operand
 * copy the following projection (gsage_linear_nt) and its weight gradient (gsage_wgrad) read.
 * relu_mask (may be NULL; H % 32 == 0): [M*n, H/32] words, bit c%32 of word c/32 of row r = hidden
 * activation (r, c) > 0 -- all the mean pool's backward needs of the hidden layer. */

/* K3 on the packed weight operand (bf16; see gsage_linear_nt_packed): one workgroup covers 64 rows x
 * 256 hidden columns, so A is read once per 256 columns instead of once per 128, and W goes from L2
 * straight into registers.  Same results and arguments as gsage_pool_mlp; Wp = gsage_pack_weight of
 * the [H, K] weight; A rows must be whole 128-byte lines, zero padded up to round_up(K, 64). */
int gsage_pool_mlp_packed(const void *A, int64_t lda, const int64_t *a_rows, const void *Wp,
                          const float *bias, int64_t M, int32_t n, int64_t H, int64_t K, int pool,
                          float *pooled, int64_t pooled_ld, int32_t *argmax, void *pooled_bf16,
                          int64_t pooled_bf16_ld, uint32_t *relu_mask, void *stream);

/* Backward routing of the max pool: the bf16 [M*n, ldo] gradient of the hidden activations,
 *     out[i*n + j, c] = (argmax[i, c] == j && pooled[i, c] > 0) ? g[i, c] : 0
 * (autograd of nn_modules.py:224-226,240: max picks one row per (segment, channel), ReLU passes
 * positive maxima only) -- the dC operand of gsage_wgrad for the MLP's weight gradient.
 * H, ldo % 8 == 0. */
int gsage_pool_route_bwd(const float *g, int64_t ldg, const float *pooled, int64_t ldp,
                         const int32_t *argmax, int64_t lda, int64_t M, int32_t n, int32_t H, void *out,
                         int out_dtype, int64_t ldo, void *stream);

/* Same for the mean pool: out[i*n + j, c] = relu_mask(i*n + j, c) ? g[i, c] / n : 0   (autograd of
 * nn_modules.py:224-226,252), plus -- when bias_part != NULL -- the MLP bias gradient as n_part
 * deterministic partial rows (summed by gsage_finalize_grads).  H % 32 == 0. */
int gsage_pool_route_mean_bwd(const float *g, int64_t ldg, const uint32_t *relu_mask, int64_t M, int32_t n,
                              int32_t H, void *out, int out_dtype, int64_t ldo, float *bias_part, int32_t n_part,
                              void *stream);

/* Bias gradient of the pooling MLP under the max pool: column sums of g * (pooled > 0) over the M
 * segments, as `n_part` deterministic partial rows part[b, c] (b < n_part; summed by
 * gsage_finalize_grads with S = n_part, stride = H).  n_part <= 1024. */
int gsage_pool_bias_partials(const float *g, int64_t ldg, const float *pooled, int64_t ldp, int64_t M,
                             int32_t H, float *part, int32_t n_part, void *stream);

/* Input gradient of a pool-aggregator level, merged and masked (autograd of nn_modules.py:224-230
 * w.r.t. the previous level's post-ReLU output Hprev [R, ldh] bf16):
 *     dH[m, c] = (Hprev[m, c] > 0) * ( (m < r_x ? DX[m, c] : 0) + (m >= r0 ? DN[m - r0, c] : 0) )
 * DX fp32 [r_x, ldx] = gradient through fc_x of the rows that were "x", DN fp32 [R - r0, ldn] =
 * gradient through the pooling MLP of the rows that were neighbours.  D % 4 == 0. */
int gsage_pool_merge_bwd(const void *Hprev, int dtype, int64_t ldh, const float *DX, int64_t ldx, int64_t r_x,
                         const float *DN, int64_t ldn, int64_t r0, void *dH, int64_t ldo, int64_t R,
                         int32_t D, void *stream);

/* ------------------------------------------------------------------------------------------
 * K4  attention aggregation  replaces AttentionAggregator.forward's weighting,
 *                            nn_modules.py:307-315 (scores, softmax over the fanout, weighted
 *                            sum of the RAW neighbour rows)
 *
 *     na[i,r,:] given (att(neibs), [M*n, Ha] fp32), xa[i,:] given (att(x), [M, Ha] fp32)
 *     s[i,r] = <na[i,r,:], xa[i,:]>;  w = softmax_r(s);  agg[i,:] = sum_r w[i,r] * neibs[i*n+r,:]
 *     neibs rows are gathered from `table` through ids when ids != NULL.
 *     ws (fp32 [M, n]) is written for backward.
 * ---------------------------------------------------------------------------------------- */
int gsage_attn_aggregate(const float *na, int64_t na_ld, const float *xa, int64_t xa_ld,
                         const void *table, int dtype, int64_t ld, const int64_t *ids, int64_t M,
                         int32_t n, int64_t Ha, int64_t D, float *agg, int64_t agg_ld, float *ws,
                         void *stream);
/* Same, plus (agg_lp != NULL) the result rounded to the table's type, [M, agg_lp_ld] with zero pad columns -- the
 * operand copy the fc_neib projection and its weight gradient read (no cast launch); agg may then be NULL. */
int gsage_attn_aggregate_lp(const float *na, int64_t na_ld, const float *xa, int64_t xa_ld, const void *table,
                            int dtype, int64_t ld, const int64_t *ids, int64_t M, int32_t n, int64_t Ha, int64_t D,
                            float *agg, int64_t agg_ld, float *ws, void *agg_lp, int64_t agg_lp_ld, void *stream);
/* Second layer of the att MLP (nn_modules.py:292-296: Linear(32, 32, bias=False) after the tanh), Ha == 32:
 *   gsage_attn_mlp2_fwd   a[m, :] = hid[m, :] W2^T                       hid, W2: `dtype` (operand copies), a: fp32
 *   gsage_attn_mlp2_bwd   da = T(dan + dax);  dhid = T((da W2) * (1 - hid^2))   -- the cast, the product and the
 *                         tanh backward of the chain d a -> d hid in one pass; W2T = the transposed operand copy */
int gsage_attn_mlp2_fwd(const void *hid, int dtype, int64_t ldh, const void *W2, int64_t ldw, float *a, int64_t lda,
                        int64_t M, int32_t Ha, void *stream);
int gsage_attn_mlp2_bwd(const float *dan, int64_t ldn, const float *dax, int64_t ldx, const void *hid, int dtype,
                        int64_t ldh, const void *W2T, int64_t ldw, void *da, int64_t ldda, void *dhid, int64_t lddh,
                        int64_t M, int32_t Ha, void *stream);
/* Backward of the weighting w.r.t. att(neibs) and att(x), one launch (autograd of the three lines
 * above):  dws[i,r] = <neibs[i*n+r,:], g[i,:]>;  ds = ws * (dws - sum_r dws*ws);
 *          dxa[i,:] = sum_r ds[i,r] * na[i*n+r,:];   dna[i*n+r,:] = ds[i,r] * xa[i,:].
 * g: fp32 [M, g_ld] gradient of agg.  Rows must be whole 16-byte chunks (ld % 8 == 0 bf16 / % 4 == 0
 * fp32, 16-byte aligned table); n <= 32. */
int gsage_attn_bwd(const float *g, int64_t g_ld, const float *ws, const float *na, int64_t na_ld,
                   const float *xa, int64_t xa_ld, const void *table, int dtype, int64_t ld,
                   const int64_t *ids, int64_t M, int32_t n, int64_t Ha, int64_t D, float *dna,
                   int64_t dna_ld, float *dxa, int64_t dxa_ld, void *stream);

/* K4 / K4' with the attention MLP inside (round 6; csrc/gsage_attn_fused.hip): one pass over a hop's child rows per
 * direction instead of three (the att.0 GEMM with its tanh, gsage_attn_mlp2_fwd and gsage_attn_aggregate_lp; backward
 * gsage_attn_bwd and gsage_attn_mlp2_bwd) -- reference nn_modules.py:293-296 + 309-315 and their autograd.  bf16 rows
 * of whole 16-byte chunks, D <= 640, fan-out 2..16, the reference's 32-wide attention MLP: gsage_attn_fused_ok says
 * whether a shape is covered (1 / 0; no GPU needed).  Child `pos` = parent * n + j is row `ids[pos]` of `table` (ids
 * NULL: row `row0 + pos`); every per-child array is indexed by pos, every per-parent array by the parent.
 *   forward   hid = bf16(tanh(row W0^T)) [M n, hid_ld], a = hid W2^T (fp32) [M n, a_ld] for the children;
 *             ws = softmax_j <a_child, xa_parent> [M n]; agg_lp = bf16(sum_j ws row_j) [M, lp_ld], the operand copy
 *             the fc_neib projection and its weight gradient read, with zero pad columns up to the next multiple of 32
 *             W0 / W2: the bf16 operand copies of att.0.weight [32, ldw0 >= 32 ceil(D / 32)] and att.2.weight [32, ldw2]
 *   backward  dws = <row, g_parent>, ds = ws (dws - sum ws dws); dxa[parent] = sum_j ds_j na_j (fp32 [M, dxa_ld]);
 *             da = bf16(ds xa_parent) [M n, da_ld]; dhid = bf16((da W2)(1 - hid^2)) [M n, dhid_ld] -- the rows of a
 *             LAST hop are nobody's parents, so da has no second term.  W2T[k][h] = W2[h][k]: transposed operand copy */
int gsage_attn_fused_ok(int dtype, int64_t ld, int64_t D, int32_t n, int64_t Ha);
int gsage_attn_fused_fwd(const void *table, int dtype, int64_t ld, const int64_t *ids, int64_t row0, const void *W0,
                         int64_t ldw0, const void *W2, int64_t ldw2, const float *xa, int64_t xa_ld, int64_t M,
                         int32_t n, int64_t D, void *hid, int64_t hid_ld, float *a, int64_t a_ld, float *ws, void *agg_lp,
                         int64_t lp_ld, void *stream);
int gsage_attn_fused_bwd(const void *table, int dtype, int64_t ld, const int64_t *ids, int64_t row0, const void *W2T,
                         int64_t ldw2t, const float *g, int64_t g_ld, const float *ws, const float *na, int64_t na_ld,
                         const float *xa, int64_t xa_ld, const void *hid, int64_t hid_ld, int64_t M, int32_t n, int64_t D,
                         void *da, int64_t da_ld, void *dhid, int64_t dhid_ld, float *dxa, int64_t dxa_ld, void *stream);

/* The trainable node-embedding prep as two row pipelines (round 6; csrc/gsage_prep_rows.hip; reference nn_modules.py:126-155
 * and its autograd).  bf16 operands, 64-wide embeddings: gsage_prep_rows_ok says whether a shape is covered (1 / 0).
 * Frontier position pos reads table row (pos < n_seed ? spare : ids[pos]) -- the seeds all read the spare row n_nodes.
 *   forward   eraw[pos] = bf16(table row); out[pos] = bf16(eraw[pos] W^T + bias): the embedding gather and the prep.fc
 *             projection in one launch (W: prep.fc's bf16 operand copy [64, ldw]; `out` points at the prep's columns of
 *             the level-0 rows)
 *   backward  v = (dhid ? dhid W0 : DATT ? DATT : 0) + (pos < r_x ? DX[pos] : 0) + w(pos) DAGG[parent(pos)]  -- the sources
 *             and the hop layout of gsage_attn_merge_bwd2, no ReLU mask (the prep is affine); dhid [R, lddh] bf16 with
 *             W0T = att.0's transposed operand copy [64, ldw0t] saves the GEMM that would have produced DATT;
 *             din0[pos] = bf16(v); bias_part[b] = column sums of v over workgroup b's rows (n_part workgroups: the
 *             prep.fc.bias gradient's partials); d = bf16(v) WpT^T (WpT[e][c] = prep.fc.weight[c][e]); then either
 *             g_table[row(pos)] += d with fp32 atomics -- the seeds' rows summed per 16-row tile first -- or
 *             (deraw != NULL) deraw[pos] = d and no atomics (the sorted, deterministic table gradient of data-parallel runs) */
int gsage_prep_rows_ok(int dtype, int64_t E);
int gsage_prep_rows_fwd(const float *table, int64_t ldt, const int64_t *ids, int64_t n_seed, int64_t spare, const void *W,
                        int64_t ldw, const float *bias, int64_t M, int64_t E, void *eraw, int64_t lde, void *out,
                        int64_t ldo, void *stream);
int gsage_prep_rows_bwd(const void *dhid, int64_t lddh, const void *W0T, int64_t ldw0t, const float *DATT, int64_t ldatt,
                        const float *DX, int64_t ldx, int64_t r_x, const float *DAGG, int64_t ldagg, const float *ws,
                        int32_t n_hops, const int64_t *off, const int32_t *fan, int64_t R, int64_t E, void *din0,
                        int64_t ldd, float *bias_part, int32_t n_part, const void *WpT, int64_t ldwpt, const int64_t *ids,
                        int64_t n_seed, int64_t spare, float *g_table, int64_t ldg, float *deraw, int64_t ldde,
                        void *stream);

/* Glue of the native attention train step (engine.FusedAttnTrainStep): what autograd ran as separate cast /
 * add / tanh-backward / expand kernels between K4, K5 and K5b.
 *   gsage_add_cast        dst[m, c] = T(a[m, c] + (b ? b[m, c] : 0))             (fp32 in, bf16 / fp32 out)
 *   gsage_tanh_bwd        out[m, c] = T(g[m, c] * (1 - hid[m, c]^2))             (the att MLP's tanh, nn_modules.py:294)
 *   gsage_attn_merge_bwd  input gradient of an attention level (autograd of nn_modules.py:307-317 w.r.t. x and
 *                         neibs), rows = hops concatenated (hop k from off[k], fan[k] children per parent):
 *       dIn[m] = mask(m) * ( DATT[m] + (m < r_x ? DX[m] : 0) + (hop(m) >= 1 ? ws[m - off[1]] * DAGG[parent(m)] : 0) )
 *       DATT: through att(.), every row; DX: through fc_x; ws: the softmax weights of every (parent, child) pair
 *       in hop order (the ws outputs of gsage_attn_aggregate, back to back); mask = (H[m, c] > 0) when H is given
 *       (the ReLU of the level below), else 1.  DATT may be NULL (no such path: a MEAN aggregator level over an
 *       embedding prep), ws may be NULL (uniform weights 1 / fan[hop]: the mean, nn_modules.py:197-198). */
int gsage_add_cast(const float *a, int64_t lda, const float *b, int64_t ldb, void *dst, int dst_dtype, int64_t ldd,
                   int64_t M, int64_t D, void *stream);
int gsage_tanh_bwd(const float *g, int64_t ldg, const void *hid, int dtype, int64_t ldh, void *out, int64_t ldo,
                   int64_t M, int64_t D, void *stream);
int gsage_attn_merge_bwd(const void *H, int h_dtype, int64_t ldh, const float *DATT, int64_t ldatt, const float *DX,
                         int64_t ldx, int64_t r_x, const float *DAGG, int64_t ldagg, const float *ws, void *out,
                         int out_dtype, int64_t ldo, int64_t R, int32_t D, int32_t n_hops, const int64_t *off,
                         const int32_t *fan, void *stream);
/* gsage_attn_merge_bwd2: the same, plus (out2_bf16 != NULL) a second, bf16 copy of the result [R, ldo2] -- the operand
 * the following GEMMs read when `out` has to stay fp32 (a bias gradient's column sums). */
int gsage_attn_merge_bwd2(const void *H, int h_dtype, int64_t ldh, const float *DATT, int64_t ldatt, const float *DX,
                         int64_t ldx, int64_t r_x, const float *DAGG, int64_t ldagg, const float *ws, void *out,
                         int out_dtype, int64_t ldo, int64_t R, int32_t D, int32_t n_hops, const int64_t *off,
                         const int32_t *fan, void *out2_bf16, int64_t ldo2,
                          void *stream);

/* L1 head of the reference's regression problems, forward + backward (Pokec: problem.py:39-42 behind models.py:90-91,100):
 *     z = E / max(||E||_2, 1e-12);  preds[i] = <z_i, W> + bias;  loss = F.l1_loss(preds [B,1], targets [B])
 * The reference passes targets.squeeze(), so [B,1] against [B] broadcasts to [B,B]: loss = mean_ij |p_i - t_j| and
 * d loss / d p_i = (1/B^2) sum_j sign(p_i - t_j) -- reproduced as is.  One output column (n_classes == 1), B <= 2048.
 * scratch: fp32, gsage_head_l1_scratch(B, D) elements; its first ceil(B / 16) rows of D + 2 floats are per-workgroup
 * partials [d fc.weight (D) | d fc.bias | loss] whose sum is the result (gsage_finalize_grads: S = ceil(B / 16),
 * stride D + 2); dE: [B, ldd] in dE_dtype. */
int gsage_head_l1(const float *E, int64_t lde, const float *W, const float *bias, const float *targets, int64_t B,
                  int64_t D, float *preds, void *dE, int dE_dtype, int64_t ldd, float *scratch, void *stream);
int gsage_head_l1_scratch(int64_t B, int64_t D);
/* The same head on ONE SHARD of a data-parallel batch (ABI 4): `targets` holds the T targets of the GLOBAL batch
 * (every rank's, T <= 8192), E / preds / dE the B rows of this rank.  loss_r = 1/(B T) sum_{i in shard} sum_j |p_i - t_j|,
 * d loss_r / d p_i = 1/(B T) sum_j sign(p_i - t_j): the AVERAGE over the ranks of loss_r and of its gradients equals
 * the single-process loss over the global batch, pair for pair (d p_i depends on p_i and the targets only). */
int gsage_head_l1_sharded(const float *E, int64_t lde, const float *W, const float *bias, const float *targets,
                          int64_t T, int64_t B, int64_t D, float *preds, void *dE, int dE_dtype, int64_t ldd,
                          float *scratch, void *stream);

/* ------------------------------------------------------------------------------------------
 * Classification head, forward + backward   replaces F.normalize(dim=1) -> fc -> F.cross_entropy
 *                                            (models.py:90-91, problem.py:34) and their autograd
 *
 *     z = E / max(||E||_2, 1e-12) (row-wise);  preds = z W^T + bias;  loss = mean CE(preds, targets)
 *     dE (bf16 or fp32, [B, ldd]), dW [C, D], db [C] = gradients of loss;  loss may be NULL.
 *     C <= 64, D <= 1024.  scratch: fp32, gsage_head_ce_scratch(B, C, D) elements = per-workgroup
 *     partials [n_wg][C*D + C + 1] (dW | db | loss); dW == NULL leaves them for gsage_finalize_grads.
 *     batch_idx (may be NULL): targets is a queue [n_batches, B], batch *batch_idx % n_batches is used. */
int gsage_head_ce(const float *E, int64_t lde, const float *W, const float *bias,
                  const int64_t *targets, int32_t B, int32_t C, int32_t D, float *preds, void *dE,
                  int dE_dtype, int64_t ldd, float *dW, float *db, float *loss, float *scratch,
                  const int64_t *batch_idx, int64_t n_batches, void *stream);
int64_t gsage_head_ce_scratch(int32_t B, int32_t C, int32_t D);

/* The whole seed level of a mean-aggregator model in one launch: segment mean of the n sampled
 * neighbours (nn_modules.py:197-198), emb = cat[fc_x(x), fc_neib(agg)] (:200-202, identity
 * activation), the classification head above (models.py:90-91, problem.py:34) and the backward down
 * to the previous level's activations.  H: bf16 [B*(1+n), 256] = previous level output (seed rows,
 * then their n neighbours each); w2 / w2t: bf16 operand copies [2,128,ldw2] / [2,256,ldw2t] of
 * (fc_x | fc_neib) and their transposes.  Writes agg (bf16 [B,256]) and dE (bf16 [B,256]) -- the
 * operands of this level's gsage_wgrad --, preds [B,C], dH (bf16 [B*(1+n),256], ReLU mask of H
 * applied) and the fc.weight | fc.bias | loss partials (layout of gsage_head_ce, scratch size
 * gsage_mean_tail_ce_scratch).  Fixed widths: previous level 256, this level 2h = 256; n <= 32;
 * C <= 64.
 * gather (may be NULL): B / 4 workgroups of ~22 us of dependent phases leave half of the chip idle at
 * B = 512, so n_workgroups extra workgroups of the same launch compute rows [0, rows) of a
 * gsage_gather_mean segment (bf16 table and output, fan-out n = 5, 10 or 15) -- part of the NEXT batch's
 * level-0 gather, which that launch then skips.  Results are those of gsage_gather_mean. */
typedef struct gsage_tail_gather_desc {
    const void *table;
    const int64_t *ids;
    void *out;
    int64_t ld, out_ld, D, rows;
    int32_t n, n_workgroups;
} gsage_tail_gather_desc;
int gsage_mean_tail_ce(const void *H, int32_t B, int32_t n, const void *w2, int64_t ldw2,
                       const void *w2t, int64_t ldw2t, const float *Wfc, const float *bfc, int32_t C,
                       const int64_t *targets, const int64_t *batch_idx, int64_t n_batches, void *agg,
                       void *dE, float *preds, void *dH, float *partial,
                       const gsage_tail_gather_desc *gather, int dtype, void *stream);
/* The same role for the NEXT gsage_linear_nt_packed launch of the calling thread (consumed by it; HOST descriptor, read
 * before gsage_gather_role_next's caller continues -- it may live on the stack until that launch call returns): one
 * more z-slice of the projection's grid (as many workgroups as one group has) gathers rows [0, rows) of the segment.
 * Only with act = ReLU and fan-out 5 or 10 (the level-0 projection of the mean engine: 416 workgroups where 768
 * fit, streaming at half of what a CU keeps in flight); n_workgroups is ignored.  NULL: no role. */
int gsage_gather_role_next(const gsage_tail_gather_desc *gather);
/* A SAMPLER role for the NEXT gsage_linear_nt_packed OR gsage_mean_tail_mfma launch of the calling thread (ABI 5;
 * consumed by whichever comes first; HOST descriptor, read before that launch call returns).  In the seed-level launch
 * (which must carry a gather role): gsage_mean_tail_mfma_sampler_wgs(B, widest) workgroups right behind the seed-level ones --
 * every workgroup of that launch owns a CU, so the caller sizes the gather role for the CUs that are left.  In the
 * projection: one more z-slice of the projection's grid runs the fused
 * multi-hop sampler (gsage_sample_hops) for a LATER batch -- address it through call_base / batch_base, the counters
 * are not touched -- ceil(B / slice size) seeds per workgroup.  K1 is ~9 us of dependent loads that move almost
 * nothing: in the projection's free workgroup slots it is off every critical path.  Only with act = ReLU, a CSR
 * adjacency and no gather role; same frontier as gsage_sample_hops (sampling reads no weights).  NULL: no role. */
int gsage_hops_role_next(const gsage_hops_desc *hops);
/* dtype = storage type of H, w2, w2t, agg, dE, dH: GSAGE_BF16, or GSAGE_F32 -- the same kernel source
 * instantiated on fp32 storage (every bf16 rounding point becomes a no-op; no gather role), used to
 * replay the reference-generated golden fixtures through this kernel at fp32 tolerance. */
int64_t gsage_mean_tail_ce_scratch(int32_t B, int32_t C);
/* The same seed level on the matrix cores (ABI 5; bf16 storage only): 16 seeds per 512-thread workgroup -- one MFMA
 * row tile -- instead of 4 per 256-thread workgroup on the VALU.  ceil(B / 16) workgroups stream the projection
 * weights (32 instead of 128 times at B = 512), both projections and the input gradients run on
 * v_mfma_f32_16x16x32_bf16, the head (logits, d z, d fc.weight) on v_mfma_f32_16x16x4_f32 (exact fp32 products, as
 * on the VALU).  Same arguments, same outputs and rounding points as gsage_mean_tail_ce with dtype = GSAGE_BF16
 * (sums run in a different order: results agree to fp32 round-off, agg / dE / dH to one bf16 rounding); the partials
 * are ceil(B / 16) rows (gsage_mean_tail_mfma_scratch).  gather: the role's workgroups are 512 threads here. */
int gsage_mean_tail_mfma(const void *H, int32_t B, int32_t n, const void *w2, int64_t ldw2,
                         const void *w2t, int64_t ldw2t, const float *Wfc, const float *bfc, int32_t C,
                         const int64_t *targets, const int64_t *batch_idx, int64_t n_batches, void *agg,
                         void *dE, float *preds, void *dH, float *partial,
                         const gsage_tail_gather_desc *gather, void *stream);
int64_t gsage_mean_tail_mfma_scratch(int32_t B, int32_t C);
/* workgroups a sampler role (gsage_hops_role_next) adds to that launch for a batch of B seeds whose widest hop holds
 * `widest` ids per seed (the product of the fan-outs so far); 0: the role's frontier does not fit the launch's LDS */
int32_t gsage_mean_tail_mfma_sampler_wgs(int64_t B, int64_t widest);

/* ------------------------------------------------------------------------------------------
 * Fused tail of train_step (models.py:101-102) and inter-layer backward routing (models.py:85-86
 * under autograd), used by the hipGraph engine.
 * ---------------------------------------------------------------------------------------- */

/* clip_grad_norm(params, max_norm) + Adam(betas, eps, L2 weight_decay) over FLAT fp32 buckets of n
 * elements: p (parameters), g (gradients; overwritten with the clipped gradient), m, v (Adam
 * moments).  lr and step are DEVICE scalars (float / int64) so a captured graph follows the LR
 * schedule and counts steps.  step_is_current == 0: this call is update number *step + 1 and
 * increments *step afterwards; != 0: *step was already advanced for this update (by
 * gsage_prep_weights' / gsage_finalize_grads' tick) and is left alone.  partial: fp32 scratch of
 * gsage_adam_partials(n) elements.  norm_out (may be NULL) receives the pre-clip gradient norm.
 * (step_is_current & 2: the caller consumes g in this call and zeroes it next -- a scatter-added
 * embedding-table gradient --, so the clipped values are not written back.  step_is_current & 4: use the
 * arithmetic of the deferred row updates (gsage_rows_*: 1-ulp sqrt and reciprocal) instead of
 * torch.optim.Adam's correctly rounded sqrt and division, so that a table updated densely here and row by
 * row there gets the same bits; every other caller leaves the bit clear and gets torch's roundings.)
 * n_partial_ready > 0: `partial` already holds that many squared-norm partials (written by
 * gsage_finalize_grads) and the norm pass is skipped.  prep_descs (DEVICE array of n_prep
 * gsage_prep_desc whose src point into p; may be NULL): the bf16 operand copies of the updated
 * weights are refreshed in the same launch.
 */
int gsage_clip_adam_step(float *p, float *g, float *m, float *v, int64_t n, float *partial,
                         const float *lr, int64_t *step, float beta1, float beta2, float eps,
                         float weight_decay, float max_norm, float *norm_out, int step_is_current,
                         int32_t n_partial_ready, const void *prep_descs, int32_t n_prep,
                         int64_t *tick1, int64_t inc1, int64_t *tick2, int64_t inc2, void *stream);
/* The same arguments as a struct (HOST memory), for gsage_gather_mean_multi_adam. */
typedef struct gsage_adam_desc {
    float *p, *g, *m, *v;
    int64_t n;
    float *partial;
    const float *lr;
    int64_t *step;
    float beta1, beta2, eps, weight_decay, max_norm;
    float *norm_out;
    int32_t step_is_current, n_partial_ready;
    const void *prep_descs;
    int32_t n_prep;
    int64_t *tick1;
    int64_t inc1;
    int64_t *tick2;
    int64_t inc2;
    /* ABI 4, gsage_gather_mean_multi_adam only: the update's workgroups form the squared norm of g THEMSELVES
     * (n_partial_ready == 0 and norm_slots given).  They are few (one per 1 024 elements, <= 1 024 of them, dispatched
     * first, all resident at once).  Each loads the elements it is about to update, publishes their partial as
     * norm_slots[its index] = (update number << 32 | float bits) with ONE device-scope store, and polls the slots of
     * the others until all carry this update's number; every workgroup then adds the same partials in the same order.
     * No counter, no reset between launches (*step must change from launch to launch: it is the update number).  For
     * the data-parallel step: the norm of the AVERAGED gradient exists only after the exchange, and a launch of its
     * own for it sat on the critical path behind the collective.  norm_slots: >= 1 024 x 8 bytes, zero-initialised. */
    uint64_t *norm_slots;
    /* ABI 5 (with norm_slots, n_partial_ready == 0): NO finalisation launch ran -- g does not exist yet.  reduce_descs:
     * DEVICE array of n_reduce (<= 16) gsage_reduce_desc (below) that together cover [0, n): every update workgroup
     * sums the partial buffers of the 1 024 elements it is about to update (buffer order 0 .. S-1: the bits
     * gsage_finalize_grads would produce), stores them to g (then the clipped values, when the clip is active) and
     * goes on as above.  Replaces the finalisation launch of the single-GPU step (~6.5 us between K5b and the update,
     * reading the same partial buffers); a data-parallel step keeps it (the exchange needs the flat bucket).  The
     * step's ticks then ride in the K5b launch (gsage_wgrad_ticks_next).  NULL: g holds the gradient. */
    const void *reduce_descs;
    int32_t n_reduce;
} gsage_adam_desc;
/* The update of a gsage_adam_desc with norm_slots as a launch of its own (per-call steps: nothing to ride with):
 * one workgroup per 1 024 elements, all resident (<= 1 024 and <= gsage_gather_adam_capacity of an LDS-free launch). */
int gsage_clip_adam_meet(const gsage_adam_desc *adam, void *stream);
/* Workgroups of gsage_gather_mean_multi_adam's launch that are resident at once on the current device when the
 * sampler role needs lds_bytes of dynamic LDS (one per CU kept as margin): the in-launch norm is admitted only when
 * the update's workgroups -- ceil(n / 1024) -- all fit; callers fall back to norm partials otherwise.  dtype: of the
 * gathered table.  0 on failure. */
int gsage_gather_adam_capacity(int dtype, int64_t lds_bytes);
/* tick1 / tick2 (may be NULL): *tick1 += inc1, *tick2 += inc2 when the kernel starts (e.g. the
 * Philox call index and batch-queue index, when nothing in the same launch reads them). */
/* Live rows for the NEXT head launch of the calling thread (gsage_head_ce, gsage_mean_tail_ce, gsage_head_l1),
 * consumed by it -- also while a command list is being recorded, the pointer then belongs to the recorded
 * launch.  n_valid: DEVICE int32, one word (or [n_batches], indexed like the target queue when the head is given
 * a batch_idx).  Rows past n_valid are padding: they get predictions, but no loss term and a zero gradient,
 * and the loss / gradient means run over n_valid rows.  Why: the reference's `iterate` (problem.py:141-153) cuts
 * an epoch into near-equal chunks that are never all of one size, while a recorded step has one geometry -- short
 * chunks are padded to it.  NULL (the default): every row is live. */
int gsage_head_n_valid_next(const int32_t *n_valid);

/* gsage_gather_mean_multi (the NEXT batch's level-0 gathers: they read features and ids only) with up
 * to two short latency-bound jobs riding in the same launch, each a few hundred workgroups that are
 * free next to the HBM-bound gather:
 *   adam (may be NULL)  the clip + Adam update of the CURRENT batch (~8 us alone).  n_partial_ready
 *                       must be > 0 (norm partials from gsage_finalize_grads) -- or 0 with norm_slots given
 *                       (the norm is formed inside the launch) -- and step_is_current != 0.
 *   hops (may be NULL)  gsage_sample_hops_philox of the batch AFTER the next one (~9 us alone) into
 *                       its own frontier buffer (hops->ids must not alias the ids being gathered).
 *                       adam's ticks must then not touch hops->call_ctr / hops->batch_idx: address the
 *                       future batch through call_base / batch_base instead.
 * At least one of the two must be given. */
int gsage_gather_mean_multi_adam(int32_t n_seg, const void *const *tables, const int64_t *const *ids,
                                 void *const *outs, const int64_t *M, const int32_t *n, int dtype,
                                 int64_t ld, int64_t D, int out_dtype, int64_t out_ld,
                                 const gsage_adam_desc *adam, const gsage_hops_desc *hops, void *stream);

/* Sums partial gradient buffers into the flat bucket and emits the squared-norm partials the
 * clip needs, in one launch: for descriptor d, flat_g[out_off + r*cols + c] =
 * sum_{s<S} src[s*stride + r*ld + c].  Covers the K5b slabs (call gsage_wgrad with out == NULL)
 * and the head partials (gsage_head_ce with dW == NULL: one row of C*D + C columns, ld = stride
 * = C*D + C + 1).  descs: DEVICE array.  partial_sq: gsage_finalize_partials(n_desc, max_elems)
 * floats.  tick (may be NULL): *tick += 1 (the Adam step counter); tick1 / tick2 (may be NULL):
 * *tick1 += inc1, *tick2 += inc2 (Philox call index, batch-queue index). */
typedef struct {
    const float *src;
    int64_t stride;
    int64_t out_off;
    int32_t S, rows, cols, ld;
} gsage_reduce_desc;
int gsage_finalize_grads(const void *descs, int32_t n_desc, int64_t max_elems, float *flat_g,
                         float *partial_sq, int64_t *tick, int64_t *tick1, int64_t inc1,
                         int64_t *tick2, int64_t inc2, void *stream);
int gsage_finalize_partials(int32_t n_desc, int64_t max_elems);
/* Pieces of the same tail for a bucket that holds a trainable embedding table (NodeEmbeddingPrep, config 4:
 * a dense 418 MB gradient by the reference's semantics):
 *   gsage_grad_sqnorm      partial[b] = sum of g[i]^2 over block b's slice (n_partial <= 1024 blocks): squared-norm
 *                          partials of a bucket slice, to sit next to gsage_finalize_grads' in one `partial` array
 *   gsage_zero_rows        table[ids[r], 0:D] = 0: undoes a gsage_scatter_add_rows once the optimizer has consumed
 *                          it -- the gradient table is zeroed by touching the rows that were written, not all of it
 *   gsage_colsum_partials  part[b, c] = sum_{i = b, b + n_part, ...} src[i, c]: a Linear's bias gradient as
 *                          deterministic partial rows (summed by gsage_finalize_grads, S = n_part, stride = D) */
int gsage_grad_sqnorm(const float *g, int64_t n, float *partial, int32_t n_partial, void *stream);
int gsage_zero_rows(float *table, int64_t ld, const int64_t *ids, int64_t M, int64_t D, void *stream);
int gsage_colsum_partials(const float *src, int64_t ld, int64_t M, int32_t D, float *part, int32_t n_part, void *stream);
int gsage_adam_partials(int64_t n);

/* Deferred ("lazy") Adam over the rows of a trainable embedding table: the reference's dense update
 * (torch.optim.Adam over nn.Embedding.weight, nn_modules.py:131-155 + models.py:44-46), applied row by row
 * and bit-identical to gsage_clip_adam_step over the whole table.  The update of a row whose gradient is zero
 * (m, v decay; p moves by -step_size * m / (sqrt(v) / sqrt(bc2) + eps); weight decay adds wd * p) depends on that
 * row alone, so it is replayed when the row is next needed instead of streaming the table every step:
 *   last[r]   number of the last update applied to row r (initialise to the optimizer's step count)
 *   seen[r]   stamp used to count every row once in the norm (initialise to 0)
 *   hist      2 * hist_cap floats: (step_size, 1/sqrt(bc2)) of update t at slot t % hist_cap, written by
 *             gsage_rows_adam (hist_cap: a power of two) -- every row must be caught up at least once every
 *             hist_cap - 1 updates
 * The update number of a call is *step + step_off (device counter, like gsage_clip_adam_step).
 *   gsage_rows_catch_up      rows ids0[0:n0] ++ ids1[0:n1] (duplicates allowed) brought up to that update --
 *                            BEFORE the forward reads them
 *   gsage_rows_catch_up_all  every row -- before anything else reads the table, m or v
 *   gsage_rows_sqnorm        partial[0:n_partial] = squared-norm partials of the listed gradient rows, each row
 *                            once (to sit beside gsage_finalize_grads' partials)
 *   gsage_rows_adam          clip (total norm from partial[0:n_partial_ready]) + update of the listed rows, each
 *                            once, and their gradient rows zeroed; records the update's constants in hist.
 *                            Must run for EVERY update (also with empty lists). */
typedef struct gsage_row_adam {
    float *p, *g, *m, *v;           /* [n_rows, E] fp32, dense rows, 16-byte aligned */
    int32_t *last, *seen;           /* [n_rows] */
    float *hist;
    const float *lr;                /* device scalar */
    const int64_t *step;            /* device counter */
    int64_t n_rows;
    int32_t E, hist_cap;            /* E % 4 == 0, E <= 256 */
    float beta1, beta2, eps, weight_decay, max_norm;
    int32_t sorted_ids;             /* ABI 4: != 0: ids0[0:n0] is sorted ascending and n1 == 0 -- the first entry of a run of
                                       equal ids does the row's work, the others are skipped by comparing neighbours:
                                       no atomics, `seen` unused, and the order in which gsage_rows_sqnorm adds its
                                       terms depends on the list alone (data-parallel replicas stay bit-identical) */
} gsage_row_adam;
int gsage_rows_catch_up(const gsage_row_adam *d, const int64_t *ids0, int64_t n0, const int64_t *ids1, int64_t n1,
                        int32_t step_off, void *stream);
int gsage_rows_catch_up_all(const gsage_row_adam *d, int32_t step_off, void *stream);
int gsage_rows_sqnorm(const gsage_row_adam *d, const int64_t *ids0, int64_t n0, const int64_t *ids1, int64_t n1,
                      int32_t step_off, float *partial, int32_t n_partial, void *stream);
int gsage_rows_adam(const gsage_row_adam *d, const int64_t *ids0, int64_t n0, const int64_t *ids1, int64_t n1,
                    int32_t step_off, const float *partial, int32_t n_partial_ready, void *stream);

/* Deterministic gradient of a trainable table (ABI 4): the reference's dense `nn.Embedding` gradient
 * (nn_modules.py:131-155 under autograd) is the sum of the gradient rows of every occurrence of a node in the frontier.
 * gsage_scatter_add_rows forms it with fp32 atomics -- an order of additions that changes from run to run and, in a
 * data-parallel run, from rank to rank (the replicas' tables would drift apart bit by bit).  Here instead:
 *   gsage_sort_rows      keys[i] = i < n0 ? ids0[i] : tail_id  (i < n0 + n_tail: a frontier's ids followed by n_tail
 *                        entries of one spare row, e.g. the row every seed reads, nn_modules.py:147-149) sorted
 *                        ascending, STABLE: ids_sorted[0:n], pos_sorted[0:n] = original positions, ascending within
 *                        equal ids.  Only the low key_bits bits are compared (ids < 2^key_bits).  The library's own
 *                        radix sort (round 6: 8 bits per pass, three launches per pass; rocPRIM's until then);
 *                        temp: gsage_sort_rows_temp_bytes(n, key_bits) bytes of 16-byte-aligned device scratch.
 *                        Recordable like any kernel of the library.
 *   gsage_segment_sum_rows   for the first entry i of every run of equal ids:
 *                        table[ids_sorted[i], 0:E] = scale * sum_{j in run, in order} rows(pos_sorted[j]),
 *                        rows(p) = p < n0 ? rows0[p * ld0 : ] : rows1[(p - n0) * ld1 : ]     (fp32, E % 4 == 0, E <= 256)
 *                        -- stores, not atomics: rows of the table that are in no run are not touched.
 * Then gsage_rows_sqnorm / gsage_rows_adam over ids_sorted with sorted_ids = 1. */
int64_t gsage_sort_rows_temp_bytes(int64_t n, int32_t key_bits);
int gsage_sort_rows(const int64_t *ids0, int64_t n0, int64_t tail_id, int64_t n_tail, int32_t key_bits,
                    int64_t *ids_sorted, int32_t *pos_sorted, void *temp, int64_t temp_bytes, void *stream);
int gsage_segment_sum_rows(const int64_t *ids_sorted, const int32_t *pos_sorted, int64_t n, const float *rows0,
                           int64_t ld0, int64_t n0, const float *rows1, int64_t ld1, int32_t E, float scale,
                           float *table, int64_t ldt, void *stream);

/* One launch converting fp32 parameters into the bf16 operand copies K5 / K5b read:
 * dst[r, c] = bf16(src[r, c]) with leading dimension dst_ld, and/or the transposed copy
 * dst_t[c, r] (leading dimension dst_t_ld), and/or the fragment-ordered copy dst_p (see
 * gsage_linear_nt_packed).  `descs` is a DEVICE array of n_desc descriptors; padding of dst / dst_t /
 * dst_p is not written (allocate them zeroed).
 * max_elems = max rows*cols over the descriptors.  Being the first launch of a step it can also
 * advance two device counters: *tick0 += inc0, *tick1 += inc1 (either pointer may be NULL). */
typedef struct {
    const float *src;
    uint16_t *dst;
    uint16_t *dst_t;
    int32_t rows, cols, dst_ld, dst_t_ld;
    uint16_t *dst_p;       /* optional: the gsage_linear_nt_packed operand of THIS matrix (one group) */
    int64_t kc_p;          /* its k chunks per column block: 4 * ceil(cols / 64) */
    int32_t dst_f32;       /* != 0: dst / dst_t are fp32 copies of the same layout (parity mode); dst_p unused */
    int32_t reserved;
} gsage_prep_desc;
int gsage_prep_weights(const void *descs, int32_t n_desc, int64_t max_elems, int64_t *tick0,
                       int64_t inc0, int64_t *tick1, int64_t inc1, void *stream);

/* Gradient of a level's ReLU output H (bf16 [R, ldh], rows = hops concatenated, hop k starting at
 * row off[k] with fan-out fan[k] relative to hop k-1):
 *     dH[m] = (H[m] > 0) * ( (m < r_x ? DG[m, 0:D] : 0)
 *                          + (hop(m) >= 1 ? DG[parent(m), dagg_off : dagg_off + D] / fan[hop(m)] : 0) )
 *     parent(m) = off[k-1] + (m - off[k]) / fan[k]
 * DG: fp32 [r_x, ldg] = the level above's (dX | dAgg); dH: bf16 [R, ldo].  off / fan: HOST arrays
 * of n_hops (<= 6) entries. */
int gsage_bwd_merge(const void *H, int dtype, int64_t ldh, const float *DG, int64_t ldg, int64_t dagg_off,
                    void *dH, int64_t ldo, int64_t R, int64_t r_x, int32_t D, int32_t n_hops,
                    const int64_t *off, const int32_t *fan, void *stream);

/* ------------------------------------------------------------------------------------------
 * Metrics of the train / eval log     replaces ProblemMetrics, problem.py:44-64 (sklearn f1_score on
 *                                     host copies of predictions and targets, called per batch at
 *                                     train.py:150)
 *
 *   gsage_metric_f1: out[0] = micro-F1, out[1] = macro-F1 of
 *       classification (multilabel == 0): argmax_c logits[i, c] against targets int64 [B]; the macro
 *                      mean runs over the classes that occur in targets or predictions (sklearn's label set)
 *       multilabel     (multilabel != 0): logits[i, c] > 0 against targets [B, ldy] (float32 when
 *                      targets_f32, else int64; non-zero = positive); macro over all C labels
 *     counts: int32 scratch [3 * C + 1] (tp | fp | fn per class, integer atomics: exact; + the number of
 *     classification targets outside [0, C)).  out: DOUBLE [3] -- micro, macro, that number: sklearn would
 *     add such a label to its label set, so a caller that sees out[2] != 0 must score the batch on the host
 *     (problem.DeviceMetrics does).  argmax: first maximum wins and a NaN logit wins over numbers (np.argmax).
 *   gsage_metric_mae: out[0] (DOUBLE) = mean |y_true[i] - y_pred[i]| over n elements (problem.py:62-64).
 * ---------------------------------------------------------------------------------------- */
int gsage_metric_f1(const float *logits, int64_t ld, const void *targets, int multilabel, int targets_f32,
                    int64_t ldy, int64_t B, int32_t C, int32_t *counts, double *out, void *stream);
int gsage_metric_mae(const float *y_true, const float *y_pred, int64_t n, double *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GSAGE_H */
