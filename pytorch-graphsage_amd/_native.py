"""
_native.py -- ctypes binding of libgsage_hip.so (C ABI declared in include/gsage.h).

The library is the product; there is no Python or CPU substitute for it.  `lib()` raises
`NativeLibraryError` when the shared object is missing or does not export the full ABI, and
every GPU op in ops.py goes through `lib()`, so a GPU run can never silently use something else.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgsage_hip.so")

F32, BF16 = 0, 1
POOL_MAX, POOL_MEAN = 0, 1
ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2
ABI_VERSION = 6

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_i32 = ctypes.c_int32
_u32 = ctypes.c_uint32
_u64 = ctypes.c_uint64
_int = ctypes.c_int
_f32 = ctypes.c_float

# name -> (restype, argtypes): must list every symbol include/gsage.h declares
SIGNATURES = {
    "gsage_abi_version": (_int, []),
    "gsage_last_error": (ctypes.c_char_p, []),
    "gsage_launch_count": (_u64, []),
    "gsage_debug_abort_trace": (_int, [_int]),
    "gsage_device_info": (_int, [ctypes.c_char_p, _int, ctypes.POINTER(_int), ctypes.POINTER(_int)]),
    "gsage_stream_create_masked": (_int, [_vp, _i32, ctypes.POINTER(_vp)]),
    "gsage_stream_destroy": (_int, [_vp]),
    "gsage_event_create": (_int, [ctypes.POINTER(_vp)]),
    "gsage_event_destroy": (None, [_vp]),
    "gsage_cmdlist_replay_pair": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int]),
    "gsage_cmdlist_begin": (_int, []),
    "gsage_cmdlist_end": (_int, [ctypes.POINTER(_vp)]),
    "gsage_cmdlist_mark": (_int, [_int]),
    "gsage_cmdlist_time_next": (_int, [_int, _int]),
    "gsage_cmdlist_elapsed": (_int, [_vp, _int, _int, ctypes.POINTER(_f32)]),
    "gsage_cmdlist_size": (_i64, [_vp]),
    "gsage_cmdlist_replay": (_int, [_vp, _vp]),
    "gsage_cmdlist_side_begin": (_int, []),
    "gsage_cmdlist_side_end": (_int, []),
    "gsage_cmdlist_join": (_int, []),
    "gsage_cmdlist_destroy": (None, [_vp]),
    "gsage_host_call": (_int, [_vp, _vp, _vp]),
    "gsage_comm_load": (_int, [ctypes.c_char_p]),
    "gsage_comm_unique_id": (_int, [_vp]),
    "gsage_comm_create": (_int, [_vp, _i32, _i32, ctypes.POINTER(_vp)]),
    "gsage_comm_destroy": (_int, [_vp]),
    "gsage_comm_all_reduce_f32": (_int, [_vp, _vp, _i64, _i32, _vp]),
    "gsage_comm_all_gather": (_int, [_vp, _vp, _vp, _i64, _vp]),
    "gsage_comm_group": (_int, [_vp, _i32, _vp]),
    "gsage_sort_rows_temp_bytes": (_i64, [_i64, _i32]),
    "gsage_sort_rows": (_int, [_vp, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _i64, _vp]),
    "gsage_segment_sum_rows": (_int, [_vp, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _i32, _f32, _vp, _i64, _vp]),
    "gsage_head_l1_sharded": (_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _int, _i64, _vp, _vp]),
    "gsage_sample_csr_sel": (_int, [_vp, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    "gsage_sample_dense": (_int, [_vp, _i64, _i64, _vp, _i64, _vp, _i32, _vp, _vp, _vp]),
    "gsage_sample_csr_philox": (_int, [_vp, _vp, _i64, _vp, _i64, _i32, _u32, _u64, _vp, _u64, _u64,
                                       _vp, _vp, _vp, _vp]),
    "gsage_sample_hops_philox": (_int, [_vp, _vp, _i64, _vp, _i64, _i32, _vp, _u32, _u64, _vp, _u64, _u64,
                                        _vp, _vp, _i64, _vp, _vp]),
    "gsage_sample_hops": (_int, [_vp, _vp]),
    "gsage_counter_add": (_int, [_vp, _u64, _vp]),
    "gsage_copy_pair": (_int, [_vp, _vp, _i64, _vp, _vp, _i64, _vp]),
    "gsage_mt_create": (_vp, [_u32]),
    "gsage_mt_destroy": (None, [_vp]),
    "gsage_mt_seed": (None, [_vp, _u32]),
    "gsage_mt_choice_i32": (_i64, [_vp, _i64, _i64, _vp]),
    "gsage_mt_permutation": (None, [_vp, _i64, _vp]),
    "gsage_mt_choice_device": (_int, [_vp, _i64, _i64, _vp, _vp]),
    "gsage_mt_choice_segments": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    "gsage_mt_choice_par": (_int, [_vp, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "gsage_mt_choice_par_scratch": (_i64, [_i32]),
    "gsage_mt_jump_table": (_int, [_vp, _i64]),
    "gsage_mt_jump_table_words": (_i64, []),
    "gsage_mt_jump_host": (_int, [_vp, _vp, _vp]),
    "gsage_head_n_valid_next": (_int, [_vp]),
    "gsage_gather_role_next": (_int, [_vp]),
    "gsage_hops_role_next": (_int, [_vp]),
    "gsage_gather_mean": (_int, [_vp, _int, _i64, _vp, _i64, _i32, _i64, _vp, _int, _i64, _vp]),
    "gsage_gather_mean_multi": (_int, [_i32, _vp, _vp, _vp, _vp, _vp, _int, _i64, _i64, _int, _i64, _vp]),
    "gsage_gather_mean_multi_adam": (_int, [_i32, _vp, _vp, _vp, _vp, _vp, _int, _i64, _i64, _int, _i64,
                                            _vp, _vp, _vp]),
    "gsage_segment_mean_bwd": (_int, [_vp, _i64, _i64, _i32, _i64, _vp, _i64, _vp]),
    "gsage_scatter_add_rows": (_int, [_vp, _i64, _vp, _i64, _i32, _i64, _f32, _vp, _i64, _vp]),
    "gsage_linear_nt": (_int, [_vp, _int, _i64, _vp, _int, _vp, _i64, _vp, _vp, _int, _i64, _i64,
                               _i64, _i64, _int, _int, _i64, _i64, _i64, _vp]),
    "gsage_packed_weight_elems": (_i64, [_i64, _i64, _i32]),
    "gsage_pack_weight": (_int, [_vp, _int, _i64, _i64, _i64, _i64, _i32, _vp, _vp]),
    "gsage_linear_nt_packed": (_int, [_vp, _i64, _vp, _int, _vp, _vp, _vp, _int, _i64, _i64, _i64, _i64, _int,
                                      _int, _i64, _i64, _vp]),
    "gsage_wgrad": (_int, [_vp, _int, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _i64, _vp,
                           _i64, _vp]),
    "gsage_wgrad_slabs": (_int, [_i64, _i64]),
    "gsage_pool_route_bwd": (_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _int, _i64, _vp]),
    "gsage_wgrad_multi": (_int, [_i32, _vp, _int, _vp]),
    "gsage_wgrad_pair_ok": (_int, [_int, _i64, _i64, _i64, _i64]),
    "gsage_wgrad_ticks_next": (_int, [_vp, _vp, _i64, _vp, _i64]),
    "gsage_clip_adam_meet": (_int, [_vp, _vp]),
    "gsage_gather_adam_capacity": (_int, [_int, _i64]),
    "gsage_head_ce": (_int, [_vp, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _int, _i64, _vp, _vp,
                             _vp, _vp, _vp, _i64, _vp]),
    "gsage_head_ce_scratch": (_i64, [_i32, _i32, _i32]),
    "gsage_mean_tail_ce": (_int, [_vp, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _i32, _vp, _vp, _i64,
                                  _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp]),
    "gsage_mean_tail_ce_scratch": (_i64, [_i32, _i32]),
    "gsage_mean_tail_mfma": (_int, [_vp, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _i32, _vp, _vp, _i64,
                                    _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gsage_mean_tail_mfma_scratch": (_i64, [_i32, _i32]),
    "gsage_mean_tail_mfma_sampler_wgs": (_i32, [_i64, _i64]),
    "gsage_clip_adam_step": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _f32, _f32, _f32, _f32,
                                    _f32, _vp, _int, _i32, _vp, _i32, _vp, _i64, _vp, _i64, _vp]),
    "gsage_finalize_grads": (_int, [_vp, _i32, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp]),
    "gsage_finalize_partials": (_int, [_i32, _i64]),
    "gsage_grad_sqnorm": (_int, [_vp, _i64, _vp, _i32, _vp]),
    "gsage_zero_rows": (_int, [_vp, _i64, _vp, _i64, _i64, _vp]),
    "gsage_colsum_partials": (_int, [_vp, _i64, _i64, _i32, _vp, _i32, _vp]),
    "gsage_rows_catch_up": (_int, [_vp, _vp, _i64, _vp, _i64, _i32, _vp]),
    "gsage_rows_catch_up_all": (_int, [_vp, _i32, _vp]),
    "gsage_rows_sqnorm": (_int, [_vp, _vp, _i64, _vp, _i64, _i32, _vp, _i32, _vp]),
    "gsage_rows_adam": (_int, [_vp, _vp, _i64, _vp, _i64, _i32, _vp, _i32, _vp]),
    "gsage_adam_partials": (_int, [_i64]),
    "gsage_prep_weights": (_int, [_vp, _i32, _i64, _vp, _i64, _vp, _i64, _vp]),
    "gsage_bwd_merge": (_int, [_vp, _int, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _i32, _i32, _vp,
                               _vp, _vp]),
    "gsage_pool_mlp": (_int, [_vp, _int, _i64, _vp, _vp, _i64, _vp, _i64, _i32, _i64, _i64, _int,
                              _vp, _i64, _vp, _vp, _i64, _vp, _vp]),
    "gsage_pool_mlp_packed": (_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i32, _i64, _i64, _int, _vp, _i64, _vp, _vp,
                                     _i64, _vp, _vp]),
    "gsage_pool_route_mean_bwd": (_int, [_vp, _i64, _vp, _i64, _i32, _i32, _vp, _int, _i64, _vp, _i32, _vp]),
    "gsage_pool_bias_partials": (_int, [_vp, _i64, _vp, _i64, _i64, _i32, _vp, _i32, _vp]),
    "gsage_pool_merge_bwd": (_int, [_vp, _int, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i32, _vp]),
    "gsage_attn_aggregate": (_int, [_vp, _i64, _vp, _i64, _vp, _int, _i64, _vp, _i64, _i32, _i64,
                                    _i64, _vp, _i64, _vp, _vp]),
    "gsage_attn_aggregate_lp": (_int, [_vp, _i64, _vp, _i64, _vp, _int, _i64, _vp, _i64, _i32, _i64,
                                       _i64, _vp, _i64, _vp, _vp, _i64, _vp]),
    "gsage_attn_fused_ok": (_int, [_int, _i64, _i64, _i32, _i64]),
    "gsage_attn_fused_fwd": (_int, [_vp, _int, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i64, _vp,
                                    _i64, _vp, _i64, _vp, _vp, _i64, _vp]),
    "gsage_attn_fused_bwd": (_int, [_vp, _int, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _i64,
                                    _i64, _i32, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "gsage_prep_rows_ok": (_int, [_int, _i64]),
    "gsage_prep_rows_fwd": (_int, [_vp, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _vp]),
    "gsage_prep_rows_bwd": (_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _i32, _vp, _vp, _i64,
                                   _i64, _vp, _i64, _vp, _i32, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _vp]),
    "gsage_attn_mlp2_fwd": (_int, [_vp, _int, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _vp]),
    "gsage_attn_mlp2_bwd": (_int, [_vp, _i64, _vp, _i64, _vp, _int, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32,
                                   _vp]),
    "gsage_add_cast": (_int, [_vp, _i64, _vp, _i64, _vp, _int, _i64, _i64, _i64, _vp]),
    "gsage_tanh_bwd": (_int, [_vp, _i64, _vp, _int, _i64, _vp, _i64, _i64, _i64, _vp]),
    "gsage_attn_merge_bwd": (_int, [_vp, _int, _i64, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _vp, _int, _i64,
                                    _i64, _i32, _i32, _vp, _vp, _vp]),
    "gsage_attn_merge_bwd2": (_int, [_vp, _int, _i64, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _vp, _int, _i64,
                                     _i64, _i32, _i32, _vp, _vp, _vp, _i64, _vp]),
    "gsage_head_l1": (_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _int, _i64, _vp, _vp]),
    "gsage_head_l1_scratch": (_int, [_i64, _i64]),
    "gsage_metric_f1": (_int, [_vp, _i64, _vp, _int, _int, _i64, _i64, _i32, _vp, _vp, _vp]),
    "gsage_metric_mae": (_int, [_vp, _vp, _i64, _vp, _vp]),
    "gsage_attn_bwd": (_int, [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _int, _i64, _vp, _i64, _i32, _i64, _i64,
                              _vp, _i64, _vp, _i64, _vp]),
}


class NativeLibraryError(RuntimeError):
    pass


_LIB = None


def lib():
    """Load libgsage_hip.so once; raise loudly if it is absent or incomplete."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            "libgsage_hip.so is missing (%s). Build it with `make hip` or "
            "`python -c 'import __graft_entry__ as g; g.build()'`; there is no fallback path."
            % LIB_PATH)
    try:
        L = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise NativeLibraryError("cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            raise NativeLibraryError("%s does not export %s (stale build?)" % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    if L.gsage_abi_version() != ABI_VERSION:
        raise NativeLibraryError("ABI version mismatch: library %d, binding %d"
                                 % (L.gsage_abi_version(), ABI_VERSION))
    _LIB = L
    return _LIB


def available():
    try:
        lib()
        return True
    except NativeLibraryError:
        return False


def check(rc, what=""):
    if rc != 0:
        msg = lib().gsage_last_error()
        raise RuntimeError("gsage %s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


def launch_count():
    return int(lib().gsage_launch_count())


class HopsDesc(ctypes.Structure):             # mirrors gsage_hops_desc (include/gsage.h)
    _fields_ = [("rowptr", _vp), ("col", _vp), ("n_rows", _i64), ("ids", _vp), ("B", _i64),
                ("n_hops", _i32), ("fan", _i32 * 5), ("max_deg", _u32), ("seed", _u64),
                ("call_ctr", _vp), ("call_base", _u64), ("rank", _u64), ("seed_queue", _vp),
                ("batch_idx", _vp), ("batch_base", _i64), ("n_batches", _i64), ("err_flag", _vp),
                ("sel", _vp), ("sel_stride", _i64), ("dense_adj", _vp), ("dense_ld", _i64)]


class AdamDesc(ctypes.Structure):             # mirrors gsage_adam_desc (include/gsage.h)
    _fields_ = [("p", _vp), ("g", _vp), ("m", _vp), ("v", _vp), ("n", _i64), ("partial", _vp),
                ("lr", _vp), ("step", _vp), ("beta1", _f32), ("beta2", _f32), ("eps", _f32),
                ("weight_decay", _f32), ("max_norm", _f32), ("norm_out", _vp),
                ("step_is_current", _i32), ("n_partial_ready", _i32), ("prep_descs", _vp),
                ("n_prep", _i32), ("tick1", _vp), ("inc1", _i64), ("tick2", _vp), ("inc2", _i64),
                ("norm_slots", _vp), ("reduce_descs", _vp), ("n_reduce", _i32)]


class RowAdamDesc(ctypes.Structure):          # mirrors gsage_row_adam (include/gsage.h)
    _fields_ = [("p", _vp), ("g", _vp), ("m", _vp), ("v", _vp), ("last", _vp), ("seen", _vp), ("hist", _vp),
                ("lr", _vp), ("step", _vp), ("n_rows", _i64), ("E", _i32), ("hist_cap", _i32), ("beta1", _f32),
                ("beta2", _f32), ("eps", _f32), ("weight_decay", _f32), ("max_norm", _f32), ("sorted_ids", _i32)]


class TailGatherDesc(ctypes.Structure):       # mirrors gsage_tail_gather_desc (include/gsage.h)
    _fields_ = [("table", _vp), ("ids", _vp), ("out", _vp), ("ld", _i64), ("out_ld", _i64), ("D", _i64),
                ("rows", _i64), ("n", _i32), ("n_workgroups", _i32)]


class WgradDesc(ctypes.Structure):            # mirrors gsage_wgrad_desc (include/gsage.h)
    _fields_ = [("dC", _vp), ("A", _vp), ("slabs", _vp), ("ldc", _i64), ("lda", _i64),
                ("a_gstride", _i64), ("M", _i64), ("Ntot", _i64), ("K", _i64), ("n_per_group", _i64),
                ("ldk", _i64), ("rows_per_split", _i64), ("a_rows", _vp)]


class CommandList(object):
    """Recorded gsage kernel launches, replayed back to back on a stream (include/gsage.h,
    "Command lists").  Usage:  with CommandList.record() as cl: <gsage kernel calls>;  cl.replay(stream)"""

    def __init__(self):
        self._h = None

    @classmethod
    def record(cls):
        return _Recorder(cls())

    def __len__(self):
        return int(lib().gsage_cmdlist_size(self._h)) if self._h else 0

    def replay(self, stream):
        check(lib().gsage_cmdlist_replay(self._h, stream), "cmdlist_replay")

    def elapsed_ms(self, slot_a, slot_b):
        """Time between two gsage_cmdlist_mark events of the last replay (blocks until mark b)."""
        ms = _f32(0.0)
        check(lib().gsage_cmdlist_elapsed(self._h, slot_a, slot_b, ctypes.byref(ms)), "cmdlist_elapsed")
        return float(ms.value)

    def __del__(self):
        if self._h and _LIB is not None:
            _LIB.gsage_cmdlist_destroy(self._h)
            self._h = None


class _Recorder(object):
    def __init__(self, cl):
        self.cl = cl

    def __enter__(self):
        check(lib().gsage_cmdlist_begin(), "cmdlist_begin")
        return self.cl

    def __exit__(self, et, ev, tb):
        h = _vp()
        rc = lib().gsage_cmdlist_end(ctypes.byref(h))
        if rc == 0:
            self.cl._h = h.value
        if et is None:
            check(rc, "cmdlist_end")
        return False


HOST_FN = ctypes.CFUNCTYPE(_int, _vp, _vp)      # gsage_host_fn: int fn(void *ctx, void *stream)


def host_call(fn, stream=None):
    """gsage_host_call: fn(stream_handle) -> None runs at this point of the list being recorded on this thread (at
    every replay, with the replay's stream -- the side stream inside a side section), or right now on `stream` when
    nothing is being recorded.  Returns the ctypes callback object: the CALLER keeps it alive as long as the list."""
    def tramp(_ctx, s):
        try:
            fn(s)
            return 0
        except Exception:                      # (an exception cannot cross the C frame)
            import traceback
            traceback.print_exc()
            return 1
    cb = HOST_FN(tramp)
    check(lib().gsage_host_call(ctypes.cast(cb, _vp), None, stream), "host_call")
    return cb


class NativeComm(object):
    """One RCCL communicator owned by libgsage_hip.so (include/gsage.h, "The step's collectives"): the collectives
    of a data-parallel step become nodes of the step's command list."""

    @staticmethod
    def load():
        """dlopen RCCL (the copy torch itself maps when there is one).  Separate from __init__ so that the ranks can
        agree that EVERY rank has it before any of them enters the collective ncclCommInitRank."""
        import torch
        cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        check(lib().gsage_comm_load(cand.encode() if os.path.exists(cand) else None), "comm_load")

    def __init__(self, rank, world, exchange_id):
        """exchange_id(bytes or None) -> bytes: hands rank 0's 128-byte id to every rank (e.g. a broadcast over the
        process group torch.distributed already has)."""
        self.load()
        buf = ctypes.create_string_buffer(128)
        err = None
        if rank == 0:
            try:
                check(lib().gsage_comm_unique_id(buf), "comm_unique_id")
            except Exception as e:             # (raised AFTER the exchange below: every rank must enter it, or the
                err = e                        #  others block in the broadcast while rank 0 moves on to the next collective)
        raw = exchange_id((b"\0" * 128 if err is not None else bytes(buf.raw)) if rank == 0 else None)
        if err is not None:
            raise err
        if len(raw) != 128 or raw == b"\0" * 128:
            raise NativeLibraryError("gsage_comm_unique_id failed on rank 0: no RCCL id to join")
        h = _vp()
        check(lib().gsage_comm_create(ctypes.create_string_buffer(raw, 128), rank, world, ctypes.byref(h)), "comm_create")
        self._h, self.rank, self.world = h.value, rank, world

    def all_reduce(self, t, average, stream):
        assert t.dtype.is_floating_point and t.element_size() == 4 and t.is_contiguous()
        check(lib().gsage_comm_all_reduce_f32(self._h, t.data_ptr(), t.numel(), 1 if average else 0, stream),
              "comm_all_reduce_f32")

    def all_gather(self, send, recv, stream):
        nbytes = send.numel() * send.element_size()
        assert send.is_contiguous() and recv.is_contiguous() and recv.numel() * recv.element_size() == nbytes * self.world
        check(lib().gsage_comm_all_gather(self._h, send.data_ptr(), recv.data_ptr(), nbytes, stream), "comm_all_gather")

    def group(self, begin, stream):
        check(lib().gsage_comm_group(self._h, 1 if begin else 0, stream), "comm_group")

    def close(self):
        if self._h and _LIB is not None:
            _LIB.gsage_comm_destroy(self._h)
        self._h = None


def device_info():
    arch = ctypes.create_string_buffer(64)
    cu, ws = _int(0), _int(0)
    rc = lib().gsage_device_info(arch, 64, ctypes.byref(cu), ctypes.byref(ws))
    if rc != 0:
        return None
    return {"arch": arch.value.decode(), "cu_count": cu.value, "wave_size": ws.value}
