"""ctypes binding of libjorldy_hip.so (the C ABI declared in include/jorldy_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a call fails,
an exception is raised.  Build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C jorldy_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libjorldy_hip.so")

JH_U8, JH_F32, JH_I64, JH_F64, JH_I32 = 0, 1, 2, 3, 4
JH_TD_DOUBLE, JH_TD_PER = 1, 2
JH_C51_DOUBLE, JH_C51_PER, JH_C51_SHIFT_MAX = 1, 2, 4


class JhError(RuntimeError):
    pass


class ColDesc(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("elems", C.c_int64)]


_vp, _i32, _i64, _f32, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double
_pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes).  Status-returning functions use c_int and are checked.
_PROTOS = {
    "jh_abi_version": (C.c_int, []),
    "jh_stream_abort_capture": (C.c_int, [_vp]),
    "jh_last_error": (C.c_char_p, []),
    "jh_device_count": (C.c_int, []),
    "jh_ctx_create": (C.c_int, [C.c_int, _pp]),
    "jh_ctx_destroy": (None, [_vp]),
    "jh_ctx_sync": (C.c_int, [_vp, _vp]),
    "jh_ctx_local_cpulist": (C.c_int, [_vp, C.c_char_p, _i64]),
    "jh_prof_enable": (C.c_int, [_i32]),
    "jh_prof_report": (C.c_int, [C.c_char_p, _i64]),
    "jh_prof_calibrate": (C.c_int, [_i32, _vp]),
    "jh_calib_stream": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "jh_host_wait_marks": (C.c_int, [_vp, _vp, _i32, _f32, _f64]),
    "jh_host_wait_words": (C.c_int, [_vp, _vp, _i32, C.c_uint32, _f64]),
    "jh_np_legacy_shuffles": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "jh_pinned_alloc": (C.c_int, [_vp, _i64, _pp, _pp]),
    "jh_pinned_free": (None, [_vp]),
    "jh_store_create": (C.c_int, [_vp, _i64, _i32, C.POINTER(ColDesc), _pp]),
    "jh_store_destroy": (None, [_vp]),
    "jh_store_push": (C.c_int, [_vp, _i64, _pp, _vp]),
    "jh_store_push_device": (C.c_int, [_vp, _i64, _pp, _vp]),
    "jh_store_stage_begin": (C.c_int, [_vp, _i64, _pp]),
    "jh_store_stage_commit": (C.c_int, [_vp, _vp]),
    "jh_store_write_rows": (C.c_int, [_vp, _i64, _vp, _pp, _vp]),
    "jh_store_gather": (C.c_int, [_vp, _i64, _vp, _i64, _i32, C.POINTER(_i32), _pp, C.POINTER(_i32), _vp]),
    "jh_store_col_ptr": (_vp, [_vp, _i32]),
    "jh_store_size": (_i64, [_vp]),
    "jh_store_index": (_i64, [_vp]),
    "jh_store_capacity": (_i64, [_vp]),
    "jh_store_clear": (None, [_vp]),
    "jh_store_set_position": (C.c_int, [_vp, _i64, _i64]),
    "jh_per_create": (C.c_int, [_vp, _i64, _f64, _pp]),
    "jh_per_destroy": (None, [_vp]),
    "jh_per_push": (C.c_int, [_vp, _i64, _vp, _vp]),
    "jh_per_push_device": (C.c_int, [_vp, _i64, _vp, _vp]),
    "jh_per_update": (C.c_int, [_vp, _i64, _vp, _vp, _i32, _vp]),
    "jh_per_sample": (C.c_int, [_vp, _i64, _f64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "jh_per_state": (C.c_int, [_vp, C.POINTER(_f64), C.POINTER(_f64), C.POINTER(_i64), C.POINTER(_i64), _vp]),
    "jh_per_tree_ptr": (_vp, [_vp]),
    "jh_per_shard_stats": (C.c_int, [_vp, _i64, _vp, _vp]),
    "jh_per_weights_sharded": (C.c_int, [_vp, _i64, _f64, _vp, _i32, _vp, _vp, _vp]),
    "jh_per_tree_size": (_i64, [_vp]),
    "jh_per_load": (C.c_int, [_vp, _vp, _f64, _i64, _i64]),
    "jh_per_dump": (C.c_int, [_vp, _vp, _vp]),
    "jh_gae": (C.c_int, [_vp, _i32, _i32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "jh_ppo_minibatch_rows": (C.c_int, [_vp, _i64, _vp, _i32, _vp, _vp, _vp, _vp]),
    "jh_mean_f32": (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
    "jh_logp_discrete": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    "jh_logp_continuous": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp]),
    "jh_ppo_loss_discrete": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _vp, _vp, _vp, _vp]),
    "jh_ppo_loss_continuous": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "jh_normal_fill": (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
    "jh_value_act": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "jh_td_loss": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _vp]),
    "jh_c51_loss": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "jh_pponet_param_count": (_i64, [_i32, _i32, _i32, _i32]),
    "jh_pponet_create": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, C.c_uint64, _pp]),
    "jh_pponet_destroy": (None, [_vp]),
    "jh_pponet_set_hyper": (C.c_int, [_vp, _f64, _f64, _f64, _f64, _f64, _vp]),
    "jh_pponet_set_lr": (C.c_int, [_vp, _f64, _vp]),
    "jh_pponet_hyper_ptr": (_vp, [_vp]),
    "jh_pponet_act_rng": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), _i32]),
    "jh_pponet_forward": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "jh_pponet_backward": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "jh_pponet_adam_step": (C.c_int, [_vp, _f32, _vp, _vp]),
    "jh_pponet_ppo_update": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _i32, _vp, _vp]),
    "jh_pponet_ppo_update_rows": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _i32, _vp, _vp]),
    "jh_ppo_loss_deferred": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "jh_ppo_loss_packed": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "jh_ppo_critic_select_strided": (C.c_int, [_vp, _i32, _vp, _f32, _f32, _vp, _i32, _vp, _vp, _vp, _vp]),
    "jh_heads_unpack": (C.c_int, [_vp, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _vp]),
    "jh_policy_act_discrete": (C.c_int, [_vp, _i32, _i32, _vp, _i32, C.c_uint64, C.c_uint64, _i32, _vp, _vp]),
    "jh_ppo_critic_select_rows": (C.c_int, [_vp, _i32, _vp, _f32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "jh_pponet_ppo_update_dp_begin": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _vp, _vp]),
    "jh_pponet_ppo_update_dp_end": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _f32, _f32, _vp, _vp]),
    "jh_pponet_ppo_update_dp_end_peer": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, C.c_float, C.c_float, _vp, _vp]),
    "jh_pponet_act_discrete": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp]),
    "jh_pponet_act_continuous": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "jh_control_create": (C.c_int, [_i32, _i32, _i32, C.c_uint64, _pp]),
    "jh_control_destroy": (None, [_vp]),
    "jh_control_obs": (C.c_int, [_vp, _vp]),
    "jh_control_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "jh_collector_create_control": (C.c_int, [_vp, _vp, _vp, _vp, C.POINTER(_i32), _pp]),
    "jh_collector_create": (C.c_int, [_vp, _vp, _vp, _vp, C.POINTER(_i32), _pp]),
    "jh_collector_create_env": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.POINTER(_i32), _pp]),
    "jh_collector_destroy": (None, [_vp]),
    "jh_rbnet_param_count_for": (_i64, [_i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32]),
    "jh_rbnet_create": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _pp]),
    "jh_rbnet_destroy": (None, [_vp]),
    "jh_rbnet_param_count": (_i64, [_vp]),
    "jh_rbnet_segment_count": (_i32, []),
    "jh_rbnet_segment": (C.c_int, [_vp, _i32, C.POINTER(_i64), C.POINTER(_i32), C.POINTER(_i32)]),
    "jh_rbnet_noise_len": (_i64, [_vp]),
    "jh_rbnet_set_hyper": (C.c_int, [_vp, _f64, _f64, _f64, _f64, _i64, _i32, _vp]),
    "jh_rbnet_optim_step": (C.c_int, [_vp, _i32, _f32, _vp]),
    "jh_rbnet_set_lr": (C.c_int, [_vp, _f64, _vp]),
    "jh_rbnet_sync_target": (C.c_int, [_vp, _vp]),
    "jh_rbnet_forward": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _vp, _vp, _vp]),
    "jh_rbnet_forward_keep": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp]),
    "jh_rbnet_learn_forward": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "jh_rbnet_prepare_noise": (C.c_int, [_vp, _vp, _vp]),
    "jh_rbnet_learn_trunk": (C.c_int, [_vp, _vp, _i32, _i32, _vp]),
    "jh_rbnet_learn_heads": (C.c_int, [_vp, _i32, _vp, _vp, _vp]),
    "jh_rbnet_backward": (C.c_int, [_vp, _vp, _vp]),
    "jh_rbnet_backward_deferred": (C.c_int, [_vp, _vp, _vp]),
    "jh_rbnet_flush_grads": (C.c_int, [_vp, _vp]),
    "jh_rbnet_learn_heads_raw": (C.c_int, [_vp, _i32, _vp, _vp]),
    "jh_rbnet_c51_step": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "jh_rbnet_adam_step": (C.c_int, [_vp, _vp]),
    "jh_tgemm_dense": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _i32, _vp, _vp]),
    "jh_tgemm_dense_group": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "jh_tgemm_set_cfg": (C.c_int, [C.c_char_p]),
    "jh_ring_create": (C.c_int, [_vp, _i64, _i32, C.POINTER(ColDesc), _i32, _pp]),
    "jh_ring_destroy": (None, [_vp]),
    "jh_ring_produce": (C.c_int, [_vp, _i64, _pp, _vp, _i32]),
    "jh_ring_drain": (C.c_int, [_vp, _vp, _vp, _i64, _vp, C.POINTER(_i64)]),
    "jh_ring_consume_host": (C.c_int, [_vp, _i64, _pp, _vp, C.POINTER(_i64)]),
    "jh_ring_reclaim": (C.c_int, [_vp, _i32]),
    "jh_ring_stats": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_f64)]),
    "jh_feed_create": (C.c_int, [_vp, _i32, _i32, _i64, _i32, _f32, _i64, _i64, _pp]),
    "jh_feed_destroy": (None, [_vp]),
    "jh_feed_tick": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i32), _vp]),
    "jh_feed_push_stacks": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "jh_feed_push_frames": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "jh_feed_emit": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _f64, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i32), _vp]),
    "jh_feed_state": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i64), _vp]),
    "jh_feed_state_bytes": (_i64, [_vp]),
    "jh_feed_save": (C.c_int, [_vp, _vp, _i64, _vp]),
    "jh_feed_load": (C.c_int, [_vp, _vp, _i64, _vp]),
    "jh_collector_stats": (C.c_int, [_vp, C.POINTER(_f64), C.POINTER(_f64), _i32]),
    "jh_collector_run": (C.c_int, [_vp, _i32, _i32, _vp]),
    "jh_collector_stats_detail": (C.c_int, [_vp, C.POINTER(_f64)]),
    "jh_collector_set_capture": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64]),
    "jh_collector_prelaunch": (C.c_int, [_vp, _i32, _vp]),
    "jh_collector_begin": (C.c_int, [_vp, _i32, _vp]),
    "jh_collector_loop": (C.c_int, [_vp, _i32, _vp]),
    "jh_collector_set_ride_along": (C.c_int, [_vp, _i32, _vp, _vp, _i64]),
    "jh_cartpole_create": (C.c_int, [_i32, C.c_uint64, _pp]),
    "jh_cartpole_destroy": (None, [_vp]),
    "jh_cartpole_obs": (C.c_int, [_vp, _vp]),
    "jh_cartpole_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "jh_comm_unique_id": (C.c_int, [_vp]),
    "jh_comm_create": (C.c_int, [_vp, _i32, _i32, _vp, _pp]),
    "jh_comm_destroy": (None, [_vp]),
    "jh_comm_info": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "jh_comm_allreduce_mean_f32": (C.c_int, [_vp, _vp, _i64, _vp]),
    "jh_comm_broadcast": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "jh_comm_allgather_f64": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "jh_peer_create": (C.c_int, [_vp, _i32, _i32, _i64, _vp]),
    "jh_peer_handle": (C.c_int, [_vp, _vp]),
    "jh_peer_connect": (C.c_int, [_vp, _vp]),
    "jh_peer_allreduce_mean_f32": (C.c_int, [_vp, _vp, _i64, _vp]),
    "jh_peer_allreduce_small_f32": (C.c_int, [_vp, _vp, _i32, _i32, _vp]),
    "jh_peer_status": (C.c_int, [_vp, _vp, _vp]),
    "jh_peer_destroy": (None, [_vp]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises JhError with build instructions if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise JhError(
            f"{LIB_PATH} not found: jorldy_amd has no CPU fallback. Build the HIP extension with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc, cross-compiles for gfx950)."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)  # AttributeError -> header / library out of sync: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.jh_abi_version() != 2:
        raise JhError(f"ABI mismatch: library reports {lib.jh_abi_version()}, binding expects 2")
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().jh_last_error()
        raise JhError(f"libjorldy_hip error {rc}: {msg.decode() if msg else '?'}")


def exported_names():
    return list(_PROTOS.keys())


# ----------------------------------------------------------------------------- context
_ctx_cache = {}


def ctx(device_index=None):
    """One jh_ctx per GPU, created lazily.  Raises JhError when no MI355X is visible."""
    import torch

    if device_index is None:
        device_index = torch.cuda.current_device() if torch.cuda.is_available() else 0
    if device_index not in _ctx_cache:
        lib = load()
        h = C.c_void_p()
        check(lib.jh_ctx_create(int(device_index), C.byref(h)))
        _ctx_cache[device_index] = h
    return _ctx_cache[device_index]


def stream_ptr():
    """Raw hipStream_t of torch's current stream (kernels are enqueued there, so they order with
    torch ops and torch.cuda.Event timing sees them)."""
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device (or host) pointer of a contiguous torch tensor / numpy array, or NULL for None."""
    if t is None:
        return C.c_void_p(0)
    if hasattr(t, "data_ptr"):
        assert t.is_contiguous(), "libjorldy_hip needs contiguous tensors"
        return C.c_void_p(t.data_ptr())
    # numpy
    assert t.flags["C_CONTIGUOUS"]
    return C.c_void_p(t.ctypes.data)
