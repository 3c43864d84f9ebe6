"""ctypes binding of libathena_mp.so -- the C ABI declared in include/athena_mp.h.

The library is hand-written HIP for gfx950; there is NO CPU fallback: if the shared object is
missing or a call fails, an exception is raised (AthenaMPError), never a silent detour.

torch is used only as the owner of device memory and streams: tensors are passed as raw device
pointers (`tensor.data_ptr()`), the current torch stream as the hipStream_t.
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("ATHENA_MP_LIB", "libathena_mp.so"))
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "athena_mp.h")


class AthenaMPError(RuntimeError):
    """A C-ABI call returned non-zero (the Fortran wrapper would call stop_program)."""


_lib = None

_i32, _i64, _vp, _f32 = C.c_int32, C.c_int64, C.c_void_p, C.c_float

# name -> argtypes (restype is int for everything except last_error)
_PROTOS = {
    "athena_mp_init": [C.c_int],
    "athena_mp_initialized": [],
    "athena_mp_finalize": [],
    "athena_mp_set_stream": [_vp],
    "athena_mp_synchronize": [],
    "athena_mp_version": [],
    "athena_mp_malloc": [C.POINTER(_vp), C.c_uint64],
    "athena_mp_free": [_vp],
    "athena_mp_memcpy_h2d": [_vp, _vp, C.c_uint64],
    "athena_mp_memcpy_d2h": [_vp, _vp, C.c_uint64],
    "athena_mp_memset_zero": [_vp, C.c_uint64],
    "athena_mp_graph_create": [_i32, _i32, _i64, _vp, _vp, _i32, _vp, _vp, C.POINTER(_vp)],
    "athena_mp_csr_from_edges": [_i32, _i64, _vp, _i32, _vp, _vp, _i64, _vp],
    "athena_mp_graph_create_from_edges": [_i32, _i64, _vp, _i32, _i32, _vp, _vp, _i64, _vp, C.POINTER(_vp)],
    "athena_mp_graph_export": [_vp, _i32, _vp, _i64, _vp],
    "athena_mp_graph_destroy": [_vp],
    "athena_mp_graph_key": [_i32, _i64, _vp, _vp, C.POINTER(C.c_uint64)],
    "athena_mp_graph_acquire": [_i32, _i64, _vp, _vp, _i32, C.POINTER(_vp)],
    "athena_mp_graph_release": [_vp],
    "athena_mp_graph_evict": [_vp],
    "athena_mp_graph_cache_stats": [C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)],
    "athena_mp_graph_dims": [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i64), C.POINTER(_i32)],
    "athena_mp_kipf_propagate_fwd": [_vp, _i32, _vp, _vp],
    "athena_mp_kipf_propagate_bwd": [_vp, _i32, _vp, _vp, _i32],
    "athena_mp_kipf_propagate_act_fwd": [_vp, _i32, _vp, _i32, _vp],
    "athena_mp_reverse_kipf_propagate_fwd": [_vp, _i32, _vp, _vp],
    "athena_mp_reverse_kipf_propagate_partial": [_vp, _i32, _vp, _vp],
    "athena_mp_reverse_kipf_propagate_partial_val": [_vp, _i32, _vp, _vp],
    "athena_mp_kipf_propagate_bwd_dual": [_vp, _i32, _vp, _vp, _vp],
    "athena_mp_kipf_propagate_fwd_dual": [_vp, _i32, _vp, _vp, _vp],
    "athena_mp_gather_rows": [_i64, _i32, _vp, _vp, _vp],
    "athena_mp_gemm_fwd": [_i64, _i32, _i32, _vp, _vp, _vp, _i32, _vp],
    "athena_mp_gemm_dw": [_i64, _i32, _i32, _vp, _vp, _vp],
    "athena_mp_gemm_dx": [_i64, _i32, _i32, _vp, _vp, _vp],
    "athena_mp_kipf_layer_fwd": [_vp, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _vp],
    "athena_mp_kipf_layer_bwd_x": [_vp, _i32, _i32, _vp, _vp, _i32, _vp],
    "athena_mp_kipf_layer_bwd": [_vp, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _vp],
    "athena_mp_pull_gemm": [_vp, _i32, _i32, _vp, _vp, _i32, _vp],
    "athena_mp_activation_fwd": [_i32, _i64, _vp, _vp],
    "athena_mp_activation_bwd": [_i32, _i64, _vp, _vp, _vp],
    "athena_mp_axpy": [_i64, _f32, _vp, _vp],
    "athena_mp_device_copy": [_vp, _vp, C.c_uint64],
    "athena_mp_duvenaud_propagate_fwd": [_vp, _i32, _i32, _vp, _vp, _vp],
    "athena_mp_duvenaud_propagate_bwd_x": [_vp, _i32, _i32, _vp, _vp],
    "athena_mp_duvenaud_propagate_bwd_e": [_vp, _i32, _i32, _vp, _vp],
    "athena_mp_duvenaud_update_fwd": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp],
    "athena_mp_duvenaud_update_act_fwd": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp],
    "athena_mp_duvenaud_update_bwd_a": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp],
    "athena_mp_duvenaud_update_bwd_w": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp],
    "athena_mp_duvenaud_update_readout_fwd": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _vp],
    "athena_mp_duvenaud_update_bwd": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp],
    "athena_mp_duvenaud_update_bwd_split": [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    "athena_mp_duvenaud_readout_update_bwd": [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp,
                                              _vp, _vp, _vp, _i32, _i32, _vp],
    "athena_mp_duvenaud_update_readout_fwd_split": [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _vp],
    "athena_mp_softmax_segsum_fwd": [_i32, _i64, _i32, _vp, _vp, _vp, _vp, _i32],
    "athena_mp_softmax_segsum_bwd": [_i32, _i64, _i32, _vp, _vp, _vp, _vp],
    "athena_mp_activation_param_fwd": [_i32, _i64, _f32, _f32, _f32, _vp, _vp],
    "athena_mp_activation_param_bwd": [_i32, _i64, _f32, _f32, _f32, _vp, _vp, _vp],
    "athena_mp_swish_fwd": [_i64, _f32, _vp, _vp],
    "athena_mp_swish_bwd": [_i64, _f32, _vp, _vp, _vp],
    "athena_mp_softmax_fwd": [_i64, _i32, _vp, _vp],
    "athena_mp_softmax_bwd": [_i64, _i32, _vp, _vp, _vp],
    "athena_mp_concat_fwd": [_i64, _i32, _i32, _vp, _vp, _vp],
    "athena_mp_concat_bwd": [_i64, _i32, _i32, _vp, _vp, _vp],
    "athena_mp_mse_loss": [_i64, _vp, _vp, _vp, _vp],
    "athena_mp_clip": [_i64, _vp, _i32, _f32, _f32, _i32, _f32],
    "athena_mp_sgd_step": [_i64, _f32, _f32, _i32, _i32, _f32, _f32, _vp, _vp, _vp],
    "athena_mp_adam_step": [_i64, _f32, _f32, _f32, _f32, _i32, _i32, _f32, _f32, _i32, _vp, _vp, _vp, _vp],
    "athena_mp_segment_sum": [_i32, _i64, _i32, _vp, _vp, _vp, _i32],
    "athena_mp_segment_sum_bwd": [_i32, _i64, _i32, _vp, _vp, _vp],
    "athena_mp_duvenaud_readout_fwd": [_i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32],
    "athena_mp_duvenaud_readout_bwd": [_i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32],
    "athena_mp_gno_aggregate_fwd": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp],
    "athena_mp_gno_aggregate_bwd_x": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp],
    "athena_mp_gno_aggregate_bwd_x_pull": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp],
    "athena_mp_gno_aggregate_bwd": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i32)],
    "athena_mp_gno_aggregate_bwd_theta": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp],
    "athena_mp_gno_aggregate_bwd_coords": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp],
    "athena_mp_gno_saved_bytes": [_vp, _i32, _i32, _i32, _i32, C.POINTER(C.c_int64)],
    "athena_mp_gno_aggregate_fwd_save": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp],
    "athena_mp_gno_aggregate_bwd_theta_saved": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    "athena_mp_comm_unique_id": [_vp],
    "athena_mp_comm_create": [_i32, _i32, _vp, C.POINTER(_vp)],
    "athena_mp_comm_create_from_file": [_i32, _i32, C.c_char_p, C.POINTER(_vp)],
    "athena_mp_comm_destroy": [_vp],
    "athena_mp_comm_info": [_vp, C.POINTER(_i32), C.POINTER(_i32), C.c_char_p, _i32],
    "athena_mp_comm_stats": [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i64), C.POINTER(_i64), _vp, _i32],
    "athena_mp_set_stall_handler": [_vp],
    "athena_mp_comm_barrier": [_vp],
    "athena_mp_allreduce_start": [_vp, _vp, _i64],
    "athena_mp_allreduce_finish": [_vp],
    "athena_mp_allreduce": [_vp, _vp, _i64],
    "athena_mp_shard_create": [_vp, _i32, _i64, _vp, _vp, C.POINTER(_vp)],
    "athena_mp_shard_create_edges": [_vp, _i32, _i64, _vp, _vp, C.POINTER(_vp)],
    "athena_mp_shard_edge_cols": [_vp, C.POINTER(_i32)],
    "athena_mp_shard_edge_reduce": [_vp, _i32, _vp],
    "athena_mp_shard_destroy": [_vp],
    "athena_mp_shard_dims": [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)],
    "athena_mp_shard_graph": [_vp, _i32, C.POINTER(_vp)],
    "athena_mp_shard_info": [_vp, C.POINTER(_i32), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_i64)],
    "athena_mp_shard_export": [_vp, _i32, _vp, _i64, C.POINTER(_i64)],
    "athena_mp_halo_start": [_vp, _i32, _i32, _vp],
    "athena_mp_halo_finish": [_vp, _i32],
    "athena_mp_halo_reduce_start": [_vp, _i32, _i32, _vp],
    "athena_mp_halo_reduce_finish": [_vp, _i32, _vp],
    "athena_mp_resident_mode": [_i32],
    "athena_mp_resident_acquire": [_vp, C.c_uint64, _i32, C.POINTER(_vp)],
    "athena_mp_resident_release": [_vp, _i32],
    "athena_mp_resident_flush": [_vp],
    "athena_mp_resident_drop": [_vp],
    "athena_mp_resident_stats": [C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)],
    "athena_mp_kipf_propagate_fwd_host": [_vp, _i32, _vp, _vp],
    "athena_mp_kipf_propagate_bwd_host": [_vp, _i32, _vp, _vp, _i32],
    "athena_mp_gemm_fwd_host": [_i64, _i32, _i32, _vp, _vp, _vp, _i32, _vp],
    "athena_mp_gemm_dw_host": [_i64, _i32, _i32, _vp, _vp, _vp],
    "athena_mp_gemm_dx_host": [_i64, _i32, _i32, _vp, _vp, _vp],
    "athena_mp_activation_fwd_host": [_i32, _i64, _vp, _vp],
    "athena_mp_activation_bwd_host": [_i32, _i64, _vp, _vp, _vp],
    "athena_mp_duvenaud_propagate_fwd_host": [_vp, _i32, _i32, _vp, _vp, _vp],
    "athena_mp_duvenaud_propagate_bwd_x_host": [_vp, _i32, _i32, _vp, _vp],
    "athena_mp_duvenaud_propagate_bwd_e_host": [_vp, _i32, _i32, _vp, _vp],
    "athena_mp_duvenaud_update_fwd_host": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp],
    "athena_mp_duvenaud_update_bwd_a_host": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp],
    "athena_mp_duvenaud_update_bwd_w_host": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp],
    "athena_mp_duvenaud_update_act_fwd_host": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp],
    "athena_mp_duvenaud_readout_fwd_host": [_i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32],
    "athena_mp_duvenaud_readout_bwd_host": [_i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32],
    "athena_mp_softmax_fwd_host": [_i64, _i32, _vp, _vp],
    "athena_mp_softmax_bwd_host": [_i64, _i32, _vp, _vp, _vp],
    "athena_mp_swish_fwd_host": [_i64, _f32, _vp, _vp],
    "athena_mp_swish_bwd_host": [_i64, _f32, _vp, _vp, _vp],
    "athena_mp_activation_param_fwd_host": [_i32, _i64, _f32, _f32, _f32, _vp, _vp],
    "athena_mp_activation_param_bwd_host": [_i32, _i64, _f32, _f32, _f32, _vp, _vp, _vp],
    "athena_mp_softmax_segsum_fwd_host": [_i32, _i64, _i32, _vp, _vp, _vp, _vp, _i32],
    "athena_mp_softmax_segsum_bwd_host": [_i32, _i64, _i32, _vp, _vp, _vp, _vp],
    "athena_mp_gno_aggregate_fwd_host": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp],
    "athena_mp_gno_aggregate_bwd_x_host": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp],
    "athena_mp_gno_aggregate_bwd_theta_host": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp],
    "athena_mp_gno_aggregate_bwd_coords_host": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp],
    "athena_mp_duvenaud_update_readout_fwd_host": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _vp],
    "athena_mp_duvenaud_update_bwd_pair_host": [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp],
    "athena_mp_gno_aggregate_bwd_pair_host": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp],
    "athena_mp_pair_stats": [C.POINTER(_i64), C.POINTER(_i64)],
}


def declared_symbols():
    """Every athena_mp_* function the public header declares."""
    with open(HEADER_PATH) as fh:
        text = fh.read()
    return sorted(set(re.findall(r"\b(athena_mp_[a-z0-9_]+)\s*\(", text)))


def load():
    """Load the shared library (torch first, so both share one HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AthenaMPError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(athena_amd/csrc/build.sh). There is no CPU fallback."
        )
    try:
        import torch  # noqa: F401  (loads torch's libamdhip64 first: same SONAME, one runtime)
    except Exception:  # pragma: no cover - torch is plumbing only
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    lib.athena_mp_last_error.restype = C.c_char_p
    lib.athena_mp_last_error.argtypes = []
    for name, argtypes in _PROTOS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = C.c_int
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().athena_mp_last_error()
        raise AthenaMPError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def call(name, *args):
    lib = load()
    check(getattr(lib, name)(*args), name)


_initialised_device = None


def init(device=0):
    global _initialised_device
    if _initialised_device != device:
        call("athena_mp_init", int(device))
        _initialised_device = device


def use_torch_stream():
    """Enqueue subsequent kernels on torch's current stream."""
    import torch

    s = torch.cuda.current_stream().cuda_stream
    call("athena_mp_set_stream", C.c_void_p(s))
