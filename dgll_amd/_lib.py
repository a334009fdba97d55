"""ctypes binding of libdgll_hip.so (the C ABI declared in include/dgll_hip.h).

There is no fallback: if the library is missing and cannot be built, importing dgll_amd raises.  The
in-tree location (dgll_amd/lib/libdgll_hip.so) is deliberate -- the driver records which in-tree .so
files the test processes mapped.
"""
import contextlib
import ctypes as C
import os

# torch bundles its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).  It must be mapped BEFORE
# libdgll_hip.so so that the library's NEEDED libamdhip64.so.7 binds to that same runtime; loaded the other way
# round the process ends up with two HIP runtimes and the second one sees no device.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdgll_hip.so")

OK = 0
F32, BF16 = 0, 1
REDUCE_SUM, REDUCE_MEAN = 0, 1
EPI_NONE, EPI_BIAS, EPI_RELU = 0, 1, 2


class BatchLoad(C.Structure):
    """include/dgll_hip.h: struct dgll_batch_load (dgll_hip_load_sampled_batch)."""
    _fields_ = [("staged_host", C.c_void_p), ("staged_entries", C.c_int64), ("staged_dev", C.c_void_p),
                ("pos_host", C.c_void_p), ("n_outer", C.c_int64), ("pos_bytes", C.c_int), ("pos_dev", C.c_void_p),
                ("indptr", C.c_void_p), ("indices", C.c_void_p),
                ("n_hops", C.c_int), ("rows", C.c_int64 * 8), ("seeds_off", C.c_int64), ("src_off", C.c_int64 * 8), ("ptr_off", C.c_int64 * 8),
                ("cache", C.c_void_p), ("ldc", C.c_int64), ("host", C.c_void_p), ("ldh", C.c_int64), ("slot", C.c_void_p),
                ("host_map", C.c_void_p), ("feat", C.c_int), ("dtype", C.c_int), ("miss_count", C.c_void_p),
                ("feat_out", C.c_void_p * 8), ("ld_feat", C.c_int64),
                ("reduced_out", C.c_void_p), ("ld_reduced", C.c_int64), ("reduce", C.c_int),
                ("ids_out", C.c_void_p),
                ("rowptr_out", C.c_void_p * 8), ("rowptr_cap", C.c_int64 * 8),
                ("labels", C.c_void_p), ("labels_out", C.c_void_p), ("labels_cap", C.c_int64), ("label_fill", C.c_int64),
                ("stage_map", C.c_void_p), ("stage_rows", C.c_void_p), ("ld_stage", C.c_int64), ("stage_cap", C.c_int64),
                ("stage_list", C.c_void_p), ("stage_count", C.c_void_p), ("stage_serial", C.c_uint), ("stage_blocks", C.c_int),
                ("upload_blocks", C.c_int)]


class DgllHipError(RuntimeError):
    pass


def _load():
    # Build when the library is missing or was built from different sources (content hash, not mtimes); hipcc
    # cross-compiles without a GPU.  A failed build raises: there is nothing to fall back to.
    from . import build as _build

    if not _build.is_current():
        _build.build()      # serialised across processes by a file lock; the library is linked to a temporary name and renamed
    if not os.path.exists(LIB_PATH):
        raise ImportError("libdgll_hip.so is missing (expected at %s); run `python -m dgll_amd.build`" % LIB_PATH)
    return C.CDLL(LIB_PATH)


lib = _load()

_vp, _i32, _i64, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_size_t

# name -> (restype, argtypes); mirrors include/dgll_hip.h one to one (tests/test_abi.py checks the header)
SIGNATURES = {
    "dgll_hip_abi_version": (_i32, []),
    "dgll_hip_last_error": (C.c_char_p, []),
    "dgll_hip_device_info": (_i32, [_i32, C.c_char_p, _i32, C.POINTER(_i32), C.POINTER(_i64)]),
    "dgll_hip_debug_tune": (_i32, [_i32, _i32]),
    "dgll_hip_csr_plan_create": (_i32, [_vp, _vp, _i64, _i64, _i32, C.POINTER(_vp)]),
    "dgll_hip_csr_plan_destroy": (None, [_vp]),
    "dgll_hip_csr_plan_workspace_bytes": (_sz, [_vp, _i32]),
    "dgll_hip_csr_plan_num_long_rows": (_i64, [_vp]),
    "dgll_hip_csr_plan_num_chunks": (_i64, [_vp]),
    "dgll_hip_spmm_csr": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _i64, _i32, _i64, _i64, _i32, _i32,
                                 _i32, _vp, _vp, _sz]),
    "dgll_hip_spmm_csr_ex": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _i64, _i32, _i64, _i64, _i32, _i32,
                                    _i32, _vp, _vp, _sz, _vp, _i32]),
    "dgll_hip_spmm_csr_gated": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _i64, _i32, _i64, _i64, _i32, _i32,
                                       _i32, _vp, _vp, _sz, _vp, _i32, _vp, _i64]),
    "dgll_hip_sage_fused_forward": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _i64, _i32, _vp, _i64, _vp, _i64, _i32,
                                           _vp, _i32, _vp, _i64, _i32, _vp, _i64, _i64, _i64, _vp, _sz]),
    "dgll_hip_sddmm_csr": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i32, _vp, _i64, _i32]),
    "dgll_hip_gat_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _i64, _i32, _i32,
                                C.c_float, _i32, _i32, _vp, _sz]),
    "dgll_hip_gat_fwd_ex": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _i64, _i32, _i32,
                                   C.c_float, _i32, _vp, _sz, _i32, _i32]),
    "dgll_hip_gat_bwd_rows": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp,
                                     _i64, _vp, _vp, _i64, _i32, _i32, C.c_float, _i32, _i32, _i32, _vp, _sz]),
    "dgll_hip_gat_bwd_rows_split": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _i64, _vp, _vp,
                                           _i64, _i32, _i32, C.c_float, _i32, _i32, _vp, _vp, _sz]),
    "dgll_hip_gat_bwd_cols": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp,
                                     _i32, _i64, _i32, _i32, C.c_float, _i32, _vp, _sz]),
    "dgll_hip_gat_fwd_strided": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i32, _vp, _i64, _i32, _vp, _i64, _i32, _i32,
                                        C.c_float, _i32, _vp, _sz]),
    "dgll_host_sampler_pool_create": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _i32, C.c_uint64, C.c_uint64, _i32, _i32, _vp, _i64,
                                             _i64, _vp, _vp, _vp, _i32]),
    "dgll_host_sampler_pool_next": (_i32, [_vp, _vp, _vp]),
    "dgll_host_sampler_pool_release": (_i32, [_vp, _i32]),
    "dgll_host_sampler_pool_destroy": (_i32, [_vp]),
    "dgll_hip_gat_fwd_rowscore": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i32, _vp, _i64, _i64, _i32, _i32, C.c_float, _i32,
                                         _vp, _sz, _i32, _i32]),
    "dgll_hip_gat_bwd_rows_rowscore": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _i64,
                                              _vp, _i32, _vp, _i64, _i64, _i32, _i32, C.c_float, _i32, _vp, _sz]),
    "dgll_hip_gat_bwd_rows_strided": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i32, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _i64,
                                             _vp, _i32, _vp, _i64, _i32, _i32, C.c_float, _i32, _vp, _sz]),
    "dgll_hip_gat_bwd_cols_strided": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _i32, _vp, _i64, _vp, _i32, _i64,
                                             _i32, _i32, C.c_float, _vp, _sz, _vp, _vp, _vp]),
    "dgll_hip_gat_bwd_strided": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i32, _vp, _vp, _i64, _vp, _i64,
                                        _i32, _vp, _vp, _i64, _vp, _i32, _vp, _i64, _vp, _vp, _i64, _i64, _i32, _i32, C.c_float,
                                        _i32, _vp, _sz]),
    "dgll_hip_gat_workspace_bytes": (_sz, [_vp, _i32, _i32]),
    "dgll_hip_gat_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i32,
                                _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _i64, _i32, _i32, C.c_float, _i32,
                                _i32, _vp, _sz]),
    "dgll_hip_gather_rows": (_i32, [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp]),
    "dgll_hip_pack_weight_bf16": (_i32, [_vp, _vp, _i32, _i64, _i64, _i32, _i32, _vp, _i64, _i32]),
    "dgll_hip_gather_rows_mapped": (_i32, [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp]),
    "dgll_hip_aggregate_rows_mapped": (_i32, [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp]),
    "dgll_hip_translate_positions": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i32, _vp]),
    "dgll_hip_load_sampled_batch": (_i32, [_vp, _vp]),
    "dgll_hip_expand_rows": (_i32, [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _i32]),
    "dgll_host_sample_neighbors": (_i32, [_vp, C.POINTER(_i32), _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _i64,
                                          C.POINTER(_i64)]),
    "dgll_host_translate_neighbors": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "dgll_host_mt_seed": (_i32, [_vp, _i64, _vp, C.POINTER(_i32)]),
    "dgll_host_sample_batch_seeded": (_i32, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32]),
    "dgll_hip_adam_flat": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _i64,
                                  C.c_float, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dgll_hip_gemm_f32": (_i32, [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _i32]),
    "dgll_hip_mm_f32": (_i32, [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _i32, _vp, _i64]),
    "dgll_hip_mm2_f32": (_i32, [_vp, _vp, _i64, _vp, _i64, _i32, _vp, _i64, _vp, _i64, _i32, _vp, _i64, _i64, _i32, _vp, _i32, _vp, _i64,
                                _vp, _i64]),
    "dgll_hip_grad_weight_f32_workspace": (_i64, [_i32, _i32, _i32]),
    "dgll_hip_grad_weight_f32": (_i32, [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _i64, _i32]),
    "dgll_hip_transform_bf16": (_i32, [_vp, _vp, _i64, _i32, _vp, _i64, _vp, _i64, _i32, _vp, _i64, _i32, _vp, _i64, _vp,
                                       _i64, _i32, _i64, _i32, _i32, _vp]),
    "dgll_hip_transform_bf16_gated": (_i32, [_vp, _vp, _i64, _i32, _vp, _i64, _vp, _i64, _i32, _vp, _i64, _i32, _vp, _i64,
                                             _vp, _i64, _i32, _i64, _i32, _i32, _vp, _vp, _i64, _vp]),
    "dgll_hip_transform_bf16_add": (_i32, [_vp, _vp, _i64, _i32, _vp, _i64, _vp, _i64, _i32, _vp, _i64, _i32, _vp, _i64,
                                           _vp, _i64, _i32, _i64, _i32, _i32, _vp, _vp, _i64, _vp, _vp, _i64]),
    "dgll_hip_transform_bf16_bits": (_i32, [_vp, _vp, _i64, _i32, _vp, _i64, _vp, _i64, _i32, _vp, _i64, _i32, _vp, _i64, _i64, _i32,
                                            _i32, _vp, _vp, _i64, _vp, _i64, _vp, _i64]),
    "dgll_hip_transform_bf16_dual": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _i64, _i32, _vp, _i64, _vp, _i64, _i64, _i32]),
    "dgll_hip_grad_weight_workspace": (_i64, [_i32, _i32, _i32]),
    "dgll_hip_grad_weight_bf16": (_i32, [_vp, _vp, _i64, _i32, _vp, _i64, _i32, _vp, _i64, _i32, _i64, _vp, _i64, _i32, _vp, _i64,
                                         _vp, _i64]),
    "dgll_hip_grad_weight_bf16_tr": (_i32, [_vp, _vp, _i64, _i32, _vp, _i64, _i32, _vp, _i64, _i32, _i64, _vp, _i64, _i32, _vp, _i64,
                                            _vp, _i64]),
    "dgll_hip_softmax_xent": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _i64, _vp, _i64, _i32]),
    "dgll_hip_softmax_xent_soft": (_i32, [_vp, _vp, _i64, _i32, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _i32]),
    "dgll_hip_softmax_xent_ex": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _i32, _i32]),
    "dgll_hip_xent_reduce": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "launch_gcn_fused_kernel": (None, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32]),
    "launch_gcn_fused_kernel_backward_optimized": (None, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32,
                                                          _i32, _i32]),
    "dgll_hip_gcn_fused_forward": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _sz]),
    "dgll_hip_gcn_fused_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "dgll_hip_segment_max": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _i32, _i64, _i32]),
    "dgll_hip_segment_max_bwd": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _i64, _i32]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here == the library does not export a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args


# ---- launch plumbing shared by the Python wrappers ---------------------------------------------------------------------------
# A wrapper needs two things from torch per launch: the raw stream the caller is on and the guarantee that the tensor's device is
# current.  torch.cuda.current_stream() builds a Stream object (2-4 us) and `with torch.cuda.device(d)` a context object (3-4 us):
# of the ~25 us a launch costs through Python that is a third, and the consumer thread of the mini-batch pipeline issues ~80 launches
# per 2 ms batch.  Both are replaced by their cheap cores.
_NULL_CTX = contextlib.nullcontext()


def raw_stream(device):
    """hipStream_t (an int) of torch's current stream on `device`."""
    idx = device.index
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device() if idx is None else idx)


def on_device(device):
    """Context that makes `device` current for the launch: a shared no-op object when it already is."""
    idx = device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NULL_CTX
    return torch.cuda.device(idx)


def last_error():
    msg = lib.dgll_hip_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(code, what):
    if code != OK:
        raise DgllHipError("%s failed (code %d): %s" % (what, code, last_error()))


def device_info(device=0):
    name = C.create_string_buffer(64)
    cus, mem = _i32(0), _i64(0)
    check(lib.dgll_hip_device_info(device, name, 64, C.byref(cus), C.byref(mem)), "dgll_hip_device_info")
    return {"arch": name.value.decode(), "compute_units": cus.value, "global_mem_bytes": mem.value}
