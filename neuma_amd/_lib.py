"""ctypes binding of libneuma_hip.so (C ABI declared in include/neuma_hip.h).

There is no CPU fallback: if the library is missing or an entry point fails, an exception is raised.
"""
import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("NEUMA_HIP_LIB", _HERE / "lib" / "libneuma_hip.so"))

c_float_p = C.c_void_p  # device pointers travel as integers


def csrc_digest() -> str:
    """sha256 (first 16 hex digits) over the library's sources (csrc/*.hip, csrc/*.h, csrc/Makefile, include/neuma_hip.h, in name
    order): recorded next to every counter file under profiles/ so that bench.py can tell a counter pass of THIS build from a
    stale one (`pmc_stale`).  Content-based: works on the GPU box, where there is no .git."""
    import hashlib
    h = hashlib.sha256()
    src = sorted(list((_HERE / "csrc").glob("*.hip")) + list((_HERE / "csrc").glob("*.h")) + [_HERE / "csrc" / "Makefile"])
    for f in src + [_HERE.parent / "include" / "neuma_hip.h"]:
        h.update(f.name.encode() + b"\0" + f.read_bytes() + b"\0")
    return h.hexdigest()[:16]


class nm_mpm_cfg(C.Structure):
    _fields_ = [("num_grids", C.c_int32), ("dt", C.c_float), ("bound", C.c_int32), ("gravity", C.c_float * 3),
                ("eps", C.c_float), ("bc", C.c_int32)]


class nm_statics(C.Structure):
    _fields_ = [("vol", C.c_void_p), ("rho", C.c_void_p), ("clip_bound", C.c_void_p), ("enabled", C.c_void_p)]


class nm_particles(C.Structure):
    _fields_ = [("x", C.c_void_p), ("v", C.c_void_p), ("C", C.c_void_p), ("F", C.c_void_p), ("stress", C.c_void_p)]


class nm_mlp(C.Structure):
    _fields_ = [("w0", C.c_void_p), ("w1", C.c_void_p), ("w2", C.c_void_p)]


class nm_raster_cfg(C.Structure):
    _fields_ = [("image_height", C.c_int32), ("image_width", C.c_int32), ("tanfovx", C.c_float), ("tanfovy", C.c_float),
                ("bg", C.c_float * 3), ("scale_modifier", C.c_float), ("viewmatrix", C.c_float * 16),
                ("projmatrix", C.c_float * 16), ("sh_degree", C.c_int32), ("campos", C.c_float * 3),
                ("prefiltered", C.c_int32), ("debug", C.c_int32), ("tile_y0", C.c_int32), ("tile_y1", C.c_int32),
                ("split_items", C.c_int32)]


class nm_lora_layer(C.Structure):
    _fields_ = [("out_f", C.c_int32), ("in_f", C.c_int32), ("r", C.c_int32), ("scaling", C.c_float), ("W", C.c_void_p),
                ("B", C.c_void_p), ("A", C.c_void_p), ("o0", C.c_void_p), ("o1", C.c_void_p)]


class nm_rollout_cfg(C.Structure):
    _fields_ = [("substeps", C.c_int32), ("plasticity_alpha", C.c_float), ("grid_cache_blocks", C.c_int32),
                ("cache_verified", C.c_int32), ("svd_adjoint", C.c_int32), ("svd_cache", C.c_void_p), ("act_cache", C.c_void_p),
                ("weights_prepared", C.c_int32), ("last_gF_zero", C.c_int32)]


COMM_ALL_GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
COMM_ALL_REDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)


COMM_EXCHANGE_PEERS = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_uint32, C.c_void_p)
COMM_ALL_RANKS = 0xFFFFFFFF     # nm_comm.peers: exchange by all-reduce


class nm_comm(C.Structure):
    _fields_ = [("world", C.c_int32), ("rank", C.c_int32), ("all_gather_i32", COMM_ALL_GATHER),
                ("all_reduce_sum_f32", COMM_ALL_REDUCE), ("user", C.c_void_p), ("exchange_peers_f32", COMM_EXCHANGE_PEERS),
                ("peers", C.c_uint32)]


# name -> (restype, argtypes); kept in one table so tests can check the exports against the header
_P, _I32, _I64, _F, _SZ = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t
SIGNATURES = {
    "nm_version": (C.c_int, []),
    "nm_last_error": (C.c_char_p, []),
    "nm_prof_enable": (C.c_int, [_I32, C.c_char_p]),
    "nm_prof_report": (C.c_int, [C.c_char_p, _SZ]),
    "nm_prof_reset": (C.c_int, []),
    "nm_material_bwd_ex": (C.c_int, [_I32, _I32, _F, _P, C.POINTER(nm_mlp), _P, _P, _P, _P, _P, _I32, _P, _SZ, _P]),
    "nm_mpm_create": (C.c_int, [C.POINTER(nm_mpm_cfg), C.POINTER(C.c_void_p)]),
    "nm_mpm_destroy": (C.c_int, [_P]),
    "nm_mpm_forward": (C.c_int, [_P, _I32, C.POINTER(nm_statics), C.POINTER(nm_particles), C.POINTER(nm_particles), _P]),
    "nm_mpm_gridcache_bytes": (_SZ, [_I32]),
    "nm_mpm_forward_ex": (C.c_int, [_P, _I32, C.POINTER(nm_statics), C.POINTER(nm_particles), C.POINTER(nm_particles), _P, _I32,
                                    _P]),
    "nm_mpm_backward_ex": (C.c_int, [_P, _I32, C.POINTER(nm_statics), C.POINTER(nm_particles), C.POINTER(nm_particles),
                                     C.POINTER(nm_particles), C.POINTER(nm_particles), _P, _I32, _P]),
    "nm_mpm_backward": (C.c_int, [_P, _I32, C.POINTER(nm_statics), C.POINTER(nm_particles), C.POINTER(nm_particles),
                                  C.POINTER(nm_particles), C.POINTER(nm_particles), _P]),
    "nm_mpm_forward_extra": (C.c_int, [_P, _I32, C.POINTER(nm_statics), C.POINTER(nm_particles), _I32,
                                       C.POINTER(nm_statics), C.POINTER(nm_particles), _P]),
    "nm_mpm_p2g": (C.c_int, [_P, _I32, C.POINTER(nm_statics), C.POINTER(nm_particles), _P]),
    "nm_mpm_active_list": (C.c_int, [_P, _P, _I32, _P]),
    "nm_mpm_dilated_list": (C.c_int, [_P, _P, _I32, _P]),
    "nm_mpm_shared_workspace": (_SZ, [_I32, _I32]),
    "nm_mpm_shared_blocks": (C.c_int, [_P, _P, _I32, _I32, _P, _I32, _P, _P, _SZ, _P]),
    "nm_mpm_blocks_pack": (C.c_int, [_P, _I32, _P, _I32, _P, _P]),
    "nm_mpm_blocks_unpack": (C.c_int, [_P, _I32, _P, _I32, _P, _P]),
    "nm_mpm_forward_finish": (C.c_int, [_P, _I32, C.POINTER(nm_statics), C.POINTER(nm_particles), C.POINTER(nm_particles), _P, _I32,
                                        _P, _P]),
    "nm_mpm_backward_begin": (C.c_int, [_P, _I32, C.POINTER(nm_statics), C.POINTER(nm_particles), C.POINTER(nm_particles),
                                        C.POINTER(nm_particles), C.POINTER(nm_particles), _P, _I32, _P]),
    "nm_mpm_backward_finish": (C.c_int, [_P, _I32, C.POINTER(nm_statics), C.POINTER(nm_particles), C.POINTER(nm_particles), _P]),
    "nm_mpm_grid_stats": (C.c_int, [_P, C.POINTER(_I32), C.POINTER(_I32), _P]),
    "nm_mpm_grid_export": (C.c_int, [_P, _P, _P, _P, _P]),
    "nm_svd3_fwd": (C.c_int, [_I32, _P, _P, _P, _P, _P]),
    "nm_svd3_bwd": (C.c_int, [_I32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nm_material_fwd": (C.c_int, [_I32, _I32, _F, _P, C.POINTER(nm_mlp), _P, _P]),
    "nm_material_bwd_workspace": (_SZ, [_I32]),
    "nm_material_bwd": (C.c_int, [_I32, _I32, _F, _P, C.POINTER(nm_mlp), _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "nm_spmm_csr": (C.c_int, [_I32, _I32, _P, _P, _P, _P, _P, _P]),
    "nm_spmm_csr_sum3": (C.c_int, [_I32, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P]),
    "nm_cov_deform": (C.c_int, [_I32, _P, _P, _P, _P]),
    "nm_bind_build_workspace": (_SZ, [_I32, _I32]),
    "nm_bind_build": (C.c_int, [_I32, _I32, _P, _P, _P, C.POINTER(C.c_float), _F, C.POINTER(_I32), _F, _I32, _P, _P, _P, _P, _P, _SZ,
                                _P]),
    "nm_bind_frame": (C.c_int, [_I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nm_raster_state_bytes": (_SZ, [C.POINTER(nm_raster_cfg), _I32, _I64]),
    "nm_raster_forward": (C.c_int, [C.POINTER(nm_raster_cfg), _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _SZ, _I64, _P, _P, _P]),
    "nm_raster_state_bytes_ex": (C.c_int, [C.POINTER(nm_raster_cfg), _I32, _I64, C.POINTER(_SZ), C.POINTER(_SZ)]),
    "nm_raster_forward_ex": (C.c_int, [C.POINTER(nm_raster_cfg), _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _SZ, _P, _SZ, _I64, _P, _P, _P,
                                       _P]),
    "nm_raster_set_hinted": (C.c_int, [_I32, _I32]),
    "nm_raster_set_reverse_px2": (C.c_int, [_I32]),
    "nm_raster_set_split": (C.c_int, [_I32, _I32, _I64]),
    "nm_raster_count_pairs": (C.c_int, [C.POINTER(nm_raster_cfg), _I32, _P, _I64, _P, _P]),
    "nm_raster_bwd_workspace": (_SZ, [_I32]),
    "nm_raster_backward": (C.c_int, [C.POINTER(nm_raster_cfg), _I32, _I32, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _P,
                                     _SZ, _P]),
    "nm_pixel_loss": (C.c_int, [_I32, _F, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P]),
    "nm_lora_merge": (C.c_int, [_I32, _I32, _I32, _F, _P, _P, _P, _P, _P]),
    "nm_lora_merge_bwd": (C.c_int, [_I32, _I32, _I32, _F, _P, _P, _P, _P, _P, _P]),
    "nm_lora_merge_layers": (C.c_int, [_I32, C.POINTER(nm_lora_layer), _P]),
    "nm_lora_merge_layers_bwd": (C.c_int, [_I32, C.POINTER(nm_lora_layer), _P]),
    "nm_rollout_workspace": (_SZ, [_I32, _I32]),
    "nm_rollout_gridcache_bytes": (_SZ, [_I32, _I32]),
    "nm_rollout_cache_status": (C.c_int, [_P, C.POINTER(nm_rollout_cfg), _P, _P]),
    "nm_rollout_forward": (C.c_int, [_P, _I32, C.POINTER(nm_rollout_cfg), C.POINTER(nm_statics), C.POINTER(nm_mlp),
                                     C.POINTER(nm_mlp), _P, _P, _P, _SZ, _P]),
    "nm_rollout_set_forward_pair": (C.c_int, [_I32]),
    "nm_rollout_backward": (C.c_int, [_P, _I32, C.POINTER(nm_rollout_cfg), C.POINTER(nm_statics), C.POINTER(nm_mlp),
                                      C.POINTER(nm_mlp), _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "nm_rollout_svdcache_bytes": (_SZ, [_I32, _I32]),
    "nm_rollout_actcache_bytes": (_SZ, [_I32, _I32]),
    "nm_rollout_shard_workspace": (_SZ, [_I32, _I32, _I32, _I32]),
    "nm_rollout_forward_sharded": (C.c_int, [_P, _I32, C.POINTER(nm_rollout_cfg), C.POINTER(nm_statics), C.POINTER(nm_mlp),
                                             C.POINTER(nm_mlp), _P, _P, _P, _SZ, C.POINTER(nm_comm), _I32, _I32, _P, _SZ, _P]),
    "nm_rollout_backward_sharded": (C.c_int, [_P, _I32, C.POINTER(nm_rollout_cfg), C.POINTER(nm_statics), C.POINTER(nm_mlp),
                                              C.POINTER(nm_mlp), _P, _P, _P, _P, _P, _P, _P, _SZ, C.POINTER(nm_comm), _I32, _I32,
                                              _P, _SZ, _P]),
    "nm_rollout_shard_status": (C.c_int, [_P, _P, _P]),
    "nm_rccl_library": (C.c_char_p, []),
    "nm_rccl_unique_id": (C.c_int, [_P]),
    "nm_rccl_create": (C.c_int, [_P, _I32, _I32, C.POINTER(_P)]),
    "nm_rccl_destroy": (C.c_int, [_P]),
    "nm_rccl_comm": (C.c_int, [_P, C.POINTER(nm_comm)]),
    "nm_rccl_all_reduce_sum_f32": (C.c_int, [_P, _P, _I64, _P]),
    "nm_rccl_all_gather_i32": (C.c_int, [_P, _P, _P, _I64, _P]),
    "nm_rccl_time_all_reduce": (C.c_int, [_P, _P, _I64, _I32, _I32, C.POINTER(C.c_float), _P]),
    "nm_rccl_exchange_peers_f32": (C.c_int, [_P, _P, _P, _I64, C.c_uint32, _P]),
    "nm_rccl_time_exchange_peers": (C.c_int, [_P, _P, _P, _I64, C.c_uint32, _I32, _I32, C.POINTER(C.c_float), _P]),
    "nm_mpm_peer_ranks": (C.c_int, [_P, _P, _I32, _I32, _I32, _P, _P]),
}

_lib = None


SVD_ADJOINT = {"reference": 0, "polar": 1}      # NM_SVD_ADJOINT_* / (NM_BWD_POLAR_ADJOINT = 2 for nm_material_bwd_ex)


class NeumaHipError(RuntimeError):
    pass


def lib():
    """Load the shared library (once).  Raises if it has not been built — no silent fallback."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise NeumaHipError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                f"(or `make -C neuma_amd/csrc`). neuma_amd has no CPU fallback.")
        # torch first: it ships its own libamdhip64, and the library must bind to the HIP runtime the process's tensors live in
        # (loaded before torch it binds to /opt/rocm's copy - two runtimes in one process, and launches fail with
        # "no ROCm-capable device is detected")
        import torch  # noqa: F401
        handle = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here == header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().nm_last_error().decode(errors="replace")
        raise NeumaHipError(f"{what or 'libneuma_hip'} failed (rc={rc}): {msg}")


def stream_ptr(device=None) -> int:
    """The caller's current HIP stream on `device`.  Also makes `device` the current HIP device of this thread: the
    library launches on whatever device is current (kernel launches carry a stream handle, and a stream belongs to one
    device), so an operator called on tensors of cuda:N while another device is current would otherwise fault."""
    import torch
    if device is not None:
        dev = device if isinstance(device, torch.device) else torch.device(device)
        if dev.type == "cuda" and dev.index is not None:
            if dev.index != torch.cuda.current_device():
                torch.cuda.set_device(dev)
            raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)      # (the handle without building a Stream object:
            if raw is not None:                                             #  this function runs ~10 times per frame)
                return raw(dev.index)
    return torch.cuda.current_stream(device).cuda_stream


_OK_DTYPES = None


def ptr(t, dtype=None):
    """Device pointer of a contiguous fp32 / int32 / uint8 (opaque buffer) HIP tensor, or None.  `dtype`: required dtype."""
    global _OK_DTYPES
    if t is None:
        return None
    if not t.is_cuda:
        raise NeumaHipError("neuma_amd operators need tensors on the GPU (no CPU path)")
    if not t.is_contiguous():
        raise NeumaHipError("tensor must be contiguous")
    if _OK_DTYPES is None:
        import torch
        _OK_DTYPES = (torch.float32, torch.int32, torch.uint8)
    if t.dtype not in _OK_DTYPES or (dtype is not None and t.dtype != dtype):
        raise NeumaHipError(f"tensor dtype {t.dtype} not accepted here (the C ABI takes float32 / int32 arrays"
                            f"{'' if dtype is None else ', this argument ' + str(dtype)})")
    return t.data_ptr()


def same_device(*tensors):
    """All non-None operands of one call must live on one device; returns it."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise NeumaHipError(f"operands on different devices: {dev} and {t.device}")
    return dev
