"""ctypes binding of ``libmsj_stage1.so`` (the C ABI in ``include/msj_stage1.h``).

The HIP extension is the only implementation behind this package: if the shared
library is missing, or no HIP device is usable, calls raise -- there is no CPU
fallback of any kind.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# The one library this package loads.  No environment variable can substitute it; measurement scripts that compare
# differently built kernels set this attribute explicitly before the first load() (bench.py --lib, which reports it).
LIB_PATH = os.path.join(_HERE, "libmsj_stage1.so")
GEN_LIB_PATH = os.path.join(_HERE, "libmsj_gen.so")


class MsjCarry(ctypes.Structure):
    """``msj_carry`` (include/msj_stage1.h): the scanners' cross-block state."""

    _fields_ = [
        ("count", ctypes.c_uint64),
        ("bytes", ctypes.c_uint64),
        ("in_string", ctypes.c_uint32),
        ("next_is_escaped", ctypes.c_uint32),
        ("prev_scalar", ctypes.c_uint32),
        ("unescaped_error", ctypes.c_uint32),
        ("utf8_error", ctypes.c_uint32),
        ("internal_error", ctypes.c_uint32),
        ("code", ctypes.c_int32),
        ("capacity_error", ctypes.c_uint32),
        ("reserved", ctypes.c_uint32 * 4),
    ]


class MsjTokensResult(ctypes.Structure):
    """``msj_tokens_result`` (include/msj_stage1.h)."""

    _fields_ = [("n", ctypes.c_uint64), ("final_depth", ctypes.c_int32), ("min_depth", ctypes.c_int32),
                ("max_depth", ctypes.c_int32), ("reserved", ctypes.c_uint32)]


class MsjDocumentsResult(ctypes.Structure):
    """``msj_documents_result`` (include/msj_stage1.h)."""

    _fields_ = [("n_documents", ctypes.c_uint64), ("n_complete", ctypes.c_uint64),
                ("tokens_complete", ctypes.c_uint64), ("resume_offset", ctypes.c_uint64)]


class MsjSegment(ctypes.Structure):
    _fields_ = [
        ("byte_base", ctypes.c_uint64),
        ("byte_len", ctypes.c_uint64),
        ("index_begin", ctypes.c_uint64),
        ("count", ctypes.c_uint64),
    ]


assert ctypes.sizeof(MsjCarry) == 64
assert ctypes.sizeof(MsjSegment) == 32

_lib = None


class HipExtensionMissing(RuntimeError):
    pass


def _share_torch_hip_runtime():
    """Make libmsj_stage1.so bind to the HIP runtime PyTorch already uses.

    The PyTorch wheel bundles its own libamdhip64.so / libhsa-runtime64.so
    (ROCm 7.0) next to the system ROCm 7.2 that libmsj_stage1.so names in its
    DT_NEEDED.  Two HIP runtimes in one process each open the device; whichever
    comes second may see no GPU, and tensors allocated by one are unknown to the
    other.  Promoting torch's runtime to the global symbol scope *before* our
    library is opened makes every hip* symbol of libmsj_stage1.so resolve to
    that one runtime (global scope is searched before a library's own
    dependencies).  Without torch in the process (the Mojo shim case) nothing is
    preloaded and the system runtime is used.
    """
    try:
        import torch  # noqa: F401
    except ImportError:
        return
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)


def load():
    """Load libmsj_stage1.so once; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    _share_torch_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise HipExtensionMissing(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C mojo_simdjson_amd/csrc)"
        )
    lib = ctypes.CDLL(LIB_PATH)
    u8p, u32p, u64p = ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)
    i32p = ctypes.POINTER(ctypes.c_int32)
    lib.msj_version.restype = ctypes.c_char_p
    lib.msj_version.argtypes = []
    lib.msj_device_count.restype = ctypes.c_int32
    lib.msj_device_count.argtypes = []
    lib.msj_tile_bytes.restype = ctypes.c_uint32
    lib.msj_tile_bytes.argtypes = []
    lib.msj_ctx_create.restype = ctypes.c_int32
    lib.msj_ctx_create.argtypes = [ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p)]
    lib.msj_ctx_destroy.restype = None
    lib.msj_ctx_destroy.argtypes = [ctypes.c_void_p]
    lib.msj_stage1.restype = ctypes.c_int32
    lib.msj_stage1.argtypes = [u8p, ctypes.c_uint64, u32p, ctypes.c_uint64, u64p, i32p, ctypes.c_uint32]
    lib.msj_stage1_ctx.restype = ctypes.c_int32
    lib.msj_stage1_ctx.argtypes = [ctypes.c_void_p] + lib.msj_stage1.argtypes
    lib.msj_stage1_device.restype = ctypes.c_int32
    lib.msj_stage1_device.argtypes = [
        ctypes.c_void_p, u8p, ctypes.c_uint64, u32p, ctypes.c_uint64, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_uint32,
    ]
    lib.msj_tokens_device.restype = ctypes.c_int32
    lib.msj_tokens_device.argtypes = [ctypes.c_void_p, u8p, ctypes.c_uint64, u32p, ctypes.c_uint64,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_void_p]
    lib.msj_token_spans_device.restype = ctypes.c_int32
    lib.msj_token_spans_device.argtypes = [ctypes.c_void_p, u8p, ctypes.c_uint64, u32p, ctypes.c_uint64,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.msj_stage2_prep_device.restype = ctypes.c_int32
    lib.msj_stage2_prep_device.argtypes = [ctypes.c_void_p, u8p, ctypes.c_uint64, u32p, ctypes.c_uint64] + [ctypes.c_void_p] * 7
    lib.msj_tokens_chain_device.restype = ctypes.c_int32
    lib.msj_tokens_chain_device.argtypes = lib.msj_tokens_device.argtypes[:-1] + [ctypes.c_void_p, ctypes.c_void_p]
    lib.msj_stage2_prep_chain_device.restype = ctypes.c_int32
    lib.msj_stage2_prep_chain_device.argtypes = lib.msj_stage2_prep_device.argtypes[:-1] + [ctypes.c_void_p, ctypes.c_void_p]
    # (round 5) the pairs form, the types prototype, the placement report: every 64-bit argument declared -- an undeclared
    # Python int goes over as a C int, and a buffer length over 2 GiB arrived truncated (found by the 3.94 GiB pairs test)
    lib.msj_stage2_prep_pairs_device.restype = ctypes.c_int32
    lib.msj_stage2_prep_pairs_device.argtypes = [ctypes.c_void_p, u8p, ctypes.c_uint64, u32p, ctypes.c_uint64] + [ctypes.c_void_p] * 8
    lib.msj_tokens_pairs_device.restype = ctypes.c_int32
    lib.msj_tokens_pairs_device.argtypes = [ctypes.c_void_p, u8p, ctypes.c_uint64, u32p, ctypes.c_uint64] + [ctypes.c_void_p] * 6
    lib.msj_stage1_types_device.restype = ctypes.c_int32
    lib.msj_stage1_types_device.argtypes = [ctypes.c_void_p, u8p, ctypes.c_uint64, u32p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_uint32]
    lib.msj_depth_from_types_device.restype = ctypes.c_int32
    lib.msj_depth_from_types_device.argtypes = [ctypes.c_void_p, u8p, ctypes.c_uint64] + [ctypes.c_void_p] * 5
    lib.msj_host_placement.restype = ctypes.c_int32
    lib.msj_host_placement.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64]
    lib.msj_debug_numa_node_of.restype = ctypes.c_int32
    lib.msj_debug_numa_node_of.argtypes = [ctypes.c_void_p]
    lib.msj_stage2_prep_segments.restype = ctypes.c_int32
    lib.msj_stage2_prep_segments.argtypes = [ctypes.c_void_p, u8p, ctypes.c_void_p, ctypes.c_uint32, u32p] + [ctypes.c_void_p] * 7 + \
        [ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p]
    lib.msj_documents_device.restype = ctypes.c_int32
    lib.msj_documents_device.argtypes = [ctypes.c_void_p, u8p, ctypes.c_uint64, ctypes.c_int32, u32p, ctypes.c_uint64,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
    lib.msj_carry_fetch.restype = ctypes.c_int32
    lib.msj_carry_fetch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(MsjCarry), ctypes.c_void_p]
    lib.msj_debug_set_wait_ticks.restype = ctypes.c_int32
    lib.msj_debug_set_wait_ticks.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
    lib.msj_host_register.restype = ctypes.c_int32
    lib.msj_host_register.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
    lib.msj_host_unregister.restype = ctypes.c_int32
    lib.msj_host_unregister.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.msj_debug_set_pipeline_min_bytes.restype = ctypes.c_int32
    lib.msj_debug_set_pipeline_min_bytes.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
    lib.msj_debug_fail_pipeline_setup.restype = ctypes.c_int32
    lib.msj_debug_fail_pipeline_setup.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    lib.msj_debug_set_span_limits.restype = ctypes.c_int32
    lib.msj_debug_set_span_limits.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32]
    lib.msj_debug_set_span_mode.restype = ctypes.c_int32
    lib.msj_debug_set_span_mode.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
    lib.msj_debug_tile_group.restype = ctypes.c_uint32
    lib.msj_debug_tile_group.argtypes = [ctypes.c_int32]
    lib.msj_fallback_count.restype = ctypes.c_uint64
    lib.msj_fallback_count.argtypes = [ctypes.c_void_p]
    lib.msj_stage1_shard_device.restype = ctypes.c_int32
    lib.msj_stage1_shard_device.argtypes = [
        ctypes.c_void_p, u8p, ctypes.c_uint64, u32p, ctypes.c_uint64, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32),
        ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_uint64, ctypes.c_void_p,
        ctypes.c_uint32,
    ]
    _lib = lib
    return lib


def load_gen():
    if not os.path.exists(GEN_LIB_PATH):
        raise HipExtensionMissing(f"{GEN_LIB_PATH} not found: run __graft_entry__.build()")
    g = ctypes.CDLL(GEN_LIB_PATH)
    g.msj_gen_unit.restype = ctypes.c_uint64
    g.msj_gen_unit.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64,
                               ctypes.c_int, ctypes.c_int, ctypes.c_int]
    g.msj_gen_extreme.restype = ctypes.c_uint64
    g.msj_gen_extreme.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int]
    return g
