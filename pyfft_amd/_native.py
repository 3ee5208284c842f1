"""ctypes binding of libmifft.so (C ABI declared in include/mifft.h).

The product path has no CPU fallback: if the HIP library is missing or cannot be loaded,
importing this module raises ImportError, and every failing call raises RuntimeError carrying
the C side's error string (the reference surfaces PyCUDA driver errors the same way,
pyfft/cuda.py:41-46).
"""

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PYFFT_AMD_DEV_BUILD=1: the `make DEV=1` library (development strategies and A/B kernel forms, pyfft_amd/csrc/Makefile)
LIB_PATH = os.path.join(_HERE, "libmifft_dev.so" if os.environ.get("PYFFT_AMD_DEV_BUILD") else "libmifft.so")

ABI_VERSION = 5

E_INVALID = -1
E_UNSUPPORTED = -2
E_NODEVICE = -3

F32, F64 = 0, 1
INTERLEAVED, SPLIT = 0, 1
VARIANT_INTERLEAVED_ONLY = 2
VARIANT_SPLIT_ONLY = 3
VARIANT_OUT_OF_PLACE_ONLY = 4     # mifft_nd_shape_supported: several work-groups per transform, interleaved, input != output
VARIANT_OUT_OF_PLACE_ANY_SIZE = 5  # ... and preferred at every buffer size
VARIANT_SPLIT_OUT_OF_PLACE = 6       # planes on both sides, out of place: several work-groups per transform (csrc/fft_nd2zp.hpp)
VARIANT_SPLIT_OUT_OF_PLACE_ANY_SIZE = 7
PASS_COL, PASS_ROW, PASS_ND = 0, 1, 2
FLAG_SRC_INTERLEAVED, FLAG_DST_INTERLEAVED = 1, 2
FLAG_STREAM_SRC, FLAG_STREAM_DST = 4, 8
FLAG_WRITE_THROUGH = 32
FLAG_PAIR_WITH_NEXT = 16
XCD2_SCRATCH_BYTES = 8 * 64 * 16 * 256 * 8
XCD2_CONTROL_BYTES = (64 + 2 * 512 * 32) * 4
FUSED2_COUNTER_STRIDE = 64          # MIFFT_FUSED2_COUNTER_STRIDE (uint32 words between two counters)


def fused2_counter_bytes(outer):
    """MIFFT_FUSED2_COUNTER_BYTES(outer)"""
    return FUSED2_COUNTER_STRIDE * 4 * (9 + 2 * int(outer))


XCD2_PREFETCH = 1
DEBUG_NO_ND2, DEBUG_FUSED_NO_NT, DEBUG_NO_WAVE, DEBUG_FORCE_WAVE, DEBUG_PERSIST, DEBUG_ALT_ROWS, DEBUG_PAIR, DEBUG_STORE = 0, 1, 2, 3, 4, 5, 6, 7
DEBUG_ROWS_ND = 8
DEBUG_NARROW_TILES = 9
DEBUG_NO_ROWFIRST = 10
DEBUG_PREFETCH = 11
FEATURE_XCD2, FEATURE_FUSED2X, FEATURE_SEQUENTIAL_LIST, FEATURE_AB_FORMS = 0, 1, 2, 3     # mifft_has_feature: parts only `make DEV=1` builds
XCD2_TRACE = 2
XCD2_TRACE_BYTES = 512 * 32 * 8


class MifftPass(ctypes.Structure):
    """struct mifft_pass (include/mifft.h)."""
    _fields_ = [
        ("kind", ctypes.c_int32),
        ("precision", ctypes.c_int32),
        ("layout", ctypes.c_int32),
        ("inverse", ctypes.c_int32),
        ("L", ctypes.c_int32),
        ("variant", ctypes.c_int32),
        ("M", ctypes.c_int64),
        ("S", ctypes.c_int64),
        ("outer", ctypes.c_int64),
        ("outer_stride_in", ctypes.c_int64),
        ("outer_stride_out", ctypes.c_int64),
        ("scale", ctypes.c_double),
        ("tw_L", ctypes.c_void_p),
        ("tw_lo", ctypes.c_void_p),
        ("tw_hi", ctypes.c_void_p),
        ("tw_shift", ctypes.c_int32),
        ("src", ctypes.c_int32),
        ("dst", ctypes.c_int32),
        ("flags", ctypes.c_int32),
    ]


class MifftCopy(ctypes.Structure):
    """struct mifft_copy (include/mifft.h)."""
    _fields_ = [
        ("precision", ctypes.c_int32),
        ("ndim", ctypes.c_int32),
        ("dims", ctypes.c_int64 * 6),
        ("src_stride", ctypes.c_int64 * 6),
        ("dst_stride", ctypes.c_int64 * 6),
        ("src_valid0", ctypes.c_int64),
        ("src_split", ctypes.c_int32),
        ("dst_split", ctypes.c_int32),
        ("conj_in", ctypes.c_int32),
        ("conj_out", ctypes.c_int32),
        ("mult", ctypes.c_void_p),
        ("scale", ctypes.c_double),
    ]


class MifftTiling(ctypes.Structure):
    """struct mifft_tiling (include/mifft.h)."""
    _fields_ = [
        ("pitch_y", ctypes.c_int64),
        ("pitch_z", ctypes.c_int64),
        ("parent_elems", ctypes.c_int64),
        ("cx", ctypes.c_int32),
        ("cy", ctypes.c_int32),
        ("cz", ctypes.c_int32),
    ]


class MifftDeviceProps(ctypes.Structure):
    """struct mifft_device_props (include/mifft.h)."""
    _fields_ = [
        ("name", ctypes.c_char * 256),
        ("gcn_arch", ctypes.c_char * 64),
        ("compute_units", ctypes.c_int32),
        ("wavefront_size", ctypes.c_int32),
        ("max_threads_per_block", ctypes.c_int32),
        ("max_grid_x", ctypes.c_int32),
        ("lds_bytes_per_block", ctypes.c_int64),
        ("total_mem_bytes", ctypes.c_int64),
        ("clock_khz", ctypes.c_int32),
        ("l2_bytes", ctypes.c_int32),
        ("llc_bytes", ctypes.c_int64),
        ("num_xcc", ctypes.c_int32),
        ("reserved0", ctypes.c_int32),
    ]


class MifftFusedSync(ctypes.Structure):
    """struct mifft_fused_sync (include/mifft.h): counters of this launch, the set it zeroes for the next one, the error word."""
    _fields_ = [
        ("counters", ctypes.c_void_p),
        ("counters_next", ctypes.c_void_p),
        ("error_word", ctypes.c_void_p),
    ]


_vp = ctypes.c_void_p
_vpp = ctypes.POINTER(ctypes.c_void_p)
_i32 = ctypes.c_int32
_sz = ctypes.c_size_t
_pass_p = ctypes.POINTER(MifftPass)
_sync_p = ctypes.POINTER(MifftFusedSync)
_buf3 = ctypes.c_void_p * 3

# name -> (restype, argtypes); every symbol include/mifft.h declares
PROTOTYPES = {
    "mifft_abi_version": (ctypes.c_int, []),
    "mifft_last_error": (ctypes.c_char_p, []),
    "mifft_debug_set": (ctypes.c_int, [_i32, _i32]),
    "mifft_debug_set_default": (ctypes.c_int, [_i32, _i32]),
    "mifft_debug_get": (ctypes.c_int, [_i32]),
    "mifft_has_feature": (ctypes.c_int, [_i32]),
    "mifft_device_count": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    "mifft_set_device": (ctypes.c_int, [ctypes.c_int]),
    "mifft_get_device": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    "mifft_device_props_get": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(MifftDeviceProps)]),
    "mifft_malloc": (ctypes.c_int, [_vpp, _sz]),
    "mifft_free": (ctypes.c_int, [_vp]),
    "mifft_memset": (ctypes.c_int, [_vp, ctypes.c_int, _sz, _vp]),
    "mifft_memcpy_h2d": (ctypes.c_int, [_vp, _vp, _sz, _vp]),
    "mifft_memcpy_d2h": (ctypes.c_int, [_vp, _vp, _sz, _vp]),
    "mifft_memcpy_d2d": (ctypes.c_int, [_vp, _vp, _sz, _vp]),
    "mifft_host_alloc": (ctypes.c_int, [_vpp, _sz]),
    "mifft_host_free": (ctypes.c_int, [_vp]),
    "mifft_memcpy_d2h_async": (ctypes.c_int, [_vp, _vp, _sz, _vp]),
    "mifft_stream_create": (ctypes.c_int, [_vpp]),
    "mifft_stream_destroy": (ctypes.c_int, [_vp]),
    "mifft_stream_sync": (ctypes.c_int, [_vp]),
    "mifft_device_sync": (ctypes.c_int, []),
    "mifft_stream_is_capturing": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int32)]),
    "mifft_stream_begin_capture": (ctypes.c_int, [_vp]),
    "mifft_stream_end_capture": (ctypes.c_int, [_vp, _vpp]),
    "mifft_graph_launch": (ctypes.c_int, [_vp, _vp]),
    "mifft_graph_destroy": (ctypes.c_int, [_vp]),
    "mifft_event_create": (ctypes.c_int, [_vpp]),
    "mifft_event_destroy": (ctypes.c_int, [_vp]),
    "mifft_event_record": (ctypes.c_int, [_vp, _vp]),
    "mifft_event_sync": (ctypes.c_int, [_vp]),
    "mifft_event_query": (ctypes.c_int, [_vp]),
    "mifft_stream_wait_event": (ctypes.c_int, [_vp, _vp]),
    "mifft_event_elapsed_ms": (ctypes.c_int, [ctypes.POINTER(ctypes.c_float), _vp, _vp]),
    "mifft_nd_max_points_for": (ctypes.c_int, [_i32]),
    "mifft_nd_shape_supported": (ctypes.c_int, [_i32, _i32, _i32, _i32, _i32]),
    "mifft_pass_supported": (ctypes.c_int, [_i32, _i32, _i32, _i32]),
    "mifft_launch_pass": (ctypes.c_int, [_pass_p, _vp, _vp, _vp, _vp, _vp]),
    "mifft_pair_split": (ctypes.c_int, [_i32, _i32, _i32, _i32, _i32]),
    "mifft_pass_pair_supported": (ctypes.c_int, [_pass_p, _pass_p]),
    "mifft_pair_kernel_supported": (ctypes.c_int, [_i32, _i32, _i32, _i32, _i32, _i32]),
    "mifft_launch_pass_pair": (ctypes.c_int, [_pass_p, _pass_p, _vp, _vp, _vp, _vp, _vp]),
    "mifft_launch_chain": (ctypes.c_int, [_pass_p, _i32, _vpp, _vpp, _vp]),
    "mifft_launch_chain_pipelined": (ctypes.c_int, [_pass_p, _i32, _vpp, _vpp, ctypes.c_int64, ctypes.c_int64,
                                                      ctypes.c_int64, _vp, _vpp, _i32, _vpp]),
    "mifft_launch_fused2": (ctypes.c_int, [_pass_p, _pass_p, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _sync_p, _i32, _vp]),
    "mifft_launch_fused2x": (ctypes.c_int, [_pass_p, _pass_p, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _sync_p, _i32, _vp]),
    "mifft_fused_pair_supported": (ctypes.c_int, [_i32, _i32, _i32, _i32, _i32]),
    "mifft_fused_pair_split": (ctypes.c_int, [_i32, _i32, _i32, _i32, _i32]),
    "mifft_launch_fused_pair": (ctypes.c_int, [_pass_p, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _sync_p, _i32, _vp]),
    "mifft_launch_xcd2": (ctypes.c_int, [_pass_p, _pass_p, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "mifft_nd_tiled_supported": (ctypes.c_int, [_i32, _i32, _i32, _i32]),
    "mifft_launch_nd_tiled": (ctypes.c_int, [_pass_p, ctypes.POINTER(MifftTiling), _vp, _vp, _vp]),
    "mifft_launch_nd_tiled_split": (ctypes.c_int, [_pass_p, ctypes.POINTER(MifftTiling), _vp, _vp, _vp, _vp, _vp]),
    "mifft_aux_copy": (ctypes.c_int, [ctypes.POINTER(MifftCopy), _vp, _vp, _vp, _vp, _vp]),
    "mifft_aux_mul_rows": (ctypes.c_int, [_i32, _vp, _vp, ctypes.c_int64, ctypes.c_int64, _vp]),
    "mifft_aux_count_mismatch": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, _vp, _vp]),
    "mifft_mixed_supported": (ctypes.c_int, [_i32, _i32]),
    "mifft_launch_mixed_rows": (ctypes.c_int, [_i32, _i32, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, _vp, _vp, _vp, _i32,
                                                ctypes.c_double, _vp]),
    "mifft_launch_mixed_lines": (ctypes.c_int, [_i32, _i32, ctypes.c_int64, ctypes.c_int64, _vp, _vp, _vp, _i32, _i32, ctypes.c_double, _vp]),
    "mifft_mixed_long_split": (ctypes.c_int, [_i32, ctypes.c_int64, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "mifft_launch_mixed_long": (ctypes.c_int, [_i32, _i32, _i32, ctypes.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, ctypes.c_double, _vp]),
    "mifft_bluestein_padded": (ctypes.c_int, [_i32, _i32, ctypes.POINTER(ctypes.c_int32)]),
    "mifft_launch_bluestein_rows": (ctypes.c_int, [_i32, _i32, _i32, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, _vp, _vp, _vp, _vp, _vp,
                                                   _i32, ctypes.c_double, _vp]),
    "mifft_mixed_nd_supported": (ctypes.c_int, [_i32, _i32, _i32, _i32]),
    "mifft_launch_mixed_nd": (ctypes.c_int, [_i32, _i32, _i32, _i32, ctypes.c_int64, _vp, _vp, _vp, _vp, _vp, _i32, ctypes.c_double, _vp]),
    "mifft_time_chain": (ctypes.c_int, [_pass_p, _i32, _vpp, _vpp, _vp, _i32, ctypes.POINTER(ctypes.c_float)]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "pyfft_amd: %s not found -- build it with `make -C pyfft_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback." % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise ImportError("pyfft_amd: cannot load %s: %s" % (LIB_PATH, e))
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.mifft_abi_version() != ABI_VERSION:
        raise ImportError("pyfft_amd: libmifft.so ABI %d != expected %d" % (lib.mifft_abi_version(), ABI_VERSION))
    return lib


lib = _load()


def _apply_debug():
    from . import _debug
    import sys
    _debug.apply_native_switches(sys.modules[__name__])


_apply_debug()


def last_error():
    msg = lib.mifft_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc, what=""):
    """Raise RuntimeError (ValueError for descriptor errors) with the C side's message."""
    if rc == 0:
        return
    msg = "%s failed (code %d): %s" % (what or "libmifft call", rc, last_error())
    if rc == E_INVALID:
        raise ValueError(msg)
    raise RuntimeError(msg)


def make_buf3(a, b, c):
    return _buf3(a, b, c)
