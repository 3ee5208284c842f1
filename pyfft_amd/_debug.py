"""Development switches, all in one place (nothing in the product path reads the environment directly).

None of these is needed in production; they exist so that the measurements quoted in DESIGN.md and docs/ can be reproduced (table: docs/switches.md)
(tools/*.py set them).  Unset = the plan's own choice.
"""
import os


def forced_strategy():
    """PYFFT_AMD_STRATEGY = auto | chain | pipelined | fused | fusedx | xcd"""
    return os.environ.get("PYFFT_AMD_STRATEGY", "auto")


def pipeline_chunk_bytes(default):
    return (int(os.environ.get("PYFFT_AMD_PIPE_MB", "0")) << 20) or default


def no_nd_generic():
    """A/B: one-launch N-D passes always on the shape's fixed instance (ignore the tuning table's "nd_generic" list)."""
    return bool(os.environ.get("PYFFT_AMD_NO_ND_GENERIC"))


def pipeline_streams(default):
    return int(os.environ.get("PYFFT_AMD_PIPE_STREAMS", "0")) or default


def slab_bytes(default):
    return (int(os.environ.get("PYFFT_AMD_SLAB_MB", "0")) << 20) or default


def no_slabs():
    return bool(os.environ.get("PYFFT_AMD_NO_SLABS"))


def no_plane_fused():
    """3-D transforms beyond a pipeline chunk: leading x / y passes slab by slab (rounds 1-5) instead of one persistent 2-D launch"""
    return bool(os.environ.get("PYFFT_AMD_NO_PLANE_FUSED"))


def no_stream_hints():
    return bool(os.environ.get("PYFFT_AMD_NO_STREAM_HINTS"))


def fused_grid_per_cu(default):
    """PYFFT_AMD_FUSED_WGS = persistent work-groups per CU of the fused kernel (development sweep)"""
    return int(os.environ.get("PYFFT_AMD_FUSED_WGS", "0")) or default


def fused_lag_factor(default):
    return int(os.environ.get("PYFFT_AMD_FUSED_LAGF", "0")) or default


def fused_ring(lag, ring):
    """PYFFT_AMD_FUSED_RING = lag,ring of the fused two-pass kernel, any size (development sweep, tools/fused_sweep.py)"""
    v = os.environ.get("PYFFT_AMD_FUSED_RING")
    if v:
        lag, ring = (int(t) for t in v.split(","))
    return lag, ring


def fused_memset():
    """PYFFT_AMD_FUSED_MEMSET=1: one counter set zeroed by a memset in front of every persistent launch (rounds 1-3; A/B)"""
    return bool(os.environ.get("PYFFT_AMD_FUSED_MEMSET"))


def no_fusedx():
    """PYFFT_AMD_NO_FUSEDX=1: never choose the per-XCD work lists by default (A/B)"""
    return bool(os.environ.get("PYFFT_AMD_NO_FUSEDX"))


def split_fusedx():
    """PYFFT_AMD_SPLIT_FUSEDX=1: split-complex fp32 N = 2^16 ... 2^18 on the per-XCD work lists (A/B; the default is the sibling-tile kernel on the global list)"""
    return bool(os.environ.get("PYFFT_AMD_SPLIT_FUSEDX"))


def no_split_rowfirst():
    """PYFFT_AMD_NO_SPLIT_ROWFIRST=1: split-complex fp32 2-D plans keep the round-3 rule (pipelined chunks; two transposing passes on
    request).  The launcher picks the kernel too, so the calling thread's native switch follows the variable whenever the planner asks."""
    from . import _native
    off = bool(os.environ.get("PYFFT_AMD_NO_SPLIT_ROWFIRST"))
    if bool(_native.lib.mifft_debug_get(_native.DEBUG_NO_ROWFIRST)) != off:
        _native.lib.mifft_debug_set(_native.DEBUG_NO_ROWFIRST, 1 if off else 0)
    return off


def no_oop_nd():
    """PYFFT_AMD_NO_OOP_ND=1: out-of-place executes run the plan's chain even where a one-launch kernel with several work-groups per
    transform exists (csrc/fft_nd2z.hpp; A/B, tests)"""
    return bool(os.environ.get("PYFFT_AMD_NO_OOP_ND"))


def no_fusedp_alt():
    """PYFFT_AMD_NO_FUSEDP_ALT=1: no persistent two-pair launch for shapes whose chain is a plane pass + a z pass (A/B, tests)"""
    return bool(os.environ.get("PYFFT_AMD_NO_FUSEDP_ALT"))


def no_tiled_kernel():
    """PYFFT_AMD_NO_TILED=1: tiled-batch plans gather / transform / scatter even where a one-launch tile kernel exists (A/B, tests)"""
    return bool(os.environ.get("PYFFT_AMD_NO_TILED"))


def no_mixed_nd():
    """PYFFT_AMD_NO_MIXED_ND=1: smooth N-D shapes one launch per axis even where the whole transform fits one tile (A/B, tests)"""
    return bool(os.environ.get("PYFFT_AMD_NO_MIXED_ND"))


def fused3_lag_ring(lag, ring):
    """PYFFT_AMD_FUSED3 = lag,ring (development sweep of the 2048 x 2048 fused kernel)"""
    v = os.environ.get("PYFFT_AMD_FUSED3")
    if v:
        lag, ring = (int(t) for t in v.split(","))
    return lag, ring


def fusedx():
    """PYFFT_AMD_FUSEDX = lag,ring (per XCD) of the XCD-local fused form forced by PYFFT_AMD_STRATEGY=fusedx (a third field,
    round 3's write-through switch, is accepted and ignored)"""
    v = os.environ.get("PYFFT_AMD_FUSEDX", "1,2,0")
    return tuple(int(t) for t in v.split(","))


def small_fused(default):
    """PYFFT_AMD_SMALL_FUSED = lag divisor of the small-batch fused form (0 = off)"""
    v = os.environ.get("PYFFT_AMD_SMALL_FUSED")
    return default if v is None else int(v)


def xcd2_flags(default):
    v = os.environ.get("PYFFT_AMD_XCD2_FLAGS")
    return default if v is None else int(v)


def apply_native_switches(native):
    """Environment -> the process defaults of libmifft's development switches (mifft_debug_set_default); called once when the library
    is loaded.  (mifft_debug_set itself changes a switch for the calling thread only: tests and tools use that.)"""
    for env, key in (("MIFFT_NO_ND2", native.DEBUG_NO_ND2), ("MIFFT_FUSED_NO_NT", native.DEBUG_FUSED_NO_NT),
                     ("MIFFT_NO_WAVE", native.DEBUG_NO_WAVE), ("MIFFT_FORCE_WAVE", native.DEBUG_FORCE_WAVE),
                     ("MIFFT_PERSIST", native.DEBUG_PERSIST)):
        if os.environ.get(env):
            native.lib.mifft_debug_set_default(key, 1)
    if os.environ.get("MIFFT_STORE"):       # streamed output stores: 1 = non-temporal, 2 = write-through, 3 = plain
        native.lib.mifft_debug_set_default(native.DEBUG_STORE, int(os.environ["MIFFT_STORE"]))
    if os.environ.get("MIFFT_ROWS_ND"):     # dense smooth rows: 1 = two-buffer row kernel only, 2 = single-buffer tile kernel wherever it fits
        native.lib.mifft_debug_set_default(native.DEBUG_ROWS_ND, int(os.environ["MIFFT_ROWS_ND"]))
    if os.environ.get("MIFFT_NARROW_TILES"):   # fp32 2^16 ... 2^18 persistent: 16-column tiles (A/B)
        native.lib.mifft_debug_set_default(native.DEBUG_NARROW_TILES, int(os.environ["MIFFT_NARROW_TILES"]))
    if os.environ.get("MIFFT_PREFETCH"):    # persistent kernels on 512-thread tiles (development builds): the loads-first work list of round 6 (A/B)
        native.lib.mifft_debug_set_default(native.DEBUG_PREFETCH, 1)
    if os.environ.get("MIFFT_PAIR"):        # pass pairs: 1 = off, 2 = the alternative y split
        native.lib.mifft_debug_set_default(native.DEBUG_PAIR, int(os.environ["MIFFT_PAIR"]))
    # the row-first switch steers the planner AND the launcher: the process default follows the environment here, so that a plan
    # prepared on one host thread and executed on another launches the kernel its planner sized the grid for (no_split_rowfirst()
    # keeps the per-thread override for tests that flip the variable at run time)
    if os.environ.get("PYFFT_AMD_NO_SPLIT_ROWFIRST"):
        native.lib.mifft_debug_set_default(native.DEBUG_NO_ROWFIRST, 1)
