"""ctypes binding of libkimi_hip.so (include/kimi_hip.h).

There is NO CPU fallback: if the library is missing, cannot be loaded, or no gfx950 device is
visible, every compute entry point raises ``HipUnavailableError``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# KIMI_HIP_LIB: developer knob -- another build of the same sources (e.g. -DKH_SWEEP_PROBE, kimimaro_amd/build.py)
LIB_PATH = os.environ.get("KIMI_HIP_LIB") or os.path.join(HERE, "libkimi_hip.so")

# every symbol include/kimi_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "kh_version", "kh_last_error", "kh_device_count", "kh_edt", "kh_edt_nd", "kh_edt_timed", "kh_label_stats", "kh_scatter_lists",
    "kh_neighbor_mask", "kh_apply_voxel_graph", "kh_edf_batch", "kh_pdrf", "kh_trace_paths", "kh_fill_f32", "kh_fill_u8",
    "kh_gather_f32", "kh_init_alive", "kh_level_keys", "kh_invalidate_cube", "kh_invalidate_ball", "kh_path_search", "kh_parental_field", "kh_path_from_parents", "kh_zero2inf", "kh_inf2zero", "kh_pdrf_field", "kh_target_max", "kh_find_target", "kh_first_label", "kh_ccl26", "kh_ccl26_graph", "kh_edt_graph_cells", "kh_edt_graph_sample", "kh_fill_voids", "kh_fill_voids_nd", "kh_host_ccl26", "kh_host_find_border_targets", "kh_host_merge_components",
    "kh_host_consolidate_paths",
]


class HipUnavailableError(RuntimeError):
    """libkimi_hip.so (or an MI355X) is not available; the product has no CPU path."""


class KimiHipError(RuntimeError):
    pass


# kh_label_t  (include/kimi_hip.h) -- 52 x 4 bytes
LABEL_T = np.dtype([
    ("segid", "<u4"), ("list_offset", "<u4"), ("count", "<u4"), ("xmin", "<u4"), ("xmax", "<u4"),
    ("source", "<u4"), ("max_loc", "<u4"), ("max_val", "<f4"), ("M", "<f4"), ("root", "<u4"),
    ("q_offset", "<u4"), ("q_capacity", "<u4"), ("heap_offset", "<u4"), ("heap_capacity", "<u4"),
    ("path_offset", "<u4"), ("path_capacity", "<u4"), ("tgt_offset", "<u4"), ("n_before", "<u4"),
    ("n_after", "<u4"), ("max_paths", "<u4"), ("n_paths", "<u4"), ("n_vertices", "<u4"),
    ("status", "<u4"), ("stat_settled", "<u4"), ("stat_heap_pushes", "<u4"),
    ("cyc_target", "<u4"), ("cyc_rail", "<u4"), ("cyc_inval", "<u4"),
    ("cyc_pop", "<u4"), ("cyc_push", "<u4"), ("cyc_fire", "<u4"),
    ("soma_mode", "<u4"), ("fsr", "<f4"), ("soma_radius", "<f4"), ("soma_scale", "<f4"), ("soma_const", "<f4"),
    ("nlev", "<u4"), ("sweep_rmax", "<f4"), ("ev_offset", "<u4"), ("ev_chunks", "<u4"), ("ev_shift", "<u4"),
    ("stat_sweep_calls", "<u4"), ("stat_sweep_bails", "<u4"), ("stat_sweep_levels", "<u4"), ("stat_sweep_events", "<u4"),
    ("stat_sweep_why", "<u4"), ("lev_window", "<u4"), ("ev_spill", "<u4"), ("stat_ghost_calls", "<u4"), ("stat_rollbacks", "<u4"),
    ("pdrf_log2e", "<u4"), ("pdrf_scale", "<f4"),
])
assert LABEL_T.itemsize == 208
SWEEP_LDS_LEVELS = 16384  # KH_SWEEP_LDS_LEVELS
SWEEP_MAX_LEVELS = 1 << 22  # labels with more levels than this use the heap emulation only
PDRF_BASE, PDRF_FINISH = -1, -2  # KH_PDRF_BASE / KH_PDRF_FINISH
PDRF_KEEP_OTHERS = 0x100         # KH_PDRF_KEEP_OTHERS


def is_pow2_exponent(e):
    """kimimaro/trace.py:343: is_power_of_two(pdrf_exponent) and pdrf_exponent < 2**16 (the repeated-squaring branch).
    The reference's test (trace.py:310-313) evaluates `num & (num - 1)`: an integral FLOAT exponent (16.0, np.float64(4))
    passes its `int(num) != num` line and then raises TypeError there; so does this mirror."""
    try:
        i = int(e)           # (a NaN raises ValueError here, as the reference's `int(num)` does)
    except TypeError:
        return False
    if i != e:
        return False
    if i == 0:               # trace.py:311: `if num != 0 and ...` short-circuits -- 0 and 0.0 take the np.power branch
        return False
    if isinstance(e, (float, np.floating)):
        raise TypeError("unsupported operand type(s) for &: 'float' and 'float' (pdrf_exponent must be an integer type, "
                        "kimimaro/trace.py:313)")
    return i > 0 and (i & (i - 1)) == 0 and i < 2 ** 16

ST_BITS = {1: "work-list overflow", 2: "invalidation heap overflow", 4: "path buffer overflow",
           8: "no rail reachable from a target", 16: "float-absorption plateau while back-tracking",
           32: "target outside the label"}

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipUnavailableError(
            "kimimaro_amd: %s is missing. Build it with `python -m kimimaro_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback." % LIB_PATH)
    try:
        # PyTorch is the device-memory plumbing and bundles its own HIP runtime: load it FIRST so that
        # libkimi_hip.so binds to the same libamdhip64 (two runtimes in one process cannot share a GPU).
        import torch  # noqa: F401
    except ImportError:
        pass
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 missing
        raise HipUnavailableError("kimimaro_amd: cannot load %s: %s" % (LIB_PATH, e)) from e
    i64, f32, vp, ci = C.c_int64, C.c_float, C.c_void_p, C.c_int
    L.kh_version.restype = ci
    L.kh_last_error.argtypes = [C.c_char_p, ci]
    L.kh_device_count.restype = ci
    L.kh_edt.argtypes = [vp, ci, i64, i64, i64, f32, f32, f32, ci, vp, vp, vp]
    L.kh_edt_nd.argtypes = [vp, ci, ci, i64, i64, i64, f32, f32, f32, ci, vp, vp, vp]
    L.kh_edt_timed.argtypes = [vp, ci, i64, i64, i64, f32, f32, f32, ci, vp, vp, vp, vp]
    L.kh_label_stats.argtypes = [vp, ci, vp, i64, i64, i64, i64, vp, vp, vp, vp, vp, vp, vp]
    L.kh_scatter_lists.argtypes = [vp, ci, i64, vp, i64, vp, vp, vp, vp]
    L.kh_neighbor_mask.argtypes = [vp, ci, i64, i64, i64, vp, vp]
    L.kh_edf_batch.argtypes = [vp, ci, ci, vp, vp, i64, i64, i64, f32, f32, f32, vp, vp, vp, vp]
    L.kh_pdrf.argtypes = [vp, ci, i64, vp, vp, vp, vp, ci, f32, vp, vp]
    L.kh_trace_paths.argtypes = [vp, ci, vp, vp, vp, i64, i64, i64, f32, f32, f32, vp, vp, vp, vp, vp, vp,
                                 f32, f32, vp, vp, vp, vp, vp, i64, i64, i64, i64, vp, vp, vp, vp, vp, vp, ci, ci, vp]
    L.kh_invalidate_ball.argtypes = [vp, vp, vp, i64, i64, i64, f32, f32, f32, vp, vp, vp, vp, vp, i64, f32, f32,
                                     vp, i64, i64, i64, i64, vp, vp, vp, vp, vp, vp]
    L.kh_apply_voxel_graph.argtypes = [vp, vp, i64, vp, vp]
    L.kh_path_search.argtypes = [vp, ci, vp, vp, i64, i64, i64, f32, f32, f32, vp, vp, vp, vp, C.c_uint64, C.c_uint64, vp, i64, vp, ci, vp]
    L.kh_parental_field.argtypes = [vp, vp, vp, i64, i64, i64, vp, vp, vp, vp, C.c_uint64, vp, ci, vp]
    L.kh_path_from_parents.argtypes = [vp, i64, C.c_uint64, vp, i64, vp, vp]
    L.kh_zero2inf.argtypes = [vp, i64, vp]
    L.kh_inf2zero.argtypes = [vp, i64, vp]
    L.kh_pdrf_field.argtypes = [vp, vp, i64, f32, ci, f32, f32, vp, vp]
    L.kh_target_max.argtypes = [vp, vp, vp, i64, vp, vp]
    L.kh_find_target.argtypes = [vp, vp, i64, i64, i64, vp, vp]
    L.kh_first_label.argtypes = [vp, i64, vp, vp]
    L.kh_level_keys.argtypes = [i64, i64, i64, f32, f32, f32, vp, vp]
    L.kh_fill_f32.argtypes = [vp, i64, f32, vp]
    L.kh_fill_u8.argtypes = [vp, i64, ci, vp]
    L.kh_gather_f32.argtypes = [vp, vp, i64, vp, vp]
    L.kh_init_alive.argtypes = [vp, ci, i64, vp, vp, vp]
    L.kh_invalidate_cube.argtypes = [vp, vp, i64, i64, i64, f32, f32, f32, vp, i64, f32, f32, vp, vp]
    L.kh_ccl26.argtypes = [vp, ci, i64, i64, i64, vp, vp, vp, vp, vp, vp, vp]
    L.kh_ccl26_graph.argtypes = [vp, ci, vp, i64, i64, i64, vp, vp, vp, vp, vp, vp, vp]
    L.kh_edt_graph_cells.argtypes = [vp, ci, vp, i64, i64, i64, ci, vp, vp]
    L.kh_edt_graph_sample.argtypes = [vp, i64, i64, i64, vp, vp]
    L.kh_fill_voids.argtypes = [vp, i64, i64, i64, vp, vp, vp, vp, vp]
    L.kh_fill_voids_nd.argtypes = [vp, ci, i64, i64, i64, vp, vp, vp, vp, vp]
    L.kh_host_ccl26.argtypes = [vp, ci, i64, i64, i64, vp]
    L.kh_host_ccl26.restype = i64
    L.kh_host_find_border_targets.argtypes = [vp, vp, i64, i64, f32, f32, i64, vp, vp]
    L.kh_host_find_border_targets.restype = i64
    L.kh_host_merge_components.argtypes = [i64, vp, vp, vp, vp, vp, vp, i64, i64, f32, f32, f32, vp, vp, vp]
    L.kh_host_merge_components.restype = i64
    L.kh_host_consolidate_paths.argtypes = [i64, vp, vp, vp, vp, vp, i64, i64, i64, vp, vp, vp, vp, vp]
    L.kh_host_consolidate_paths.restype = i64
    for name in SYMBOLS:
        getattr(L, name)
        if name not in ("kh_version", "kh_device_count", "kh_host_ccl26", "kh_last_error",
                        "kh_host_find_border_targets", "kh_host_merge_components", "kh_host_consolidate_paths"):
            getattr(L, name).restype = ci
    _lib = L
    return L


def last_error():
    buf = C.create_string_buffer(512)
    lib().kh_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


def check(rc):
    if rc == 0:
        return
    msg = last_error()
    if rc == 3:
        raise HipUnavailableError(msg)
    raise KimiHipError("libkimi_hip error %d: %s" % (rc, msg))


def require_gpu():
    """Raise unless the library loads AND a gfx950 device is visible."""
    L = lib()
    if L.kh_device_count() <= 0:
        raise HipUnavailableError(
            "kimimaro_amd: no gfx950 (MI355X) device visible; the product path has no CPU fallback.")
    return L


def describe_status(bits):
    return ", ".join(v for k, v in ST_BITS.items() if bits & k) or "ok"
