"""ctypes binding of libmonoport_hip.so -- the C-ABI declared in include/monoport_hip.h.

There is deliberately NO fallback: if the shared library is missing or fails to load, importing
the product path raises, so a GPU test can never pass on a silent eager/PyTorch detour.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MONOPORT_HIP_LIB: measurement hook (tools/ablate.py side builds); the product loads lib/libmonoport_hip.so
LIB_PATH = os.environ.get("MONOPORT_HIP_LIB") or os.path.join(_HERE, "lib", "libmonoport_hip.so")

c_int = ctypes.c_int
c_i64 = ctypes.c_int64
c_f32 = ctypes.c_float
c_vp = ctypes.c_void_p
_pint = ctypes.POINTER(ctypes.c_int)
_pf32 = ctypes.POINTER(ctypes.c_float)



class GnOut(ctypes.Structure):
    """mp_gn_out (include/monoport_hip.h): where a producing kernel leaves GroupNorm statistics."""
    _fields_ = [("acc", c_vp), ("partial", c_vp), ("partial_doubles", c_i64)]


class GnIn(ctypes.Structure):
    """mp_gn_in: the GroupNorm a consuming kernel applies to its input while staging it."""
    _fields_ = [("acc", c_vp), ("gamma", c_vp), ("beta", c_vp), ("eps", c_f32), ("ss", c_vp)]


class Conv3x3Args(ctypes.Structure):
    """mp_conv3x3_args"""
    _fields_ = [("x", c_vp), ("n", c_int), ("cin", c_int), ("h", c_int), ("w", c_int), ("gn", GnIn),
                ("relu", c_int), ("reflect", c_int), ("packed", c_vp), ("wmax", c_vp), ("cout", c_int),
                ("y", c_vp), ("y2", c_vp), ("res", c_vp), ("y2_channels", c_int), ("y2_offset", c_int),
                ("fin", GnOut), ("fin2", GnOut), ("packed_wino", c_vp)]


class Conv1x1Args(ctypes.Structure):
    """mp_conv1x1_args"""
    _fields_ = [("x1", c_vp), ("gn1", GnIn), ("relu1", c_int), ("x2", c_vp), ("n", c_int), ("c1", c_int),
                ("c2", c_int), ("cout", c_int), ("hw", c_i64), ("packed", c_vp), ("f16", c_int),
                ("wmax", c_vp), ("bias", c_vp), ("res", c_vp), ("y", c_vp), ("y_hwc", c_vp), ("fin", GnOut)]


class ConvKArgs(ctypes.Structure):
    """mp_convk_args"""
    _fields_ = [("x", c_vp), ("n", c_int), ("cin", c_int), ("h", c_int), ("w", c_int), ("gn", GnIn),
                ("relu", c_int), ("reflect", c_int), ("packed", c_vp), ("bias", c_vp), ("cout", c_int),
                ("ks", c_int), ("stride", c_int), ("y", c_vp), ("fin", GnOut)]


class PlanGnApplyArgs(ctypes.Structure):
    """mp_plan_gn_apply_args"""
    _fields_ = [("x", c_vp), ("gn", GnIn), ("relu", c_int), ("n", c_int), ("c", c_int), ("hw", c_i64),
                ("res", c_vp), ("y", c_vp), ("fin", GnOut)]


class PlanPoolArgs(ctypes.Structure):
    """mp_plan_pool_args"""
    _fields_ = [("x", c_vp), ("n", c_int), ("c", c_int), ("h", c_int), ("w", c_int), ("y", c_vp), ("fin", GnOut)]


class PlanUpsampleArgs(ctypes.Structure):
    """mp_plan_upsample_args"""
    _fields_ = [("x", c_vp), ("n", c_int), ("c", c_int), ("h", c_int), ("w", c_int), ("add", c_vp), ("y", c_vp),
                ("fin", GnOut)]


class PlanMemsetArgs(ctypes.Structure):
    """mp_plan_memset_args"""
    _fields_ = [("ptr", c_vp), ("bytes", c_i64), ("value", c_int)]


class ReconEarly(ctypes.Structure):
    """mp_recon_early"""
    _fields_ = [("expect_level0", c_vp), ("flags_dev", c_vp), ("flags_host", c_vp), ("event", c_vp)]


class PlanWaitArgs(ctypes.Structure):
    """mp_plan_wait_args"""
    _fields_ = [("waiter_slot", c_int), ("signaller_slot", c_int)]


# MP_PLAN_* command kinds (include/monoport_hip.h)
PLAN_CONVK, PLAN_GN_APPLY, PLAN_CONV3X3, PLAN_CONV1X1, PLAN_AVGPOOL2, PLAN_UPSAMPLE2X, PLAN_MEMSET, PLAN_WAIT = range(1, 9)

# name -> (restype, argtypes); kept in one table so tests can check the exported surface against
# the header (tests/test_abi.py)
SIGNATURES = {
    "mp_version": (c_int, []),
    "mp_create": (c_int, [c_int, ctypes.POINTER(c_vp)]),
    "mp_destroy": (None, [c_vp]),
    "mp_last_error": (ctypes.c_char_p, [c_vp]),
    "mp_stream_release": (c_int, [c_vp, c_vp]),
    "mp_stream_create_cu_mask": (c_int, [c_vp, c_int, c_int, ctypes.POINTER(c_vp)]),
    "mp_stream_destroy": (c_int, [c_vp, c_vp]),
    "mp_stream_cu_count": (c_int, [c_vp, c_vp]),
    "mp_memory_stats": (c_int, [c_vp, ctypes.POINTER(c_i64)]),
    "mp_max_frames": (c_int, []),
    "mp_mlp_create": (c_int, [c_vp, c_int, _pint, c_int, _pint]),
    "mp_mlp_load": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_int, c_int, c_vp]),
    "mp_mlp_destroy": (c_int, [c_vp, c_int]),
    "mp_mlp_set_precision": (c_int, [c_vp, c_int, c_int]),
    "mp_feat_pack_hwc": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_int, c_int, c_vp]),
    "mp_skip_table": (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_int, c_vp, c_vp]),
    "mp_skip_table_batch": (c_int, [c_vp, c_int, c_int, c_vp, c_int, c_int, c_int, c_vp, c_vp]),
    "mp_skip_table_release": (c_int, [c_vp, c_vp, c_vp]),
    "mp_index": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_i64, c_vp, c_vp]),
    "mp_orthogonal": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "mp_query": (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_int, c_vp, c_i64, c_i64, c_i64, c_vp,
                         c_f32, c_vp, c_vp]),
    "mp_mlp_forward": (c_int, [c_vp, c_int, c_vp, c_i64, c_vp, c_vp]),
    "mp_query_counted": (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_int, c_vp, c_i64, c_vp, c_vp,
                                 c_f32, c_vp, c_vp]),
    "mp_query_counted_batch": (c_int, [c_vp, c_int, c_int, c_vp, c_int, c_int, c_int, c_vp, c_i64, c_vp,
                                       c_vp, c_f32, c_vp, c_vp]),
    "mp_recon": (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_int, c_vp, c_f32, _pf32, _pf32, _pint,
                         c_int, c_f32, c_vp, c_vp, c_vp]),
    "mp_recon_batch": (c_int, [c_vp, c_int, c_int, c_vp, c_int, c_int, c_int, c_vp, c_f32, _pf32, _pf32,
                               _pint, c_int, c_f32, c_vp, c_vp, c_vp]),
    "mp_recon_batch_ex": (c_int, [c_vp, c_int, c_int, c_vp, c_int, c_int, c_int, c_vp, c_f32, _pf32, _pf32,
                                  _pint, c_int, c_f32, c_int, c_vp, c_vp, c_vp]),
    "mp_recon_batch_early": (c_int, [c_vp, c_int, c_int, c_vp, c_int, c_int, c_int, c_vp, c_f32, _pf32, _pf32,
                                     _pint, c_int, c_f32, c_int, c_vp, c_vp, ctypes.POINTER(ReconEarly), c_vp]),
    "mp_concat3_add": (c_int, [c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_i64, c_vp, c_vp]),
    "mp_prepare_inputs": (c_int, [c_vp, c_vp, c_i64, _pf32, _pf32, c_vp, c_vp, c_vp]),
    "mp_octree_select": (c_int, [c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_f32, c_vp,
                                 c_vp, c_vp]),
    "mp_octree_select_box": (c_int, [c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_f32, c_vp,
                                     c_vp, c_vp]),
    "mp_octree_conflicts": (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp,
                                    c_vp]),
    "mp_lattice_points": (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_int, _pf32, _pf32, c_vp, c_vp]),
    "mp_scatter_nodes": (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_vp]),
    "mp_forward_vertices": (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mp_forward_vertices_batch": (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mp_paint_batch": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_i64, c_int, c_f32, c_f32, c_f32, c_f32,
                               c_vp, c_vp]),
    "mp_vertex_points": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, _pf32, c_vp, c_vp]),
    "mp_paint": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_i64, c_int, c_f32, c_f32, c_f32,
                         c_f32, c_vp, c_vp]),
    "mp_visualize": (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp]),
    "mp_marching_cubes": (c_int, [c_vp, c_vp, c_int, c_f32, _pf32, _pf32, c_vp, c_i64, c_vp, c_i64,
                                  c_vp, c_vp]),
    "mp_group_norm": (c_int, [c_vp, c_vp, c_int, c_int, c_i64, c_int, c_vp, c_vp, c_f32, c_int, c_vp,
                              c_vp]),
    "mp_upsample_bicubic2x": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "mp_conv3x3_pack": (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_vp]),
    "mp_conv3x3_pack_wino": (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_vp]),
    "mp_conv3x3_wino_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "mp_conv3x3_stat_slices": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "mp_conv3x3_tune": (None, [c_int]),
    "mp_query_tune": (None, [c_int]),
    "mp_conv3x3_gn": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_int, c_int, c_vp, c_int, c_vp,
                              c_vp, c_vp]),
    "mp_scale_shift_add": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_i64, c_vp, c_vp]),
    "mp_conv3x3_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "mp_conv3x3_pack16": (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp]),
    "mp_conv3x3_gn16": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_int, c_int, c_vp, c_vp, c_int,
                                c_vp, c_vp, c_vp]),
    "mp_conv1x1_pack": (c_int, [c_vp, c_vp, c_int, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "mp_conv1x1": (c_int, [c_vp, c_vp, c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_i64, c_vp, c_int,
                           c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mp_gn_stat_slices": (c_int, []),
    "mp_gn_stats": (c_int, [c_vp, c_vp, c_int, c_int, c_i64, c_int, c_vp, c_vp]),
    "mp_gn_finalize": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_i64, c_vp, c_vp, c_f32, c_vp,
                               c_vp]),
    "mp_conv3x3_ex": (c_int, [c_vp, ctypes.POINTER(Conv3x3Args), c_vp]),
    "mp_conv1x1_ex": (c_int, [c_vp, ctypes.POINTER(Conv1x1Args), c_vp]),
    "mp_conv1x1_stat_slices": (c_int, [c_i64]),
    "mp_gn_acc_replicas": (c_int, []),
    "mp_convk_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "mp_convk_packed_floats": (c_i64, [c_int, c_int, c_int]),
    "mp_convk_stat_slices": (c_int, [c_int, c_int, c_int, c_int]),
    "mp_convk_pack": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp]),
    "mp_convk": (c_int, [c_vp, ctypes.POINTER(ConvKArgs), c_vp]),
    "mp_avgpool2_gn": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, ctypes.POINTER(GnOut), c_vp]),
    "mp_upsample_bicubic2x_gn": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp,
                                         ctypes.POINTER(GnOut), c_vp]),
    "mp_gn_apply": (c_int, [c_vp, c_vp, ctypes.POINTER(GnIn), c_int, c_int, c_int, c_i64, c_vp, c_vp,
                            ctypes.POINTER(GnOut), c_vp]),
    "mp_plan_create": (c_int, [c_vp, c_int, ctypes.POINTER(c_vp)]),
    "mp_plan_add": (c_int, [c_vp, c_int, c_vp, c_i64, c_int]),
    "mp_plan_size": (c_int, [c_vp]),
    "mp_plan_run": (c_int, [c_vp, c_vp]),
    "mp_plan_destroy": (None, [c_vp]),
    "mp_mfma_clock_probe": (c_int, [c_vp, ctypes.c_float, ctypes.POINTER(ctypes.c_double), c_vp]),
    "mp_profile_begin": (c_int, [c_vp, c_int]),
    "mp_profile_end": (c_int, [c_vp, _pf32, c_int, _pint]),
}

_lib = None


def load():
    """Load the shared library (no GPU needed for this step) and declare every signature."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "monoport_amd: %s is missing -- build it with `python -m monoport_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        # torch ships its own libamdhip64; it must be in the process BEFORE our library resolves
        # the same SONAME, or two HIP runtimes coexist and device pointers / streams stop being
        # interchangeable (symptom: "no HIP device visible" from mp_create)
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        # measurement hook (tools/, DESIGN_HISTORY.md section 5): MONOPORT_QUERY_SMALL_TILES moves the gate
        # between the 32- and 64-point query kernels; the results do not depend on it
        if os.environ.get("MONOPORT_QUERY_SMALL_TILES"):
            lib.mp_query_tune(int(os.environ["MONOPORT_QUERY_SMALL_TILES"]))
        _lib = lib
    return _lib


class MonoportError(RuntimeError):
    """Raised for any non-zero status of the C-ABI (so pipeline stage threads surface it through
    the ExceptionWrapper path of RTL/dataloader.py:1042-1047)."""


def check(ctx_handle, rc, what):
    if rc != 0:
        msg = load().mp_last_error(ctx_handle)
        raise MonoportError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
