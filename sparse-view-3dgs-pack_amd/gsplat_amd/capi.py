"""ctypes view of the C ABI declared in include/gsplat.h.

This module only describes the ABI (structs + prototypes) and loads a shared library that
implements it.  The product loads ``csrc/libgsplat_hip.so`` with the ``gs_`` prefix (see
``gsplat_amd/_lib.py``); it raises if that library is missing - there is no CPU fallback.
(The test-suite binds the same prototypes, under its own symbol prefix, onto its CPU checker.)
"""
import ctypes as C
import os

c_f_p = C.c_void_p  # all device/host array pointers travel as raw addresses


class GsView(C.Structure):
    _fields_ = [
        ("image_height", C.c_int32),
        ("image_width", C.c_int32),
        ("tanfovx", C.c_float),
        ("tanfovy", C.c_float),
        ("scale_modifier", C.c_float),
        ("sh_degree", C.c_int32),
        ("prefiltered", C.c_int32),
        ("antialiasing", C.c_int32),
        ("debug", C.c_int32),
        ("tile_cull", C.c_int32),
        ("bg", C.c_void_p),
        ("viewmatrix", C.c_void_p),
        ("projmatrix", C.c_void_p),
        ("campos", C.c_void_p),
    ]


class GsGaussians(C.Structure):
    _fields_ = [
        ("P", C.c_int32),
        ("M", C.c_int32),
        ("means3D", C.c_void_p),
        ("shs", C.c_void_p),
        ("colors_precomp", C.c_void_p),
        ("opacities", C.c_void_p),
        ("scales", C.c_void_p),
        ("rotations", C.c_void_p),
        ("cov3D_precomp", C.c_void_p),
        ("extra_channel", C.c_void_p),
        ("raw_activations", C.c_int32),
        ("extra_gain", C.c_void_p),
        ("shs_rest", C.c_void_p),
    ]


class GsScratch(C.Structure):
    _fields_ = [
        ("geom", C.c_void_p),
        ("geom_bytes", C.c_size_t),
        ("img", C.c_void_p),
        ("img_bytes", C.c_size_t),
        ("binning", C.c_void_p),
        ("binning_bytes", C.c_size_t),
        ("binning_capacity", C.c_int64),
        ("tile_order_hint", C.c_void_p),
        ("tile_depth_limit", C.c_void_p),
        ("tile_order_out", C.c_void_p),
        ("tile_depth_limit_out", C.c_void_p),
        ("binned", C.c_int32),
        ("defer_tile_order", C.c_int32),
        ("step_tag", C.c_void_p),
        ("tile_depth_limit_slack", C.c_void_p),
        ("status_host", C.c_void_p),
    ]


class GsGrads(C.Structure):
    _fields_ = [
        ("dL_dmeans3D", C.c_void_p),
        ("dL_dmeans2D", C.c_void_p),
        ("dL_dsh", C.c_void_p),
        ("dL_dcolors", C.c_void_p),
        ("dL_dopacity", C.c_void_p),
        ("dL_dscales", C.c_void_p),
        ("dL_drotations", C.c_void_p),
        ("dL_dcov3D", C.c_void_p),
        ("dL_dextra", C.c_void_p),
    ]


class GsLgdwtParams(C.Structure):
    _fields_ = [("lambda_dssim", C.c_float), ("n_pix", C.c_float), ("n_band1", C.c_float), ("n_band2", C.c_float),
                ("dwt_w", C.c_float * 8), ("patch_w", C.c_float * 3), ("patch_weight", C.c_float),
                ("patch_elems_per_sel", C.c_float), ("dwt_enable", C.c_int32), ("patch_enable", C.c_int32),
                ("reset_sums", C.c_int32), ("custom_base", C.c_int32), ("w_l1", C.c_float), ("w_ssim", C.c_float)]


class GsAdamSeg(C.Structure):
    _fields_ = [("begin", C.c_int64), ("end", C.c_int64), ("lr_a", C.c_float), ("lr_b", C.c_float),
                ("period", C.c_int32), ("split", C.c_int32), ("step", C.c_int32), ("row_width", C.c_int32)]


class GsStepState(C.Structure):
    _fields_ = [("xyz", C.c_void_p), ("features", C.c_void_p), ("opacity", C.c_void_p), ("scaling", C.c_void_p),
                ("rotation", C.c_void_p), ("m", C.c_void_p * 5), ("v", C.c_void_p * 5), ("lr", C.c_float * 6),
                ("step", C.c_int32 * 5), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("max_radii2D", C.c_void_p), ("xyz_gradient_accum", C.c_void_p), ("denom", C.c_void_p),
                ("coef_dev", C.c_void_p), ("rows_override", C.c_void_p), ("grad_out", C.c_void_p * 5),
                ("fail_flag", C.c_void_p), ("phase", C.c_int32), ("rows_clean", C.c_int32),
                ("phase1_done", C.c_void_p),
                # 4th blended channel (gs_backward_step_x): its raw row, the global gain, their moments and rates
                ("extra", C.c_void_p), ("extra_m", C.c_void_p), ("extra_v", C.c_void_p), ("gain", C.c_void_p),
                ("gain_m", C.c_void_p), ("gain_v", C.c_void_p), ("lr_extra", C.c_float), ("lr_gain", C.c_float),
                ("step_extra", C.c_int32), ("step_gain", C.c_int32), ("grad_out_extra", C.c_void_p),
                ("grad_out_gain", C.c_void_p), ("grad_mask", C.c_void_p), ("dormant", C.c_void_p), ("sparse", C.c_int32),
                ("grad_out_rest", C.c_void_p)]


_P = C.c_void_p
_I32 = C.c_int32
_I64 = C.c_int64
_F = C.c_float
_SZ = C.c_size_t

# name -> (restype, argtypes); the single source of truth for "every symbol include/gsplat.h
# declares" (tests/test_abi.py checks this table against the header and against the .so).
PROTOTYPES = {
    "abi_version": (C.c_int, []),
    "struct_bytes": (C.c_size_t, [C.c_int32]),
    "build_info": (C.c_char_p, []),
    "scratch_bytes": (C.c_int, [_I32, _I32, _I32, _I64, C.POINTER(_SZ), C.POINTER(_SZ)]),
    "forward_geometry": (C.c_int, [C.POINTER(GsView), C.POINTER(GsGaussians), C.POINTER(GsScratch), _P, _P, _P]),
    "forward_render": (C.c_int, [C.POINTER(GsView), C.POINTER(GsGaussians), C.POINTER(GsScratch), _P, _P, _P]),
    "backward": (C.c_int, [C.POINTER(GsView), C.POINTER(GsGaussians), _P, C.POINTER(GsScratch), _I64, _P, _P,
                           C.POINTER(GsGrads), _P, _SZ, _P]),
    "forward_render_x": (C.c_int, [C.POINTER(GsView), C.POINTER(GsGaussians), C.POINTER(GsScratch), _P, _P, _P, _P]),
    "backward_x": (C.c_int, [C.POINTER(GsView), C.POINTER(GsGaussians), _P, C.POINTER(GsScratch), _I64, _P, _P, _P,
                             C.POINTER(GsGrads), _P, _SZ, _P]),
    "forward_render_fsgs": (C.c_int, [C.POINTER(GsView), C.POINTER(GsGaussians), C.POINTER(GsScratch), _P, _P, _P, _P]),
    "backward_fsgs": (C.c_int, [C.POINTER(GsView), C.POINTER(GsGaussians), _P, C.POINTER(GsScratch), _I64, _P, _P, _P,
                                C.POINTER(GsGrads), _P, _SZ, _P]),
    "step_uninstanced": (C.c_int, [C.POINTER(GsView), C.POINTER(GsGaussians), _P, C.POINTER(GsScratch),
                                   C.POINTER(GsStepState), _P]),
    "backward_step": (C.c_int, [C.POINTER(GsView), C.POINTER(GsGaussians), _P, C.POINTER(GsScratch), _I64, _P, _P,
                                C.POINTER(GsStepState), _P, _SZ, _P]),
    "backward_step_x": (C.c_int, [C.POINTER(GsView), C.POINTER(GsGaussians), _P, C.POINTER(GsScratch), _I64, _P, _P, _P,
                                  C.POINTER(GsStepState), _P, _SZ, _P]),
    "backward_from_rows": (C.c_int, [C.POINTER(GsView), C.POINTER(GsGaussians), _P, C.POINTER(GsScratch), _P, _I32,
                                     C.POINTER(GsGrads), _P, _SZ, _P]),
    "mark_visible": (C.c_int, [_I32, _P, _P, _P, _P, _P]),
    "export_geom": (C.c_int, [C.POINTER(GsScratch), _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "export_binning": (C.c_int, [C.POINTER(GsScratch), _I64, _P, _P, _P]),
    "export_img": (C.c_int, [C.POINTER(GsScratch), _I32, _I32, _P, _P, _P, _P]),
    "export_tile_order": (C.c_int, [C.POINTER(GsScratch), _I32, _I32, _P, _P]),
    "export_tile_stop_depth": (C.c_int, [C.POINTER(GsScratch), _I32, _I32, _P, _P]),
    "forward_status": (C.c_int, [C.POINTER(GsScratch), _P, _P]),
    "forward_tile_order": (C.c_int, [C.POINTER(GsView), C.POINTER(GsScratch), _P]),
    "forward_bin": (C.c_int, [C.POINTER(GsView), C.POINTER(GsGaussians), C.POINTER(GsScratch), _P, _P]),
    "export_binning_region": (C.c_int, [C.POINTER(GsScratch), _I32, _I32, _I64, _P, _P, _P]),
    "debug_blend_stats": (C.c_int, [C.POINTER(GsScratch), _I32, _I32, _I32, _P, _P]),
    "tile_depth_limit_floats": (C.c_size_t, [_I32, _I32]),
    "knn_tmp_bytes": (_SZ, [_I32]),
    "knn_mean_dist2": (C.c_int, [_P, _I32, _P, _P, _SZ, _P]),
    "knn_mean_dist2_idx": (C.c_int, [_P, _I32, _P, _P, _P, _SZ, _P]),
    "l1_fwd": (C.c_int, [_P, _P, _I64, _P, _P]),
    "l1_bwd": (C.c_int, [_P, _P, _I64, _F, _P, _I32, _P]),
    "dwt_haar_fwd": (C.c_int, [_P, _I32, _I32, _I32, _P, _P, _P, _P, _P]),
    "dwt_haar_bwd": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _P, _P]),
    "dwt2_l1_fwd": (C.c_int, [_P, _P, _I32, _I32, _I32, _P, _P]),
    "dwt2_l1_bwd": (C.c_int, [_P, _P, _I32, _I32, _I32, _P, _P, _I32, _P]),
    "elf_map": (C.c_int, [_P, _I32, _I32, _I32, _P, _P, _P]),
    "patch_means": (C.c_int, [_P, _I32, _I32, _I32, _P, _P]),
    "patch_dwt_fwd": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _P, _P, _P]),
    "patch_dwt_bwd": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _P, _P, _P, _I32, _P]),
    "ssim_fwd": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _F, _F, _P, _P, _P, _P, _P]),
    "ssim_bwd": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _F, _F, _P, _P, _P, _P, _P, _P]),
    "dwt_partials_count": (C.c_int64, [_I32, _I32, _I32]),
    "l1_dwt2_patch_fwd_clamp_p": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _P, _P, _P, _P]),
    "l1_partials_count": (C.c_int64, [_I64]),
    "l1_fwd_p": (C.c_int, [_P, _P, _I64, _P, _P]),
    "depth_l1_partials_count": (C.c_int64, [_I64]),
    "depth_l1": (C.c_int, [_P, _P, _P, _I64, _P, _F, _P, _P, _P]),
    "lgdwt_combine_pp": (C.c_int, [_P, _P, _I64, _P, _I64, _P, _I64, _P, C.POINTER(GsLgdwtParams), _P, _P]),
    "l1_bwd_dev": (C.c_int, [_P, _P, _I64, _P, _P, _I32, _P]),
    "l1_dwt2_fwd": (C.c_int, [_P, _P, _I32, _I32, _I32, _P, _P, _P]),
    "l1_dwt2_fwd_clamp": (C.c_int, [_P, _P, _I32, _I32, _I32, _P, _P, _P, _P]),
    "l1_dwt2_patch_fwd_clamp": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P]),
    "l1_dwt2_patch_bwd": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _I32, _P]),
    "l1_dwt2_bwd": (C.c_int, [_P, _P, _I32, _I32, _I32, _P, _P, _P, _I32, _P]),
    "ssim_fwd_sum": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _F, _F, _P, _P, _P, _P, _P]),
    "ssim_bwd_uniform": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _I32, _P, _P]),
    "lgdwt_combine": (C.c_int, [_P, _P, C.POINTER(GsLgdwtParams), _P, _P]),
    "lgdwt_combine_p": (C.c_int, [_P, _P, _I64, _P, C.POINTER(GsLgdwtParams), _P, _P]),
    "ssim_partials_count": (C.c_int64, [_I32, _I32, _I32, _I32]),
    "ssim_fwd_partials": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _F, _F, _P, _P, _P, _P, _P]),
    "adam_step": (C.c_int, [_P, _P, _P, _P, _I64, C.POINTER(GsAdamSeg), _I32, _F, _F, _F, _I32, _P]),
    "adam_step_gated": (C.c_int, [_P, _P, _P, _P, _I64, C.POINTER(GsAdamSeg), _I32, _F, _F, _F, _I32, _P, _P]),
    "adam_step_masked": (C.c_int, [_P, _P, _P, _P, _I64, C.POINTER(GsAdamSeg), _I32, _F, _F, _F, _I32, _P, _P, _P]),
    "activations_fwd": (C.c_int, [_P, _P, _P, _I32, _P, _P, _P, _P]),
    "activations_bwd": (C.c_int, [_P, _P, _P, _I32, _P, _P, _P, _P, _P, _P, _P]),
    "densify_stats": (C.c_int, [_P, _P, _I32, _P, _P, _P, _P]),
    "profile_enable": (C.c_int, [_I32]),
    "profile_only": (C.c_int, [_I32]),
    "profile_reset": (C.c_int, []),
    "profile_stage_count": (C.c_int, []),
    "rows_pack": (C.c_int, [_P, _I32, _I32, C.POINTER(C.c_int32), _P, _P, _I32, _P, _P]),
    "rows_unpack": (C.c_int, [_P, _I32, _I32, C.POINTER(C.c_int32), _P, _P, _I32, _P, _P]),
    "export_row_mask": (C.c_int, [C.POINTER(GsScratch), _I32, _P, _P]),
    "profile_stage_name": (C.c_char_p, [_I32]),
    "profile_read": (C.c_int, [C.POINTER(C.c_double), C.POINTER(_I64), _I32]),
}

# entry points only the device library has to provide (the CPU oracle is timed with a wall clock)
# (and the fused 4-channel pass is a product-side fusion of two reference passes: its parity target is the
# reference's two 3-channel passes, so the checker does not need it)
DEVICE_ONLY = ("export_row_mask", "backward_step", "backward_step_x", "step_uninstanced", "export_tile_order", "export_tile_stop_depth", "forward_status", "forward_bin", "export_binning_region", "debug_blend_stats", "adam_step_gated", "tile_depth_limit_floats", "profile_enable", "profile_only", "profile_reset", "profile_stage_count", "profile_stage_name", "profile_read",
               "forward_render_x", "backward_x", "forward_tile_order")

ERRORS = {-1: "GS_E_NULL", -2: "GS_E_SHAPE", -3: "GS_E_SCRATCH", -4: "GS_E_OVERFLOW", -5: "GS_E_UNSUPPORTED"}


class GsError(RuntimeError):
    def __init__(self, fn, code):
        self.code = code
        what = ERRORS.get(code, "hipError_t %d" % code if code > 0 else "error %d" % code)
        super().__init__("%s failed: %s" % (fn, what))


class CApi:
    """A loaded implementation of the gsplat C ABI."""

    def __init__(self, path, prefix="gs_", optional=()):
        if not os.path.exists(path):
            raise ImportError(
                "gsplat native library not found at %s - build it first (python __graft_entry__.py "
                "or `make -C sparse-view-3dgs-pack_amd/csrc`); there is no fallback path." % path)
        self.path = path
        self.prefix = prefix
        self.lib = C.CDLL(path)
        self.missing = []
        for name, (res, args) in PROTOTYPES.items():
            try:
                f = getattr(self.lib, prefix + name)
            except AttributeError:
                if name in optional:
                    self.missing.append(name)
                    continue
                raise ImportError("%s does not export %s%s" % (path, prefix, name))
            f.restype = res
            f.argtypes = args
            setattr(self, "_" + name, f)

    def call(self, name, *args):
        """Call an int-status entry point; raise GsError on a non-zero status."""
        rc = getattr(self, "_" + name)(*args)
        if rc != 0:
            raise GsError(self.prefix + name, rc)

    def raw(self, name):
        return getattr(self, "_" + name)


def read_profile(api):
    """{stage name: (total ms, launches)} from the device library's HIP-event timers."""
    n = api.raw("profile_stage_count")()
    ms = (C.c_double * n)()
    cnt = (C.c_int64 * n)()
    api.call("profile_read", ms, cnt, n)
    return {api.raw("profile_stage_name")(i).decode(): (ms[i], cnt[i]) for i in range(n) if cnt[i] > 0}
