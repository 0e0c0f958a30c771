"""ctypes binding of libbts_render.so (the C ABI in include/bts_render.h).

The library is the product: there is NO Python/torch fallback for anything it exports.  If it is missing or a call
fails, a ``BtsNativeError`` is raised."""
import ctypes as C
import os
import threading

PKG = os.path.dirname(os.path.abspath(__file__))
# The product always loads libbts_render.so from the package directory.  The profiling / A-B tools load another build of the same sources
# (the instrumented probe build, an older revision's kernels) through BTS_RENDER_LIB -- honoured ONLY together with
# BTS_ALLOW_LIB_OVERRIDE=1 and announced on stderr, so that a stale variable in somebody's shell cannot silently swap the library.
_OVERRIDE = os.environ.get("BTS_RENDER_LIB") if os.environ.get("BTS_ALLOW_LIB_OVERRIDE") == "1" else None
LIB_PATH = _OVERRIDE or os.path.join(PKG, "libbts_render.so")
ABI_VERSION = 9
BTS_MAX_SCALES = 4
BTS_MAX_LOSS_VIEWS = 16

BTS_MAX_VIEWS = 8
ERRORS = {-1: "BTS_E_INVALID", -2: "BTS_E_UNSUPPORTED", -3: "BTS_E_LAUNCH", -4: "BTS_E_WORKSPACE"}


class BtsNativeError(RuntimeError):
    pass


class BtsFieldCfg(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("n", "H", "W", "C", "d_hidden", "n_blocks", "nv", "num_freqs", "code_mode", "inv_z",
                                         "learn_empty", "empty_empty")] + \
               [("freq_factor", C.c_float), ("d_min", C.c_float), ("d_max", C.c_float), ("feat_shift", C.c_int32), ("enc_render_view", C.c_int32),
                ("tile_blocks", C.c_int32)]       # ABI 9


class BtsFieldTensors(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("feat_nhwc", "proj_nhwc", "K_enc", "w2c_enc", "imgs_nhwc4", "K_r", "w2c_r",
                                          "empty_feature", "mlp_params")]


class BtsRenderArgs(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("rays_per_sample", "K", "hard_alpha_cap", "white_bkgd")] + \
               [(k, C.c_void_p) for k in ("rays", "z_samp", "rgb", "depth", "weights", "alphas", "invalid", "rgb_samps",
                                          "sigma_raw", "trans", "invalid_wsum", "invalid_any", "sigma_noise", "jitter", "z_samp_out")] + \
               [("lindisp", C.c_int32), ("reserved_", C.c_int32)]


class BtsRenderGrads(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("g_rgb", "g_depth", "g_weights", "g_alphas", "d_proj_nhwc", "d_mlp_params",
                                          "d_empty_proj", "d_proj_tiles")]


class BtsLossArgs(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("rgb", "depth", "weights", "invalid", "rgb_gt", "parts", "g_rgb", "g_depth")] + \
               [(k, C.c_int32) for k in ("n_patches", "patch_h", "patch_w", "nv", "K", "invalid_policy", "edge_aware_smoothness")] + \
               [("scale_rgb", C.c_float), ("scale_eas", C.c_float)] + [(k, C.c_void_p) for k in ("invalid_wsum", "invalid_any")]


class BtsTrainScale(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("feat_nchw", "jitter", "rgb", "depth", "invalid_wsum", "invalid_any", "proj_nhwc", "sampled_tiles",
                                          "z_samp", "sigma_raw", "trans", "rgb_samps", "loss_parts", "g_rgb", "g_depth", "gs_rgb", "gs_depth",
                                          "d_proj_nhwc", "d_proj_tiles", "d_feat_nchw")] + \
               [("feat_shift", C.c_int32), ("feat_channels_last", C.c_int32)]


class BtsTrainStep(C.Structure):
    _fields_ = [("cfg", BtsFieldCfg), ("v", C.c_int32), ("id_encoder", C.c_int32), ("ids_render", C.c_int32 * BTS_MAX_VIEWS),
                ("n_loss", C.c_int32), ("ids_loss", C.c_int32 * BTS_MAX_LOSS_VIEWS)] + \
               [(k, C.c_int32) for k in ("P", "ph", "pw", "K", "lindisp", "hard_alpha_cap", "invalid_policy", "edge_aware_smoothness", "n_scales",
                                         "concurrent_scales")] + \
               [(k, C.c_float) for k in ("z_near", "z_far", "img_scale", "img_shift")] + \
               [("loss_matrix", C.c_float * (9 * 3 * BTS_MAX_SCALES))] + \
               [(k, C.c_void_p) for k in ("images", "Ks", "poses_c2w", "patch_v", "patch_y", "patch_x", "mlp_params", "empty_feature", "rays", "rgb_gt",
                                          "loss_vals", "cams", "imgs_nhwc4", "bwd_workspace")] + \
               [("bwd_workspace_bytes", C.c_size_t)] + \
               [(k, C.c_void_p) for k in ("d_empty_proj", "d_mlp_params", "d_empty_feature")] + \
               [("scale", BtsTrainScale * BTS_MAX_SCALES)]


class BtsEvalFrame(C.Structure):
    _fields_ = [("cfg", BtsFieldCfg), ("v", C.c_int32), ("id_encoder", C.c_int32), ("ids_render", C.c_int32 * BTS_MAX_VIEWS)] + \
               [(k, C.c_int32) for k in ("K", "lindisp", "hard_alpha_cap", "norm_dir")] + \
               [(k, C.c_float) for k in ("z_near", "z_far", "img_scale", "img_shift")] + \
               [(k, C.c_void_p) for k in ("images", "Ks", "poses_c2w", "feat_nchw", "mlp_params", "empty_feature", "jitter", "cams", "imgs_nhwc4",
                                          "proj_nhwc", "inv_K", "rays", "rgb", "depth", "depth_z", "weights", "alphas", "invalid")] + \
               [("feat_channels_last", C.c_int32), ("reserved_", C.c_int32)]      # ABI 9 (appended)


class BtsConv3x3(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("N", "H", "W", "C", "up2", "elu", "out_nchw", "reserved_")] + \
               [(k, C.c_void_p) for k in ("x", "weight", "bias", "y")]


# every symbol include/bts_render.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_I = C.c_int32
SYMBOLS = {
    "bts_abi_version": (C.c_int, []),
    "bts_last_error": (C.c_char_p, []),
    "bts_supported": (C.c_int, [C.POINTER(BtsFieldCfg)]),
    "bts_mlp_param_count": (C.c_int64, [C.POINTER(BtsFieldCfg)]),
    "bts_render_fwd": (C.c_int, [C.POINTER(BtsFieldCfg), C.POINTER(BtsFieldTensors), C.POINTER(BtsRenderArgs), _P]),
    "bts_render_bwd_workspace": (C.c_size_t, [C.POINTER(BtsFieldCfg), C.POINTER(BtsRenderArgs)]),
    "bts_render_bwd": (C.c_int, [C.POINTER(BtsFieldCfg), C.POINTER(BtsFieldTensors), C.POINTER(BtsRenderArgs),
                                 C.POINTER(BtsRenderGrads), _P, C.c_size_t, _P]),
    "bts_project_features": (C.c_int, [C.POINTER(BtsFieldCfg), _P, _P, _I, _P, _P]),
    "bts_project_features_bwd": (C.c_int, [C.POINTER(BtsFieldCfg), _P, _P, _P, _I, _P, _P, _P]),
    "bts_proj_tile_count": (C.c_int64, [C.POINTER(BtsFieldCfg)]),
    "bts_project_features_bwd_tiles": (C.c_int, [C.POINTER(BtsFieldCfg), _P, _P, _P, _P, _I, _P, _P, _I, _P]),
    "bts_project_features_cl": (C.c_int, [C.POINTER(BtsFieldCfg), _P, _P, _I, _P, _P, _P]),
    "bts_project_features_bwd_cl": (C.c_int, [C.POINTER(BtsFieldCfg), _P, _P, _P, _P, _I, _P, _P, _I, _P]),
    "bts_mark_sampled_tiles": (C.c_int, [C.POINTER(BtsFieldCfg), _P, _P, C.POINTER(BtsRenderArgs), _P, _P]),
    "bts_project_features_tiles": (C.c_int, [C.POINTER(BtsFieldCfg), _P, _P, _I, _P, _P, _P]),
    "bts_field_query": (C.c_int, [C.POINTER(BtsFieldCfg), C.POINTER(BtsFieldTensors), _P, _I, _I, _P, _P, _P, _P]),
    "bts_occupancy_profile": (C.c_int, [C.POINTER(BtsFieldCfg), C.POINTER(BtsFieldTensors), _P, _I, _I, C.c_float, _I, _P, _P, _P]),
    "bts_nchw_to_nhwc": (C.c_int, [_P, _P, _I, _I, _I, _I, _P]),
    "bts_nhwc_to_nchw": (C.c_int, [_P, _P, _I, _I, _I, _I, _P]),
    "bts_pack_rgb": (C.c_int, [_P, _P, _I, _I, _I, C.c_float, C.c_float, _P]),
    "bts_gen_rays": (C.c_int, [_P, _P, _I, _I, _I, C.c_float, C.c_float, _I, _P, _P]),
    "bts_patch_rays": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, C.c_float, C.c_float, _I, _P, _P, _P]),
    "bts_photometric_loss": (C.c_int, [C.POINTER(BtsLossArgs), _P]),
    "bts_sample_coarse": (C.c_int, [_P, _P, C.c_int64, _I, _I, _P, _P]),
    "bts_distance_to_z": (C.c_int, [_P, _P, _I, _I, _I, _P, _P]),
    "bts_invert_small": (C.c_int, [_P, _P, _I, _I, _P]),
    "bts_train_step_fwd": (C.c_int, [C.POINTER(BtsTrainStep), _P]),
    "bts_train_step_bwd": (C.c_int, [C.POINTER(BtsTrainStep), _P, _P]),
    "bts_eval_frame": (C.c_int, [C.POINTER(BtsEvalFrame), _P]),
    "bts_conv3x3_fwd": (C.c_int, [C.POINTER(BtsConv3x3), _P]),
    "bts_conv3x3_bwd_workspace": (C.c_size_t, [C.POINTER(BtsConv3x3)]),
    "bts_conv3x3_bwd": (C.c_int, [C.POINTER(BtsConv3x3), _P, _P, C.c_size_t, _P, _P, _P, _P]),
}

_lock = threading.Lock()
_lib = None


def load():
    """Loads the shared library (once).  Raises BtsNativeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise BtsNativeError(
                f"{LIB_PATH} not found: the HIP renderer has not been built. Run `python -m behindthescenes_amd.build` "
                "(needs hipcc, no GPU). There is no fallback path.")
        if _OVERRIDE:
            import sys
            print(f"behindthescenes_amd: loading {LIB_PATH} instead of the package's libbts_render.so (BTS_RENDER_LIB + BTS_ALLOW_LIB_OVERRIDE=1)",
                  file=sys.stderr)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise BtsNativeError(f"{LIB_PATH} does not export {name}; rebuild it") from e
            fn.restype, fn.argtypes = res, args
        # (A/B tools load an older revision's kernels through BTS_RENDER_LIB: the structs only ever grew at their ends, so a library
        # of an older ABI reads the prefix it knows -- opt-in, tools only)
        if lib.bts_abi_version() != ABI_VERSION and not (_OVERRIDE and os.environ.get("BTS_ALLOW_OLDER_ABI") == "1"
                                                         and lib.bts_abi_version() < ABI_VERSION):
            raise BtsNativeError(f"ABI mismatch: library {lib.bts_abi_version()} vs binding {ABI_VERSION}; rebuild")
        _lib = lib
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().bts_last_error().decode(errors="replace")
        raise BtsNativeError(f"{what} failed: {ERRORS.get(rc, rc)}: {msg}")
