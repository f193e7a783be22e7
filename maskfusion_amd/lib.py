"""ctypes loader for libmaskfusion_amd.so (C ABI: include/maskfusion_amd.h)."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmaskfusion_amd.so")

MF_OK = 0
MF_N_TIMINGS = 18
TIMING_LABELS = ["Preprocess", "odomInit", "odom", "indexMap", "Fuse::Data", "Fuse::Update", "Fuse::Copy",
                 "IndexMap::ACTIVE", "Run", "icpIterations", "icpCoarse", "icpFine",
                 "mmGlobalProjection", "mmEdgeLabels", "mmBackgroundFuseClean", "mmHostStall", "mmObjectFuseClean", "mmHostWaitMs"]


MF_N_PASSES = 12
PASS_LABELS = ["bgGlobalProjection", "bgIndexMap", "bgFuseData", "bgFuseUpdate", "bgIndexMap2", "bgClean", "bgAppend", "bgPredict",
               "objGlobalProjection", "objFuseClean", "objPredict", "compaction"]


class MFError(RuntimeError):
    pass


class ModelInfo(C.Structure):
    """mf_model_info_t"""
    _fields_ = [("id", C.c_int32), ("class_id", C.c_int32), ("surfels", C.c_uint32), ("confidence_threshold", C.c_float),
                ("is_static", C.c_int32), ("age", C.c_uint32)]


class Config(C.Structure):
    """mf_config (include/maskfusion_amd.h)."""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float), ("device", C.c_int32), ("time_delta", C.c_int32), ("conf_global", C.c_float),
                ("conf_object", C.c_float), ("depth_cutoff", C.c_float), ("icp_weight", C.c_float),
                ("fast_odom", C.c_int32), ("so3", C.c_int32), ("pyramid", C.c_int32),
                ("max_depth_processed", C.c_float), ("outlier_coefficient", C.c_float), ("num_gsurfels", C.c_int32),
                ("num_osurfels", C.c_int32), ("enable_multiple_models", C.c_int32), ("model_spawn_offset", C.c_int32),
                ("track_all_models", C.c_int32), ("max_models", C.c_int32), ("rgb_only", C.c_int32),
                ("pose_log_capacity", C.c_int32), ("reserved", C.c_int32 * 3)]


# every symbol declared in include/maskfusion_amd.h (tests/test_abi.py checks the header against this table)
SYMBOLS = {
    "mf_default_config": (C.c_int, [C.POINTER(Config), C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float]),
    "mf_create": (C.c_int, [C.POINTER(Config), C.POINTER(C.c_void_p)]),
    "mf_destroy": (None, [C.c_void_p]),
    "mf_last_error": (C.c_char_p, [C.c_void_p]),
    "mf_process_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64,
                                   C.c_void_p, C.c_float, C.c_int32]),
    "mf_process_frame_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float]),
    "mf_sync": (C.c_int, [C.c_void_p]),
    "mf_set_mask_class_ids": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "mf_predict": (C.c_int, [C.c_void_p]),
    "mf_preallocate_models": (C.c_int, [C.c_void_p, C.c_uint32]),
    "mf_set_tick": (C.c_int, [C.c_void_p, C.c_int32]),
    "mf_get_tick": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "mf_num_models": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "mf_get_pose": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "mf_get_surfel_count": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_uint32)]),
    "mf_model_state_dev": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "mf_get_icp_stats": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "mf_get_track_stats": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "mf_get_gn_condition": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "mf_get_pose_log": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]),
    "mf_export_poses": (C.c_int, [C.c_void_p, C.c_char_p]),
    "mf_save_ply": (C.c_int, [C.c_void_p, C.c_char_p]),
    "mf_k_intensity": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "mf_k_pyrdown_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "mf_k_derivative_images": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "mf_k_so3_prealign": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "mf_k_rgb_residual": (C.c_int, [C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                    C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mf_k_rgb_step": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p,
                                C.c_void_p, C.c_float, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "mf_download_map": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]),
    "mf_get_last_fillin": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "mf_model_info": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "mf_download_segmentation": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mf_export_segmentation_png": (C.c_int, [C.c_void_p, C.c_char_p]),
    "mf_write_png_gray8": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32]),
    "mf_segmentation_labels": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mf_k_segmentation_labels": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mf_k_geometric_edges": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                       C.c_float, C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_void_p]),
    "mf_set_param": (C.c_int, [C.c_void_p, C.c_char_p, C.c_double]),
    "mf_get_param": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_double)]),
    "mf_get_timings": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mf_get_pass_timings": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mf_get_stream": (C.c_void_p, [C.c_void_p]),
    "mf_get_input_stream": (C.c_void_p, [C.c_void_p]),
    "mf_debug_read": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]),
    "mf_debug_read_model": (C.c_int, [C.c_void_p, C.c_int32, C.c_char_p, C.c_void_p, C.c_uint64]),
    "mf_stage_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mf_stage_frame_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mf_end_frame": (C.c_int, [C.c_void_p, C.c_int64]),
    "mf_track_models": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "mf_fuse_models": (C.c_int, [C.c_void_p, C.c_int32, C.c_float, C.c_int32]),
    "mf_predict_models": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64]),
    "mf_models_state_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "mf_get_model_ids": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "mf_model_initialise": (C.c_int, [C.c_void_p, C.c_int32]),
    "mf_model_override_pose": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "mf_model_fusion_weight": (C.c_int, [C.c_void_p, C.c_int32, C.c_float, C.POINTER(C.c_float)]),
    "mf_model_perform_tracking": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_int32, C.c_int32,
                                            C.c_float, C.c_int64, C.c_int32]),
    "mf_model_predict_indices": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_int32]),
    "mf_model_fuse": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float]),
    "mf_model_clean": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float]),
    "mf_model_combined_predict": (C.c_int, [C.c_void_p, C.c_int32, C.c_float, C.c_int32, C.c_int32, C.c_int32]),
    "mf_model_upload_map": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32]),
    "mf_export_projection_keys_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "mf_import_projection_keys_dev": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mf_perform_segmentation": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                          C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mf_perform_segmentation_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "mf_perform_segmentation_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mf_fuse_background": (C.c_int, [C.c_void_p, C.c_float]),
    "mf_export_segmentation_dev": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mf_import_segmentation_dev": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mf_spawn_object_model": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "mf_drop_model": (C.c_int, [C.c_void_p, C.c_int32]),
    "mf_model_update_static_pose": (C.c_int, [C.c_void_p, C.c_int32]),
    "mf_update_object_params": (C.c_int, [C.c_void_p]),
    "mf_make_nonstatic": (C.c_int, [C.c_void_p, C.c_int32]),
    "mf_make_static": (C.c_int, [C.c_void_p, C.c_int32]),
    "mf_set_trackable_class_ids": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "mf_k_bilateral": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "mf_k_pyrdown_f": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "mf_k_vmap_nmap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float,
                                 C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "mf_k_model_pyramid": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                     C.c_int32, C.c_void_p]),
    "mf_k_gn_solve": (C.c_int, [C.c_void_p] * 10 + [C.c_void_p]),
    "mf_k_icp_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int32,
                                C.c_int32, C.c_void_p, C.c_void_p]),
}

_lib = None


def load():
    """Loads the HIP extension.  Raises MFError if it has not been built (python maskfusion_amd/build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MFError(f"{LIB_PATH} is missing: build it with `python maskfusion_amd/build.py` "
                      "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)  # AttributeError here = ABI mismatch, fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L
