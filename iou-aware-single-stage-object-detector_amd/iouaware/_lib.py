"""ctypes binding of libiouaware_hip.so (the C-ABI declared in include/iouaware.h).

The library is the product: there is NO CPU / eager fallback.  If it is
missing, importing an op raises; the kernels themselves only run on a gfx950
device.
"""
import ctypes as C
import os

IA_MAX_LEVELS = 8
IA_MAX_ANCHORS = 16
IA_MAX_NMS_PRE = 4096
IA_MAX_CANDIDATES = 8192
IA_MAX_PER_IMG = 1024
IA_F32, IA_BF16, IA_F16, IA_F64 = 0, 1, 2, 3
IA_LAYOUT_NCHW, IA_LAYOUT_NHWC = 0, 1
IA_CLS_SIGMOID, IA_CLS_SOFTMAX = 0, 1
IA_LOSS_SLOTS = 64
IA_MAX_TARGET_BATCH = 16

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'csrc')
SO_PATH = os.path.abspath(os.path.join(_CSRC, 'libiouaware_hip.so'))


class HeadGeom(C.Structure):
    _fields_ = [('num_levels', C.c_int32), ('num_anchors', C.c_int32),
                ('num_classes', C.c_int32), ('nms_pre', C.c_int32),
                ('H', C.c_int32 * IA_MAX_LEVELS), ('W', C.c_int32 * IA_MAX_LEVELS),
                ('stride', C.c_int32 * IA_MAX_LEVELS),
                ('base_anchors', ((C.c_float * 4) * IA_MAX_ANCHORS) * IA_MAX_LEVELS),
                ('means', C.c_float * 4), ('stds', C.c_float * 4), ('layout', C.c_int32),
                ('cls_activation', C.c_int32)]


class LevelPtrs(C.Structure):
    _fields_ = [('cls', C.c_void_p * IA_MAX_LEVELS), ('reg', C.c_void_p * IA_MAX_LEVELS),
                ('iou', C.c_void_p * IA_MAX_LEVELS)]


class LevelPixStrides(C.Structure):
    _fields_ = [('cls', C.c_int64 * IA_MAX_LEVELS), ('reg', C.c_int64 * IA_MAX_LEVELS),
                ('iou', C.c_int64 * IA_MAX_LEVELS)]


class HeadTargets(C.Structure):
    _fields_ = [('labels', C.c_void_p * IA_MAX_LEVELS), ('label_weights', C.c_void_p * IA_MAX_LEVELS),
                ('bbox_targets', C.c_void_p * IA_MAX_LEVELS),
                ('bbox_weights', C.c_void_p * IA_MAX_LEVELS), ('counts', C.c_void_p),
                ('avg_factor_dev', C.c_void_p), ('avg_factor', C.c_float)]


class HeadLossCfg(C.Structure):
    _fields_ = [('gamma', C.c_float), ('alpha', C.c_float), ('loss_weight_cls', C.c_float),
                ('beta', C.c_float), ('loss_weight_bbox', C.c_float),
                ('attach_iou_target', C.c_int32), ('exact_large_logits', C.c_int32),
                ('grad_rows_start_at_reg', C.c_int32)]


class ImageDesc(C.Structure):
    _fields_ = [('src', C.c_void_p), ('src_h', C.c_int32), ('src_w', C.c_int32),
                ('dst_h', C.c_int32), ('dst_w', C.c_int32), ('flip', C.c_int32)]


class WinoGeom(C.Structure):
    _fields_ = [('num_levels', C.c_int32), ('batch', C.c_int32),
                ('H', C.c_int32 * IA_MAX_LEVELS), ('W', C.c_int32 * IA_MAX_LEVELS)]


class Conv3x3Desc(C.Structure):
    """ia_conv3x3_desc"""
    _fields_ = [('num_levels', C.c_int32), ('batch', C.c_int32), ('groups', C.c_int32),
                ('cin', C.c_int32), ('cout', C.c_int32), ('x_stride', C.c_int32), ('y_stride', C.c_int32),
                ('H', C.c_int32 * IA_MAX_LEVELS), ('W', C.c_int32 * IA_MAX_LEVELS),
                ('x', (C.c_void_p * IA_MAX_LEVELS) * 2), ('y', (C.c_void_p * IA_MAX_LEVELS) * 2)]


class WinoSeg(C.Structure):
    _fields_ = [('c0', C.c_int32), ('n', C.c_int32), ('dst_channels', C.c_int32),
                ('dst_offset', C.c_int32), ('dst', C.c_void_p * IA_MAX_LEVELS)]


_vp, _i, _f, _sz, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_int64
_G, _P = C.POINTER(HeadGeom), C.POINTER(LevelPtrs)

# name -> (restype, argtypes); mirrors include/iouaware.h one to one
SIGNATURES = {
    'ia_version': (C.c_char_p, []),
    'ia_geom_sizes': (_i, [_G, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    'ia_decode_fuse_rowmax': (_i, [_G, _P, _i, _i, _vp, _vp]),
    'ia_select_topk_workspace_bytes': (_sz, [_G, _i]),
    'ia_select_topk': (_i, [_G, _vp, _i, _vp, _vp, _sz, _vp]),
    'ia_decode_fuse_rowmax_grouped': (_i, [_G, _P, _i, _i, _vp, _vp, _sz, _vp]),
    'ia_select_topk_grouped': (_i, [_G, _vp, _i, _vp, _vp, _sz, _vp]),
    'ia_decode_stage': (_i, [_G, _P, _i, _i, _vp, _vp, _i, _vp, _sz, _vp]),
    'ia_get_bboxes_status_offset': (_sz, [_G, _i]),
    'ia_debug_fused_spin_limit': (_i, [_i64]),
    'ia_profile_stage_events': (_i, [_vp, _vp]),
    'ia_gather_decode': (_i, [_G, _P, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    'ia_multiclass_nms_workspace_bytes': (_sz, [_i, _i, _i]),
    'ia_multiclass_nms': (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _vp, _sz, _vp, _vp, _vp,
                               _vp, _vp, _vp, _vp]),
    'ia_multiclass_soft_nms_workspace_bytes': (_sz, [_i, _i, _i]),
    'ia_multiclass_soft_nms': (_i, [_vp, _vp, _i, _i, _i, _f, _f, _i, _f, _f, _i, _vp, _sz, _vp, _vp,
                                    _vp, _vp, _vp, _vp, _vp]),
    'ia_soft_nms': (_i, [_vp, _i, _f, _i, _f, _f, _vp, _vp, _vp, _vp]),
    'ia_get_bboxes_workspace_bytes': (_sz, [_G, _i]),
    'ia_get_bboxes': (_i, [_G, _P, _i, _i, _vp, _vp, _i, _f, _f, _i, _vp, _sz, _vp, _vp, _vp,
                           _vp, _vp]),
    'ia_multiclass_nms_lazy_workspace_bytes': (_sz, [_i, _i, _i]),
    'ia_multiclass_nms_lazy': (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _i, _vp, _sz, _vp, _vp, _vp,
                                    _vp, _vp]),
    'ia_get_bboxes_lazy': (_i, [_G, _P, _i, _i, _vp, _vp, _i, _f, _f, _i, _i, _vp, _sz, _vp, _vp, _vp,
                                _vp, _vp]),
    'ia_get_bboxes_workspace_layout': (_i, [_G, _i, C.POINTER(_sz * 8)]),
    'ia_nms_workspace_bytes': (_sz, [_i]),
    'ia_nms': (_i, [_vp, _i, _f, _vp, _vp, _vp, _sz, _vp]),
    'ia_nms_f64_workspace_bytes': (_sz, [_i]),
    'ia_nms_f64': (_i, [_vp, _i, _f, _vp, _vp, _vp, _sz, _vp]),
    'ia_image_transform': (_i, [C.POINTER(ImageDesc), _i, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                _i, _i, _i, _i, _vp, _vp]),
    'ia_wino_tiles': (_i, [C.POINTER(WinoGeom), C.POINTER(C.c_int32)]),
    'ia_wino_input_transform': (_i, [C.POINTER(WinoGeom), C.POINTER(C.c_void_p), _i, _i, _vp, _vp, _i,
                                     _vp, _vp]),
    'ia_wino_grad_output_transform': (_i, [C.POINTER(WinoGeom), C.POINTER(C.c_void_p), _i, _vp, _vp]),
    'ia_wino_weight_transform': (_i, [_vp, _i, _i, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _i,
                                      C.POINTER(C.c_double), _vp, _vp]),
    'ia_wino_weight_grad': (_i, [_vp, _i, _i, C.POINTER(C.c_double), _vp, C.c_int64, C.c_int64,
                                 C.c_int64, C.c_int64, _vp]),
    'ia_bn_fold_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    'ia_bn_fold_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    'ia_relu_bwd_bias_grad_workspace_bytes': (C.c_size_t, [C.c_int64, _i]),
    'ia_relu_bwd_bias_grad': (_i, [_vp, _vp, C.c_int64, _i, _vp, _vp, _vp, C.c_size_t, _vp]),
    'ia_wino_output_transform': (_i, [C.POINTER(WinoGeom), _vp, _i, _i, _vp, _i, _i,
                                      C.POINTER(WinoSeg), _vp]),
    'ia_linear_bias_act': (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp, _sz, _vp]),
    'ia_linear_bias_act_bf16': (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp, _sz, _vp]),
    'ia_batched_gemm': (_i, [_vp, _vp, _vp, _i, _i64, _i, _i, _vp, _sz, _vp]),
    'ia_batched_gemm_stream': (_i, [_vp, _vp, _vp, _i, _i64, _i, _i, _vp]),
    'ia_linear_bias_act_wt': (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp, _sz, _vp]),
    'ia_gemm_tuning': (_i, [_i]),
    'ia_im2col3x3_bytes': (_sz, [_i, _i, _i, _i, _i, _i]),
    'ia_im2col3x3_nhwc': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'ia_conv1x1_strided': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'ia_gemm_table_add': (_i, [_i64, _i64, _i64, _i, _i, _i, _i]),
    'ia_gemm_table_clear': (_i, []),
    'ia_gemm_table_dump': (_i, [_vp, _i]),
    'ia_gemm_table_stats': (_i, [_vp]),
    'ia_gemm_library_version': (_i, []),
    'ia_gemm_tn': (_i, [_vp, _vp, _vp, _i, _i64, _i, _i, _vp, _sz, _vp]),
    'ia_focal_loss_fwd': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _f, _f, _vp, _vp]),
    'ia_focal_loss_bwd': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _vp, _vp, _vp]),
    'ia_smooth_l1_fwd': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _f, _vp, _vp]),
    'ia_smooth_l1_bwd': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _f, _f, _vp, _vp, _vp]),
    'ia_focal_loss_balanced_fwd': (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _vp, _vp]),
    'ia_focal_loss_balanced_bwd': (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _vp, _f,
                                        _vp, _vp, _vp]),
    'ia_smooth_l1_balanced_fwd': (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _f, _f, _vp, _vp]),
    'ia_smooth_l1_balanced_bwd': (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _f, _f, _f, _vp, _vp,
                                       _vp]),
    'ia_iou_bce_fwd': (_i, [_G, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    'ia_iou_bce_bwd': (_i, [_G, _i, _vp, _vp, _i, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp]),
    'ia_anchor_targets': (_i, [_G, _vp, _vp, _vp, _i, _i, _vp, _f, _f, _f, _f, _vp, _vp, _vp, _vp,
                               _vp, _vp, _vp]),
    'ia_anchor_targets_ptrs': (_i, [_G, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                    C.POINTER(C.c_int32), _i, C.POINTER(C.c_int32), _f, _f, _f, _f,
                                    _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'ia_head_loss_workspace_bytes': (_sz, [_G, _i]),
    'ia_head_loss_fwd': (_i, [_G, _P, _i, _i, C.POINTER(HeadTargets), C.POINTER(HeadLossCfg), _vp,
                              _sz, _vp, _vp]),
    'ia_head_loss_bwd': (_i, [_G, _P, _i, _i, C.POINTER(HeadTargets), C.POINTER(HeadLossCfg), _vp,
                              _vp, _vp, _P, _vp]),
    'ia_head_loss_fwd_nhwc': (_i, [_G, _P, C.POINTER(LevelPixStrides), _i, C.POINTER(HeadTargets),
                                   C.POINTER(HeadLossCfg), _vp, _sz, _vp, _vp]),
    'ia_head_loss_bwd_nhwc': (_i, [_G, _P, C.POINTER(LevelPixStrides), _i, C.POINTER(HeadTargets),
                                   C.POINTER(HeadLossCfg), _vp, _vp, _P, C.POINTER(LevelPixStrides),
                                   _vp]),
    'ia_grouped_conv3x3_pack': (_i, [_vp, _vp, _i, _i, _vp]),
    'ia_grouped_conv3x3_nhwc': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'ia_sigmoid_focal_loss_fwd': (_i, [_vp, _vp, _i, _i, _f, _f, _vp, _vp]),
    'ia_sigmoid_focal_loss_bwd': (_i, [_vp, _vp, _vp, _i, _i, _f, _f, _vp, _vp]),
    'ia_sigmoid_focal_loss_fwd_dt': (_i, [_vp, _i, _vp, _i, _i, _f, _f, _vp, _vp]),
    'ia_sigmoid_focal_loss_bwd_dt': (_i, [_vp, _i, _vp, _vp, _i, _i, _f, _f, _vp, _vp]),
    'ia_channel_affine_act': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _vp]),
    'ia_channel_affine_act_nhwc': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _vp]),
    'ia_nhwc_to_nchw': (_i, [_vp, _vp, _i, _i, _i, _i64, _vp]),
    'ia_upsample2x_add_nhwc': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'ia_conv1x1_stream': (_i, [_vp, _vp, _vp, _vp, _vp, C.c_int64, _i, _i, _i, _vp]),
    'ia_stem_conv7x7s2': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    'ia_stem_conv7x7s2_bf16': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    'ia_conv1x1_wide': (_i, [_vp, _vp, _vp, _vp, _vp, C.c_int64, _i, _i, _i, _vp]),
    'ia_conv1x1_chain': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp]),
    'ia_conv3x3_bf16_packed_bytes': (C.c_size_t, [_i, _i, _i]),
    'ia_conv3x3_bf16_pack': (_i, [_vp, _i, _i, _i, _vp, _vp]),
    'ia_conv3x3_bf16_levels': (_i, [_vp, _vp, _vp, _i, _vp]),
    'ia_upsample2x_add_nhwc_dt': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'ia_affine_relu_maxpool_nhwc_dt': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'ia_affine_relu_maxpool_nhwc': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'ia_test_math': (_i, [_i, _vp, _vp, _vp, _i64, _vp]),
}

_lib = None


class IouAwareLibraryError(RuntimeError):
    pass


def lib():
    """The loaded shared library.  Raises (never falls back) when absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise IouAwareLibraryError(
                'libiouaware_hip.so not found at %s: build it with '
                '`python iou-aware-single-stage-object-detector_amd/csrc/build.py` '
                '(hipcc, gfx950). There is no CPU fallback.' % SO_PATH)
        h = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)     # AttributeError if the .so is stale: intended
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


_ERRORS = {-1: 'argument error', -2: 'workspace too small',
           -3: 'more than %d boxes / candidates in one call (IA_E_LIMIT_BOXES)' % 8192,
           -4: 'nms_pre above %d (IA_E_LIMIT_NMS_PRE)' % 4096,
           -5: 'max_per_img above %d (IA_E_LIMIT_PER_IMG)' % 1024}


def check(rc, what):
    if rc != 0:
        raise IouAwareLibraryError('%s failed: %s' % (what, _ERRORS.get(rc, 'hipError_t %d' % rc)))
