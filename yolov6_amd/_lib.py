"""ctypes binding of libyolov6_hip.so (the C ABI declared in include/yolov6_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, a
RuntimeError is raised (the reference's assigner fallback catches exactly RuntimeError,
yolov6/models/losses/loss.py:105).  torch is used only for device memory and streams.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# Y6_LIB_PATH: developer override (e.g. a ceiling-probe build from tools/build_probe_libs.py); never a fallback
LIB_PATH = os.environ.get("Y6_LIB_PATH") or os.path.join(_HERE, "lib", "libyolov6_hip.so")

Y6_F16, Y6_F32, Y6_U8 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_SILU, ACT_HARDSWISH = 0, 1, 2, 3
ACT_BY_NAME = {None: ACT_NONE, "relu": ACT_RELU, "silu": ACT_SILU, "hardswish": ACT_HARDSWISH}
MAX_LEVELS = 4


class Tensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
                ("cstride", C.c_int32), ("coff", C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [("inp", Tensor), ("out", Tensor), ("w_packed", C.c_void_p), ("w_oihw", C.c_void_p),
                ("bias", C.c_void_p), ("post_scale", C.c_void_p), ("post_shift", C.c_void_p), ("res", Tensor),
                ("res_alpha", C.c_void_p), ("ksize", C.c_int32), ("stride", C.c_int32), ("act", C.c_int32),
                ("variant", C.c_int32)]


class ConvI8Desc(C.Structure):
    _fields_ = [("conv", ConvDesc), ("dequant", C.c_void_p), ("in_amax", C.c_float), ("q_in", Tensor), ("q_out", Tensor),
                ("q_out_amax", C.c_float), ("acc_out", C.c_void_p)]


class ConvGeometry(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("tile_h", "tile_w", "tiles_x", "tiles_y", "items", "cout_blocks", "block_pixels", "halo_h",
                                         "halo_w", "halo_pieces", "halo_pieces_max", "row_pitch")] + \
               [("lds_bytes", C.c_uint64), ("lds_limit", C.c_uint64)]


class ConvTDesc(C.Structure):
    _fields_ = [("inp", Tensor), ("out", Tensor), ("w_packed", C.c_void_p), ("bias", C.c_void_p)]


class StemDesc(C.Structure):
    _fields_ = [("in_nchw", C.c_void_p), ("in_dtype", C.c_int32), ("B", C.c_int32), ("Cin", C.c_int32),
                ("H", C.c_int32), ("W", C.c_int32), ("out", Tensor), ("w_oihw_f32", C.c_void_p),
                ("bias", C.c_void_p), ("post_scale", C.c_void_p), ("post_shift", C.c_void_p), ("act", C.c_int32),
                ("q_out", Tensor), ("q_out_amax", C.c_float)]


class DecodeDesc(C.Structure):
    _fields_ = [("n_levels", C.c_int32), ("cls", Tensor * MAX_LEVELS), ("reg", Tensor * MAX_LEVELS),
                ("stride", C.c_float * MAX_LEVELS), ("use_dfl", C.c_int32), ("reg_max", C.c_int32),
                ("proj", C.c_void_p), ("grid_cell_offset", C.c_float), ("out", C.c_void_p), ("nc", C.c_int32)]


class LetterboxDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("H", C.c_int32), ("W", C.c_int32), ("dst", C.c_void_p), ("out_h", C.c_int32),
                ("out_w", C.c_int32), ("new_h", C.c_int32), ("new_w", C.c_int32), ("top", C.c_int32), ("left", C.c_int32),
                ("planar", C.c_int32), ("reverse_channels", C.c_int32), ("pad", C.c_int32 * 3)]


class PwS2Desc(C.Structure):
    _fields_ = [("pw", ConvDesc), ("s2", ConvDesc)]


class StemS2Desc(C.Structure):
    _fields_ = [("stem", StemDesc), ("s2", ConvDesc)]


class NmsSink(C.Structure):
    _fields_ = [("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("conf_thres", C.c_float),
                ("classes", C.c_void_p), ("n_classes", C.c_int32), ("multi_label", C.c_int32)]


class PredDecodeDesc(C.Structure):
    _fields_ = [("n_levels", C.c_int32), ("cls_feat", Tensor * MAX_LEVELS), ("reg_feat", Tensor * MAX_LEVELS),
                ("w_cls", C.c_void_p * MAX_LEVELS), ("w_reg", C.c_void_p * MAX_LEVELS),
                ("b_cls", C.c_void_p * MAX_LEVELS), ("b_reg", C.c_void_p * MAX_LEVELS),
                ("stride", C.c_float * MAX_LEVELS), ("use_dfl", C.c_int32), ("reg_max", C.c_int32),
                ("proj", C.c_void_p), ("grid_cell_offset", C.c_float), ("out", C.c_void_p), ("nc", C.c_int32),
                ("first_anchor", C.c_int32), ("total_anchors", C.c_int32), ("cand", NmsSink)]


class NmsDesc(C.Structure):
    _fields_ = [("pred", C.c_void_p), ("B", C.c_int32), ("A", C.c_int32), ("nc", C.c_int32),
                ("conf_thres", C.c_float), ("iou_thres", C.c_float), ("classes", C.c_void_p),
                ("n_classes", C.c_int32), ("agnostic", C.c_int32), ("multi_label", C.c_int32),
                ("max_det", C.c_int32), ("max_nms", C.c_int32), ("max_wh", C.c_float), ("out_dets", C.c_void_p),
                ("out_index", C.c_void_p), ("out_count", C.c_void_p), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_size_t), ("candidates_ready", C.c_int32)]


class TalDesc(C.Structure):
    _fields_ = [("pd_scores", C.c_void_p), ("pd_bboxes", C.c_void_p), ("anc_points", C.c_void_p),
                ("gt_labels", C.c_void_p), ("gt_bboxes", C.c_void_p), ("mask_gt", C.c_void_p), ("B", C.c_int32),
                ("A", C.c_int32), ("C", C.c_int32), ("G", C.c_int32), ("topk", C.c_int32), ("alpha", C.c_float),
                ("beta", C.c_float), ("eps", C.c_float), ("target_labels", C.c_void_p),
                ("target_bboxes", C.c_void_p), ("target_scores", C.c_void_p), ("fg_mask", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


class AtssDesc(C.Structure):
    _fields_ = [("anc_bboxes", C.c_void_p), ("n_level_bboxes", C.c_int32 * MAX_LEVELS), ("n_levels", C.c_int32),
                ("gt_labels", C.c_void_p), ("gt_bboxes", C.c_void_p), ("mask_gt", C.c_void_p),
                ("pd_bboxes", C.c_void_p), ("B", C.c_int32), ("A", C.c_int32), ("C", C.c_int32), ("G", C.c_int32),
                ("topk", C.c_int32), ("target_labels", C.c_void_p), ("target_bboxes", C.c_void_p),
                ("target_scores", C.c_void_p), ("fg_mask", C.c_void_p), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_size_t)]


class LossDesc(C.Structure):
    _fields_ = [("pred_scores", C.c_void_p), ("pred_distri", C.c_void_p), ("pred_bboxes", C.c_void_p),
                ("anchor_points_s", C.c_void_p), ("stride", C.c_void_p), ("target_labels", C.c_void_p),
                ("target_bboxes", C.c_void_p), ("target_scores", C.c_void_p), ("fg_mask", C.c_void_p),
                ("B", C.c_int32), ("A", C.c_int32), ("C", C.c_int32), ("use_dfl", C.c_int32), ("reg_max", C.c_int32),
                ("iou_type", C.c_int32), ("w_class", C.c_float), ("w_iou", C.c_float), ("w_dfl", C.c_float),
                ("out", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("box_mode", C.c_int32),
                ("norm_mode", C.c_int32)]


class DistillDesc(C.Structure):
    _fields_ = [("scores_s", C.c_void_p), ("scores_t", C.c_void_p), ("distri_s", C.c_void_p), ("distri_t", C.c_void_p),
                ("fg_mask", C.c_void_p), ("target_scores", C.c_void_p), ("BA", C.c_int32), ("C", C.c_int32), ("reg_max", C.c_int32),
                ("temperature", C.c_float), ("acc", C.c_void_p), ("coef", C.c_void_p), ("dscores", C.c_void_p), ("ddistri", C.c_void_p)]


class BnTrainDesc(C.Structure):
    _fields_ = [("x", Tensor), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("running_mean", C.c_void_p),
                ("running_var", C.c_void_p), ("num_batches_tracked", C.c_void_p), ("momentum", C.c_float), ("eps", C.c_float),
                ("scale", C.c_void_p), ("shift", C.c_void_p), ("mean", C.c_void_p), ("invstd", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("workspace_clean", C.c_int32)]


class BnTrainMultiDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("d", BnTrainDesc * 3)]


class BnActDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("x", Tensor * 3), ("scale", C.c_void_p * 3), ("shift", C.c_void_p * 3), ("res", Tensor),
                ("res_alpha", C.c_void_p), ("out", Tensor), ("act", C.c_int32)]


class BnActBwdDesc(C.Structure):
    _fields_ = [("fwd", BnActDesc), ("mean", C.c_void_p * 3), ("invstd", C.c_void_p * 3), ("gamma", C.c_void_p * 3),
                ("dout", Tensor), ("dx", Tensor * 3), ("dx_dil", C.c_int32 * 3), ("dx_acc", C.c_int32 * 3),
                ("dgamma", C.c_void_p * 3), ("dbeta", C.c_void_p * 3), ("dres", Tensor), ("dres_acc", C.c_int32),
                ("dalpha", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("workspace_clean", C.c_int32)]


class WgradTDesc(C.Structure):
    _fields_ = [("src", Tensor), ("nchw", C.c_int32), ("src_dtype", C.c_int32), ("sy", C.c_int32), ("sx", C.c_int32),
                ("oy", C.c_int32), ("ox", C.c_int32), ("R", C.c_int32), ("Q", C.c_int32), ("dst", C.c_void_p)]


class WgradDesc(C.Structure):
    _fields_ = [("mode", C.c_int32), ("a", C.c_void_p), ("M", C.c_int32), ("N", C.c_int32), ("B", C.c_int32), ("Q", C.c_int32),
                ("rows", C.c_int32), ("a_rows", C.c_int32), ("a_channels", C.c_int32), ("plane_channels", C.c_int32),
                ("plane", C.c_void_p * 6), ("plane_rows", C.c_int32 * 6),
                ("drow", C.c_int32 * 6), ("out", C.c_void_p), ("sm", C.c_int32), ("sn", C.c_int32), ("st", C.c_int32),
                ("flops", C.c_double), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


class WgradNhwcDesc(C.Structure):
    _fields_ = [("ksize", C.c_int32), ("dy", Tensor), ("x", Tensor), ("M", C.c_int32), ("N", C.c_int32), ("out", C.c_void_p),
                ("sm", C.c_int32), ("sn", C.c_int32), ("st", C.c_int32), ("stride", C.c_int32), ("flops", C.c_double), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_size_t)]


class WgradStemDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("in_dtype", C.c_int32), ("B", C.c_int32), ("Cin", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("Cout", C.c_int32), ("dy3", Tensor), ("dy1", Tensor), ("out3", C.c_void_p), ("out1", C.c_void_p), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_size_t)]


class WgradFlatGeom(C.Structure):
    _fields_ = [("row_pitch", C.c_int32), ("plane", C.c_int32), ("flat_positions", C.c_int32), ("chunk", C.c_int32), ("chunks", C.c_int32),
                ("tile_m", C.c_int32), ("tile_n", C.c_int32), ("tiles", C.c_int32), ("x_positions", C.c_int32), ("stages", C.c_int32),
                ("slices", C.c_int32), ("chunks_per_slice", C.c_int32), ("lds_bytes", C.c_uint64), ("partial_bytes", C.c_uint64)]


class PackJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("kind", C.c_int32), ("Cout", C.c_int32), ("Cin", C.c_int32),
                ("K", C.c_int32), ("first", C.c_uint64)]


class PackBatchDesc(C.Structure):
    _fields_ = [("jobs", C.c_void_p), ("njobs", C.c_int32), ("total", C.c_uint64)]


class SppfBwdDesc(C.Structure):
    _fields_ = [("x", Tensor), ("y1", Tensor), ("y2", Tensor), ("dy1", Tensor), ("dy2", Tensor), ("dy3", Tensor),
                ("dx", Tensor), ("dx_acc", C.c_int32)]


class DgradS2Desc(C.Structure):
    _fields_ = [("dy3", Tensor), ("dy1", Tensor), ("dx", Tensor), ("w3_packed", C.c_void_p), ("w1_packed", C.c_void_p),
                ("accumulate", C.c_int32)]


class SppfQDesc(C.Structure):
    _fields_ = [("x", Tensor), ("y1", Tensor), ("y2", Tensor), ("y3", Tensor), ("q1", Tensor), ("q2", Tensor), ("q3", Tensor),
                ("q_amax", C.c_float), ("pad", C.c_int32 * 3)]


class HeadPackDesc(C.Structure):
    _fields_ = [("n_levels", C.c_int32), ("cls", Tensor * 4), ("reg", Tensor * 4), ("scores", C.c_void_p),
                ("distri", C.c_void_p), ("dscores", C.c_void_p), ("ddistri", C.c_void_p), ("nc", C.c_int32), ("nreg", C.c_int32)]


class HeadAbDesc(C.Structure):
    _fields_ = [("n_levels", C.c_int32), ("nc", C.c_int32), ("na", C.c_int32), ("cls", Tensor * 4), ("reg", Tensor * 4),
                ("reg_fwd", Tensor * 4), ("anchors", C.c_float * 24), ("scores", C.c_void_p), ("distri", C.c_void_p),
                ("dscores", C.c_void_p), ("ddistri", C.c_void_p)]


class LossGradDesc(C.Structure):
    _fields_ = [("fwd", LossDesc), ("grad_scale", C.c_void_p), ("dpred_scores", C.c_void_p), ("dpred_distri", C.c_void_p)]


WG_3X3S1, WG_1X1, WG_3X3S2, WG_CONVT = 0, 1, 2, 3
TOP_NAMES = {1: "bn_stats", 2: "bnact_fwd", 3: "bnact_bwd", 4: "wgrad_transpose", 5: "wgrad", 6: "pack", 7: "pool_bwd",
             8: "head_pack", 9: "head_unpack", 10: "s2d", 11: "bias_grad", 12: "fill", 13: "add", 14: "conv_i8", 15: "absmax", 16: "quantize",
             17: "pred_decode", 18: "pw_s2", 19: "stem_s2", 20: "avgpool3", 21: "sppf", 22: "conv_dgrad_s2"}

IOU_TYPES = {"giou": 0, "diou": 1, "ciou": 2, "siou": 3}

# public struct of include/yolov6_hip.h -> its ctypes mirror above (load() checks every size against y6_abi_sizeof)
STRUCTS = {
    "y6_tensor": Tensor, "y6_conv_desc": ConvDesc, "y6_conv_geometry": ConvGeometry, "y6_conv_i8_desc": ConvI8Desc, "y6_convt_desc": ConvTDesc, "y6_stem_desc": StemDesc,
    "y6_pw_s2_desc": PwS2Desc, "y6_stem_s2_desc": StemS2Desc, "y6_letterbox_desc": LetterboxDesc, "y6_decode_desc": DecodeDesc,
    "y6_pred_decode_desc": PredDecodeDesc, "y6_nms_sink": NmsSink, "y6_nms_desc": NmsDesc, "y6_tal_desc": TalDesc, "y6_atss_desc": AtssDesc,
    "y6_loss_desc": LossDesc, "y6_distill_desc": DistillDesc, "y6_bn_train_desc": BnTrainDesc, "y6_bn_train_multi_desc": BnTrainMultiDesc, "y6_bnact_desc": BnActDesc,
    "y6_bnact_bwd_desc": BnActBwdDesc, "y6_wgrad_t_desc": WgradTDesc, "y6_wgrad_desc": WgradDesc, "y6_wgrad_nhwc_desc": WgradNhwcDesc, "y6_wgrad_stem_desc": WgradStemDesc, "y6_wgrad_flat_geom": WgradFlatGeom,
    "y6_pack_job": PackJob, "y6_pack_batch_desc": PackBatchDesc, "y6_sppf_bwd_desc": SppfBwdDesc, "y6_dgrad_s2_desc": DgradS2Desc, "y6_sppf_q_desc": SppfQDesc, "y6_head_pack_desc": HeadPackDesc,
    "y6_head_ab_desc": HeadAbDesc, "y6_loss_grad_desc": LossGradDesc,
}

# symbol -> (restype, argtypes); also the list the CPU test checks the .so exports against
SIGNATURES = {
    "y6_abi_version": (C.c_int, []),
    "y6_abi_sizeof": (C.c_size_t, [C.c_char_p]),
    "y6_last_error": (C.c_char_p, []),
    "y6_device_info": (C.c_int, [C.POINTER(C.c_int), C.c_char_p, C.c_size_t]),
    "y6_packed_weight_elems": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "y6_pack_conv_weight": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "y6_pack_convt2x2_weight": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "y6_conv2d": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "y6_conv2d_i8": (C.c_int, [C.POINTER(ConvI8Desc), C.c_void_p]),
    "y6_conv2d_i8_variant": (C.c_int, [C.POINTER(ConvI8Desc)]),
    "y6_packed_weight_i8_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "y6_pack_conv_weight_i8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "y6_absmax": (C.c_int, [C.POINTER(Tensor), C.c_void_p, C.c_void_p]),
    "y6_quantize_i8": (C.c_int, [C.POINTER(Tensor), C.c_float, C.POINTER(Tensor), C.c_void_p]),
    "y6_conv_variants": (C.c_int, []),
    "y6_conv_variant_name": (C.c_char_p, [C.c_int]),
    "y6_conv_variant_supports": (C.c_int, [C.POINTER(ConvDesc), C.c_int]),
    "y6_conv_launch_geometry": (C.c_int, [C.POINTER(ConvDesc), C.c_int, C.POINTER(ConvGeometry)]),
    "y6_dma_probe": (C.c_int, [C.c_void_p, C.c_uint, C.c_uint, C.c_ulonglong, C.c_void_p, C.c_void_p]),
    "y6_convt2x2": (C.c_int, [C.POINTER(ConvTDesc), C.c_void_p]),
    "y6_stem_conv": (C.c_int, [C.POINTER(StemDesc), C.c_void_p]),
    "y6_stem_twin_supported": (C.c_int, [C.POINTER(StemDesc)]),
    "y6_sppf_pool": (C.c_int, [C.POINTER(Tensor)] * 4 + [C.c_void_p]),
    "y6_sppf_pool_q": (C.c_int, [C.POINTER(SppfQDesc), C.c_void_p]),
    "y6_nchw_to_nhwc": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Tensor), C.c_void_p]),
    "y6_nhwc_to_nchw": (C.c_int, [C.POINTER(Tensor), C.c_void_p, C.c_int, C.c_void_p]),
    "y6_head_decode": (C.c_int, [C.POINTER(DecodeDesc), C.c_void_p]),
    "y6_letterbox": (C.c_int, [C.POINTER(LetterboxDesc), C.c_void_p]),
    "y6_fused_pw_s2_supported": (C.c_int, [C.POINTER(PwS2Desc)]),
    "y6_fused_pw_s2": (C.c_int, [C.POINTER(PwS2Desc), C.c_void_p]),
    "y6_fused_stem_s2_supported": (C.c_int, [C.POINTER(StemS2Desc)]),
    "y6_fused_stem_s2": (C.c_int, [C.POINTER(StemS2Desc), C.c_void_p]),
    "y6_head_pred_decode_supported": (C.c_int, [C.POINTER(PredDecodeDesc)]),
    "y6_head_pred_decode": (C.c_int, [C.POINTER(PredDecodeDesc), C.c_void_p]),
    "y6_nms_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "y6_nms": (C.c_int, [C.POINTER(NmsDesc), C.c_void_p]),
    "y6_tal_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "y6_tal_assign": (C.c_int, [C.POINTER(TalDesc), C.c_void_p]),
    "y6_atss_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "y6_atss_assign": (C.c_int, [C.POINTER(AtssDesc), C.c_void_p]),
    "y6_bbox_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "y6_loss_workspace_bytes": (C.c_size_t, []),
    "y6_loss_forward": (C.c_int, [C.POINTER(LossDesc), C.c_void_p]),
    "y6_distill_forward": (C.c_int, [C.POINTER(DistillDesc), C.c_void_p]),
    "y6_distill_backward": (C.c_int, [C.POINTER(DistillDesc), C.c_void_p]),
    "y6_distill_cw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "y6_bn_stats_workspace_bytes": (C.c_size_t, [C.c_int]),
    "y6_bn_stats_workspace_bytes_for": (C.c_size_t, [C.c_int, C.c_long]),
    "y6_bnact_bwd_workspace_bytes_for": (C.c_size_t, [C.c_int, C.c_long]),
    "y6_bn_train_stats": (C.c_int, [C.POINTER(BnTrainDesc), C.c_void_p]),
    "y6_bn_train_stats_multi": (C.c_int, [C.POINTER(BnTrainMultiDesc), C.c_void_p]),
    "y6_bnact_forward": (C.c_int, [C.POINTER(BnActDesc), C.c_void_p]),
    "y6_bnact_bwd_workspace_bytes": (C.c_size_t, [C.c_int]),
    "y6_bnact_backward": (C.c_int, [C.POINTER(BnActBwdDesc), C.c_void_p]),
    "y6_wgrad_transpose": (C.c_int, [C.POINTER(WgradTDesc), C.c_void_p]),
    "y6_wgrad": (C.c_int, [C.POINTER(WgradDesc), C.c_void_p]),
    "y6_wgrad_nhwc": (C.c_int, [C.POINTER(WgradNhwcDesc), C.c_void_p]),
    "y6_wgrad_nhwc_supported": (C.c_int, [C.POINTER(WgradNhwcDesc)]),
    "y6_wgrad_nhwc_route": (C.c_int, [C.POINTER(WgradNhwcDesc)]),
    "y6_wgrad_stem": (C.c_int, [C.POINTER(WgradStemDesc), C.c_void_p]),
    "y6_wgrad_stem_supported": (C.c_int, [C.POINTER(WgradStemDesc)]),
    "y6_wgrad_stem_workspace_bytes": (C.c_size_t, [C.c_int]),
    "y6_wgrad_flat_geometry": (C.c_int, [C.POINTER(WgradNhwcDesc), C.POINTER(WgradFlatGeom)]),
    "y6_pack_job_elems": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "y6_pack_weights_batched": (C.c_int, [C.POINTER(PackBatchDesc), C.c_void_p]),
    "y6_sppf_pool_backward": (C.c_int, [C.POINTER(SppfBwdDesc), C.c_void_p]),
    "y6_head_pack": (C.c_int, [C.POINTER(HeadPackDesc), C.c_void_p]),
    "y6_head_unpack_backward": (C.c_int, [C.POINTER(HeadPackDesc), C.c_void_p]),
    "y6_head_ab_pack": (C.c_int, [C.POINTER(HeadAbDesc), C.c_void_p]),
    "y6_head_ab_unpack_backward": (C.c_int, [C.POINTER(HeadAbDesc), C.c_void_p]),
    "y6_plan_add_head_ab_pack": (C.c_int, [C.c_void_p, C.POINTER(HeadAbDesc)]),
    "y6_plan_add_head_ab_unpack_backward": (C.c_int, [C.c_void_p, C.POINTER(HeadAbDesc)]),
    "y6_space_to_depth2": (C.c_int, [C.POINTER(Tensor), C.POINTER(Tensor), C.c_void_p]),
    "y6_subsample2": (C.c_int, [C.POINTER(Tensor), C.POINTER(Tensor), C.c_void_p]),
    "y6_dgrad_s2_supported": (C.c_int, [C.POINTER(DgradS2Desc)]),
    "y6_dgrad_s2": (C.c_int, [C.POINTER(DgradS2Desc), C.c_void_p]),
    "y6_plan_add_dgrad_s2": (C.c_int, [C.c_void_p, C.POINTER(DgradS2Desc)]),
    "y6_plan_add_subsample2": (C.c_int, [C.c_void_p, C.POINTER(Tensor), C.POINTER(Tensor)]),
    "y6_avgpool3": (C.c_int, [C.POINTER(Tensor), C.POINTER(Tensor), C.c_int, C.c_int, C.c_void_p]),
    "y6_plan_add_avgpool3": (C.c_int, [C.c_void_p, C.POINTER(Tensor), C.POINTER(Tensor), C.c_int, C.c_int]),
    "y6_channel_sum": (C.c_int, [C.POINTER(Tensor), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "y6_tensor_add": (C.c_int, [C.POINTER(Tensor), C.POINTER(Tensor), C.c_int, C.c_void_p]),
    "y6_loss_forward_backward": (C.c_int, [C.POINTER(LossGradDesc), C.c_void_p]),
    "y6_loss_backward": (C.c_int, [C.POINTER(LossGradDesc), C.c_void_p]),
    "y6_grad_finite_check": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "y6_sgd_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int,
                              C.c_void_p, C.c_void_p, C.c_void_p]),
    "y6_sgd_step_grouped": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                      C.c_float, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "y6_scaler_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_void_p]),
    "y6_plan_add_bn_train_stats": (C.c_int, [C.c_void_p, C.POINTER(BnTrainDesc)]),
    "y6_plan_add_bn_train_stats_multi": (C.c_int, [C.c_void_p, C.POINTER(BnTrainMultiDesc)]),
    "y6_plan_add_bnact_forward": (C.c_int, [C.c_void_p, C.POINTER(BnActDesc)]),
    "y6_plan_add_bnact_backward": (C.c_int, [C.c_void_p, C.POINTER(BnActBwdDesc)]),
    "y6_plan_add_wgrad_transpose": (C.c_int, [C.c_void_p, C.POINTER(WgradTDesc)]),
    "y6_plan_add_wgrad": (C.c_int, [C.c_void_p, C.POINTER(WgradDesc)]),
    "y6_plan_add_wgrad_nhwc": (C.c_int, [C.c_void_p, C.POINTER(WgradNhwcDesc)]),
    "y6_plan_add_wgrad_stem": (C.c_int, [C.c_void_p, C.POINTER(WgradStemDesc)]),
    "y6_plan_add_pack_batch": (C.c_int, [C.c_void_p, C.POINTER(PackBatchDesc)]),
    "y6_plan_add_sppf_backward": (C.c_int, [C.c_void_p, C.POINTER(SppfBwdDesc)]),
    "y6_plan_add_head_pack": (C.c_int, [C.c_void_p, C.POINTER(HeadPackDesc)]),
    "y6_plan_add_head_unpack_backward": (C.c_int, [C.c_void_p, C.POINTER(HeadPackDesc)]),
    "y6_plan_add_space_to_depth2": (C.c_int, [C.c_void_p, C.POINTER(Tensor), C.POINTER(Tensor)]),
    "y6_plan_add_channel_sum": (C.c_int, [C.c_void_p, C.POINTER(Tensor), C.c_void_p, C.c_void_p, C.c_size_t]),
    "y6_plan_add_tensor_add": (C.c_int, [C.c_void_p, C.POINTER(Tensor), C.POINTER(Tensor), C.c_int]),
    "y6_plan_add_fill_zero": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "y6_plan_create": (C.c_void_p, []),
    "y6_plan_destroy": (None, [C.c_void_p]),
    "y6_plan_add_conv": (C.c_int, [C.c_void_p, C.POINTER(ConvDesc)]),
    "y6_plan_add_conv_i8": (C.c_int, [C.c_void_p, C.POINTER(ConvI8Desc)]),
    "y6_plan_add_absmax": (C.c_int, [C.c_void_p, C.POINTER(Tensor), C.c_void_p]),
    "y6_plan_add_quantize_i8": (C.c_int, [C.c_void_p, C.POINTER(Tensor), C.c_float, C.POINTER(Tensor)]),
    "y6_plan_add_convt": (C.c_int, [C.c_void_p, C.POINTER(ConvTDesc)]),
    "y6_plan_add_stem": (C.c_int, [C.c_void_p, C.POINTER(StemDesc)]),
    "y6_plan_add_sppf": (C.c_int, [C.c_void_p] + [C.POINTER(Tensor)] * 4),
    "y6_plan_add_sppf_q": (C.c_int, [C.c_void_p, C.POINTER(SppfQDesc)]),
    "y6_plan_add_decode": (C.c_int, [C.c_void_p, C.POINTER(DecodeDesc)]),
    "y6_plan_add_pred_decode": (C.c_int, [C.c_void_p, C.POINTER(PredDecodeDesc)]),
    "y6_plan_set_nms_sink": (C.c_int, [C.c_void_p, C.POINTER(NmsSink)]),
    "y6_plan_side_pending": (C.c_int, [C.c_void_p]),
    "y6_plan_add_pw_s2": (C.c_int, [C.c_void_p, C.POINTER(PwS2Desc)]),
    "y6_plan_add_stem_s2": (C.c_int, [C.c_void_p, C.POINTER(StemS2Desc)]),
    "y6_plan_add_nchw2nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Tensor)]),
    "y6_plan_add_nhwc2nchw": (C.c_int, [C.c_void_p, C.POINTER(Tensor), C.c_void_p, C.c_int]),
    "y6_plan_num_ops": (C.c_int, [C.c_void_p]),
    "y6_plan_autotune": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "y6_plan_copy_variants": (C.c_int, [C.c_void_p, C.c_void_p]),
    "y6_plan_rebind": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "y6_plan_rebind_input": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "y6_plan_rebind_output": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "y6_plan_run": (C.c_int, [C.c_void_p, C.c_void_p]),
    "y6_plan_run_range": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "y6_plan_mark_side": (C.c_int, [C.c_void_p]),
    "y6_plan_set_schedule": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int32), C.c_int]),
    "y6_plan_capture": (C.c_int, [C.c_void_p, C.c_void_p]),
    "y6_plan_timing_begin": (C.c_int, [C.c_void_p, C.c_int]),
    "y6_plan_run_timed": (C.c_int, [C.c_void_p, C.c_void_p]),
    "y6_plan_timing_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int]),
    "y6_plan_op_info": (C.c_int, [C.c_void_p, C.c_int] + [C.POINTER(C.c_int32)] * 4 + [C.POINTER(C.c_double)] * 2),
    "y6_plan_profile": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int32),
                                  C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]),
}

_lib = None


def load():
    """Load (once) and return the shared library; RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own copy of the HIP runtime; it must be the first one in the process, so that
    # libyolov6_hip.so's libamdhip64 dependency resolves to the SAME runtime (two runtimes in one
    # process cannot both own the device: launches fail with "no ROCm-capable device is detected")
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"yolov6_amd: {LIB_PATH} is missing - build it with `python yolov6_amd/csrc/build.py` "
            "(there is no CPU or PyTorch fallback for the hot path)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.y6_abi_version() != 1:
        raise RuntimeError("yolov6_amd: libyolov6_hip.so ABI version mismatch - rebuild")
    bad = [(n, C.sizeof(t), int(lib.y6_abi_sizeof(n.encode()))) for n, t in STRUCTS.items()
           if C.sizeof(t) != int(lib.y6_abi_sizeof(n.encode()))]
    if bad:      # a descriptor field added on one side only: kernels would read past (or short of) what Python filled in
        raise RuntimeError("yolov6_amd: struct layout differs between yolov6_amd/_lib.py and libyolov6_hip.so (name, ctypes, C): "
                           f"{bad} - rebuild the library / update the mirror")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().y6_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"yolov6_hip {what} failed (code {rc}): {msg}")


def current_stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu_tensor(t, name="tensor"):
    if not t.is_cuda:
        raise RuntimeError(f"yolov6_amd: {name} must live on a ROCm device (got {t.device}); the hot path has no CPU fallback")
