"""ctypes binding of libpointseg_hip.so -- the only door between the Python host code and the HIP kernels.

The signatures below are include/pointseg.h verbatim.  The library is built in-tree by compile_op.sh
(csrc/Makefile); a missing library is a hard error: there is no CPU fallback anywhere in this package.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpointseg_hip.so")
PS_ABI_VERSION = 6  # include/pointseg.h

PS_MAX_LAYERS = 8
c_f32p = ctypes.POINTER(ctypes.c_float)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_vp = ctypes.c_void_p


class PsPyramid(ctypes.Structure):
    _fields_ = [
        ("num_layers", ctypes.c_int32),
        ("K", ctypes.c_int32),
        ("B", ctypes.c_int64),
        ("n", ctypes.c_int64 * (PS_MAX_LAYERS + 1)),
        ("xyz", c_vp * PS_MAX_LAYERS),
        ("neigh_idx", c_vp * PS_MAX_LAYERS),
        ("sub_idx", c_vp * PS_MAX_LAYERS),
        ("interp_idx", c_vp * PS_MAX_LAYERS),
        ("order", c_vp * PS_MAX_LAYERS),
        ("built", ctypes.c_uint64),  # ps_pyramid_build's stamp: sub_idx IS the prefix of neigh_idx (0 = caller-filled, compared every step)
    ]


class PsRandlaConfig(ctypes.Structure):
    _fields_ = [
        ("num_layers", ctypes.c_int32),
        ("k_n", ctypes.c_int32),
        ("num_classes", ctypes.c_int32),
        ("in_channels", ctypes.c_int32),
        ("d_out", ctypes.c_int32 * PS_MAX_LAYERS),
    ]


class PsTrainOptions(ctypes.Structure):
    _fields_ = [
        ("learning_rate", ctypes.c_float),
        ("keep_prob", ctypes.c_float),
        ("mlp_bf16", ctypes.c_int32),
        ("fused_att", ctypes.c_int32),
        ("fused_locse", ctypes.c_int32),
        ("num_ignored", ctypes.c_int32),
        ("ignored_label_inds", ctypes.c_int32 * 8),
        ("deterministic", ctypes.c_int32),
        ("fused_convbn", ctypes.c_int32),
        ("overlap_wgrad", ctypes.c_int32),
        ("act_bf16", ctypes.c_int32),
    ]


# int (*ps_allreduce_fn)(void* user, void* buf, int64_t count, int dtype, void* hip_stream)
PS_ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p)


class PsTimingRow(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 48), ("ms", ctypes.c_double), ("launches", ctypes.c_int64)]


# name -> (restype, argtypes); every symbol include/pointseg.h declares
PROTOTYPES = {
    "ps_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(c_vp)]),
    "ps_destroy": (ctypes.c_int, [c_vp]),
    "ps_set_stream": (ctypes.c_int, [c_vp, c_vp]),
    "ps_synchronize": (ctypes.c_int, [c_vp]),
    "ps_set_deferred_checks": (ctypes.c_int, [c_vp, ctypes.c_int]),
    "ps_last_error": (ctypes.c_char_p, []),
    "ps_version": (ctypes.c_char_p, []),
    "ps_abi_version": (ctypes.c_int, []),
    "ps_set_train_gemm_bf16": (ctypes.c_int, [c_vp, ctypes.c_int]),
    "ps_set_train_act_bf16": (ctypes.c_int, [c_vp, ctypes.c_int]),
    "ps_set_att_bf16x3": (ctypes.c_int, [c_vp, ctypes.c_int]),
    "ps_set_train_gemm_b3": (ctypes.c_int, [c_vp, ctypes.c_int]),
    "ps_timing_begin": (ctypes.c_int, [c_vp]),
    "ps_timing_select": (ctypes.c_int, [c_vp, ctypes.c_char_p]),
    "ps_timing_end": (ctypes.c_int, [c_vp, ctypes.POINTER(PsTimingRow), ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    "ps_knn_batch": (ctypes.c_int, [c_vp, c_vp, c_vp] + [ctypes.c_int64] * 5 + [c_vp, ctypes.c_int]),
    "ps_knn_batch_i64": (ctypes.c_int, [c_vp, c_vp, c_vp] + [ctypes.c_int64] * 5 + [c_vp, ctypes.c_int]),
    "ps_pyramid_build": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, c_i32p, ctypes.c_int32,
                                        ctypes.POINTER(PsPyramid)]),
    "ps_grid_subsample": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp, ctypes.c_int64, c_vp, ctypes.c_int64, ctypes.c_float,
                                         c_i64p, c_vp, c_vp, c_vp]),
    "ps_randla_create": (ctypes.c_int, [c_vp, ctypes.POINTER(PsRandlaConfig), ctypes.POINTER(c_vp)]),
    "ps_randla_destroy": (ctypes.c_int, [c_vp]),
    "ps_randla_weight_count": (ctypes.c_int64, [c_vp]),
    "ps_randla_set_weights": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64]),
    "ps_randla_forward": (ctypes.c_int, [c_vp, ctypes.POINTER(PsPyramid), c_vp, c_vp]),
    "ps_randla_tap": (ctypes.c_int, [c_vp, ctypes.c_int, c_vp, ctypes.c_int64]),
    "ps_randla_keep_taps": (ctypes.c_int, [c_vp, ctypes.c_int]),
    "ps_op_gather_neighbour": (ctypes.c_int, [c_vp, c_vp, c_vp] + [ctypes.c_int64] * 5 + [c_vp]),
    "ps_op_relative_pos_encoding": (ctypes.c_int, [c_vp, c_vp, c_vp] + [ctypes.c_int64] * 3 + [c_vp]),
    "ps_op_random_sample": (ctypes.c_int, [c_vp, c_vp, c_vp] + [ctypes.c_int64] * 5 + [c_vp]),
    "ps_op_nearest_interpolation": (ctypes.c_int, [c_vp, c_vp, c_vp] + [ctypes.c_int64] * 4 + [c_vp]),
    "ps_op_conv1x1": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp] + [ctypes.c_int64] * 3 + [ctypes.c_int, c_vp]),
    "ps_op_att_pool": (ctypes.c_int, [c_vp, c_vp, c_vp] + [ctypes.c_int64] * 3 + [c_vp]),
    "ps_op_probs_to_volume": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, ctypes.c_int64, c_vp, c_vp] + [ctypes.c_int64] * 4 + [c_vp, c_vp]),
    "ps_op_linear_wgrad": (ctypes.c_int, [c_vp, c_vp, c_vp] + [ctypes.c_int64] * 3 + [c_vp, c_vp]),
    "ps_op_bn_train_fwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_float, ctypes.c_int] + [c_vp] * 5),
    "ps_op_bn_train_bwd": (ctypes.c_int, [c_vp] * 7 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_int] + [c_vp] * 3),
    "ps_volume_to_cloud": (ctypes.c_int, [c_vp, c_vp, c_vp] + [ctypes.c_int64] * 3 + [ctypes.POINTER(ctypes.c_int64)] + [c_vp] * 4),
    "ps_volume_to_cloud_dev": (ctypes.c_int, [c_vp, c_vp, c_vp] + [ctypes.c_int64] * 3 + [ctypes.POINTER(ctypes.c_int64)] + [c_vp] * 4),
    "ps_grid_subsample_dev": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp, ctypes.c_int64, c_vp, ctypes.c_int64, ctypes.c_float, ctypes.c_int64,
                                             c_i64p, c_vp, c_vp, c_vp]),
    "ps_op_half_to_float": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp]),
    "ps_op_bn_train_sums": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, ctypes.c_int64, c_vp]),
    "ps_op_bn_train_apply": (ctypes.c_int, [c_vp] * 5 + [ctypes.c_int64] * 3 + [ctypes.c_float, ctypes.c_int] + [c_vp] * 4),
    "ps_op_bn_train_bwd_sums": (ctypes.c_int, [c_vp] * 7 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_int] + [c_vp] * 2),
    "ps_op_bn_train_bwd_apply": (ctypes.c_int, [c_vp] * 9 + [ctypes.c_int64] * 3 + [ctypes.c_int, c_vp]),
    "ps_op_scatter_add_rows": (ctypes.c_int, [c_vp, c_vp, c_vp] + [ctypes.c_int64] * 4 + [c_vp]),
    "ps_op_softmax_pool_fwd": (ctypes.c_int, [c_vp, c_vp, c_vp] + [ctypes.c_int64] * 3 + [c_vp, c_vp]),
    "ps_op_softmax_pool_bwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp] + [ctypes.c_int64] * 3 + [c_vp, c_vp]),
    "ps_op_softmax_pool_bwd_scores": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp] + [ctypes.c_int64] * 3 + [c_vp, c_vp]),
    "ps_op_att_pool_train_supported": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int64]),
    "ps_op_att_pool_train_supported_ex": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int64, ctypes.c_int]),
    "ps_op_att_pool_gemm_supported": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int64]),
    "ps_op_att_pool_gemm_fwd": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp] + [ctypes.c_int64] * 3 + [c_vp]),
    "ps_op_att_pool_gemm_bwd": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp, c_vp] + [ctypes.c_int64] * 3 + [c_vp, ctypes.c_int64, ctypes.c_int, c_vp, ctypes.c_int64]),
    "ps_op_att_pool_gemm_fwd_split": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c_vp, ctypes.c_int64, c_vp,
                                                     ctypes.c_int64, ctypes.c_int64, c_vp]),
    "ps_op_att_pool_gemm_bwd_split": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c_vp, ctypes.c_int64, c_vp, c_vp,
                                                     ctypes.c_int64, ctypes.c_int64, c_vp, ctypes.c_int64, c_vp, ctypes.c_int64, ctypes.c_int, c_vp, ctypes.c_int64]),
    "ps_op_linear_wgrad_split": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c_vp, ctypes.c_int64,
                                                c_vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c_vp]),
    "ps_op_att_pool_train_fwd": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp] + [ctypes.c_int64] * 3 + [c_vp]),
    "ps_op_att_pool_train_bwd": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp, c_vp] + [ctypes.c_int64] * 3 + [c_vp, ctypes.c_int64, c_vp]),
    "ps_op_conv_bn_train_supported": (ctypes.c_int, [ctypes.c_int64]),
    "ps_op_conv_bn_train_sums": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp, c_vp, ctypes.c_int64, ctypes.c_int64, c_vp]),
    "ps_op_conv_bn_train_apply": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp, c_vp, ctypes.c_int64, ctypes.c_int64, c_vp, c_vp, c_vp, c_vp,
                                                 ctypes.c_int64]),
    "ps_op_conv_bn_train_bwd_sums": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp, c_vp, ctypes.c_int64, ctypes.c_int64, c_vp, c_vp, c_vp, c_vp, c_vp,
                                                    ctypes.c_int64, c_vp]),
    "ps_op_conv_bn_train_bwd_apply": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp, c_vp, ctypes.c_int64, ctypes.c_int64, c_vp, c_vp, c_vp, c_vp, c_vp,
                                                     c_vp, c_vp, ctypes.c_int64, ctypes.c_int, c_vp, ctypes.c_int64]),
    "ps_op_conv_bn_train_bwd_sums2": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp, c_vp, ctypes.c_int64, ctypes.c_int64, c_vp, c_vp, c_vp, c_vp, c_vp,
                                                     ctypes.c_int64, c_vp]),
    "ps_op_conv_bn_train_bwd_apply_w": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp, c_vp, ctypes.c_int64, ctypes.c_int64, c_vp, c_vp, c_vp, c_vp, c_vp,
                                                       ctypes.c_float, c_vp, ctypes.c_int64, ctypes.c_int, c_vp, ctypes.c_int64, c_vp, c_vp]),
    "ps_op_convbn_train_supported": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int64]),
    "ps_op_convbn_train_sums": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp, c_vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c_vp]),
    "ps_op_convbn_train_apply": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp, c_vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c_vp, c_vp, c_vp,
                                                ctypes.c_int, c_vp, ctypes.c_int64]),
    "ps_op_convbn_train_apply_add": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp, c_vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c_vp, c_vp, c_vp,
                                                    c_vp, ctypes.c_int64, c_vp, ctypes.c_int64]),
    "ps_op_convbn_train_bwd_sums": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp, c_vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c_vp, c_vp, c_vp,
                                                   c_vp, ctypes.c_int, c_vp, ctypes.c_int64, c_vp]),
    "ps_op_convbn_train_bwd_apply": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp, c_vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c_vp, c_vp, c_vp,
                                                    c_vp, ctypes.c_int, c_vp, ctypes.c_float, c_vp, ctypes.c_int64, ctypes.c_int, c_vp, ctypes.c_int64, c_vp,
                                                    c_vp]),
    "ps_op_locse_train_supported": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int64]),
    "ps_op_locse_train_sums": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c_vp, c_vp, ctypes.c_int64, c_vp]),
    "ps_op_locse_train_apply": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c_vp, c_vp, ctypes.c_int64, c_vp, c_vp,
                                               c_vp, c_vp, ctypes.c_int64]),
    "ps_op_locse_train_bwd": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c_vp, c_vp, ctypes.c_int64, c_vp, c_vp,
                                             c_vp, c_vp, c_vp, ctypes.c_int64, c_vp]),
    "ps_op_att_pool_train_fwd_split": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c_vp, ctypes.c_int64, c_vp, ctypes.c_int64, ctypes.c_int64, c_vp]),
    "ps_op_att_pool_train_bwd_split": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c_vp, ctypes.c_int64, c_vp, c_vp, ctypes.c_int64, ctypes.c_int64,
                                                      c_vp, ctypes.c_int64, c_vp, ctypes.c_int64, c_vp]),
    "ps_op_inverse_index_workspace": (ctypes.c_int64, [ctypes.c_int64, ctypes.c_int64]),
    "ps_op_inverse_index": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c_vp, c_vp, c_vp]),
    "ps_op_gather_reduce_rows": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp, c_vp, ctypes.c_int64, ctypes.c_int64, c_vp, ctypes.c_int64, ctypes.c_int]),
    "ps_op_gather_reduce_rows_ordered": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp, c_vp, ctypes.c_int64, ctypes.c_int64, c_vp, ctypes.c_int64, ctypes.c_int,
                                                        c_vp, ctypes.c_int64]),
    "ps_op_random_sample_bwd_inv": (ctypes.c_int, [c_vp] * 7 + [ctypes.c_int64] * 5 + [c_vp, c_vp, c_vp]),
    "ps_op_random_sample_ties": (ctypes.c_int, [c_vp, c_vp, c_vp] + [ctypes.c_int64] * 5 + [c_vp, c_vp]),
    "ps_op_att_pool_train_bwd_split_rows": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c_vp, ctypes.c_int64, c_vp, c_vp,
                                                           ctypes.c_int64, ctypes.c_int64, c_vp, ctypes.c_int64, c_vp, ctypes.c_int64, c_vp]),
    "ps_op_random_sample_bwd": (ctypes.c_int, [c_vp] * 5 + [ctypes.c_int64] * 5 + [c_vp]),
    "ps_op_add_lrelu": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int64, c_vp]),
    "ps_op_add_lrelu_bwd": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int64, c_vp]),
    "ps_op_axpy": (ctypes.c_int, [c_vp, ctypes.c_float, c_vp, ctypes.c_int64, c_vp]),
    "ps_op_mul": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int64, c_vp]),
    "ps_op_weighted_ce": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, ctypes.c_int64, ctypes.c_int64, c_vp, c_vp]),
    "ps_op_adam": (ctypes.c_int, [c_vp] * 5 + [ctypes.c_int64] + [ctypes.c_float] * 4 + [ctypes.c_int64]),
    "ps_op_dropout": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, ctypes.c_uint32, ctypes.c_float, c_vp, c_vp]),
    # row-strided / accumulating variants (the training step's concat buffers, include/pointseg.h)
    "ps_op_gather_neighbour_ex": (ctypes.c_int, [c_vp, c_vp, c_vp] + [ctypes.c_int64] * 5 + [c_vp, ctypes.c_int64]),
    "ps_op_conv1x1_ex": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp, c_vp] + [ctypes.c_int64] * 3 + [ctypes.c_int, ctypes.c_int, c_vp, ctypes.c_int64]),
    "ps_op_linear_wgrad_ex": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp, ctypes.c_int64] + [ctypes.c_int64] * 3 + [c_vp, c_vp]),
    "ps_op_bn_train_fwd_ex": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_float, ctypes.c_int, c_vp, ctypes.c_int64]
                              + [c_vp] * 4),
    "ps_op_bn_train_fwd_mov": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_float, ctypes.c_int, c_vp, ctypes.c_int64]
                               + [c_vp] * 6 + [ctypes.c_float]),
    "ps_op_bn_train_bwd_ex": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64] + [c_vp] * 5 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_int] + [c_vp] * 3),
    "ps_op_bn_train_apply_ex": (ctypes.c_int, [c_vp] * 5 + [ctypes.c_int64] * 3 + [ctypes.c_float, ctypes.c_int, c_vp, ctypes.c_int64] + [c_vp] * 3),
    "ps_op_bn_train_bwd_sums_ex": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64] + [c_vp] * 5 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_int] + [c_vp] * 2),
    "ps_op_bn_train_bwd_apply_ex": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64] + [c_vp] * 7 + [ctypes.c_int64] * 3 + [ctypes.c_int, c_vp]),
    "ps_op_scatter_add_rows_ex": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp] + [ctypes.c_int64] * 4 + [c_vp]),
    # the training step behind one call (csrc/trainer.hip)
    "ps_trainer_create": (ctypes.c_int, [c_vp, ctypes.POINTER(PsRandlaConfig), ctypes.POINTER(PsTrainOptions), ctypes.POINTER(c_vp)]),
    "ps_trainer_destroy": (ctypes.c_int, [c_vp]),
    "ps_trainer_param_count": (ctypes.c_int64, [c_vp]),
    "ps_trainer_buffer_count": (ctypes.c_int64, [c_vp]),
    "ps_trainer_layout_rows": (ctypes.c_int, [c_vp]),
    "ps_trainer_layout": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, c_i64p, c_i64p, c_i64p, ctypes.POINTER(ctypes.c_int)]),
    "ps_trainer_bind": (ctypes.c_int, [c_vp] * 6),
    "ps_trainer_set_collective": (ctypes.c_int, [c_vp, PS_ALLREDUCE_FN, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "ps_trainer_set_options": (ctypes.c_int, [c_vp, ctypes.POINTER(PsTrainOptions)]),
    "ps_trainer_set_step": (ctypes.c_int, [c_vp, ctypes.c_int64]),
    "ps_trainer_get_step": (ctypes.c_int64, [c_vp]),
    "ps_trainer_pool_peak_bytes": (ctypes.c_int64, [c_vp]),
    "ps_trainer_set_profile": (ctypes.c_int, [c_vp, ctypes.c_int]),
    "ps_trainer_profile": (ctypes.c_int, [c_vp, ctypes.POINTER(PsTimingRow), ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    "ps_trainer_collective_stats": (ctypes.c_int, [c_vp, c_i64p, c_i64p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "ps_randla_backward": (ctypes.c_int, [c_vp, ctypes.POINTER(PsPyramid), c_vp, c_vp, c_vp, c_vp, c_vp]),
    "ps_randla_train_step": (ctypes.c_int, [c_vp, ctypes.POINTER(PsPyramid), c_vp, c_vp, c_vp, c_vp, c_vp]),
}

_lib = None


class PointSegError(RuntimeError):
    pass


def lib():
    """The loaded library.  Raises if it has not been built -- never falls back to anything else."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PointSegError("libpointseg_hip.so is missing: build it with `sh point-unet_amd/compile_op.sh` "
                                "(or __graft_entry__.build()); there is no CPU fallback")
        try:  # share torch's HIP runtime (same SONAME libamdhip64.so.7) when torch is in the process
            import torch  # noqa: F401
        except Exception:  # pragma: no cover
            pass
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.ps_abi_version() != PS_ABI_VERSION:  # (the ctypes Structures below mirror include/pointseg.h at this revision)
            raise PointSegError("libpointseg_hip.so has struct layout revision %d, this package was written against %d: rebuild the library"
                                % (handle.ps_abi_version(), PS_ABI_VERSION))
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        raise PointSegError("libpointseg_hip: %s (code %d)" % (lib().ps_last_error().decode(), rc))
