"""ctypes binding of libfdgan_hip.so (include/fdgan_hip.h).

This is the ONLY way the Python host code reaches the GPU kernels: plain pointers,
sizes and a hipStream_t cross the boundary; torch is used for device memory and
streams, never for the math.  There is no CPU fallback: if the shared library is
missing the import of any product module fails loudly (FdganLibraryError).
"""
import ctypes as C
import os

from . import buildid

_HERE = os.path.dirname(os.path.abspath(__file__))
# FDGAN_LIB: tuning aid -- an experiment build of the same ABI (`FDGAN_BUILD_TAG=x python __graft_entry__.py` writes
# variants/libfdgan_hip_x.so); unset everywhere outside tools/
LIB_PATH = os.environ.get("FDGAN_LIB") or os.path.join(_HERE, "libfdgan_hip.so")

FD_OK, FD_EINVAL, FD_EUNSUPPORTED, FD_ELAUNCH, FD_ESTATE = 0, -1, -2, -3, -4
FD_BF16, FD_F32, FD_F16 = 0, 1, 2   # fp16: forward activations / filter images; bf16: gradients (include/fdgan_hip.h)
ACT_NONE, ACT_RELU, ACT_LEAKY02, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4
WLAYOUT_CHUNK32, WLAYOUT_X64 = 0, 1
ABI_VERSION = 16


class FdganLibraryError(RuntimeError):
    pass


class FdTensor(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("n", C.c_int64), ("h", C.c_int64), ("w", C.c_int64), ("c", C.c_int64),
                ("stride", C.c_int64 * 4), ("dtype", C.c_int32), ("_pad", C.c_int32)]


class FdPrologue(C.Structure):
    _fields_ = [("mean", C.c_void_p), ("var", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("eps", C.c_float), ("act", C.c_int32), ("pool2", C.c_int32), ("momentum", C.c_float),
                ("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("num_batches_tracked", C.c_void_p),
                ("count", C.c_int64)]


class FdPackJob(C.Structure):
    _fields_ = [("w", C.c_void_p), ("packed", C.c_void_p), ("cout", C.c_int32), ("cin", C.c_int32), ("ksize", C.c_int32),
                ("transposed", C.c_int32), ("flip", C.c_int32), ("layout", C.c_int32), ("dtype", C.c_int32), ("_pad", C.c_int32),
                ("first_unit", C.c_int64)]


class FdZeroJob(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("bytes", C.c_int64), ("first_group", C.c_int64)]


class FdReduceJob(C.Structure):
    _fields_ = [("part", C.c_void_p), ("out", C.c_void_p), ("numel", C.c_int64), ("nsplit", C.c_int32), ("accumulate", C.c_int32),
                ("first_group", C.c_int64)]


class FdTrReduceJob(C.Structure):
    _fields_ = [("part", C.c_void_p), ("out", C.c_void_p), ("item_stride", C.c_int64), ("items", C.c_int32), ("zt", C.c_int32),
                ("kyg", C.c_int32), ("kyn", C.c_int32), ("ks", C.c_int32), ("nw", C.c_int32), ("taps", C.c_int32), ("cin", C.c_int32),
                ("cout", C.c_int32), ("accumulate", C.c_int32), ("first_group", C.c_int64), ("groups", C.c_int64)]


class FdConvDesc(C.Structure):
    _fields_ = [("ksize", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("epilogue_act", C.c_int32),
                ("upsample2", C.c_int32), ("cout", C.c_int32), ("w_layout", C.c_int32)]


class FdStats(C.Structure):
    _fields_ = [("partial", C.c_void_p), ("capacity_floats", C.c_int64), ("mean", C.c_void_p), ("var", C.c_void_p),
                ("counter", C.c_void_p), ("count", C.c_int64)]


class FdConvInfo(C.Structure):
    _fields_ = [("stats_rows", C.c_int64), ("stats_cpad", C.c_int64), ("grid_x", C.c_int64), ("grid_y", C.c_int64),
                ("lds_bytes", C.c_int64), ("fused_finalize", C.c_int64)]


# name -> (restype, argtypes); every symbol include/fdgan_hip.h declares.
SIGNATURES = {
    "fdgan_last_error": (C.c_char_p, []),
    "fdgan_version": (C.c_int, []),
    "fdgan_build_id": (C.c_char_p, []),
    "fdgan_set_cu_budget": (C.c_int, [C.c_int]),
    "fdgan_device_arch": (C.c_char_p, []),
    "fdgan_packed_weight_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "fdgan_conv_weight_layout": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "fdgan_pack_conv_weight": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p, C.c_size_t, C.c_void_p]),
    "fdgan_pack_units": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "fdgan_pack_conv_weights": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "fdgan_conv2d_fwd_info": (C.c_int, [C.POINTER(FdTensor), C.POINTER(FdTensor), C.c_int, C.POINTER(FdConvDesc),
                                        C.POINTER(FdPrologue), C.POINTER(FdConvInfo)]),
    "fdgan_conv2d_fwd": (C.c_int, [C.POINTER(FdTensor), C.c_void_p, C.c_void_p, C.POINTER(FdPrologue),
                                   C.POINTER(FdTensor), C.POINTER(FdStats), C.POINTER(FdConvDesc), C.c_void_p]),
    "fdgan_bn_finalize": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    "fdgan_nchw_f32_to_nhwc": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                         C.POINTER(FdTensor), C.c_void_p]),
    "fdgan_nhwc_to_nchw_f32": (C.c_int, [C.POINTER(FdTensor), C.c_void_p, C.c_void_p]),
    "fdgan_copy_nhwc": (C.c_int, [C.POINTER(FdTensor), C.POINTER(FdTensor), C.c_void_p]),
    "fdgan_scatter_dehaze": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_void_p,
                             C.c_void_p, C.c_void_p, C.POINTER(FdTensor), C.c_void_p]),
    "fdgan_maxpool3s2_nhwc": (C.c_int, [C.POINTER(FdTensor), C.POINTER(FdPrologue), C.POINTER(FdTensor), C.c_void_p, C.c_int64,
                              C.POINTER(C.c_int64), C.c_void_p]),
    "fdgan_pyramid_pool4": (C.c_int, [C.POINTER(FdTensor), C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.POINTER(FdTensor), C.c_void_p]),
    "fdgan_bn_dropout_nhwc": (C.c_int, [C.POINTER(FdTensor), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                              C.POINTER(FdTensor), C.c_void_p]),
    "fdgan_maxpool3s2_bwd": (C.c_int, [C.POINTER(FdTensor), C.POINTER(FdPrologue), C.POINTER(FdTensor), C.POINTER(FdTensor), C.c_void_p]),
    "fdgan_pyramid_pool4_bwd": (C.c_int, [C.POINTER(FdTensor), C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.POINTER(FdTensor),
                                          C.POINTER(FdTensor), C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p]),
    "fdgan_bn_dropout_bwd": (C.c_int, [C.POINTER(FdTensor), C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.POINTER(FdTensor),
                                       C.POINTER(FdTensor), C.c_void_p, C.c_void_p, C.c_void_p]),
    "fdgan_scatter_dehaze_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_float, C.c_float,
                                           C.c_void_p, C.c_void_p, C.POINTER(FdTensor), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "fdgan_allreduce_unique_id": (C.c_int, [C.c_void_p]),
    "fdgan_allreduce_comm_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "fdgan_allreduce_sum_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "fdgan_allreduce_comm_destroy": (C.c_int, [C.c_void_p]),
    "fdgan_plan_create": (C.c_void_p, []),
    "fdgan_plan_destroy": (None, [C.c_void_p]),
    "fdgan_plan_begin": (C.c_int, [C.c_void_p]),
    "fdgan_plan_end": (C.c_int, [C.c_void_p]),
    "fdgan_plan_num_launches": (C.c_int64, [C.c_void_p]),
    "fdgan_plan_launch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "fdgan_ssim_fwd_w": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    "fdgan_ssim_bwd_w": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int,
                                   C.c_float, C.c_void_p, C.c_void_p]),
    "fdgan_blur_gauss_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "fdgan_blur_gauss_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_float, C.c_int,
                                       C.c_void_p]),
    "fdgan_mul_mask_nhwc": (C.c_int, [C.POINTER(FdTensor), C.POINTER(FdTensor), C.c_int, C.c_void_p]),
    "fdgan_plan_set_slot": (C.c_int, [C.c_void_p, C.c_int]),
    "fdgan_plan_record_wait": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "fdgan_plan_launch_multi": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int]),
    "fdgan_fill_zero": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    "fdgan_fill_zero_many": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "fdgan_add_transposed_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "fdgan_plan_instantiate_graph": (C.c_int, [C.c_void_p, C.c_void_p]),
    "fdgan_plan_kernel_name": (C.c_char_p, [C.c_void_p, C.c_int64]),
    "fdgan_plan_time_launches": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_int64]),
    "fdgan_plan_read_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "fdgan_plan_profile": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_int64]),
    "fdgan_debug_timing": (C.c_int, [C.c_void_p]),
    "fdgan_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float,
                                  C.c_float, C.c_int64, C.c_void_p]),
    "fdgan_ssim_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p]),
    "fdgan_ssim_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                 C.c_float, C.c_void_p, C.c_void_p]),
    "fdgan_loss_f32": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                 C.POINTER(C.c_int64), C.c_void_p]),
    "fdgan_sum_partials": (C.c_int, [C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_void_p]),
    "fdgan_mse_nhwc_fwd": (C.c_int, [C.POINTER(FdTensor), C.POINTER(FdTensor), C.c_float, C.c_void_p, C.c_int64,
                                     C.POINTER(C.c_int64), C.c_void_p]),
    "fdgan_mse_nhwc_bwd": (C.c_int, [C.POINTER(FdTensor), C.POINTER(FdTensor), C.c_void_p, C.c_float, C.c_int,
                                     C.POINTER(FdTensor), C.c_void_p]),
    "fdgan_cx_rows_fwd": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    "fdgan_cx_rows_bwd": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "fdgan_maxpool2_nhwc": (C.c_int, [C.POINTER(FdTensor), C.POINTER(FdTensor), C.c_void_p]),
    "fdgan_maxpool2_bwd_nhwc": (C.c_int, [C.POINTER(FdTensor), C.POINTER(FdTensor), C.POINTER(FdTensor), C.c_void_p]),
    "fdgan_blur15_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int,
                                   C.c_void_p]),
    "fdgan_blur15_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int,
                                   C.c_void_p]),
    "fdgan_laplacian3_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    "fdgan_laplacian3_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    "fdgan_laplacian_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p]),
    "fdgan_fusion_input_nchw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p]),
    "fdgan_fusion_input_nhwc": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.POINTER(FdTensor),
                                          C.c_int, C.c_void_p]),
    "fdgan_conv2d_bwd_weight": (C.c_int, [C.POINTER(FdTensor), C.POINTER(FdPrologue), C.POINTER(FdTensor),
                                          C.POINTER(FdConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                          C.c_void_p]),
    "fdgan_bn_act_bwd": (C.c_int, [C.POINTER(FdTensor), C.POINTER(FdTensor), C.POINTER(FdPrologue), C.c_void_p, C.c_int64,
                                   C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p]),
    "fdgan_bn_act_bwd_acc": (C.c_int, [C.POINTER(FdTensor), C.POINTER(FdTensor), C.POINTER(FdPrologue), C.POINTER(FdTensor), C.c_void_p,
                                       C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p]),
    "fdgan_bn_act_bwd_dx": (C.c_int, [C.POINTER(FdTensor), C.POINTER(FdTensor), C.POINTER(FdPrologue), C.POINTER(FdTensor), C.c_int, C.c_void_p,
                                      C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p]),
    "fdgan_bn_bwd_finalize": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_void_p]),
    "fdgan_bn_bwd_finalize_sink": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int,
                                             C.c_void_p, C.c_void_p, C.c_void_p]),
    "fdgan_bn_bwd_finalize_raw": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_float,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "fdgan_bn_bwd_finalize_coef": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.POINTER(FdPrologue), C.c_int64,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "fdgan_conv2d_bwd_data": (C.c_int, [C.POINTER(FdTensor), C.c_void_p, C.POINTER(FdTensor), C.POINTER(FdPrologue),
                                        C.POINTER(FdTensor), C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int64),
                                        C.POINTER(C.c_int64), C.POINTER(FdConvDesc), C.c_void_p]),
    "fdgan_conv1x1_bwd_data_weight": (C.c_int, [C.POINTER(FdTensor), C.c_void_p, C.POINTER(FdTensor), C.POINTER(FdPrologue),
                                      C.POINTER(FdTensor), C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                      C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.POINTER(FdTensor), C.c_void_p, C.c_void_p,
                                      C.POINTER(C.c_int64), C.c_void_p]),
    "fdgan_wgrad_reduce_batch": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "fdgan_conv2d_bwd_weight_job": (C.c_int, [C.POINTER(FdTensor), C.POINTER(FdPrologue), C.POINTER(FdTensor),
                                              C.POINTER(FdConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                              C.POINTER(FdTrReduceJob), C.c_int, C.c_void_p]),
    "fdgan_wgrad_tr_reduce_batch": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "fdgan_bn_bwd_coef": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(FdPrologue), C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    "fdgan_affine_accumulate": (C.c_int, [C.POINTER(FdTensor), C.c_void_p, C.c_void_p, C.POINTER(FdTensor), C.c_void_p]),
    "fdgan_affine_accumulate_out": (C.c_int, [C.POINTER(FdTensor), C.c_void_p, C.c_void_p, C.POINTER(FdTensor), C.POINTER(FdTensor), C.c_void_p]),
    "fdgan_kernel_timer_arm": (C.c_int, [C.c_char_p, C.c_int, C.c_int]),
    "fdgan_kernel_timer_read": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                          C.POINTER(C.c_float), C.c_char_p]),
    "fdgan_bn_bwd_apply": (C.c_int, [C.POINTER(FdTensor), C.POINTER(FdTensor), C.POINTER(FdPrologue), C.c_void_p,
                                     C.c_void_p, C.POINTER(FdTensor), C.c_int, C.c_void_p]),
    "fdgan_conv2d_bwd_data_direct": (C.c_int, [C.POINTER(FdTensor), C.c_void_p, C.c_int, C.c_int, C.POINTER(FdConvDesc),
                                               C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    "fdgan_conv2d_bwd_data_direct_nhwc": (C.c_int, [C.POINTER(FdTensor), C.c_void_p, C.c_int, C.c_int, C.POINTER(FdConvDesc),
                                                    C.POINTER(FdTensor), C.c_void_p]),
    "fdgan_out_act_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int,
                                    C.POINTER(FdTensor), C.c_void_p]),
    "fdgan_grad_ew": (C.c_int, [C.c_int, C.POINTER(FdTensor), C.POINTER(FdTensor), C.POINTER(FdTensor), C.c_void_p]),
}

_lib = None


def load():
    """Loads libfdgan_hip.so (built in-tree by __graft_entry__.build()).  No fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FdganLibraryError(
            "%s not found: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
            "The FD-GAN hot path has no CPU fallback." % LIB_PATH)
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise FdganLibraryError("cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise FdganLibraryError("%s does not export %s" % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    if lib.fdgan_version() != ABI_VERSION:
        raise FdganLibraryError("ABI version mismatch: library %d, binding %d" % (lib.fdgan_version(), ABI_VERSION))
    # the library must have been built from the sources lying next to it (fdgan_hip/buildid.py); FDGAN_ALLOW_STALE_LIB=1 is for
    # tools/ that A/B a variant built from another commit's sources
    built = (lib.fdgan_build_id() or b"").decode().split(":")[0]
    have = buildid.source_build_id()
    if have is not None and built != have and os.environ.get("FDGAN_ALLOW_STALE_LIB") != "1":
        raise FdganLibraryError(
            "%s is stale: built from sources %s, the tree next to it hashes to %s -- rebuild with `python __graft_entry__.py`"
            % (LIB_PATH, built or "<none>", have))
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != FD_OK:
        msg = load().fdgan_last_error()
        raise RuntimeError("libfdgan_hip %s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))
