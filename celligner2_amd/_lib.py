"""ctypes binding of libc2mi.so (the C-ABI declared in include/c2mi.h).

The library is the product: if it cannot be loaded the import fails loudly --
there is no CPU or eager-PyTorch fallback anywhere in this package.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("C2MI_LIB") or os.path.join(_HERE, "lib", "libc2mi.so")  # C2MI_LIB: ablation builds

F32, BF16, F32X3, BF16X3P, F16 = 0, 1, 2, 3, 4
MMD_MAX_PARTS = 64

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `make -C celligner2_amd/csrc` (or __graft_entry__.build()); "
        "celligner2_amd has no fallback path"
    )
lib = C.CDLL(LIB_PATH)

i64, i32, f32, f64, vp, sz = C.c_int64, C.c_int, C.c_float, C.c_double, C.c_void_p, C.c_size_t


class AdamState(C.Structure):
    _fields_ = [("step", i64), ("lr", f64), ("beta1", f64), ("beta2", f64), ("eps", f64), ("weight_decay", f64),
                ("clip_value", f64)]


class SlabSum(C.Structure):
    _fields_ = [("slabs", vp), ("out", vp), ("slab_stride", i64), ("n", i64), ("nslab", i32)]


class AdamTensor(C.Structure):
    _fields_ = [("p", vp), ("g", vp), ("m", vp), ("v", vp), ("shadow_bf16", vp), ("n", i64), ("shadow_lo", vp), ("shadow_f16", i32),
                ("shadow_cols", i32), ("shadow_ld", i64)]


_SIGS = {
    "c2_last_error": (C.c_char_p, []),
    "c2_version": (i32, []),
    "c2_prep_input": (i32, [vp, i64, i64, i64, vp, i32, i64, i32, vp, vp]),
    "c2_col_sum_ws": (sz, [i64, i64]),
    "c2_col_sum": (i32, [vp, i64, i64, i64, i32, vp, vp, vp]),
    "c2_col_mean_from_sums": (i32, [vp, i64, i64, vp, vp]),
    "c2_gemv_bias": (i32, [vp, i64, i64, i64, vp, vp, vp, vp]),
    "c2_rank1_add": (i32, [vp, i64, i64, i64, vp, vp, vp]),
    "c2_drop_pending_sums": (i32, []),
    "c2_linear_fwd": (i32, [vp, i64, vp, i64, i32, i64, i64, i64, vp, i64, i32, i64, vp, vp]),
    "c2_linear_dgrad": (i32, [vp, i64, vp, i64, i32, i64, i64, i64, vp, i64, i32, i64, vp, vp]),
    "c2_linear_wgrad": (i32, [vp, i64, vp, i64, i32, i64, i64, i64, vp, i64, i32, i64, vp, vp, vp, vp]),
    "c2_linear_fwd_rows": (i32, [vp, i64, i64, vp, vp, i64, i32, i64, i64, i64, vp, i64, i32, i64, vp, vp]),
    "c2_linear_wgrad_rows": (i32, [vp, i64, vp, i64, i64, vp, i64, i64, i64, vp, i64, i32, i64, vp, vp, vp, vp]),
    "c2_linear_wgrad_rows_at": (i32, [vp, i64, vp, i64, i64, vp, i64, i64, i64, vp, i64, i32, i64, vp, vp, vp, vp]),
    "c2_linear_wgrad_bt": (i32, [vp, i64, vp, i64, i64, i64, i64, vp, i64, i32, i64, vp, vp, vp, vp]),
    "c2_gemm_tile": (i32, [i32, i64, i64, i64]),
    "c2_gemm_is_tiny": (i32, [i64, i64, i64]),
    "c2_split_bf16": (i32, [vp, i64, i64, i64, vp, i64, vp]),
    "c2_sum_slabs_2d": (i32, [vp, i32, i64, i64, i64, i64, vp, i64, vp]),
    "c2_sum_slabs": (i32, [vp, i32, i64, vp, i64, vp]),
    "c2_sum_slabs_multi": (i32, [C.POINTER(SlabSum), i32, vp]),
    "c2_epilogue_fwd": (i32, [vp, i64, i32, i64, vp, vp, i32, vp, i64, vp, f32, i32, i64, i64, vp, i32, i64, vp, vp, vp, i64, vp]),
    "c2_epilogue_bwd": (i32, [vp, i64, i32, i64, vp, i64, vp, vp, f32, i32, vp, vp, i64, i64, vp, i32, i64, vp, i64, vp]),
    "c2_bn_ws": (sz, [i64, i64]),
    "c2_bn_stats": (i32, [vp, i64, i32, i64, vp, vp, i32, vp, i64, i64, i64, vp, vp, vp, vp]),
    "c2_bn_fwd": (i32, [vp, vp, i32, i64, i64, vp, vp, vp, vp, vp, f32, i32, f32, vp, f32, i64, i64, vp, i32, i64, vp,
                        vp, vp]),
    "c2_bn_bwd_reduce": (i32, [vp, i64, i32, i64, vp, i64, vp, vp, f32, vp, vp, vp, i64, i64, vp, vp, vp, vp, vp, vp]),
    "c2_bn_bwd": (i32, [vp, vp, vp, i64, i32, vp, vp, i64, i64, vp, i32, i64, vp, vp]),
    "c2_bias_cond_grad_ws": (sz, [i64, i64, i32]),
    "c2_bias_cond_grad": (i32, [vp, i32, i64, vp, i32, i64, i64, vp, vp, i64, vp, vp, vp]),
    "c2_reduce_ws": (sz, []),
    "c2_reparam_kl_fwd": (i32, [vp, vp, vp, i64, i64, i64, f32, i32, vp, vp, i32, i64, vp, vp, vp]),
    "c2_reparam_kl_bwd": (i32, [vp, i64, i32, i64, vp, i64, vp, vp, vp, vp, i64, i64, i64, f32, i32, vp, vp, vp, i32, i64, vp]),
    "c2_recon_nb_ws": (sz, [i64, i64, i32]),
    "c2_recon_nb_fwd_bwd": (i32, [vp, i64, vp, i64, vp, i64, vp, vp, vp, i32, i64, i64, f32, vp, vp, i32, i64, vp, vp, vp, vp, vp, vp]),
    "c2_recon_nb_logits_f16_ok": (i32, [i64, i64]),
    "c2_linear_fwd_f16out": (i32, [vp, i64, vp, i64, i32, i64, i64, i64, vp, i64, vp, vp]),
    "c2_recon_nb_fwd_bwd_dbias": (i32, [vp, i32, i64, vp, i64, vp, i64, vp, vp, vp, i32, i64, i64, f32, vp, vp, i32, i64, vp, vp, vp, vp, vp, vp, vp]),
    "c2_recon_zinb_fwd_bwd": (i32, [vp, i64, vp, i64, vp, i64, vp, i64, vp, vp, vp, i32, i64, i64, f32, vp, vp, vp, i32, i64,
                                    i64, vp, vp, vp, vp, vp]),
    "c2_recon_zinb_fwd_bwd_rc": (i32, [vp, i64, vp, i64, vp, i64, vp, i64, vp, vp, vp, i32, i64, i64, f32, vp, vp, vp, i32,
                                       i64, i64, vp, vp, vp, vp, vp, vp]),
    "c2_nb_row_const": (i32, [vp, i64, vp, i64, i64, i64, vp, vp]),
    "c2_recon_mse_fwd_bwd": (i32, [vp, i64, vp, i64, vp, i64, i64, i64, f32, vp, vp, i32, i64, vp, vp]),
    "c2_classifier_loss_fwd_bwd": (i32, [vp, i64, vp, i64, i32, f32, vp, vp, i32, i64, vp, vp]),
    "c2_partition_ws": (sz, [i64]),
    "c2_partition": (i32, [vp, i64, i32, i32, i32, i32, vp, vp, vp, vp, vp]),
    "c2_mmd_ws": (sz, [i64, i64, i64]),
    "c2_mmd_fwd_bwd": (i32, [vp, i32, i64, vp, vp, i64, i64, i64, i64, f32, vp, vp, i64, vp, vp]),
    "c2_mmd_static_floats": (sz, []),
    "c2_mmd_fwd_bwd_static": (i32, [vp, i32, i64, vp, vp, i64, i64, i64, i64, f32, vp, vp, i64, vp]),
    "c2_adam_step": (i32, [C.POINTER(AdamTensor), i32, f64, f64, f64, f64, f64, f64, i64, vp]),
    "c2_adam_hyper_bytes": (sz, []),
    "c2_adam_prepare": (i32, [vp, vp, vp]),
    "c2_adam_step_dev": (i32, [C.POINTER(AdamTensor), i32, vp, vp]),
    "c2_combine_losses_dev": (i32, [vp, vp, vp, vp, vp]),
    "c2_scale4_dev": (i32, [vp, vp, vp, vp]),
    "c2_dropout_masks_dev": (i32, [vp, i64, f32, C.c_uint64, vp, C.c_uint64, C.c_uint64, vp]),
    "c2_scale4_dev_scale2": (i32, [vp, vp, vp, vp, vp, i64, vp, vp, i64, vp]),
    "c2_split_bf16_planes": (i32, [vp, i64, i64, i64, vp, vp, i64, vp]),
    "c2_linear_fwd_planes": (i32, [vp, vp, i64, vp, vp, i64, i64, i64, i64, vp, i64, i32, i64, vp, vp]),
    "c2_linear_dgrad_planes": (i32, [vp, vp, i64, vp, vp, i64, i64, i64, i64, vp, i64, i32, i64, vp, vp]),
    "c2_linear_wgrad_planes": (i32, [vp, vp, i64, vp, vp, i64, i64, i64, i64, vp, i64, i32, i64, vp, vp, vp, vp]),
    "c2_linear_bwd_tiny": (i32, [vp, i64, vp, i64, vp, i64, i64, i64, i64, vp, i64, vp, i64, vp, vp]),
    "c2_noise_dev": (i32, [vp, i64, f32, vp, i64, C.c_uint64, C.c_uint64, vp, C.c_uint64, C.c_uint64, vp]),
    "c2_graph_launch": (i32, [vp, vp]),
    "c2_cast_bf16": (i32, [vp, vp, i64, vp]),
    "c2_cast_f16": (i32, [vp, vp, i64, vp]),
    "c2_transpose_bf16": (i32, [vp, i64, i64, i64, vp, i64, vp]),
    "c2_gather_rows": (i32, [vp, vp, vp, vp, vp, i64, i32, i32, vp, i64, vp, vp, vp, vp, vp, vp, i32, i64, i32, vp, vp,
                             vp, i64, vp, vp, i64, vp]),
    "c2_gather_rows_ld": (i32, [vp, i64, vp, i64, vp, vp, vp, i64, i32, i32, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, i32,
                                i64, i32, vp, vp, vp, i64, vp, vp, i64, vp]),
    "c2_combine_losses": (i32, [vp, f32, f32, f32, f32, vp, vp]),
    "c2_defer_loss_sums": (i32, [i32]),
    "c2_scale4": (i32, [vp, f32, f32, f32, f32, vp, vp]),
    "c2_decoder_output": (i32, [vp, i64, i64, i64, i32, vp, i64, vp]),
    "c2_scale": (i32, [vp, vp, vp, i64, vp]),
    "c2_axpy": (i32, [vp, vp, vp, i64, vp]),
    "c2_dropout_masks": (i32, [vp, i64, f32, C.c_uint64, C.c_uint64, vp]),
}
for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)  # AttributeError here = header and library out of sync
    _fn.restype = _res
    _fn.argtypes = _args

EXPORTED = sorted(_SIGS)


class C2Error(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise C2Error(lib.c2_last_error().decode())


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def stream():
    """Raw handle of torch's current stream on the current device (the C call: torch.cuda.current_stream() builds a
    Stream object through several Python layers, ~9 us each on a ~100-launch step)."""
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def dt_code(dtype):
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    if dtype == torch.float16:
        return F16
    raise TypeError(f"unsupported dtype {dtype}")


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise C2Error("celligner2_amd kernels need tensors on an MI355X device (no CPU fallback)")
