"""ctypes binding of libsdnet_hip.so (C ABI declared in include/sdnet_hip.h)."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import torch

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("SDNET_HIP_LIB", _HERE / "csrc" / "libsdnet_hip.so"))

_lib = None

c_f32p = C.c_void_p   # device pointers travel as integers
c_i64 = C.c_int64
c_int = C.c_int
c_float = C.c_float
c_size = C.c_size_t
c_vp = C.c_void_p


class SdError(RuntimeError):
    pass


class LossDesc(C.Structure):
    _fields_ = [
        ("anchor_hm", c_vp), ("a_sb", c_i64), ("a_sc", c_i64),
        ("part_hm", c_vp), ("p_sb", c_i64), ("p_sc", c_i64),
        ("offsets", c_vp), ("o_sb", c_i64), ("o_sc", c_i64),
        ("embeddings", c_vp), ("e_sb", c_i64), ("e_sc", c_i64),
        ("t_anchor_hm", c_vp), ("ta_sb", c_i64), ("ta_sc", c_i64),
        ("t_part_hm", c_vp), ("tp_sb", c_i64), ("tp_sc", c_i64),
        ("anchor_inds", c_vp), ("part_inds", c_vp),
        ("anchor_offsets", c_vp), ("part_offsets", c_vp), ("t_embeddings", c_vp),
        ("anchor_mask", c_vp), ("part_mask", c_vp),
        ("B", c_int), ("M", c_int), ("N", c_int), ("h", c_int), ("w", c_int), ("K", c_int), ("P", c_int),
        ("hm_loss_fn", c_int),
        ("hm_weight", c_float), ("offset_weight", c_float), ("embedding_weight", c_float),
    ]


class ConvDesc(C.Structure):
    _fields_ = [(n, c_int) for n in ("B", "Hi", "Wi", "Cin", "Ho", "Wo", "Cout", "R", "S", "stride", "pad")]


_MAP = [c_vp, c_i64, c_i64]          # pointer, batch stride, channel stride
_PEAK_OUT = [c_vp, c_vp, c_vp, c_vp, c_vp]

_SIGNATURES = {
    "sd_version": (c_int, []),
    "sd_last_error": (C.c_char_p, []),
    "sd_build_flags": (c_int, []),
    "sd_clamped_sigmoid": (c_int, [c_vp, c_vp, c_i64, c_vp]),
    "sd_nms5": (c_int, _MAP + [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "sd_topk_workspace_bytes": (c_size, [c_int] * 5),
    "sd_topk": (c_int, _MAP + [c_int] * 5 + _PEAK_OUT + [c_vp, c_size, c_vp]),
    "sd_transpose_and_gather": (c_int, _MAP + [c_int, c_int, c_i64, c_vp, c_int, c_vp, c_vp]),
    "sd_hypot": (c_int, [c_vp, c_vp, c_i64, c_vp]),
    "sd_decode_peaks_workspace_bytes": (c_size, [c_int] * 5),
    "sd_decode_peaks": (c_int, _MAP + [c_int] * 5 + _PEAK_OUT + [c_vp, c_size, c_vp]),
    "sd_decode_workspace_bytes": (c_size, [c_int] * 7),
    "sd_decode_packed_words": (c_size, [c_int] * 3),
    "sd_decode": (c_int, _MAP * 4 + [c_int] * 7 + [c_float, c_float, c_int, c_vp, c_vp, c_size, c_vp]),
    "sd_decode_state_bytes": (c_size, [c_int] * 5),
    "sd_decode_fused_supported": (c_int, [c_int] * 7),
    "sd_decode_fused_recommended": (c_int, [c_int] * 8),
    "sd_stream_synchronize": (c_int, [c_vp]),
    "sd_selfcheck_sigmoid": (c_int, [c_vp, c_vp]),
    "sd_decode_fused_workspace_bytes": (c_size, [c_int] * 7),
    "sd_decode_fused": (c_int, _MAP * 4 + [c_int] * 7 + [c_float, c_float, c_int, c_vp, c_vp, c_size, c_vp, c_size, c_vp]),
    "sd_decode_group": (c_int, [c_vp] * 6 + _MAP * 2 + [c_int] * 5 + [c_float, c_float, c_vp, c_vp]),
    "sd_preprocess_workspace_bytes": (c_size, [c_int] * 4),
    "sd_preprocess_images": (c_int, [c_vp] + [c_int] * 5 + [c_vp, c_vp, c_int, c_vp, c_vp, c_int, c_vp, C.POINTER(c_float), C.POINTER(c_float),
                                                c_vp, c_vp, c_size, c_vp]),
    "sd_preprocess_jitter_workspace_bytes": (c_size, [c_int] * 5),
    "sd_preprocess_images_jitter": (c_int, [c_vp] + [c_int] * 5 + [c_vp, c_vp, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, C.POINTER(c_float), C.POINTER(c_float),
                                            c_vp, c_vp, c_size, c_vp]),
    "sd_render_targets": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_float, c_vp, c_vp]),
    "sd_loss_workspace_bytes": (c_size, [c_int] * 5),
    "sd_loss_fwd": (c_int, [C.POINTER(LossDesc), c_vp, c_vp, c_size, c_vp]),
    "sd_loss_bwd": (c_int, [C.POINTER(LossDesc), c_vp, c_vp, c_vp, c_vp]),
    "sd_conv2d_fwd_workspace_bytes": (c_size, [C.POINTER(ConvDesc)]),
    "sd_conv2d_fwd": (c_int, [c_vp, c_vp, c_vp, C.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_size, c_vp]),
    "sd_conv2d_fwd_sb_supported": (c_int, [C.POINTER(ConvDesc), c_int]),
    "sd_conv2d_fwd_sb_workspace_bytes": (c_size, [C.POINTER(ConvDesc), c_int]),
    "sd_conv2d_fwd_sb_state_bytes": (c_size, [C.POINTER(ConvDesc), c_int]),
    "sd_conv2d_fwd_sb": (c_int, [c_vp, c_vp, c_vp, C.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_size, c_vp, c_size, c_vp]),
    "sd_conv2d_stem_fwd_workspace_bytes": (c_size, [C.POINTER(ConvDesc)]),
    "sd_conv2d_stem_fwd": (c_int, [c_vp, c_vp, c_vp, C.POINTER(ConvDesc), c_vp, c_vp, c_int, c_int, c_vp, c_size, c_vp]),
    "sd_cast_f32_to_bf16": (c_int, [c_vp, c_vp, c_i64, c_vp]),
    "sd_conv2d_fwd_bf16_workspace_bytes": (c_size, [C.POINTER(ConvDesc)]),
    "sd_conv2d_fwd_bf16": (c_int, [c_vp, c_vp, c_vp, C.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_size, c_vp]),
    "sd_conv2d_fwd_bf16_head_supported": (c_int, [C.POINTER(ConvDesc), c_int]),
    "sd_head_split_bf16_bytes": (c_size, []),
    "sd_head_split_bf16": (c_int, [c_vp, c_vp, c_int, c_vp, c_vp]),
    "sd_conv2d_fwd_bf16_head": (c_int, [c_vp, c_vp, C.POINTER(ConvDesc), c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_vp]),
    "sd_cast_bf16_to_f32": (c_int, [c_vp, c_vp, c_i64, c_vp]),
    "sd_conv2d_fwd_bf16_bn_stats_workspace_bytes": (c_size, [c_vp]),
    "sd_conv2d_fwd_bf16_bn_stats": (c_int, [c_vp, c_vp, c_vp, c_vp, c_float, c_float, c_vp, c_vp, c_vp, c_vp, c_vp, c_size, c_vp]),
    "sd_conv2d_wgrad_bf16_workspace_bytes": (c_size, [c_vp]),
    "sd_conv2d_wgrad_bf16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_size, c_vp]),
    "sd_conv2d_transpose_weights_bf16": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "sd_conv2d_dgrad_bf16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp]),
    "sd_bn_apply_bf16": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp]),
    "sd_bn_bwd_bf16": (c_int, [c_vp, c_vp, c_vp, c_int, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_size, c_vp]),
    "sd_bn_train_stats_bf16": (c_int, [c_vp, c_i64, c_int, c_float, c_float, c_vp, c_vp, c_vp, c_vp, c_vp, c_size, c_vp]),
    "sd_col_sum_bf16": (c_int, [c_vp, c_i64, c_int, c_vp, c_int, c_vp, c_size, c_vp]),
    "sd_upsample2x_bwd_bf16": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "sd_maxpool3x3s2_fwd_bf16": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "sd_stem_bn_relu_maxpool_fwd_bf16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "sd_head_fwd_bf16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "sd_conv2d_dgrad": (c_int, [c_vp, c_vp, c_vp, C.POINTER(ConvDesc), c_vp, c_vp]),
    "sd_conv2d_transpose_weights": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "sd_conv2d_transpose_weights_batched": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "sd_conv2d_wgrad_workspace_bytes": (c_size, [C.POINTER(ConvDesc)]),
    "sd_conv2d_wgrad": (c_int, [c_vp, c_vp, c_vp, C.POINTER(ConvDesc), c_int, c_vp, c_size, c_vp]),
    "sd_conv2d_stem_wgrad_workspace_bytes": (c_size, [C.POINTER(ConvDesc)]),
    "sd_conv2d_stem_wgrad": (c_int, [c_vp, c_vp, c_vp, C.POINTER(ConvDesc), c_int, c_vp, c_size, c_vp]),
    "sd_conv2d_stem_wgrad_bf16mm": (c_int, [c_vp, c_vp, c_vp, C.POINTER(ConvDesc), c_int, c_vp, c_size, c_vp]),
    "sd_conv2d_stem_wgrad_bf16": (c_int, [c_vp, c_vp, c_vp, C.POINTER(ConvDesc), c_int, c_vp, c_size, c_vp]),
    "sd_col_reduce_workspace_bytes": (c_size, [c_i64, c_int]),
    "sd_bn_train_stats": (c_int, [c_vp, c_i64, c_int, c_float, c_float, c_vp, c_vp, c_vp, c_vp, c_vp, c_size, c_vp]),
    "sd_bn_apply": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp]),
    "sd_bn_fold": (c_int, [c_vp, c_vp, c_vp, c_vp, c_float, c_int, c_vp, c_vp, c_vp]),
    "sd_bn_bwd": (c_int, [c_vp, c_vp, c_vp, c_int, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_size, c_vp]),
    "sd_col_sum": (c_int, [c_vp, c_i64, c_int, c_vp, c_int, c_vp, c_size, c_vp]),
    "sd_maxpool3x3s2_fwd": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "sd_maxpool3x3s2_bwd": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "sd_upsample2x_bwd": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "sd_head_fwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "sd_head_bwd_workspace_bytes": (c_size, [c_int] * 4),
    "sd_head_bwd": (c_int, [c_vp] * 6 + [c_int] * 5 + [c_vp, c_size, c_vp]),
    "sd_head_bwd_bf16": (c_int, [c_vp] * 6 + [c_int] * 5 + [c_vp, c_size, c_vp]),
    "sd_adam_step": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_float, c_float, c_float, c_float, c_float, c_vp]),
    "sd_conv2d_kernel_name": (C.c_char_p, [c_vp, c_int]),
    "sd_set_option": (c_int, [C.c_char_p, c_int]),
    "sd_decode_set_option": (c_int, [C.c_char_p, c_int]),
    "sd_conv2d_stem_fwd_bn_stats_workspace_bytes": (c_size, [c_vp]),
    "sd_conv2d_stem_fwd_bn_stats": (c_int, [c_vp, c_vp, c_vp, c_vp, c_float, c_float, c_vp, c_vp, c_vp, c_vp, c_vp, c_size, c_vp]),
    "sd_conv2d_stem_fwd_bn_stats_bf16mm": (c_int, [c_vp, c_vp, c_vp, c_vp, c_float, c_float, c_vp, c_vp, c_vp, c_vp, c_vp, c_size, c_vp]),
    "sd_conv2d_stem_fwd_bn_stats_bf16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_float, c_float, c_vp, c_vp, c_vp, c_vp, c_vp, c_size, c_vp]),
    "sd_conv2d_dgrad_half_res": (c_int, [c_vp] * 6),
    "sd_bn_relu_maxpool_fwd": (c_int, [c_vp, c_int, c_int, c_int, c_int] + [c_vp] * 7),
    "sd_bn_relu_maxpool_fwd_bf16": (c_int, [c_vp, c_int, c_int, c_int, c_int] + [c_vp] * 7),
    "sd_maxpool_bn_relu_bwd": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int] + [c_vp] * 7 + [c_int, c_vp, c_size, c_vp]),
    "sd_maxpool_bn_relu_bwd_bf16": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int] + [c_vp] * 7 + [c_int, c_vp, c_size, c_vp]),
    "sd_maxpool_bn_relu_bwd_bf16_dx16": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int] + [c_vp] * 7 + [c_int, c_vp, c_size, c_vp]),
    "sd_conv2d_fwd_bn_stats_workspace_bytes": (c_size, [c_vp]),
    "sd_conv2d_fwd_bn_stats": (c_int, [c_vp, c_vp, c_vp, c_vp, c_float, c_float, c_vp, c_vp, c_vp, c_vp, c_vp, c_size, c_vp]),
    "sd_conv2d_dgrad_bn_reduce_workspace_bytes": (c_size, [c_vp]),
    "sd_conv2d_dgrad_bn_reduce": (c_int, [c_vp] * 5 + [c_vp, c_vp, c_int] + [c_vp] * 6 + [c_int, c_vp, c_vp, c_size, c_vp]),
    "sd_bn_bwd_finalize": (c_int, [c_vp, c_int, c_i64, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_vp]),
    "sd_bn_bwd_apply": (c_int, [c_vp, c_vp, c_vp, c_int, c_i64, c_int] + [c_vp] * 8),
    "sd_bn_finalize_scratch_rows": (c_int, [c_int]),
    "sd_bn_finalize_stats": (c_int, [c_vp, c_int, c_i64, c_int, c_float, c_float, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "sd_allreduce_unique_id": (c_int, [c_vp]),
    "sd_allreduce_init": (c_int, [c_vp, c_int, c_int, C.POINTER(c_vp)]),
    "sd_allreduce_run": (c_int, [c_vp, c_vp, c_i64, c_vp]),
    "sd_allreduce_destroy": (c_int, [c_vp]),
    "sd_mfma_bf16_stream_flops": (C.c_double, [c_int]),
    "sd_mfma_bf16_stream": (c_int, [c_vp, c_vp, c_int, c_vp]),
    "sd_comm_sim_copy": (c_int, [c_vp, c_vp, c_size, c_size, c_int, c_float, c_vp]),
    "sd_range_enabled": (c_int, []),
    "sd_range_library": (C.c_char_p, []),
    "sd_range_push": (c_int, [C.c_char_p]),
    "sd_range_pop": (c_int, []),
    "sd_range_mark": (c_int, [C.c_char_p]),
}


def declared_symbols():
    return sorted(_SIGNATURES)


def lib():
    """Load the shared library once.  Fails loudly: there is no fallback path."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise SdError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"(or `make -C {LIB_PATH.parent}`); structuredetector_amd has no CPU / eager fallback.")
        handle = C.CDLL(str(LIB_PATH))
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)      # AttributeError if the .so is stale
            fn.restype = res
            fn.argtypes = args
        flags = handle.sd_build_flags()
        if flags and os.environ.get("SDNET_ALLOW_ABLATION") != "1":
            raise SdError(f"{LIB_PATH} is a timing-only ablation build (sd_build_flags = {flags}: SD_ABLATE_* switches make the conv "
                          "kernels compute wrong results on purpose); rebuild without EXTRA=-DSD_ABLATE_... "
                          "(set SDNET_ALLOW_ABLATION=1 only for timing experiments)")
        _lib = handle
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().sd_last_error().decode(errors="replace")
        raise SdError(f"{what or 'libsdnet_hip'} failed (code {rc}): {msg}")


def stream() -> int:
    """Raw hipStream_t of torch's current stream on the current device (what every C-ABI call is handed)."""
    try:
        return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())     # ~0.3 us (current_stream().cuda_stream: ~3 us)
    except AttributeError:
        return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise SdError("structuredetector_amd ops need tensors on the GPU (device 'cuda'); got a "
                          f"{t.device} tensor and there is no CPU fallback")


def map_view(t: torch.Tensor):
    """(B,C,h,w) fp32 tensor -> (tensor kept alive, ptr, batch stride, channel stride) with contiguous rows.
    Channel-slice views of the head output (network.py:77-84) pass through without a copy."""
    if t.dtype != torch.float32:
        t = t.float()
    B, Cc, h, w = t.shape
    s0, s1, s2, s3 = t.stride()
    ptr = t.data_ptr()
    if not (s3 == 1 and s2 == w and s1 >= h * w and (B == 1 or s0 >= h * w) and s1 % 4 == 0 and s0 % 4 == 0 and ptr % 16 == 0):
        t = t.contiguous()
        s0, s1 = t.stride(0), t.stride(1)
        ptr = t.data_ptr()
    return t, ptr, s0, s1


_ws_cache: dict = {}


def workspace(nbytes: int, device) -> torch.Tensor:
    """Per-(device, stream) grow-only scratch buffer (torch-allocated; the C ABI never allocates)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), stream())
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


_state_cache: dict = {}


def zero_state(nbytes: int, device, tag: str = "conv") -> torch.Tensor:
    """Per-(device, stream, tag) persistent ZEROED buffer for the kernels that keep arrival tickets / hand-off records in caller-owned
    memory (sd_conv2d_fwd_sb, sd_decode_fused): zero at first use, left zero by every launch, never shared between streams."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), stream(), tag)
    buf = _state_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            torch.cuda.current_stream().synchronize()      # a launch may still be using the old one
        buf = torch.zeros(max(nbytes, 1 << 16), dtype=torch.uint8, device=device)
        _state_cache[key] = buf
    return buf
