"""ctypes binding of ``libavdiff_hip.so`` (C ABI declared in ``include/avdiff_hip.h``).

PyTorch is plumbing here: it owns device memory and the stream; every compute call goes through the C ABI
with raw device pointers.  There is NO fallback: if the shared library is missing or a tensor is not on a
ROCm device the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional

import torch

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("AVDIFF_HIP_LIB", _HERE / "csrc" / "libavdiff_hip.so"))

ACT_NONE, ACT_GELU, ACT_SILU, ACT_TANH, ACT_RELU, ACT_LEAKY_RELU = 0, 1, 2, 3, 4, 5
EINVAL, EUNSUPPORTED, ELAUNCH, EWORKSPACE = -1, -2, -3, -4


class AvdError(RuntimeError):
    """A HIP-side failure (launch error, unsupported shape, workspace)."""


class EmbedDesc(C.Structure):
    _fields_ = [("target_kind", C.c_int), ("target_first", C.c_int), ("B", C.c_int), ("d", C.c_int),
                ("tdim", C.c_int), ("C", C.c_int), ("T", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("p0", C.c_int), ("p1", C.c_int), ("p2", C.c_int), ("Nt", C.c_int), ("Np", C.c_int),
                ("temb_freqs", C.c_void_p), ("temb_add", C.c_int)]


class BlockWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "norm1_scale", "in_proj_weight", "in_proj_bias", "out_proj_weight", "out_proj_bias",
        "norm2_scale", "fc1_weight", "fc1_bias", "fc2_weight", "fc2_bias",
        "in_proj_weight_n", "fc1_weight_n",
        "in_proj_weight3", "out_proj_weight3", "fc1_weight3", "fc2_weight3",
        "norm1_bias", "norm2_bias")] + [("f16x2_scale", C.c_float * 8), ("in_proj_weight3n", C.c_void_p), ("fc1_weight3n", C.c_void_p)]


class CoreWeights(C.Structure):
    _fields_ = [("d", C.c_int), ("n_layers", C.c_int), ("n_heads", C.c_int), ("mlp_hidden", C.c_int),
                ("norm_eps", C.c_float), ("blocks", C.POINTER(BlockWeights)), ("final_norm_scale", C.c_void_p),
                ("norm_kind", C.c_int), ("final_norm_bias", C.c_void_p), ("split_terms", C.c_int), ("attn_mode", C.c_int)]


class HeadWeights(C.Structure):
    _fields_ = [("d_in", C.c_int), ("hidden", C.c_int), ("d_out", C.c_int), ("n_shared", C.c_int),
                ("ln_eps", C.c_float), ("act", C.c_int),
                ("input_proj_weight", C.c_void_p), ("input_proj_bias", C.c_void_p),
                ("shared_lin_weight", C.POINTER(C.c_void_p)), ("shared_lin_bias", C.POINTER(C.c_void_p)),
                ("shared_ln_weight", C.POINTER(C.c_void_p)), ("shared_ln_bias", C.POINTER(C.c_void_p)),
                ("out_proj_weight", C.c_void_p), ("out_proj_bias", C.c_void_p),
                ("split_terms", C.c_int), ("input_proj_weight3", C.c_void_p), ("shared_lin_weight3", C.POINTER(C.c_void_p)),
                ("out_proj_weight3", C.c_void_p), ("f16x2_scale", C.POINTER(C.c_float))]


class StepDesc(C.Structure):
    _fields_ = [("embed", EmbedDesc), ("core", C.POINTER(CoreWeights)), ("head", C.POINTER(HeadWeights)),
                ("adapt_w", C.c_void_p), ("adapt_b", C.c_void_p), ("alpha_bar", C.c_void_p), ("T_train", C.c_int),
                ("guidance", C.c_float), ("eta", C.c_float), ("split_streams", C.c_int)]


class VaeDecodeDesc(C.Structure):
    _fields_ = [("B", C.c_int), ("Cv", C.c_int), ("Tp", C.c_int), ("Hp", C.c_int), ("Wp", C.c_int),
                ("T", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("base", C.c_int), ("n_blocks", C.c_int), ("out_ch", C.c_int), ("out_tanh", C.c_int),
                ("gn_eps", C.c_float),
                ("from_lat_w", C.c_void_p), ("from_lat_b", C.c_void_p),
                ("conv_w", C.POINTER(C.c_void_p)), ("conv_b", C.POINTER(C.c_void_p)),
                ("gn_w", C.POINTER(C.c_void_p)), ("gn_b", C.POINTER(C.c_void_p)),
                ("to_img_w", C.c_void_p), ("to_img_b", C.c_void_p),
                ("conv_w3", C.POINTER(C.c_void_p)), ("conv_terms", C.c_int),
                ("conv_w_scale", C.POINTER(C.c_float)), ("conv_a_scale", C.POINTER(C.c_float)),
                ("conv0_lat_w3", C.c_void_p), ("conv0_lat_btab", C.c_void_p), ("conv0_lat_w_scale", C.c_float),
                ("conv0_lat_packed", C.c_int)]


class VaeEncodeDesc(C.Structure):
    _fields_ = [("B", C.c_int), ("in_ch", C.c_int), ("T", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("t_down", C.c_int), ("s_down", C.c_int),
                ("base", C.c_int), ("n_blocks", C.c_int), ("lat_ch", C.c_int), ("gn_eps", C.c_float),
                ("conv_w", C.POINTER(C.c_void_p)), ("conv_b", C.POINTER(C.c_void_p)),
                ("gn_w", C.POINTER(C.c_void_p)), ("gn_b", C.POINTER(C.c_void_p)),
                ("to_lat_w", C.c_void_p), ("to_lat_b", C.c_void_p),
                ("conv_w3", C.POINTER(C.c_void_p)), ("conv_terms", C.c_int),
                ("conv_w_scale", C.POINTER(C.c_float)), ("conv_a_scale", C.POINTER(C.c_float)),
                ("conv0_pk_w3", C.c_void_p)]


ABI_VERSION = 7
_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> (restype, argtypes); must list every symbol include/avdiff_hip.h declares
SIGNATURES = {
    "avd_abi_version": (_I, []),
    "avd_last_error": (C.c_char_p, []),
    "avd_device_arch": (_I, [C.c_char_p, _I]),
    "avd_rmsnorm_f32": (_I, [_P, _P, _P, _L, _I, _F, _P]),
    "avd_gemm_bias_act_f32": (_I, [_P, _L, _P, _P, _P, _L, _P, _L, _L, _I, _I, _I, _P]),
    "avd_gemm_rmsfold_f32": (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _I, _P, _I, _F, _P, _P]),
    "avd_tune_set": (_I, [C.c_char_p, _L]),
    "avd_attn_fwd_f32": (_I, [_P, _P, _I, _I, _I, _I, _F, _I, _P, _P]),
    "avd_layernorm_act_f32": (_I, [_P, _P, _P, _P, _L, _I, _F, _I, _P]),
    "avd_timestep_embedding_f32": (_I, [_P, _P, _P, _I, _I, _F, _P]),
    "avd_tube_patch_f32": (_I, [_P, _P] + [_I] * 8 + [_P]),
    "avd_tube_unpatch_f32": (_I, [_P, _P] + [_I] * 8 + [_P]),
    "avd_audio_tokens_f32": (_I, [_P, _P] + [_I] * 5 + [_P]),
    "avd_audio_untokens_f32": (_I, [_P, _P, _P] + [_I] * 5 + [_P]),
    "avd_ddim_step_f32": (_I, [_P, _P, _P, _P, _P, _I, _F, _P, _P, _I, _L, _P]),
    "avd_cfg_unpatch_ddim_f32": (_I, [_P, _P, _P, _P, _P, _I, _F, _F, _P, _P] + [_I] * 8 + [_P]),
    "avd_cfg_untoken_ddim_audio_f32": (_I, [_P, _P, _P, _P, _P, _I, _F, _F, _P, _P] + [_I] * 5 + [_P]),
    "avd_embed_workspace_floats": (_L, [C.POINTER(EmbedDesc)]),
    "avd_embed_cfg_pair_f32": (_I, [C.POINTER(EmbedDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    "avd_core_workspace_bytes": (_L, [C.POINTER(CoreWeights), _I, _I]),
    "avd_core_forward_f32": (_I, [C.POINTER(CoreWeights), _P, _P, _I, _I, _I, _I, _P, _P, _L, _P]),
    "avd_head_workspace_bytes": (_L, [C.POINTER(HeadWeights), _L]),
    "avd_head_forward_f32": (_I, [C.POINTER(HeadWeights), _P, _L, _L, _L, _L, _P, _P, _L, _P]),
    "avd_step_workspace_bytes": (_L, [C.POINTER(StepDesc)]),
    "avd_denoise_step_f32": (_I, [C.POINTER(StepDesc), _P, _P, _P, _P, _P, _P, _P, _L, _P]),
    "avd_sched_advance": (_I, [_P, _I, _P, _P, _P, _I, _P]),
    "avd_split3_bytes": (_L, [_L, _I]),
    "avd_split3_f32": (_I, [_P, _P, _L, _I, _P]),
    "avd_rmsnorm_split3_f32": (_I, [_P, _P, _P, _L, _I, _F, _P]),
    "avd_attn_fwd_split3_f32": (_I, [_P, _P, _I, _I, _I, _I, _F, _I, _P]),
    "avd_qkv3_bytes": (_L, [_I, _I, _I]),
    "avd_gemm_bf16x3_qkv3_f32": (_I, [_P, _P, _P, _P, _L, _I, _I, _I, _F, _I, _P]),
    "avd_attn_fwd_qkv3_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "avd_attn_fp8_workspace_bytes": (_L, [_I, _I, _I]),
    "avd_attn_fwd_fp8_f32": (_I, [_P, _P, _L, _P, _P, _I, _I, _I, _I, _P]),
    "avd_attn_fwd_fp8_f16x2_f32": (_I, [_P, _P, _L, _P, _P, _I, _I, _I, _I, _F, _F, _P]),
    "avd_gemm_bf16x3_f32": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _P]),
    "avd_weight_bounds_f32": (_I, [_P, _L, _I, _P, _P]),
    "avd_split_f16x2_f32": (_I, [_P, _P, _L, _I, _F, _P]),
    "avd_rmsnorm_split_f16x2_f32": (_I, [_P, _P, _P, _L, _I, _F, _F, _P]),
    "avd_gemm_f16x2_f32": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _F, _F, _P]),
    "avd_gemm_f16x2_qkv_f32": (_I, [_P, _P, _P, _P, _L, _I, _I, _I, _F, _F, _F, _P]),
    "avd_attn_fwd_qkv_f16x2_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _F, _F, _P]),
    "avd_conv3_weight_bytes": (_L, []),
    "avd_conv3_weight_f32": (_I, [_P, _P, _P]),
    "avd_conv3_weight_f16x2_f32": (_I, [_P, _P, _F, _P]),
    "avd_vae_decode_workspace_bytes": (_L, [C.POINTER(VaeDecodeDesc)]),
    "avd_vae_decode_f32": (_I, [C.POINTER(VaeDecodeDesc), _P, _P, _P, _L, _P]),
    "avd_vae_encode_workspace_bytes": (_L, [C.POINTER(VaeEncodeDesc)]),
    "avd_vae_encode_f32": (_I, [C.POINTER(VaeEncodeDesc), _P, _P, _P, _L, _P]),
    "avd_conv1d_act_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "avd_avgpool_frames_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "avd_crossfade_f32": (_I, [_P, _P, _P, _I, _I, _I, _L, _P]),
    "avd_crossfade_u8": (_I, [_P, _P, _P, _I, _I, _I, _L, _P]),
    "avd_prof_enable": (_I, [_I]),
    "avd_prof_num_tags": (_I, []),
    "avd_prof_tag_name": (C.c_char_p, [_I]),
    "avd_prof_report": (_I, [C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double), _I]),
}

def prof_enable(on: bool) -> None:
    check(lib().avd_prof_enable(1 if on else 0))


def prof_report():
    """{kernel name: (launches, total_ms, algorithmic work)} for the launches recorded since prof_enable(True)."""
    n = lib().avd_prof_num_tags()
    cnt, ms, work = (C.c_int64 * max(n, 1))(), (C.c_double * max(n, 1))(), (C.c_double * max(n, 1))()
    check(lib().avd_prof_report(cnt, ms, work, n))
    return {lib().avd_prof_tag_name(i).decode(): (int(cnt[i]), float(ms[i]), float(work[i])) for i in range(n)}


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load the HIP library (once).  Raises if it has not been built — there is no CPU path."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise AvdError(
                f"{LIB_PATH} not found: build it with `make -C multimodal_diffusion_amd/csrc` "
                "(or __graft_entry__.build()); this package has no CPU/PyTorch fallback")
        handle = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError if the .so is stale / missing a symbol
            fn.restype, fn.argtypes = res, args
        if handle.avd_abi_version() != ABI_VERSION:
            raise AvdError(f"ABI version mismatch: library {handle.avd_abi_version()}, binding {ABI_VERSION}")
        _lib = handle
    return _lib


def check(rc: int) -> None:
    if rc == 0:
        return
    msg = lib().avd_last_error().decode("utf-8", "replace")
    if rc == EINVAL:
        raise ValueError(msg)
    raise AvdError(f"[avd {rc}] {msg}")


def stream_ptr(device: torch.device) -> int:
    """torch's current stream on `device`.  The library launches on the CURRENT device (one device per call, per-device lazy
    state inside), so a tensor living on another device than the current one is refused rather than silently mis-launched."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx != torch.cuda.current_device():
        raise AvdError(f"tensors are on cuda:{idx} but the current device is cuda:{torch.cuda.current_device()}: "
                       "call torch.cuda.set_device / use torch.cuda.device(...) around the call")
    return torch.cuda.current_stream(device).cuda_stream


# matrix-pipe modes of the MMDiT core / engine -> product terms of the split-operand kernels (avd_core_weights.split_terms)
# "auto" (the modules' default): the exact three-plane bf16x3 kernels wherever the library's own rule engages them (RMSNorm, no
# key-padding mask, >= 2,048 rows 2BN — 6,144 for the head and in the other split modes —, widths the 256-column tiles cover) and the fp32 MFMA kernels — with the norms folded into their
# neighbours — everywhere else; same fp32-level results either way
MATMUL_TERMS = {"auto": 6, "f32": 0, "bf16x3": 6, "bf16x3_strict": 9, "bf16": 1, "f16x2": 3}


def dev_f32(t: torch.Tensor, name: str = "tensor") -> torch.Tensor:
    """Validate a tensor handed to a kernel: ROCm device, fp32, contiguous (made so if not)."""
    if not t.is_cuda:
        raise AvdError(f"{name} is on {t.device}: the HIP hot path needs ROCm device tensors (no CPU fallback)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32 (the reference path is fp32), got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def dev_i64(t: torch.Tensor, device: torch.device, name: str = "timesteps") -> torch.Tensor:
    if t.dtype != torch.long:
        t = t.long()
    if t.device != device:
        t = t.to(device)
    return t if t.is_contiguous() else t.contiguous()


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()
