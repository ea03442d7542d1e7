"""GPU-side image pre / post-processing around the hot path.

Mirrors basicsr/utils/img_util.py:9-35 (`img2tensor`) + the `/255.` of
inference_femasr.py:55, and img_util.py:38-94 (`tensor2img`), but on the device:
the uint8 image crosses PCIe (12x less than the fp32 HR tensor the reference
copies back, img_util.py:66) and the clamp / scale / round-half-even runs in a
HIP kernel.  No CPU fallback: raises if the tensors are not on a GPU.
"""
import ctypes

import torch

from . import _lib


def u8_to_input(img_u8_hwc: torch.Tensor, bgr: bool = False) -> torch.Tensor:
    """uint8 (H,W,3) cuda tensor (RGB, or BGR if bgr) -> float32 (1,3,H,W) RGB in [0,1]."""
    if img_u8_hwc.device.type != 'cuda':
        raise _lib.FemasrError('u8_to_input: tensor must be on the GPU (no CPU fallback)')
    if img_u8_hwc.dtype != torch.uint8 or img_u8_hwc.dim() != 3 or img_u8_hwc.shape[2] != 3:
        raise ValueError(f'expected uint8 (H,W,3), got {img_u8_hwc.dtype} {tuple(img_u8_hwc.shape)}')
    lib = _lib.load()
    src = img_u8_hwc.contiguous()
    h, w, _ = src.shape
    out = torch.empty((1, 3, h, w), dtype=torch.float32, device=src.device)
    stream = torch.cuda.current_stream(src.device).cuda_stream
    _lib.check(lib.femasr_image_u8_to_f32(ctypes.c_void_p(stream), _lib.ptr(src), h, w, int(bgr), _lib.ptr(out)))
    return out


def output_to_u8(out_nchw: torch.Tensor, bgr: bool = False) -> torch.Tensor:
    """float32 (1,3,H,W) RGB -> uint8 (H,W,3) (RGB, or BGR if bgr): clamp, *255, round half to even."""
    if out_nchw.device.type != 'cuda':
        raise _lib.FemasrError('output_to_u8: tensor must be on the GPU (no CPU fallback)')
    if out_nchw.dim() != 4 or out_nchw.shape[0] != 1 or out_nchw.shape[1] != 3:
        raise ValueError(f'expected (1,3,H,W), got {tuple(out_nchw.shape)}')
    lib = _lib.load()
    src = out_nchw.detach().to(torch.float32).contiguous()
    _, _, h, w = src.shape
    out = torch.empty((h, w, 3), dtype=torch.uint8, device=src.device)
    stream = torch.cuda.current_stream(src.device).cuda_stream
    _lib.check(lib.femasr_image_f32_to_u8(ctypes.c_void_p(stream), _lib.ptr(src), h, w, int(bgr), _lib.ptr(out)))
    return out
