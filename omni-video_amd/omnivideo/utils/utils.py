"""Video output helpers with the reference's names (reference: omnivideo/utils/utils.py).

The tensor half of save_video / cache_video - clamp to the value range, normalise to [0, 1],
scale by 255, truncate to uint8, frames in [T, H, W, 3] order (utils.py:37-44, 177-184) - runs
as one HIP kernel on the device, so 4x fewer bytes cross PCIe than with the reference's
fp32 -> uint8 conversion on whatever device the tensor lives on followed by `.cpu()`.
The container half (libx264 through imageio) is left to imageio; when it is not installed the
functions raise, unless the caller passes its own `writer(path, frames_uint8_numpy, fps)`.
"""
import binascii
import logging
import os
import os.path as osp

import torch

from .. import _hip as H

__all__ = ["save_video", "cache_video", "video_to_frames", "transform_frames_to_tensor", "best_output_size"]


def rand_name(length=8, suffix=""):
    """reference :14-20"""
    name = binascii.b2a_hex(os.urandom(length)).decode("utf-8")
    if suffix:
        name += suffix if suffix.startswith(".") else "." + suffix
    return name


def video_to_frames(tensor, normalize=True, value_range=(-1, 1)):
    """[1, 3, T, H, W] or [3, T, H, W] fp32 on the HIP device -> uint8 [T, H, W, 3] on the device.
    Equals the reference's clamp + make_grid(normalize=True, value_range) + `* 255` + uint8 cast
    for a batch of one video (a grid of one image is the image itself)."""
    if tensor.dim() == 5:
        if tensor.shape[0] != 1:
            raise NotImplementedError("grids of several videos are not built: pass one video")
        tensor = tensor[0]
    if tensor.dim() != 4 or tensor.shape[0] != 3:
        raise ValueError("expected a [3, T, H, W] video")
    lo, hi = float(min(value_range)), float(max(value_range))
    if not normalize:
        # without normalisation the reference multiplies the clamped values by 255 directly;
        # the same kernel does that with the range [0, 1] after the clamp, which is only the
        # same when the value range already is [0, 1]
        if (lo, hi) != (0.0, 1.0):
            raise NotImplementedError("normalize=False is built for value_range (0, 1) only")
    return H.video_to_uint8(tensor.to(torch.float32).contiguous(), channels_last=False, value_range=(lo, hi))


def transform_frames_to_tensor(frames, target_size=(480, 832), device=None):
    """Source-video frames -> encoder input (reference: tools/inference/generate_omni_v2v.py:306-343).
    frames: uint8 [T, H, W, 3] (numpy array, list of arrays, or tensor); returns fp32 [T, 3, h, w]
    in [-1, 1] on the HIP device - a view of the planar [3, T, h, w] buffer, so the reference's
    `frames.transpose(0, 1)` before `vae.encode` (:614) is again contiguous.

    Centre crop to the target aspect ratio (crop size :318-323, offsets as torchvision's
    center_crop rounds them), then one launch pair that resamples exactly as Pillow's
    Image.resize(BILINEAR) does on uint8 and maps bytes through ToTensor/Normalize."""
    import numpy as np
    if isinstance(frames, (list, tuple)):
        frames = np.stack([np.asarray(f) for f in frames])
    t = torch.as_tensor(frames)
    if t.dtype != torch.uint8 or t.dim() != 4 or t.shape[3] != 3:
        raise ValueError("expected uint8 frames [T, H, W, 3]")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    t = t.to(dev).contiguous()
    h, w = int(t.shape[1]), int(t.shape[2])
    ratio = float(target_size[1]) / float(target_size[0])
    if w < h * ratio:
        ch, cw = int(float(w) / ratio), w
    else:
        ch, cw = h, int(float(h) * ratio)
    top, left = int(round((h - ch) / 2.0)), int(round((w - cw) / 2.0))
    # ToTensor divides by 255 in fp32, Normalize subtracts 0.5 and divides by 0.5: 256 possible results
    lut = torch.arange(256, dtype=torch.float32).div(255).sub(0.5).div(0.5).to(dev)
    out = H.frames_resize_normalize(t, (top, left, ch, cw), (int(target_size[0]), int(target_size[1])), lut)
    return out.transpose(0, 1)


def _imageio_writer(path, frames, fps):
    try:
        import imageio
    except ImportError as e:        # pragma: no cover - depends on the installation
        raise RuntimeError("writing a video file needs imageio (with ffmpeg); pass writer=... otherwise") from e
    w = imageio.get_writer(path, fps=fps, codec="libx264", quality=8)
    for frame in frames:
        w.append_data(frame)
    w.close()


def _write(frames_or_tensor, cache_file, fps, normalize, value_range, writer):
    t = frames_or_tensor
    frames = t if t.dtype == torch.uint8 else video_to_frames(t, normalize, value_range)
    (writer or _imageio_writer)(cache_file, frames.cpu().numpy(), fps)


def save_video(tensor, save_file=None, fps=30, suffix=".mp4", nrow=8, normalize=True, value_range=(-1, 1), writer=None):
    """reference :23-53.  `tensor` may also be the uint8 frames of Wan2_1_VAE.decode_frames."""
    cache_file = osp.join("/tmp", rand_name(suffix=suffix)) if save_file is None else save_file
    try:
        _write(tensor, cache_file, fps, normalize, value_range, writer)
    except Exception as e:
        logging.info(f"save_video failed, error: {e}")


def cache_video(tensor, save_file=None, fps=30, suffix=".mp4", nrow=8, normalize=True, value_range=(-1, 1), retry=5,
                writer=None):
    """reference :160-198: returns the file name, or None after `retry` failures."""
    cache_file = osp.join("/tmp", rand_name(suffix=suffix)) if save_file is None else save_file
    error = None
    for _ in range(retry):
        try:
            _write(tensor, cache_file, fps, normalize, value_range, writer)
            return cache_file
        except Exception as e:
            error = e
    print(f"cache_video failed, error: {error}", flush=True)
    return None


def best_output_size(w, h, dw, dh, expected_area):
    """reference :135-158: the (width, height), multiples of (dw, dh), of at most `expected_area`
    pixels whose aspect ratio is closest to w / h."""
    ratio = w / h
    ow = (expected_area * ratio) ** 0.5
    oh = expected_area / ow
    cands = []
    ow1 = int(ow // dw * dw)
    cands.append((ow1, int(expected_area / ow1 // dh * dh)))
    oh2 = int(oh // dh * dh)
    cands.append((int(expected_area / oh2 // dw * dw), oh2))
    for cw, ch in cands:
        assert cw % dw == 0 and ch % dh == 0 and cw * ch <= expected_area

    def skew(c):
        r = c[0] / c[1]
        return max(ratio / r, r / ratio)
    return cands[0] if skew(cands[0]) < skew(cands[1]) else cands[1]

