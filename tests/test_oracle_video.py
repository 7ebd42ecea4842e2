"""Frame pre/post-processing: the CPU restatement against Pillow's own outputs (golden fixture
written by oracle/gen_golden.py `frames`, and Pillow directly when it imports), and the
library's host-side tap tables against the restatement."""
import ctypes
import os

import numpy as np
import pytest
import torch

from omnivideo.utils import synthetic as syn
from oracle import video as V


@pytest.fixture(scope="module")
def G(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "frames.npz")))


@pytest.mark.parametrize("name", list(syn.FRAME_CASES))
def test_resize_restatement_matches_pillow_golden(G, name):
    T, Hh, W, th, tw = syn.FRAME_CASES[name]
    frames = syn.make_frames(name, T, Hh, W)
    top, left, ch, cw = V.center_crop_box(Hh, W, (th, tw))
    assert top >= 0 and left >= 0 and top + ch <= Hh and left + cw <= W
    for f, want in zip(frames, G[name]):
        got = V.resize_bilinear_u8(f[top:top + ch, left:left + cw], th, tw)
        assert np.array_equal(got, want)


def test_resize_restatement_matches_pillow_live():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(4)
    for (Hh, W, oh, ow) in [(31, 47, 16, 24), (50, 50, 50, 30), (19, 23, 40, 52), (120, 214, 48, 80)]:
        img = rng.integers(0, 256, (Hh, W, 3), dtype=np.uint8)
        want = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
        assert np.array_equal(V.resize_bilinear_u8(img, oh, ow), want)


def test_center_crop_box_rounds_half_to_even():
    # (37 - 34) / 2 = 1.5 -> 2 ; (39 - 36) / 2 = 1.5 -> 2 ; (h - ch) / 2 = 2.5 -> 2
    assert V.center_crop_box(37, 61, (16, 28)) == (2, 0, 34, 61)
    assert V.center_crop_box(45, 70, (16, 28)) == (2, 0, 40, 70)
    assert V.center_crop_box(480, 832, (480, 832)) == (0, 0, 480, 832)


def test_transform_and_uint8_round_trip():
    """normalised tensor -> cache_video's uint8 gives the resized bytes back (b/255 maps to b
    or b-1 under the truncating cast, never further)."""
    name = "down_wide"
    T, Hh, W, th, tw = syn.FRAME_CASES[name]
    x = V.transform_frames_to_tensor(syn.make_frames(name, T, Hh, W), (th, tw))
    assert tuple(x.shape) == (T, 3, th, tw) and float(x.min()) >= -1 and float(x.max()) <= 1
    back = V.video_to_uint8(x.transpose(0, 1)).numpy().astype(int)
    top, left, ch, cw = V.center_crop_box(Hh, W, (th, tw))
    res = np.stack([V.resize_bilinear_u8(f[top:top + ch, left:left + cw], th, tw) for f in syn.make_frames(name, T, Hh, W)])
    d = res.astype(int) - back
    assert d.min() >= 0 and d.max() <= 1


@pytest.mark.parametrize("in_size,out_size", [(80, 40), (45, 24), (30, 48), (40, 40), (356, 40), (61, 28), (1280, 832), (7, 3)])
def test_library_tap_tables_equal_restatement(in_size, out_size):
    """ovd_resize_coeffs is host arithmetic: callable without a GPU."""
    import omnivideo._hip as H
    L = H.lib()
    k = L.ovd_resize_coeffs(in_size, out_size, None, None, 0)
    bounds, kk = V.bilinear_coeffs(in_size, out_size)
    assert k == kk.shape[1]
    b = np.zeros((out_size, 2), dtype=np.int32)
    c = np.full((out_size, k), -7, dtype=np.int32)
    assert L.ovd_resize_coeffs(in_size, out_size, b.ctypes.data_as(ctypes.c_void_p), c.ctypes.data_as(ctypes.c_void_p), k) == k
    assert np.array_equal(b, bounds) and np.array_equal(c, kk)
    assert (c.sum(axis=1) - (1 << 22)).__abs__().max() <= k        # taps sum to one in 22-bit fixed point
    # a too-small buffer writes nothing and reports the needed width; oversized requests are refused
    c2 = np.full((out_size, 1), -7, dtype=np.int32)
    if k > 1:
        assert L.ovd_resize_coeffs(in_size, out_size, b.ctypes.data_as(ctypes.c_void_p), c2.ctypes.data_as(ctypes.c_void_p), 1) == k
        assert (c2 == -7).all()
    assert L.ovd_resize_coeffs(4000, 10, None, None, 0) == -1 and L.ovd_resize_coeffs(0, 10, None, None, 0) == -1
