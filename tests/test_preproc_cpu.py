"""CPU tests of the input pre-step: the oracle's integer restatement of Pillow's bicubic resize
against Pillow itself (live, where importable -- it is in this image) and against the committed fixtures; the library's
host-side table builder (fmmt_resize_table, C) against the oracle's tables for both flavours; properties of the
unpinned cv2 flavour."""
import ctypes

import numpy as np
import pytest

from facialmmt_amd import _lib, synth
from oracle import preproc as P


def _crops(name, n, S, seed):
    return synth.randint(name, (n, S, S, 3), 0, 256, seed=seed).astype(np.uint8)


@pytest.mark.parametrize("S", [112, 160, 96, 64, 223])
def test_oracle_resize_equals_pillow_live(S):
    Image = pytest.importorskip("PIL.Image")
    img = _crops("crop_live", 2, S, seed=S)
    want = np.stack([np.asarray(Image.fromarray(i, "RGB").resize((224, 224), Image.BICUBIC)) for i in img])
    assert np.array_equal(P.resize_u8(img, "pil"), want)


def test_oracle_resize_equals_pillow_fixture(golden):
    z = golden.files["preproc"]
    for S in (112, 160):
        img = _crops("crop", 2, S, seed=7)
        assert np.array_equal(P.resize_u8(img, "pil"), z[f"pil_resize_{S}"])
    assert np.array_equal(P.frames_from_u8(_crops("crop", 2, 112, seed=7)[:1], "pil"), z["pil_frames_112"])   # ToTensor + Normalize (torch, float32)


@pytest.mark.parametrize("mode", ["pil", "cv2"])
@pytest.mark.parametrize("S", [112, 160, 100, 224])
def test_library_table_equals_oracle_table(mode, S):
    lib = _lib.load()
    tab = (ctypes.c_int32 * (224 * 8))()
    lut = (ctypes.c_float * 256)()
    code = {"pil": _lib.RESIZE_PIL, "cv2": _lib.RESIZE_CV2}[mode]
    assert lib.fmmt_resize_table(code, S, 224, ctypes.addressof(tab), ctypes.addressof(lut)) == 0
    got = np.array(list(tab), np.int32).reshape(224, 8)
    idx, w = P.table(mode, S, 224)
    assert np.array_equal(got[:, 4:], w)
    assert np.array_equal(got[:, :4][w != 0], idx[w != 0])            # indices of zero-weight padding taps are free
    assert got[:, :4].min() >= 0 and got[:, :4].max() < S
    assert np.array_equal(np.array(list(lut), np.float32), P.normalize_lut())
    assert 0 < lib.fmmt_resize_band_rows(ctypes.addressof(tab), 224) <= 8
    assert lib.fmmt_resize_table(code, 300, 224, ctypes.addressof(tab), None) == -1       # down-scaling: refused
    assert lib.fmmt_resize_table(5, S, 224, ctypes.addressof(tab), None) == -1


def test_cv2_flavour_properties():
    """unpinned flavour: what must hold for any correct cubic resize with replicate border"""
    const = np.full((1, 112, 112, 3), 137, np.uint8)
    assert np.array_equal(P.resize_u8(const, "cv2"), np.full((1, 224, 224, 3), 137, np.uint8))
    img = _crops("crop_cv", 2, 112, seed=3)
    out = P.resize_u8(img, "cv2")
    assert np.array_equal(P.resize_u8(img[:, ::-1], "cv2"), out[:, ::-1])                  # flips commute with the resize
    assert np.array_equal(P.resize_u8(img[:, :, ::-1], "cv2"), out[:, :, ::-1])
    assert np.array_equal(P.resize_u8(img.transpose(0, 2, 1, 3), "cv2"), out.transpose(0, 2, 1, 3))
    # against a float evaluation of the same a = -0.75 kernel: rounding differences only
    idx, w = P.cv2_table(112, 224)
    wf = w.astype(np.float64) / 2048.0
    a = img.astype(np.float64)
    h = sum(a[:, :, idx[:, k], :] * wf[:, k][:, None] for k in range(4))
    v = sum(h[:, idx[:, k], :, :] * wf[:, k][:, None, None] for k in range(4))
    assert np.abs(np.clip(np.rint(v), 0, 255) - out).max() <= 1
    assert np.abs(wf.sum(1) - 1).max() <= 2 / 2048.0


def test_patch_order_matches_conv_weight_layout():
    import torch
    frames = synth.tensor("frames", (2, 3, 224, 224), seed=1)
    w = synth.tensor("pe_w", (96, 3, 4, 4), seed=2)
    want = torch.nn.functional.conv2d(frames, w, stride=4).flatten(2).transpose(1, 2).reshape(-1, 96)
    got = torch.from_numpy(P.patch_cols(frames.numpy())) @ w.reshape(96, 48).t()
    assert torch.allclose(got, want, atol=1e-4, rtol=1e-4)
