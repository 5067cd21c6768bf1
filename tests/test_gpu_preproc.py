"""GPU parity of the fused input pre-step (fmmt_patch_embed_u8): uint8 crops -> patch matrix of
Normalize(ToTensor(bicubic_224(crop))).  Integer work: BIT-EXACT against the oracle (oracle/preproc.py, itself held
bit-exactly to Pillow's own output -- tests/golden/preproc.npz) for both flavours, several source sizes and both
activation dtypes; the Swin model fed uint8 crops equals the model fed the float frames the reference would build."""
import numpy as np
import pytest
import torch

from facialmmt_amd import ops, synth
from oracle import preproc as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _crops(name, n, S, seed):
    return synth.randint(name, (n, S, S, 3), 0, 256, seed=seed).astype(np.uint8)


@pytest.mark.parametrize("mode", ["pil", "cv2"])
@pytest.mark.parametrize("S", [112, 160, 100, 224, 57])
def test_patch_matrix_bit_exact(dev, mode, S):
    img = _crops("crop_gpu", 3, S, seed=S + (7 if mode == "pil" else 11))
    img[0, :3] = 255                                          # saturated and black borders: overshoot must clip
    img[0, -3:] = 0
    want = P.patch_cols_from_u8(img, mode)
    got = ops.patch_embed_u8(torch.from_numpy(img).to(dev), mode, torch.float32)
    assert got.shape == (3 * 3136, 48)
    assert np.array_equal(got.cpu().numpy(), want)
    got16 = ops.patch_embed_u8(torch.from_numpy(img).to(dev), mode, torch.bfloat16)
    assert torch.equal(got16.cpu(), torch.from_numpy(want).bfloat16())


def test_against_pillow_fixture(golden, dev):
    z = golden.files["preproc"]
    for S in (112, 160):
        img = _crops("crop", 2, S, seed=7)
        got = ops.patch_embed_u8(torch.from_numpy(img).to(dev), "pil", torch.float32).cpu().numpy()
        frames = P.normalize_lut()[z[f"pil_resize_{S}"]].transpose(0, 3, 1, 2)                 # Pillow's own resize output
        assert np.array_equal(got, P.patch_cols(np.ascontiguousarray(frames)))
    got = ops.patch_embed_u8(torch.from_numpy(_crops("crop", 2, 112, seed=7)[:1]).to(dev), "pil", torch.float32).cpu().numpy()
    assert np.array_equal(got, P.patch_cols(z["pil_frames_112"]))                            # incl. torch's ToTensor + Normalize


def test_rejects_bad_input(dev):
    from facialmmt_amd import _lib
    with pytest.raises(ValueError):
        ops.patch_embed_u8(torch.zeros(2, 112, 100, 3, dtype=torch.uint8, device=dev), "pil", torch.float32)
    with pytest.raises(ValueError):
        ops.patch_embed_u8(torch.zeros(2, 112, 112, 3, device=dev), "pil", torch.float32)
    with pytest.raises(ValueError):
        ops.patch_embed_u8(torch.zeros(2, 112, 112, 3, dtype=torch.uint8, device=dev), "lanczos", torch.float32)
    with pytest.raises(_lib.FmmtError):
        ops.patch_embed_u8(torch.zeros(2, 300, 300, 3, dtype=torch.uint8, device=dev), "pil", torch.float32)   # down-scaling
    with pytest.raises(_lib.FmmtError):
        ops.patch_embed_u8(torch.zeros(2, 112, 112, 3, dtype=torch.uint8), "pil", torch.float32)               # CPU tensor


@pytest.mark.parametrize("mode", ["pil", "cv2"])
def test_swin_on_uint8_equals_swin_on_reference_frames(dev, mode):
    """SwinTransformer.forward(uint8 crops) == forward(the float frames the reference's pipeline builds from them):
    bit-identical, because the patch matrices are."""
    import os
    from facialmmt_amd.modules.SwinTransformer import Swin_Transformer as S
    from facialmmt_amd.modules.SwinTransformer.backbone_def import BackboneFactory
    m = BackboneFactory("SwinTransformer", os.path.join(os.path.dirname(S.__file__), "swin_conf.yaml")).get_backbone()
    synth.fill_state_dict(m, seed=100)
    m.to(dev).eval()
    m.input_resize = mode
    img = _crops("crop_swin", 4, 112, seed=5)
    frames = torch.from_numpy(P.frames_from_u8(img, mode)).to(dev)
    with torch.no_grad():
        a = m(torch.from_numpy(img).to(dev))
        b = m(frames)
    assert torch.equal(a, b)
    m.train()
    x = torch.from_numpy(img).to(dev)
    out = m(x)
    out.square().sum().backward()                                  # integer input: no input gradient, parameters get theirs
    assert m.patch_embed.proj.weight.grad is not None and torch.isfinite(m.patch_embed.proj.weight.grad).all()


def test_full_size_batch_is_chunk_independent(dev):
    """bench size (640 crops): every crop is an independent unit -- the patch matrix of the batch is the concatenation of
    the per-chunk matrices, bit for bit, and equals the oracle on a sample of crops"""
    g = torch.Generator().manual_seed(0)
    img = torch.randint(0, 256, (640, 112, 112, 3), generator=g, dtype=torch.uint8)
    full = ops.patch_embed_u8(img.to(dev), "pil", torch.bfloat16)
    part = ops.patch_embed_u8(img[100:164].to(dev), "pil", torch.bfloat16)
    assert torch.equal(full[100 * 3136:164 * 3136], part)
    pick = [0, 333, 639]
    want = torch.from_numpy(P.patch_cols_from_u8(img[pick].numpy(), "pil")).bfloat16()
    got = torch.cat([full[i * 3136:(i + 1) * 3136] for i in pick]).cpu()
    assert torch.equal(got, want)
