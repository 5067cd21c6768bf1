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


@pytest.mark.parametrize("mode,S,dtype", [("pil", 112, torch.bfloat16), ("cv2", 160, torch.bfloat16), ("pil", 57, torch.float32), ("cv2", 224, torch.float32)])
def test_pre_step_projection_and_layernorm_in_one_launch(dev, monkeypatch, mode, S, dtype):
    """fmmt_patch_embed_u8_ln_fwd (uint8 crops -> LayerNorm(PatchEmbed projection), the patch matrix fed to the MFMA out of LDS): the patch matrix it
    writes is the oracle's, bit for bit; y / x_pre / statistics equal the two launches it replaces bit for bit (same tile arithmetic on the same
    values: patch_ln_core.h); at inference the patch matrix is not written and y is unchanged; parameter gradients equal the two-launch
    form's (same backward on the same saved tensors)."""
    from facialmmt_amd import _lib
    from facialmmt_amd.modules.SwinTransformer import Swin_Transformer as SW
    img = _crops("crop_fused", 3, S, seed=S + 3)
    img[0, :3] = 255
    img[0, -3:] = 0
    x = torch.from_numpy(img).to(dev)
    pe = SW.PatchEmbed(224, 4, 3, 96, torch.nn.LayerNorm)
    synth.fill_state_dict(pe, seed=31, prefix="pe.")
    pe.to(dev)
    lib = _lib.load()
    calls = []
    real = lib.fmmt_patch_embed_u8_ln_fwd
    monkeypatch.setattr(lib, "fmmt_patch_embed_u8_ln_fwd", lambda *a: (calls.append(a[12]), real(*a))[1])
    outs = []
    for fused in (True, False):
        monkeypatch.setattr(ops, "_PATCH_U8_LN", fused)
        pe.zero_grad(set_to_none=True)
        y = pe.forward_u8(x, mode, dtype)
        w = torch.randn(y.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(9)).to(dtype)
        grads = torch.autograd.grad(y, list(pe.parameters()), w)
        outs.append((y.detach(), grads))
    assert len(calls) == 1 and calls[0] is not None          # one fused call, training: the patch matrix pointer was handed over
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)
    monkeypatch.setattr(ops, "_PATCH_U8_LN", True)
    with torch.no_grad():
        y_inf = pe.forward_u8(x, mode, dtype)
    assert len(calls) == 2 and calls[1] is None and torch.equal(y_inf, outs[0][0])
    # the raw entry point: patch matrix against the oracle, bit-exact
    code, tab, lut = ops.resize_tables(mode, S, dev)
    cols = torch.empty(3 * 3136, 48, dtype=dtype, device=dev)
    y2 = torch.empty(3 * 3136, 96, dtype=dtype, device=dev)
    w2d = pe.proj.weight.detach().view(96, 48).to(dtype).contiguous()
    rc = real(_lib.BF16 if dtype == torch.bfloat16 else _lib.F32, code, 3, S, x.data_ptr(), tab.data_ptr(), lut.data_ptr(), w2d.data_ptr(),
              pe.proj.bias.detach().float().data_ptr(), pe.norm.weight.detach().float().data_ptr(), pe.norm.bias.detach().float().data_ptr(), 1e-5,
              cols.data_ptr(), None, y2.data_ptr(), None, None, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    want = torch.from_numpy(P.patch_cols_from_u8(img, mode))
    assert torch.equal(cols.cpu(), want.to(dtype))
    assert torch.equal(y2.view_as(y_inf), y_inf)
    # refused, not mis-computed
    assert real(2, code, 3, S, x.data_ptr(), tab.data_ptr(), lut.data_ptr(), w2d.data_ptr(), None, pe.norm.weight.data_ptr(), pe.norm.bias.data_ptr(), 1e-5,
                None, None, y2.data_ptr(), None, None, 0) == -1
    assert real(1, code, 3, 300, x.data_ptr(), tab.data_ptr(), lut.data_ptr(), w2d.data_ptr(), None, pe.norm.weight.data_ptr(), pe.norm.bias.data_ptr(), 1e-5,
                None, None, y2.data_ptr(), None, None, 0) == -1
