"""The device-side frame filter (facialmmt_amd/train_step.select_frames) against the literal loop
restatement of train.py:75-114 (oracle/train_glue.py), plus the property SURVEY.md 8c names: with
threshold 0 every real frame is kept and the mask is unchanged (single utterance, where the reference's
multi-utterance boundary quirk cannot act)."""
import torch

from facialmmt_amd.train_step import select_frames
from oracle.train_glue import select_frames_loop


def _case(g, B, Lv, nmax, peaky):
    num = torch.randint(max(1, nmax // 2), nmax + 1, (B,), generator=g)
    nF = int(num.sum())
    logits = torch.randn(nF, 7, generator=g) * (3.0 if peaky else 0.3)
    preds = torch.softmax(logits, -1)
    vis = torch.randn(B, Lv, 16, generator=g)
    mask = (torch.arange(Lv).view(1, Lv) < num.view(B, 1)).float()
    return preds, vis, mask, num


def test_matches_literal_loop():
    g = torch.Generator().manual_seed(0)
    for trial in range(60):
        B = int(torch.randint(1, 5, (1,), generator=g))
        preds, vis, mask, num = _case(g, B, 12, 12, peaky=(trial % 3 != 0))
        thr = [0.2, 0.5, 0.99, 0.0][trial % 4]
        a, am = select_frames(preds, vis, mask, num, thr)
        b, bm = select_frames_loop(preds, vis, mask, num.tolist(), thr)
        assert torch.equal(am, bm), (trial, am, bm)
        assert torch.allclose(a, b, atol=0, rtol=0), trial


def test_threshold_zero_keeps_everything_single_utterance():
    g = torch.Generator().manual_seed(1)
    preds, vis, mask, num = _case(g, 1, 10, 10, peaky=True)
    out, m = select_frames(preds, vis, mask, num, 0.0)
    assert torch.equal(m, mask)
    n = int(num[0])
    assert torch.equal(out[0, :n, :16], vis[0, :n]) and torch.equal(out[0, :n, 16:], preds[:n])


def test_gradient_flows_to_the_emotion_features():
    g = torch.Generator().manual_seed(2)
    preds, vis, mask, num = _case(g, 2, 8, 8, peaky=True)
    preds = preds.clone().requires_grad_(True)
    out, m = select_frames(preds, vis, mask, num, 0.2)
    out[..., 16:].sum().backward()
    kept = int(m.sum())
    assert preds.grad is not None and int((preds.grad.abs().sum(1) > 0).sum()) == kept
