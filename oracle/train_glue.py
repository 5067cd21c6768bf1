"""ORACLE (test infrastructure).  Literal restatement, loops and all, of the target-task glue between the
Swin model and the multimodal model: the frame-importance filter and emotion-concat of
train.py:75-114,122-131.  The reference's closure cannot be imported (it needs pytorch_lightning), so
parity for this glue is UNPINNED by reference outputs; this restatement follows the
source line by line and pins the build's vectorised device version (facialmmt_amd/train_step.py).

Reference quirk kept on purpose: when several utterances share a batch, the utterance boundary advances
by (num_imgs - 1) instead of num_imgs (train.py:96,109), so utterance u owns the selected faces with
global index in [sum_{i<u} n_i - u, sum_{i<=u} n_i - u) and reads vision feature row (global - sum_{i<u}(n_i - 1)).
With the reference's own batch size (1 dialogue, trg_batch_size=1, main.py:56) the quirk is invisible."""
from __future__ import annotations

import torch


def select_frames_loop(preds, vision_inputs, vision_mask, num_imgs, threshold, num_labels=7):
    """preds (sum F, 7) Gumbel-softmax outputs; vision_inputs (B, Lv, D); vision_mask (B, Lv); num_imgs list.
    Returns (vision_inputs_concat (B, Lv, D+7), new_vision_mask (B, Lv))."""
    B, Lv, D = vision_inputs.shape
    importance = torch.diagonal(preds @ preds.t())
    high = torch.nonzero(importance.gt(threshold)).squeeze(1)
    emo = torch.zeros(B, Lv, num_labels, dtype=preds.dtype)
    if len(high) > 0:
        tmp = high
        new_mask = torch.zeros_like(vision_mask)
        margin = 0
        for u in range(B):
            k = 0
            for j in range(len(tmp)):
                if tmp[j] < num_imgs[u] + margin:
                    new_mask[u][k] = 1
                    k += 1
                else:
                    break
            margin = margin + num_imgs[u] - 1
            tmp = tmp[k:]
        new_inputs = torch.zeros_like(vision_inputs)
        jj = 0
        margin = 0
        for u in range(B):
            for f in range(Lv):
                if new_mask[u][f] != 0:
                    emo[u][f] = preds[high[jj]]
                    new_inputs[u][f] = vision_inputs[u][high[jj] - margin]
                    jj += 1
                else:
                    break
            margin = margin + num_imgs[u] - 1
        return torch.cat((new_inputs, emo), dim=-1), new_mask
    jj = 0
    for i in range(B):
        for j in range(Lv):
            if vision_mask[i][j] == 1:
                emo[i][j] = preds[jj]
                jj += 1
            else:
                break
    return torch.cat((vision_inputs, emo), dim=-1), vision_mask
