"""The argparse namespace the reference uses as model config (main.py:12-105 defaults + the fields
main.py:131-145 derives from the data), as a plain SimpleNamespace factory.  Only model-relevant fields."""
from __future__ import annotations

import os
from types import SimpleNamespace

_HERE = os.path.dirname(os.path.abspath(__file__))


def default_args(**overrides) -> SimpleNamespace:
    a = SimpleNamespace(
        num_labels=7, plm_name="roberta-large", choice_modality="T+A+V",
        pretrainedtextmodel_path="pretrained_model/roberta-large",
        backbone_type="SwinTransformer",
        backbone_conf_file=os.path.join(_HERE, "modules", "SwinTransformer", "swin_conf.yaml"),
        tau=1.0, FacialEmoImpor_threshold=0.2,
        aux_lr=5e-5, trg_lr=7e-6, weight_decay=0.01, warm_up=0.1, clip=0.8,
        aux_batch_size=150, trg_batch_size=1, aux_accumulation_steps=1, trg_accumulation_steps=4,
        crossmodal_layers_TA=2, crossmodal_num_heads_TA=12, crossmodal_attn_dropout_TA=0.1,
        crossmodal_layers_TA_V=2, crossmodal_num_heads_TA_V=12, crossmodal_attn_dropout_TA_V=0.1,
        audio_utt_Transformernum=5, vision_utt_Transformernum=2,
        hidden_size=768, num_attention_heads=12, intermediate_size=3072, hidden_act="gelu",
        hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, layer_norm_eps=1e-12, initializer_range=0.02,
        seed=1111,
        # derived from the data in the reference (main.py:131-145); synthetic MELD-shaped defaults
        get_text_utt_max_lens=38, get_audio_utt_max_lens=128, get_vision_utt_max_lens=160,
        audio_featExtr_dim=300, vision_featExtr_dim=512,
    )
    for k, v in overrides.items():
        setattr(a, k, v)
    return a
