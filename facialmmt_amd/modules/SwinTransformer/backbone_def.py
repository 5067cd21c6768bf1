"""YAML -> backbone factory; mirrors the reference's modules/SwinTransformer/backbone_def.py:8-53
(class name, constructor arguments, YAML schema, fixed qkv_bias/ape/patch_norm/use_checkpoint)."""
from __future__ import annotations

import yaml

from .Swin_Transformer import SwinTransformer


class BackboneFactory:
    def __init__(self, backbone_type: str, backbone_conf_file: str):
        self.backbone_type = backbone_type
        with open(backbone_conf_file) as f:
            self.backbone_param = yaml.load(f, Loader=yaml.FullLoader)[backbone_type]

    def get_backbone(self):
        if self.backbone_type != "SwinTransformer":
            raise ValueError(f"unknown backbone type {self.backbone_type!r} (only 'SwinTransformer' is built)")
        c = self.backbone_param
        return SwinTransformer(img_size=c["img_size"], patch_size=c["patch_size"], in_chans=c["in_chans"],
                               embed_dim=c["embed_dim"], depths=c["depths"], num_heads=c["num_heads"],
                               window_size=c["window_size"], mlp_ratio=c["mlp_ratio"], qkv_bias=True, qk_scale=None,
                               drop_rate=c["drop_rate"], drop_path_rate=c["drop_path_rate"], ape=False,
                               patch_norm=True, use_checkpoint=False)
