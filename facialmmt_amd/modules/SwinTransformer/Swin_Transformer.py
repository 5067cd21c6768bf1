"""Swin-tiny facial-sequence encoder behind the reference's nn.Module API, computed by libfmmt_hip.

Host-side mirror of the reference's modules/SwinTransformer/Swin_Transformer.py: same class names,
constructor signatures, sub-module attribute names and state_dict keys (a reference state_dict loads
with strict=True), same AssertionErrors for shape violations.  The arithmetic is not here: every
forward is a sequence of calls into the C ABI (facialmmt_amd/ops.py -> include/fmmt.h).

What differs by design (results identical, see oracle/ and tests/):
  * torch.roll / window_partition / window_reverse (ref :33-62,244,261) do not exist here (the two module-level helper
    functions of the reference have no counterpart in this file: nothing would call them): the attention
    kernel addresses shifted windows directly in token order;
  * residual adds and DropPath scaling are epilogues of the proj / fc2 GEMMs; GELU is fc1's epilogue;
  * PatchMerging's 2x2 gather + concat (ref :316-323) is folded into its LayerNorm kernel;
  * PatchEmbed's 4x4/stride-4 convolution (ref :407,419) is an im2col gather + the same GEMM kernel.
Unsupported-by-kernel configurations (window_size != 7, head_dim != 32, dropout p > 0 inside the
fused ops, act_layer != GELU, ape=True) raise instead of silently taking another path.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ... import ops


def to_2tuple(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


def _require(cond: bool, msg: str):
    if not cond:
        raise NotImplementedError("facialmmt_amd HIP path: " + msg)


DROP_PATH_ONE_DRAW = True    # module constant (probes patch it): every stochastic-depth multiplier of a forward from one draw (SwinTransformer._draw_drop_paths)


class DropPath(nn.Module):
    """Stochastic depth, per sample (timm semantics: keep mask / keep_prob).  On the HIP path the
    multiplier is handed to the GEMM epilogue as a per-sample vector; `sample_scale` draws it."""

    def __init__(self, drop_prob: float = 0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)
        self._rows = None           # multipliers drawn ahead for this forward by the model (SwinTransformer._draw_drop_paths): a list, consumed in call order

    def sample_scale(self, n: int, device):
        if self.drop_prob == 0.0 or not self.training:
            return None
        if self._rows:
            row = self._rows.pop(0)
            if row.shape[0] == n and row.device == torch.device(device):
                return row
        keep = 1.0 - self.drop_prob
        return torch.empty(n, dtype=torch.float32, device=device).bernoulli_(keep).div_(keep)

    def forward(self, x):
        s = self.sample_scale(x.shape[0], x.device)
        return x if s is None else x * s.to(x.dtype).view(-1, *([1] * (x.dim() - 1)))

    def extra_repr(self):
        return f"drop_prob={self.drop_prob}"


class Flatten(nn.Module):
    def forward(self, input):
        return input.view(input.size(0), -1)


class Mlp(nn.Module):
    """fc1 -> GELU -> fc2 (ref :14-30) as two GEMM launches with fused epilogues."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        _require(act_layer is nn.GELU, "Mlp activation must be nn.GELU (erf form)")
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x, res=None, rowscale=None, rows_per_scale=1):
        _require(self.drop.p == 0.0 or not self.training, "Mlp dropout p > 0 in training")
        return ops.mlp(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, res, rowscale, rows_per_scale)


class WindowAttention(nn.Module):
    """W-MSA / SW-MSA with relative position bias (ref :65-144)."""

    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.dim = dim
        self.window_size = window_size
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        ws_h, ws_w = window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws_h - 1) * (2 * ws_w - 1), num_heads))
        # pair-wise relative position index: (di + ws-1) * (2ws-1) + (dj + ws-1)
        t = torch.arange(ws_h * ws_w)
        ti, tj = t // ws_w, t % ws_w
        rel = (ti[:, None] - ti[None, :] + ws_h - 1) * (2 * ws_w - 1) + (tj[:, None] - tj[None, :] + ws_w - 1)
        self.register_buffer("relative_position_index", rel)
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)
        self.softmax = nn.Softmax(dim=-1)
        self._index_i32 = None

    def _index(self, device):
        if self._index_i32 is None or self._index_i32.device != device:
            self._index_i32 = self.relative_position_index.to(device=device, dtype=torch.int32).contiguous()
        return self._index_i32

    def _check(self):
        _require(tuple(self.window_size) == (7, 7), "window_size must be 7 (kernel constant)")
        _require(self.dim // self.num_heads == 32, "head_dim must be 32")
        _require(not self.training or (self.attn_drop.p == 0.0 and self.proj_drop.p == 0.0),
                 "attention / projection dropout p > 0 in training")

    def forward_tokens(self, x, n_img, H, W, shift, mask, res=None, rowscale=None, mask_is_shift=False):
        """x: (n_img, H*W, C) in token order -> same shape; `res + rowscale * attn(x)` if res is given.
        mask_is_shift: the caller guarantees `mask` is the standard SW-MSA mask of (H, W, shift)."""
        self._check()
        qkv = ops.linear(x, self.qkv.weight, self.qkv.bias)
        o = ops.window_attn_core(qkv.view(-1, 3 * self.dim), self.relative_position_bias_table, self._index(x.device),
                                 mask, n_img, H, W, self.num_heads, shift, float(self.scale), mask_is_shift)
        return ops.linear(o.view(n_img, H * W, self.dim), self.proj.weight, self.proj.bias, res, rowscale, H * W)

    def forward(self, x, mask=None):
        """x: (num_windows*B, 49, C) already partitioned; mask (nW,49,49) or None (ref :113-144)."""
        B_, N, C = x.shape
        assert N == self.window_size[0] * self.window_size[1] and C == self.dim, "input feature has wrong size"
        return self.forward_tokens(x, B_, self.window_size[0], self.window_size[1], 0, mask)

    def extra_repr(self) -> str:
        return f'dim={self.dim}, window_size={self.window_size}, num_heads={self.num_heads}'

    def flops(self, N):
        return N * self.dim * 3 * self.dim + 2 * self.num_heads * N * (self.dim // self.num_heads) * N + N * self.dim * self.dim


def build_shift_mask(H, W, window_size, shift_size):
    """(nW, ws*ws, ws*ws) of {0,-100} for SW-MSA (ref :208-227): region ids on shifted coordinates."""
    def region(n):
        r = torch.zeros(n, dtype=torch.long)
        r[n - window_size:n - shift_size] = 1
        r[n - shift_size:] = 2
        return r
    rid = region(H)[:, None] * 3 + region(W)[None, :]
    ids = rid.view(H // window_size, window_size, W // window_size, window_size).permute(0, 2, 1, 3).reshape(-1, window_size * window_size)
    same = ids[:, None, :] == ids[:, :, None]
    return torch.where(same, torch.tensor(0.0), torch.tensor(-100.0))


class SwinTransformerBlock(nn.Module):
    """x + DropPath(W-MSA(LN(x))) ; x + DropPath(Mlp(LN(x)))  (ref :163-270)."""

    def __init__(self, dim, input_resolution, num_heads, window_size=7, shift_size=0,
                 mlp_ratio=4., qkv_bias=True, qk_scale=None, drop=0., attn_drop=0., drop_path=0.,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim = dim
        self.input_resolution = input_resolution
        self.num_heads = num_heads
        self.window_size = window_size
        self.shift_size = shift_size
        self.mlp_ratio = mlp_ratio
        if min(self.input_resolution) <= self.window_size:
            self.shift_size = 0
            self.window_size = min(self.input_resolution)
        assert 0 <= self.shift_size < self.window_size, "shift_size must in 0-window_size"
        _require(norm_layer is nn.LayerNorm, "norm_layer must be nn.LayerNorm")
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, window_size=to_2tuple(self.window_size), num_heads=num_heads,
                                    qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        attn_mask = None
        if self.shift_size > 0:
            H, W = self.input_resolution
            attn_mask = build_shift_mask(H, W, self.window_size, self.shift_size)
        self.register_buffer("attn_mask", attn_mask)
        self._mask_checked = None      # (data_ptr, version) of the attn_mask buffer last verified to be the standard mask

    def _mask_is_standard(self) -> bool:
        """True iff the attn_mask buffer (possibly overwritten by load_state_dict) equals the mask the
        constructor builds; verified once per buffer version, so the kernel may derive it arithmetically."""
        m = self.attn_mask
        if m is None:
            return False
        key = (m.data_ptr(), m._version)
        if self._mask_checked is None or self._mask_checked[0] != key:
            H, W = self.input_resolution
            ok = bool(torch.equal(m.detach().float().cpu(), build_shift_mask(H, W, self.window_size, self.shift_size)))
            self._mask_checked = (key, ok)
        return self._mask_checked[1]

    def _scale(self, n, device):
        return self.drop_path.sample_scale(n, device) if isinstance(self.drop_path, DropPath) else None

    def forward(self, x):
        H, W = self.input_resolution
        B, L, C = x.shape
        assert L == H * W, "input feature has wrong size"
        a = self.attn
        std = self._mask_is_standard()
        if ops.window_block_fusable(x, C, a.num_heads, a.window_size, self.shift_size, self.attn_mask, std):
            # stages 0 / 1: norm1 -> qkv -> (S)W-MSA -> proj -> residual + DropPath as one op (stage 0: ONE launch, csrc/wblock.hip), whose
            # backward re-forms q / k / v inside the attention-backward kernel instead of keeping or recomputing a qkv tensor
            a._check()
            x = ops.window_block(x, self.norm1.weight, self.norm1.bias, self.norm1.eps, a.qkv.weight, a.qkv.bias, a.proj.weight, a.proj.bias,
                                 a.relative_position_bias_table, a._index(x.device), self.attn_mask, B, H, W, a.num_heads, self.shift_size,
                                 float(a.scale), self._scale(B, x.device))
        else:
            x, xn = ops.residual_layer_norm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
            x = a.forward_tokens(xn, B, H, W, self.shift_size, self.attn_mask, res=x, rowscale=self._scale(B, x.device), mask_is_shift=std)
        mlp = self.mlp
        if ops.mlp_ln_fusable(x, mlp.fc1.weight, mlp.fc2.weight, mlp.fc1.bias, mlp.fc2.bias) and (mlp.drop.p == 0.0 or not self.training):
            # stages 0 / 1: norm2 -> fc1 -> GELU -> fc2 -> DropPath -> residual as ONE launch (csrc/mlp_fused.hip, LayerNorm on the fragments)
            return ops.mlp_ln(x, self.norm2.weight, self.norm2.bias, self.norm2.eps, mlp.fc1.weight, mlp.fc1.bias, mlp.fc2.weight, mlp.fc2.bias,
                              self._scale(B, x.device), L)
        x, xn = ops.residual_layer_norm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        return mlp(xn, res=x, rowscale=self._scale(B, x.device), rows_per_scale=L)

    def extra_repr(self) -> str:
        return (f"dim={self.dim}, input_resolution={self.input_resolution}, num_heads={self.num_heads}, "
                f"window_size={self.window_size}, shift_size={self.shift_size}, mlp_ratio={self.mlp_ratio}")

    def flops(self):
        H, W = self.input_resolution
        nW = H * W / self.window_size / self.window_size
        return 2 * self.dim * H * W + nW * self.attn.flops(self.window_size * self.window_size) + 2 * H * W * self.dim * self.dim * self.mlp_ratio


class PatchMerging(nn.Module):
    """2x2 neighbour concat -> LayerNorm(4C) -> Linear(4C, 2C, bias=False)  (ref :291-328)."""

    def __init__(self, input_resolution, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.input_resolution = input_resolution
        self.dim = dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = norm_layer(4 * dim)

    def forward(self, x):
        H, W = self.input_resolution
        B, L, C = x.shape
        assert L == H * W, "input feature has wrong size"
        assert H % 2 == 0 and W % 2 == 0, f"x size ({H}*{W}) are not even."
        _require(H == W, "square token grid")
        xn = ops.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps, merge_hw=H)      # (B, L/4, 4C)
        return ops.linear(xn, self.reduction.weight)

    def extra_repr(self) -> str:
        return f"input_resolution={self.input_resolution}, dim={self.dim}"

    def flops(self):
        H, W = self.input_resolution
        return H * W * self.dim + (H // 2) * (W // 2) * 4 * self.dim * 2 * self.dim


class BasicLayer(nn.Module):
    """One stage: `depth` blocks (shift 0, ws//2 alternating) + optional PatchMerging (ref :340-389)."""

    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio=4., qkv_bias=True,
                 qk_scale=None, drop=0., attn_drop=0., drop_path=0., norm_layer=nn.LayerNorm, downsample=None,
                 use_checkpoint=False):
        super().__init__()
        self.dim = dim
        self.input_resolution = input_resolution
        self.depth = depth
        self.use_checkpoint = use_checkpoint
        _require(not use_checkpoint, "activation checkpointing (288 GB of HBM: activations are kept)")
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim=dim, input_resolution=input_resolution, num_heads=num_heads,
                                 window_size=window_size, shift_size=0 if (i % 2 == 0) else window_size // 2,
                                 mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop,
                                 attn_drop=attn_drop,
                                 drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path,
                                 norm_layer=norm_layer)
            for i in range(depth)])
        self.downsample = downsample(input_resolution, dim=dim, norm_layer=norm_layer) if downsample is not None else None

    def forward(self, x):
        for blk in self.blocks:
            x = blk(x)
        if self.downsample is not None:
            x = self.downsample(x)
        return x

    def extra_repr(self) -> str:
        return f"dim={self.dim}, input_resolution={self.input_resolution}, depth={self.depth}"

    def flops(self):
        return sum(b.flops() for b in self.blocks) + (self.downsample.flops() if self.downsample is not None else 0)


class PatchEmbed(nn.Module):
    """4x4 non-overlapping patches -> Linear(48, 96) -> LayerNorm  (ref :392-422).  `proj` stays an
    nn.Conv2d so that the state_dict key/shape (96,3,4,4) is the reference's."""

    def __init__(self, img_size=224, patch_size=4, in_chans=3, embed_dim=96, norm_layer=None):
        super().__init__()
        img_size = to_2tuple(img_size)
        patch_size = to_2tuple(patch_size)
        self.img_size = img_size
        self.patch_size = patch_size
        self.patches_resolution = [img_size[0] // patch_size[0], img_size[1] // patch_size[1]]
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        self.in_chans = in_chans
        self.embed_dim = embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None

    def forward_u8(self, x, resize="pil", dtype=None):
        """x: (B, S, S, 3) uint8 face crops in image layout.  The reference's host-side pre-step -- bicubic resize to
        224x224, ToTensor, Normalize(.5,.5) (utils/util.py:43-52 'pil' / utils/dataset.py:47-69 'cv2') -- is fused into the
        patch gather (ops.patch_embed_u8): same result as forward(Normalize(ToTensor(resize(x)))), without the float image."""
        _require(tuple(self.img_size) == (224, 224) and tuple(self.patch_size) == (4, 4) and self.in_chans == 3,
                 "PatchEmbed kernel is built for 3x224x224 images and 4x4 patches")
        B = x.shape[0]
        w2d = self.proj.weight.view(self.embed_dim, -1)
        dt = dtype or self.proj.weight.dtype
        if ops.patch_embed_u8_ln_fusable(x, w2d, self.norm, dt):   # pre-step, projection, bias and LayerNorm: one launch
            return ops.patch_embed_u8_ln(x, resize, w2d, self.proj.bias, self.norm.weight, self.norm.bias, self.norm.eps, dt).view(B, self.num_patches, self.embed_dim)
        cols = ops.patch_embed_u8(x, resize, dt)
        if ops.patch_proj_ln_fusable(cols, w2d, self.norm):    # projection + bias + LayerNorm: one launch
            return ops.patch_proj_ln(cols, w2d, self.proj.bias, self.norm.weight, self.norm.bias, self.norm.eps).view(B, self.num_patches, self.embed_dim)
        y = ops.linear(cols, w2d, self.proj.bias).view(B, self.num_patches, self.embed_dim)
        if self.norm is not None:
            y = ops.layer_norm(y, self.norm.weight, self.norm.bias, self.norm.eps)
        return y

    def forward(self, x):
        B, C, H, W = x.shape
        assert H == self.img_size[0] and W == self.img_size[1], \
            f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."
        _require(tuple(self.img_size) == (224, 224) and tuple(self.patch_size) == (4, 4) and C == 3,
                 "PatchEmbed kernel is built for 3x224x224 images and 4x4 patches")
        cols = ops.patch_im2col(x)                                                   # (B*3136, 48)
        w2d = self.proj.weight.view(self.embed_dim, -1)
        if ops.patch_proj_ln_fusable(cols, w2d, self.norm):
            return ops.patch_proj_ln(cols, w2d, self.proj.bias, self.norm.weight, self.norm.bias, self.norm.eps).view(B, self.num_patches, self.embed_dim)
        y = ops.linear(cols, w2d, self.proj.bias).view(B, self.num_patches, self.embed_dim)
        if self.norm is not None:
            y = ops.layer_norm(y, self.norm.weight, self.norm.bias, self.norm.eps)
        return y

    def flops(self):
        Ho, Wo = self.patches_resolution
        f = Ho * Wo * self.embed_dim * self.in_chans * (self.patch_size[0] * self.patch_size[1])
        return f + (Ho * Wo * self.embed_dim if self.norm is not None else 0)


class SwinTransformer(nn.Module):
    """Backbone + FaceX-Zoo style embedding head LN -> flatten -> Linear(49*768,512) -> BatchNorm1d (ref :434-541)."""

    def __init__(self, img_size=224, patch_size=4, in_chans=3, embed_dim=96, depths=[2, 2, 6, 2],
                 num_heads=[3, 6, 12, 24], window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0.1, norm_layer=nn.LayerNorm, ape=False,
                 patch_norm=True, use_checkpoint=False, **kwargs):
        super().__init__()
        self.num_layers = len(depths)
        self.embed_dim = embed_dim
        self.ape = ape
        self.patch_norm = patch_norm
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))
        self.mlp_ratio = mlp_ratio
        _require(not ape, "absolute position embedding (ape=True)")
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                                      norm_layer=norm_layer if self.patch_norm else None)
        self.patches_resolution = self.patch_embed.patches_resolution
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(BasicLayer(
                dim=int(embed_dim * 2 ** i),
                input_resolution=(self.patches_resolution[0] // (2 ** i), self.patches_resolution[1] // (2 ** i)),
                depth=depths[i], num_heads=num_heads[i], window_size=window_size, mlp_ratio=self.mlp_ratio,
                qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate, attn_drop=attn_drop_rate,
                drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])], norm_layer=norm_layer,
                downsample=PatchMerging if (i < self.num_layers - 1) else None, use_checkpoint=use_checkpoint))
        self.output_layer = nn.Sequential(norm_layer(self.num_features), Flatten(), nn.Linear(49 * 768, 512), nn.BatchNorm1d(512))
        self.apply(self._init_weights)
        # keep probabilities of the DropPath draws of one forward, in call order (a block draws twice: attention branch, Mlp branch); not part of the state_dict
        keep = [1.0 - blk.drop_path.drop_prob for layer in self.layers for blk in layer.blocks
                if isinstance(blk.drop_path, DropPath) and blk.drop_path.drop_prob > 0.0 for _ in range(2)]
        self.register_buffer("_dp_keep", torch.tensor(keep, dtype=torch.float32).view(-1, 1), persistent=False)
        self.register_buffer("_dp_inv", (1.0 / torch.tensor(keep, dtype=torch.float64)).float().view(-1, 1), persistent=False)

    def _draw_drop_paths(self, n, device):
        """Every stochastic-depth multiplier of this forward in ONE draw: rand, compare, select over a (draws, n) matrix whose rows the DropPath modules then
        hand out in call order -- instead of bernoulli_ + div_ per draw (44 launches of ~5 us in front of the blocks of Swin-tiny's forward, which runs alone on
        the GPU).  Same distribution: row r is 1 / keep_r with probability keep_r, else 0."""
        dps = [blk.drop_path for layer in self.layers for blk in layer.blocks if isinstance(blk.drop_path, DropPath) and blk.drop_path.drop_prob > 0.0]
        for dp in dps:
            dp._rows = None
        if not DROP_PATH_ONE_DRAW or not self.training or not dps or self._dp_keep.numel() != 2 * len(dps) or self._dp_keep.device != torch.device(device):
            return
        keep = self._dp_keep
        pool = torch.where(torch.rand(keep.shape[0], n, dtype=torch.float32, device=device) < keep, self._dp_inv, 0.0)          # three launches
        for i, dp in enumerate(dps):
            dp._rows = [pool[2 * i], pool[2 * i + 1]]

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'absolute_pos_embed'}

    @torch.jit.ignore
    def no_weight_decay_keywords(self):
        return {'relative_position_bias_table'}

    def _head(self, x):
        ln, _, fc, bn = self.output_layer
        x = ops.layer_norm(x, ln.weight, ln.bias, ln.eps)
        x = ops.linear(x.reshape(x.shape[0], -1), fc.weight, fc.bias)
        use_batch = self.training or not bn.track_running_stats
        if self.training and bn.track_running_stats:
            bn.num_batches_tracked += 1
        return ops.batch_norm_1d(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, use_batch)

    # uint8 input (B, S, S, 3): which library's bicubic the fused pre-step reproduces ('pil': Aff-Wild2 path, pinned
    # bit-exactly; 'cv2': MELD path, unpinned) and the activation dtype it produces (None: the parameters' dtype)
    input_resize = "pil"
    input_dtype = None

    def forward_features(self, x):
        _require(self.pos_drop.p == 0.0 or not self.training, "pos_drop p > 0 in training")
        if x.dtype == torch.uint8:
            x = self.patch_embed.forward_u8(x, self.input_resize, self.input_dtype)
        else:
            x = self.patch_embed(x)
        self._draw_drop_paths(x.shape[0], x.device)
        for layer in self.layers:
            x = layer(x)
        return self._head(x)

    def forward(self, x):
        if len(x) == 1:                              # BatchNorm needs two samples (ref :535-538)
            return self.forward_features(torch.cat((x, x), dim=0))[:1]
        return self.forward_features(x)
