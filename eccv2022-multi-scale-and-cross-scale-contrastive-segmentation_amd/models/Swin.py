"""Swin Transformer backbone (T / S / B / L) -- drop-in for the reference (models/Swin.py:21-743):
same constructor keywords (``backbone_config[name]``), same module tree and therefore the same
state_dict keys (``patch_embed.proj``, ``layers.{i}.blocks.{j}.{norm1,attn.{qkv,proj,
relative_position_bias_table,relative_position_index},norm2,mlp.{fc1,fc2}}``,
``layers.{i}.downsample.{reduction,norm}``, ``norm{i}``), NCHW feature maps out.

Differences in HOW (not what): window attention goes through ``scaled_dot_product_attention`` with
the relative-position bias (+ the shifted-window mask) as an additive mask, so QK^T, bias, mask,
softmax and PV are one fused kernel on ROCm instead of five ops; the shifted-window mask is cached
per padded resolution instead of being rebuilt on the host every forward (reference :448-466).

Third-party arithmetic: the reference takes ``DropPath`` / ``trunc_normal_`` / ``to_2tuple`` from
``timm`` (unpinned in env_dgx.yml).  DropPath here follows timm's published semantics (per-sample
Bernoulli keep mask scaled by 1/keep_prob, identity in eval); stochastic-depth parity with the
reference is UNPINNED (no reference test pins it) -- model goldens run in eval mode.
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..utils import printlog
from .amax import carry, tag, tag_of
from .ops import FusedLayerNorm, TokenLinear, tagged_gelu

_COMMON = dict(window_size=7, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0,
               drop_path_rate=0.3, ape=False, patch_norm=True, out_indices=[0, 1, 2, 3], pretrained=True)


def _variant(embed_dim, depths, heads, name):
    return dict(_COMMON, embed_dim=embed_dim, depths=depths, num_heads=heads,
                out_channels=[embed_dim * 2 ** i for i in range(4)], name=name)


backbone_config = {
    "swinT": _variant(96, [2, 2, 6, 2], [3, 6, 12, 24], "swin_tiny"),
    "swinS": _variant(96, [2, 2, 18, 2], [3, 6, 12, 24], "swin_small"),
    "swinB": _variant(128, [2, 2, 18, 2], [4, 8, 16, 32], "swin_base"),
    "swinL": _variant(192, [2, 2, 18, 2], [6, 12, 24, 48], "swin_large"),
}


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


class TokenMap:
    """A backbone output kept TOKEN-MAJOR: ``tokens`` [B, H * W, C] -- what the stage's output LayerNorm wrote -- standing for the
    [B, C, H, W] map the reference returns (models/Swin.py:452-455: view, permute, contiguous).  A consumer that is a 1x1
    convolution reads the tokens as they are (a GEMM does not care which operand index is contiguous; models/ops.py
    conv1x1_from_tokens) and its data gradient comes back token-major, which is what the LayerNorm's backward wants: the layout
    copies of both directions disappear.  Anybody else calls ``nchw()`` and gets the reference's tensor (made once).  Only handed
    out when the owner of the backbone asks for it (SwinTransformer.token_outputs, set by models/UPerNet.py for its own decoder)."""

    def __init__(self, tokens, H, W):
        self.tokens, self.H, self.W = tokens, int(H), int(W)
        self._nchw = None

    @property
    def shape(self):
        b, _, c = self.tokens.shape
        return torch.Size((b, c, self.H, self.W))

    def nchw(self):
        if self._nchw is None:
            t = self.tokens
            self._nchw = carry(t, t.view(-1, self.H, self.W, t.shape[-1]).permute(0, 3, 1, 2).contiguous())
        return self._nchw


def as_nchw(x):
    """A backbone output as the reference's NCHW tensor (TokenMap or tensor)."""
    return x.nchw() if isinstance(x, TokenMap) else x


class DropPath(nn.Module):
    """Stochastic depth per sample (timm semantics)."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return x * mask / keep

    def add_to(self, shortcut, x):
        """``shortcut + self(x)`` in ONE pass (same mask draw): the per-sample factor mask / keep is a [B, 1, ...] tensor, so the
        residual sum is a single addcmul forward and a single scaling backward instead of mul, div, add (+ two multiplies
        backward) over the whole activation -- 35 ms of a Swin-L step were element-wise ATen kernels."""
        if self.drop_prob == 0.0 or not self.training:
            return shortcut + x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return _ScaledResidual.apply(shortcut, x, mask / keep, 1.0 / keep)


    def factors(self, x):
        """(scale, bound) for a fused ``shortcut + scale * branch``: the per-sample factors mask / keep as a [B] tensor (the same
        draw as ``forward`` / ``add_to``) and 1 / keep; (None, 1.0) when the module is inactive."""
        if self.drop_prob == 0.0 or not self.training:
            return None, 1.0
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return (mask / keep).reshape(-1), 1.0 / keep


class _ScaledResidual(torch.autograd.Function):
    """shortcut + x * scale (scale [B, 1, ...], no gradient); backward: the incoming gradient goes to the shortcut as it
    is, scaled to x -- with its absmax tag (|g scale| <= |g| / keep), so that the Linear behind x finds its operand scale
    without a pass over the gradient."""

    @staticmethod
    def forward(ctx, shortcut, x, scale, bound):
        ctx.save_for_backward(scale)
        ctx.bound = float(bound)
        return torch.addcmul(shortcut, x, scale)

    @staticmethod
    def backward(ctx, g):
        (scale,) = ctx.saved_tensors
        gx = g * scale
        if g.is_cuda:
            t = tag_of(g)
            if t is not None:
                tag(gx, t * ctx.bound)
        return g, gx, None, None


def _residual(shortcut, y, drop_path):
    """shortcut + drop_path(y) (reference models/Swin.py:318, :321), fused when drop_path is this module's DropPath."""
    return drop_path.add_to(shortcut, y) if isinstance(drop_path, DropPath) else shortcut + drop_path(y)


def _factors(drop_path, x):
    """(fusable, scale, bound): the residual sum can ride in a GEMM epilogue when drop_path is Identity or this module's DropPath."""
    if isinstance(drop_path, DropPath):
        if x.dim() < 3 or x.numel() // (x.shape[0] * x.shape[-1]) < 32:     # (the epilogue wants >= 32 rows per sample; checked
            return False, None, 1.0                                          # BEFORE the mask is drawn: one draw per residual sum)
        return (True,) + drop_path.factors(x)
    return isinstance(drop_path, nn.Identity), None, 1.0


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        self.fc1 = TokenLinear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = TokenLinear(hidden_features or in_features, out_features or in_features)
        self.drop = nn.Dropout(drop)

    def _fusable(self, x):
        from .ops import fused_mlp_ok
        return (type(self.act) is nn.GELU and self.act.approximate == 'none' and (self.drop.p == 0.0 or not self.training)
                and fused_mlp_ok(x, self.fc1, self.fc2))

    def forward(self, x):
        if self._fusable(x):
            from .ops import fused_mlp
            return fused_mlp(x, self.fc1, self.fc2)         # GELU and its derivative inside the GEMM epilogues (ops._FusedMlp)
        h = self.fc1(x)
        # (the absmax tag of an f16x3 fc1's output also bounds fc2's input, and the tag of fc2's data gradient the
        # gradient GELU hands back to fc1)
        a = tagged_gelu(h) if type(self.act) is nn.GELU and self.act.approximate == 'none' else self.act(h)
        return self.drop(self.fc2(self.drop(a)))

    def add_to(self, shortcut, x, drop_path):
        """shortcut + drop_path(self(x)) (reference models/Swin.py:321); on the fused path the sum is fc2's epilogue."""
        if self._fusable(x):
            ok, scale, bound = _factors(drop_path, x)
            if ok:
                from .ops import fused_mlp
                return fused_mlp(x, self.fc1, self.fc2, shortcut=shortcut, scale=scale, bound=bound)
        return _residual(shortcut, self(x), drop_path)


def window_partition(x, ws):
    """(B, H, W, C) -> (B * nW, ws, ws, C)"""
    B, H, W, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, C)


def window_reverse(windows, ws, H, W):
    """(B * nW, ws, ws, C) -> (B, H, W, C)"""
    B = windows.shape[0] // ((H // ws) * (W // ws))
    x = windows.view(B, H // ws, W // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, -1)


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        wh, ww = window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * wh - 1) * (2 * ww - 1), num_heads))
        coords = torch.stack(torch.meshgrid(torch.arange(wh), torch.arange(ww), indexing='ij')).flatten(1)
        rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += wh - 1
        rel[:, :, 1] += ww - 1
        rel[:, :, 0] *= 2 * ww - 1
        self.register_buffer("relative_position_index", rel.sum(-1))
        self.qkv = TokenLinear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = TokenLinear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)
        self.softmax = nn.Softmax(dim=-1)

    def forward(self, x, mask=None):
        """x: (B * nW, N, C); mask: (nW, N, N) additive or None."""
        B_, N, C = x.shape
        h = self.num_heads
        qkv = self.qkv(x).view(B_, N, 3, h, C // h).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        bias = self.relative_position_bias_table[self.relative_position_index.view(-1)].view(N, N, h)
        bias = bias.permute(2, 0, 1).unsqueeze(0)                        # 1, h, N, N
        if mask is not None:
            nW = mask.shape[0]
            bias = (bias + mask.view(nW, 1, N, N)).repeat(B_ // nW, 1, 1, 1)   # B_, h, N, N (window-major)
        out = F.scaled_dot_product_attention(q, k, v, attn_mask=bias.to(q.dtype),
                                             dropout_p=self.attn_drop.p if self.training else 0.0,
                                             scale=self.scale)
        return self.proj_drop(self.proj(out.transpose(1, 2).reshape(B_, N, C)))


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, num_heads, window_size=7, shift_size=0, mlp_ratio=4., qkv_bias=True, qk_scale=None,
                 drop=0., attn_drop=0., drop_path=0., act_layer=nn.GELU, norm_layer=FusedLayerNorm):
        super().__init__()
        assert 0 <= shift_size < window_size, "shift_size must in 0-window_size"
        self.dim, self.num_heads, self.window_size, self.shift_size = dim, num_heads, window_size, shift_size
        self.mlp_ratio = mlp_ratio
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, window_size=to_2tuple(window_size), num_heads=num_heads,
                                    qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.H = self.W = None

    hip_attention = True       # class switch (UPerNet graph key 'hip_attention'): False = the library path below

    def _hip_path(self, x):
        a = self.attn
        return (self.hip_attention and x.is_cuda and x.dtype == torch.float32 and self.window_size == 7
                and a.dim == 32 * a.num_heads and a.attn_drop.p == 0.0 and not torch.is_autocast_enabled()
                and a.qkv.weight.dtype == torch.float32)

    def _proj_residual(self, shortcut, att):
        """shortcut + drop_path(proj_drop(proj(att))) (reference models/Swin.py:318); the sum in the projection GEMM's epilogue when
        nothing sits between the two (no projection dropout)."""
        a = self.attn
        if a.proj_drop.p == 0.0 or not self.training:
            from .ops import linear_residual, linear_residual_ok
            if linear_residual_ok(att, a.proj):
                ok, scale, bound = _factors(self.drop_path, att)
                if ok:
                    return linear_residual(att, a.proj, shortcut, scale=scale, bound=bound)
        return _residual(shortcut, a.proj_drop(a.proj(att)), self.drop_path)

    def forward(self, x, mask_matrix):
        B, L, C = x.shape
        H, W, ws = self.H, self.W, self.window_size
        assert L == H * W, "input feature has wrong size"
        shortcut = x
        if self._hip_path(x):
            # The qkv projection is token-wise, so it commutes with pad / roll / window_partition: it runs on the tokens
            # in their natural order and the attention kernel gathers each window's 49 tokens itself (and writes the
            # result back in natural order): no pad, roll, partition, reverse, roll, crop copies (csrc/dcl_winattn.hip).
            from .ops import window_attention
            a = self.attn
            n1, shortcut = self.norm1.with_shortcut(x) if isinstance(self.norm1, FusedLayerNorm) else (self.norm1(x), x)
            qkv = a.qkv(n1)
            qb = a.qkv.bias if a.qkv.bias is not None else torch.zeros(3 * C, dtype=x.dtype, device=x.device)
            N = ws * ws
            bias = a.relative_position_bias_table[a.relative_position_index.view(-1)].view(N, N, a.num_heads)
            att = window_attention(qkv, qb, bias.permute(2, 0, 1), H, W, a.num_heads, self.shift_size, a.scale)
            qbuf = tag_of(qkv)
            if qbuf is not None:
                tag(att, qbuf)              # a softmax-weighted mean of v rows (or of the bias row): |att| <= max|qkv|
            x = self._proj_residual(shortcut, att)
            n2, x = self.norm2.with_shortcut(x) if isinstance(self.norm2, FusedLayerNorm) else (self.norm2(x), x)
            return self.mlp.add_to(x, n2, self.drop_path) if isinstance(self.mlp, Mlp) else _residual(x, self.mlp(n2), self.drop_path)
        x = self.norm1(x).view(B, H, W, C)
        pad_r, pad_b = (ws - W % ws) % ws, (ws - H % ws) % ws
        if pad_r or pad_b:
            x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))
        Hp, Wp = H + pad_b, W + pad_r
        if self.shift_size > 0:
            x = torch.roll(x, shifts=(-self.shift_size, -self.shift_size), dims=(1, 2))
        windows = window_partition(x, ws).view(-1, ws * ws, C)
        windows = self.attn(windows, mask=mask_matrix if self.shift_size > 0 else None)
        x = window_reverse(windows.view(-1, ws, ws, C), ws, Hp, Wp)
        if self.shift_size > 0:
            x = torch.roll(x, shifts=(self.shift_size, self.shift_size), dims=(1, 2))
        if pad_r or pad_b:
            x = x[:, :H, :W, :].contiguous()
        x = _residual(shortcut, x.view(B, H * W, C), self.drop_path)
        return _residual(x, self.mlp(self.norm2(x)), self.drop_path)


class PatchMerging(nn.Module):
    def __init__(self, dim, norm_layer=FusedLayerNorm):
        super().__init__()
        self.dim = dim
        self.reduction = TokenLinear(4 * dim, 2 * dim, bias=False)
        self.norm = norm_layer(4 * dim)

    def forward(self, x, H, W):
        B, L, C = x.shape
        assert L == H * W, "input feature has wrong size"
        x = x.view(B, H, W, C)
        if H % 2 or W % 2:
            x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
            x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
        else:
            # the same tensor as the concatenation of the four strided slices (channel group k = 2 dx + dy holds pixel
            # (2 i + dy, 2 j + dx)), as ONE permutation: one copy forward and one backward, where autograd's backward of the
            # four slices is four zero-fills + strided copies and three adds over the full map (1.8 ms of a Swin-L step)
            x = x.view(B, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 4, 2, 5).reshape(B, H // 2, W // 2, 4 * C)
        return self.reduction(self.norm(x.view(B, -1, 4 * C)))


class BasicLayer(nn.Module):
    def __init__(self, dim, depth, num_heads, window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop=0.,
                 attn_drop=0., drop_path=0., norm_layer=FusedLayerNorm, downsample=None, use_checkpoint=False):
        super().__init__()
        self.window_size, self.shift_size, self.depth = window_size, window_size // 2, depth
        self.use_checkpoint = use_checkpoint
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim=dim, num_heads=num_heads, window_size=window_size,
                                 shift_size=0 if i % 2 == 0 else window_size // 2, mlp_ratio=mlp_ratio,
                                 qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop, attn_drop=attn_drop,
                                 drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path,
                                 norm_layer=norm_layer) for i in range(depth)])
        self.downsample = downsample(dim=dim, norm_layer=norm_layer) if downsample is not None else None
        self._mask_cache = {}

    def _shift_mask(self, H, W, device):
        """Additive mask (0 / -100) separating the wrapped-around regions of shifted windows."""
        ws, ss = self.window_size, self.shift_size
        Hp, Wp = int(np.ceil(H / ws)) * ws, int(np.ceil(W / ws)) * ws
        key = (Hp, Wp, str(device))
        if key not in self._mask_cache:
            region = torch.zeros((1, Hp, Wp, 1))
            cnt = 0
            for hs in (slice(0, -ws), slice(-ws, -ss), slice(-ss, None)):
                for wsl in (slice(0, -ws), slice(-ws, -ss), slice(-ss, None)):
                    region[:, hs, wsl, :] = cnt
                    cnt += 1
            mw = window_partition(region, ws).reshape(-1, ws * ws)
            diff = mw.unsqueeze(1) - mw.unsqueeze(2)
            self._mask_cache[key] = torch.where(diff != 0, torch.tensor(-100.0), torch.tensor(0.0)).to(device)
        return self._mask_cache[key]

    def forward(self, x, H, W):
        attn_mask = self._shift_mask(H, W, x.device)
        for blk in self.blocks:
            blk.H, blk.W = H, W
            if self.use_checkpoint:
                from torch.utils import checkpoint
                x = checkpoint.checkpoint(blk, x, attn_mask)
            else:
                x = blk(x, attn_mask)
        if self.downsample is not None:
            return x, H, W, self.downsample(x, H, W), (H + 1) // 2, (W + 1) // 2
        return x, H, W, x, H, W


class PatchEmbed(nn.Module):
    def __init__(self, patch_size=4, in_chans=3, embed_dim=96, norm_layer=None):
        super().__init__()
        self.patch_size = to_2tuple(patch_size)
        self.in_chans, self.embed_dim = in_chans, embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None

    def forward(self, x):
        _, _, H, W = x.size()
        ph, pw = self.patch_size
        if W % pw:
            x = F.pad(x, (0, pw - W % pw))
        if H % ph:
            x = F.pad(x, (0, 0, 0, ph - H % ph))
        x = self.proj(x)
        if self.norm is not None:
            Wh, Ww = x.size(2), x.size(3)
            x = self.norm(x.flatten(2).transpose(1, 2)).transpose(1, 2).reshape(-1, self.embed_dim, Wh, Ww)
        return x


class SwinTransformer(nn.Module):
    def __init__(self, pretrain_img_size=224, patch_size=4, in_chans=3, embed_dim=96, depths=[2, 2, 6, 2],
                 num_heads=[3, 6, 12, 24], window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0.2, norm_layer=FusedLayerNorm, ape=False, patch_norm=True,
                 out_indices=(0, 1, 2, 3), frozen_stages=-1, use_checkpoint=False, pretrained=True, name='swinT',
                 **kwargs):
        super().__init__()
        self.name = name
        self.pretrain_img_size = pretrain_img_size
        self.num_layers = len(depths)
        self.embed_dim, self.ape, self.patch_norm = embed_dim, ape, patch_norm
        self.out_indices, self.frozen_stages = out_indices, frozen_stages
        self.patch_embed = PatchEmbed(patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                                      norm_layer=norm_layer if patch_norm else None)
        if ape:
            pis, ps = to_2tuple(pretrain_img_size), to_2tuple(patch_size)
            self.absolute_pos_embed = nn.Parameter(torch.zeros(1, embed_dim, pis[0] // ps[0], pis[1] // ps[1]))
            nn.init.trunc_normal_(self.absolute_pos_embed, std=.02)
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(BasicLayer(
                dim=int(embed_dim * 2 ** i), depth=depths[i], num_heads=num_heads[i], window_size=window_size,
                mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate, attn_drop=attn_drop_rate,
                drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])], norm_layer=norm_layer,
                downsample=PatchMerging if i < self.num_layers - 1 else None, use_checkpoint=use_checkpoint))
        self.num_features = [int(embed_dim * 2 ** i) for i in range(self.num_layers)]
        for i in out_indices:
            self.add_module(f'norm{i}', norm_layer(self.num_features[i]))
        self._freeze_stages()
        self.init_weights(pretrained)

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.patch_embed.eval()
            for p in self.patch_embed.parameters():
                p.requires_grad = False
        if self.frozen_stages >= 1 and self.ape:
            self.absolute_pos_embed.requires_grad = False
        if self.frozen_stages >= 2:
            self.pos_drop.eval()
            for i in range(0, self.frozen_stages - 1):
                self.layers[i].eval()
                for p in self.layers[i].parameters():
                    p.requires_grad = False

    def init_weights(self, pretrained=True):
        """Reference quirk kept (Swin.py:670-672): the custom init runs ONLY when pretrained weights
        are about to be loaded; with pretrained=False modules keep PyTorch's default init."""
        if not pretrained:
            return

        def _init(m):
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)
        self.apply(_init)
        self.load_pretrained()

    def load_pretrained(self, strict=False):
        path = os.path.join("pytorch_checkpoints", "swin_imagenet", f"{self.name}_patch4_window7_224.pth")
        if not os.path.isfile(path):
            raise ValueError(f'cannot find swin imagenet checkpoint {path}')
        state = torch.load(path, map_location='cpu')['model']
        state = self._interpolate_position_bias(state)
        missing = self.load_state_dict(state, strict=strict)
        printlog(f'loaded pretrained {path}; missing {len(missing.missing_keys)} keys')

    def _interpolate_position_bias(self, state_dict):
        own = self.state_dict()
        for key in [k for k in state_dict if "relative_position_bias_table" in k]:
            src, dst = state_dict[key], own[key]
            (L1, h1), (L2, h2) = src.size(), dst.size()
            if h1 == h2 and L1 != L2:
                s1, s2 = int(L1 ** 0.5), int(L2 ** 0.5)
                resized = F.interpolate(src.permute(1, 0).view(1, h1, s1, s1), size=(s2, s2), mode='bicubic')
                state_dict[key] = resized.view(h2, L2).permute(1, 0)
        return state_dict

    def forward(self, x):
        x = self.patch_embed(x)
        Wh, Ww = x.size(2), x.size(3)
        if self.ape:
            x = x + F.interpolate(self.absolute_pos_embed, size=(Wh, Ww), mode='bicubic')
        x = self.pos_drop(x.flatten(2).transpose(1, 2))
        outs = []
        for i, layer in enumerate(self.layers):
            x_out, H, W, x, Wh, Ww = layer(x, Wh, Ww)
            if i in self.out_indices:
                x_out = getattr(self, f'norm{i}')(x_out)
                tm = TokenMap(x_out, H, W)
                # token_outputs: every level but the last stays token-major (the decoder's lateral 1x1 convolutions read tokens;
                # the last level feeds pooling / 3x3 convolutions).  Otherwise the reference's NCHW tensors (same values in
                # another order: the norm's absmax tag travels with the copy the decoder convolves)
                keep = self.token_outputs and self.training and torch.is_grad_enabled() and x_out.is_cuda and i != max(self.out_indices)
                outs.append(tm if keep else tm.nchw())
        return tuple(outs)

    token_outputs = False      # set by the owner (models/UPerNet.py); see TokenMap

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        return self
