"""HGNetv2 backbone, channel-last. Structure and parameter names after the reference's
src/d_fine/arch/hgnetv2.py (stem :115-166, HG_Block :189-275 with the default ``agg="se"``
aggregation, HG_Stage :278-329); the unused ESE / padding="same" branches are not built.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .configs import BACKBONES, ModelSpec
from . import ops
from .layers import ConvNorm


class Stem(nn.Module):
    def __init__(self, c_in, c_mid, c_out, kw):
        super().__init__()
        self.stem1 = ConvNorm(c_in, c_mid, 3, stride=2, **kw)
        self.stem2a = ConvNorm(c_mid, c_mid // 2, 2, **kw)
        self.stem2b = ConvNorm(c_mid // 2, c_mid, 2, **kw)
        self.stem2a.pad_br = self.stem2b.pad_br = 1  # F.pad(x, (0, 1, 0, 1)) folded into the conv's tap bounds
        self.stem3 = ConvNorm(c_mid * 2, c_mid, 3, stride=2, **kw)
        self.stem4 = ConvNorm(c_mid, c_out, 1, **kw)

    def forward(self, x):
        if x.shape[1] == 4 and self.stem1.conv.in_channels == 3 and not hasattr(self.stem1, "conv_bn_fused"):
            # the image arrives zero-padded to 4 channels (TrainStep does, on the bf16 path): 8-byte tap loads in the implicit
            # GEMM and its weight gradient instead of 2-byte ones. The PARAMETER keeps its [24, 3, 3, 3] shape: the padded
            # copy is a differentiable view of it (pad's backward slices the gradient back)
            w4 = F.pad(self.stem1.conv.weight, (0, 0, 0, 0, 0, 1)).contiguous(memory_format=torch.channels_last)
            y, ws, hs = self.stem1._conv(x, w4)
            u = self.stem1._norm_tail(y, ws, hs)
        else:
            u = self.stem1(x)
        c = u.shape[1]
        if ops.stem_pool_supported(u) and ops.concat_buffer_supported(u, [c, c]) and u.dtype == torch.bfloat16:
            # pooled map and conv branch are written straight into the two halves of one buffer (no torch.cat on the largest
            # maps of the model: [B, 48, 320, 320] at 640x640), and read their halves of stem3's input gradient in place
            B, _, H, W = u.shape
            buf = ops.ConcatBuffer(B, H, W, 2 * c, torch.bfloat16, u.device)
            pooled = ops.stem_pool(u, into=buf)  # HIP: max-pool of the zero-extended map, no padded copy
            branch = self.stem2b(self.stem2a(u), into=buf)  # 2x2 convs on the zero-extended map (no padded copy)
            return self.stem4(self.stem3(ops.concat_read(buf, [pooled, branch])))
        branch = self.stem2b(self.stem2a(u))
        if ops.stem_pool_supported(u):
            pooled = ops.stem_pool(u)
        else:
            pooled = F.max_pool2d(F.pad(u, (0, 1, 0, 1)), 2, stride=1, ceil_mode=True)
        return self.stem4(self.stem3(torch.cat([pooled, branch], 1)))


CHAIN_BLOCKS = os.environ.get("DFINE_CHAIN_BLOCKS", "1") != "0"  # measurement switches
FUSE_RESIDUAL = os.environ.get("DFINE_FUSE_RESIDUAL", "1") != "0"


class LightConv(nn.Module):
    """1x1 conv+norm (no act) then depthwise kxk conv+norm+act (hgnetv2.py:83-112)."""

    def __init__(self, c_in, c_out, k, kw):
        super().__init__()
        self.conv1 = ConvNorm(c_in, c_out, 1, **{**kw, "act": None})
        self.conv2 = ConvNorm(c_out, c_out, k, groups=c_out, **kw)

    def forward(self, x, into=None):
        return self.conv2(self.conv1(x), into=into)


class Block(nn.Module):
    def __init__(self, c_in, c_mid, c_out, n_layers, k, light, residual, kw):
        super().__init__()
        make = (lambda ci: LightConv(ci, c_mid, k, kw)) if light else (lambda ci: ConvNorm(ci, c_mid, k, **kw))
        self.layers = nn.ModuleList(make(c_in if i == 0 else c_mid) for i in range(n_layers))
        total = c_in + n_layers * c_mid
        self.part_channels = [c_in] + [c_mid] * n_layers
        self.aggregation = nn.Sequential(ConvNorm(total, c_out // 2, 1, **kw), ConvNorm(c_out // 2, c_out, 1, **kw))
        self.residual = residual

    def make_buffer(self, B, H, W, device):
        return ops.ConcatBuffer(B, H, W, sum(self.part_channels), torch.bfloat16, device)

    def forward(self, x, buf=None, out_into=None):
        """``buf``: this block's concat buffer when ``x`` already IS its first slice (the producer -- the previous block's
        last norm, the stage's downsample -- wrote it there: no copy-in, and that producer's backward picks the slice of
        the concat gradient up on load); ``out_into``: the NEXT block's buffer, whose first slice the output becomes. The
        residual is added by the last norm kernel after its activation (ConvNorm ``post``)."""
        if buf is not None or ops.concat_buffer_supported(x, self.part_channels):
            # no torch.cat: every layer's norm kernel writes its output into its channel slice of one buffer, the next layer
            # reads that slice in place (pixel pitch = total channels), and in backward each layer adds its slice of the
            # aggregation's input gradient on load (ops.ConcatBuffer)
            if buf is None:
                B, _, H, W = x.shape
                buf = self.make_buffer(B, H, W, x.device)
                parts = [ops.concat_copy_in(x, buf)]
            else:
                parts = [x]
            cur = x
            for layer in self.layers:
                cur = layer(cur, into=buf)
                parts.append(cur)
            y = self.aggregation[0](ops.concat_read(buf, parts))
            if FUSE_RESIDUAL or not self.residual:
                return self.aggregation[1](y, into=out_into, post=x if self.residual else None)
            y = self.aggregation[1](y) + x
            return y if out_into is None else ops.concat_copy_in(y, out_into)
        feats = [x]
        for layer in self.layers:
            feats.append(layer(feats[-1]))
        y = self.aggregation(torch.cat(feats, 1))
        y = y + x if self.residual else y
        return y if out_into is None else ops.concat_copy_in(y, out_into)


class Stage(nn.Module):
    def __init__(self, s, kw):
        super().__init__()
        if s.downsample:  # depthwise 3x3 stride 2 + norm, no activation
            self.downsample = ConvNorm(s.c_in, s.c_in, 3, stride=2, groups=s.c_in, **{**kw, "act": None})
        self.blocks = nn.Sequential(*[
            Block(s.c_in if i == 0 else s.c_out, s.c_mid, s.c_out, s.n_layers, s.kernel, s.light, i > 0, kw)
            for i in range(s.n_blocks)
        ])

    def forward(self, x):
        blocks = list(self.blocks)
        chained = CHAIN_BLOCKS and all(ops.concat_buffer_supported(x, b.part_channels) for b in blocks)
        buf = None
        if hasattr(self, "downsample"):
            if chained:  # the downsample's norm writes the first block's first slice
                conv = self.downsample.conv
                B, _, H, W = x.shape
                out = lambda n, i: (n + 2 * conv.padding[i] - conv.kernel_size[i]) // conv.stride[i] + 1
                buf = blocks[0].make_buffer(B, out(H, 0), out(W, 1), x.device)
                x = self.downsample(x, into=buf)
            else:
                x = self.downsample(x)
        for i, block in enumerate(blocks):
            nxt = None
            if chained and i + 1 < len(blocks):  # this block's last norm writes the next block's first slice
                nxt = blocks[i + 1].make_buffer(x.shape[0], x.shape[2], x.shape[3], x.device)
            x = block(x, buf, nxt) if chained else block(x)
            buf = nxt
        return x


class HGNetv2(nn.Module):
    def __init__(self, spec: ModelSpec):
        super().__init__()
        arch = BACKBONES[spec.backbone]
        kw = dict(act="relu", lab=spec.use_lab, norm_name="bn", frozen_norm=spec.frozen_backbone_norm)
        self.stem = Stem(*arch.stem, kw)
        self.stages = nn.ModuleList(Stage(s, kw) for s in arch.stages)
        self.return_idx = tuple(spec.return_idx)
        if spec.frozen_backbone_norm:  # freeze_at=0 with freeze_stem_only: the stem does not train
            for p in self.stem.parameters():
                p.requires_grad_(False)

    def forward(self, x):
        x = self.stem(x)
        outs = []
        for i, stage in enumerate(self.stages):
            x = stage(x)
            if i in self.return_idx:
                outs.append(x)
        return outs
