"""Hybrid encoder: 1x1 projections, one transformer layer (AIFI) on the coarsest map, then a
top-down FPN and bottom-up PAN built from RepNCSPELAN4 blocks. Behaviour and parameter
names after the reference's src/d_fine/arch/hybrid_encoder.py (forward :443-486).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

import os

from .configs import ModelSpec
from .layers import ConvNorm, conv_pair, LayerNorm, Linear, PackedSelfAttention


class RepVgg(nn.Module):
    """silu(conv3x3+norm + conv1x1+norm), training form of VGGBlock (:106-121)."""

    def __init__(self, c):
        super().__init__()
        self.conv1 = ConvNorm(c, c, 3)
        self.conv2 = ConvNorm(c, c, 1)

    def convert_to_deploy(self):
        """3x3 + 1x1 branches and their norms folded into ONE 3x3 conv with bias (VGGBlock.convert_to_deploy,
        hybrid_encoder.py:123-156)."""
        if hasattr(self, "conv"):
            return
        from .layers import fuse_conv_norm

        k3, b3 = fuse_conv_norm(self.conv1.conv, getattr(self.conv1, self.conv1.norm_name))
        k1, b1 = fuse_conv_norm(self.conv2.conv, getattr(self.conv2, self.conv2.norm_name))
        c = k3.shape[0]
        conv = nn.Conv2d(c, c, 3, 1, padding=1).to(k3.device).to(memory_format=torch.channels_last)
        conv.weight.data = (k3 + F.pad(k1, [1, 1, 1, 1])).contiguous(memory_format=torch.channels_last)
        conv.bias.data = b3 + b1
        self.conv = conv
        delattr(self, "conv1")
        delattr(self, "conv2")

    def forward(self, x):
        if hasattr(self, "conv"):  # deployed
            from . import ops

            return ops.conv_bias_act(x, self.conv, "silu")
        if hasattr(self.conv1, "conv_bn_fused") or hasattr(self.conv2, "conv_bn_fused"):
            return self.conv1(x, addend=self.conv2(x), act="silu")
        from . import ops

        (y3, ws3, hs3), (y1, ws1, hs1) = conv_pair(self.conv1, self.conv2, x)  # one input gradient for the two branches
        n3, n1 = getattr(self.conv1, self.conv1.norm_name), getattr(self.conv2, self.conv2.norm_name)
        if self.training and hs3 and hs1 and ops.batch_norm_pair_act_supported(y3, y1, n3, n1, ws3, ws1):
            # both norms + the sum + SiLU in ONE launch (two backward): the 1x1 branch's output and the shared dz stay on chip
            self.conv1._pending_batches += 1
            self.conv2._pending_batches += 1
            return ops.batch_norm_pair_act(y3, y1, n3, n1, "silu", ws3, ws1)
        return self.conv1._norm_tail(y3, ws3, hs3, addend=self.conv2._norm_tail(y1, ws1, hs1), act="silu")  # fused norm+add+silu


class CSP(nn.Module):
    """Two 1x1 branches, one through n RepVgg blocks, summed (CSPLayer :209-239 with
    expansion 1.0, hence no conv3)."""

    def __init__(self, c_in, c_out, n):
        super().__init__()
        self.conv1 = ConvNorm(c_in, c_out, 1, act="silu")
        self.conv2 = ConvNorm(c_in, c_out, 1, act="silu")
        self.bottlenecks = nn.Sequential(*[RepVgg(c_out) for _ in range(n)])

    def forward(self, x):
        if hasattr(self.conv1, "conv_bn_fused") or hasattr(self.conv2, "conv_bn_fused"):
            return self.bottlenecks(self.conv1(x)) + self.conv2(x)
        (ya, ws_a, hs_a), (yb, ws_b, hs_b) = conv_pair(self.conv1, self.conv2, x)
        # the branch sum rides in the second projection's norm kernel (``post``: added after the activation, one rounding)
        return self.conv2._norm_tail(yb, ws_b, hs_b, post=self.bottlenecks(self.conv1._norm_tail(ya, ws_a, hs_a)))


ELAN_BUFFER = os.environ.get("DFINE_ELAN_BUFFER", "1") != "0"  # measurement switch


class RepNCSPELAN4(nn.Module):
    def __init__(self, c_in, c_out, c_mid, c_branch, n):
        super().__init__()
        self.cv1 = ConvNorm(c_in, c_mid, 1, act="silu")
        self.cv2 = nn.Sequential(CSP(c_mid // 2, c_branch, n), ConvNorm(c_branch, c_branch, 3, act="silu"))
        self.cv3 = nn.Sequential(CSP(c_branch, c_branch, n), ConvNorm(c_branch, c_branch, 3, act="silu"))
        self.cv4 = ConvNorm(c_mid + 2 * c_branch, c_out, 1, act="silu")

    def forward(self, x):
        from . import ops

        c_mid, c_branch = self.cv1.out_channels, self.cv2[1].out_channels
        if ELAN_BUFFER and ops.concat_buffer_supported(x, [c_mid // 2, c_mid // 2, c_branch, c_branch]):
            # no torch.cat, no chunk: cv1 and the two branches' last norms write their channel slices of one buffer, the first
            # branch reads the second half of cv1's slice in place, and every producer picks its slice of cv4's input
            # gradient up from the buffer (ops.ConcatBuffer)
            B, _, H, W = x.shape
            buf = ops.ConcatBuffer(B, H, W, c_mid + 2 * c_branch, torch.bfloat16, x.device)
            y = self.cv1(x, into=buf)
            b = ops.concat_slot_slice(y, buf, c_mid // 2, c_mid // 2)
            c = self.cv2[1](self.cv2[0](b), into=buf)
            d = self.cv3[1](self.cv3[0](c), into=buf)
            return self.cv4(ops.concat_read(buf, [y, c, d]))
        y = self.cv1(x)
        a, b = y.chunk(2, dim=1)
        c = self.cv2(b)
        d = self.cv3(c)
        return self.cv4(torch.cat([a, b, c, d], 1))


class SCDown(nn.Module):
    def __init__(self, c, k, s):
        super().__init__()
        self.cv1 = ConvNorm(c, c, 1)
        self.cv2 = ConvNorm(c, c, k, stride=s, groups=c)

    def forward(self, x):
        return self.cv2(self.cv1(x))


class EncoderLayer(nn.Module):
    """Post-norm transformer layer, GELU FFN, dropout 0 (:243-290)."""

    def __init__(self, dim, n_head, ffn):
        super().__init__()
        self.self_attn = PackedSelfAttention(dim, n_head)
        self.linear1 = Linear(dim, ffn)
        self.linear2 = Linear(ffn, dim)
        self.norm1 = LayerNorm(dim)
        self.norm2 = LayerNorm(dim)

    def forward(self, x, pos):
        x = self.norm1(x, self.self_attn(x + pos, x))
        return self.norm2(x, self.linear2(F.gelu(self.linear1(x))))


class _LayerStack(nn.Module):  # gives the reference's key path ``encoder.0.layers.0``
    def __init__(self, layer):
        super().__init__()
        self.layers = nn.ModuleList([layer])


def sincos_position_embedding(w, h, dim, temperature=10000.0, device=None):
    """[1, w*h, dim] table, reference semantics (:425-441): called with (w, h), built from
    meshgrid(arange(w), arange(h), indexing="ij"), yet consumed by row-major (y*W+x) tokens --
    so for token t the first half encodes t // h and the second t % h. Kept as is."""
    gw, gh = torch.meshgrid(torch.arange(int(w), dtype=torch.float32, device=device),
                            torch.arange(int(h), dtype=torch.float32, device=device), indexing="ij")
    omega = 1.0 / (temperature ** (torch.arange(dim // 4, dtype=torch.float32, device=device) / (dim // 4)))
    ow, oh = gw.flatten()[:, None] * omega[None], gh.flatten()[:, None] * omega[None]
    return torch.cat([ow.sin(), ow.cos(), oh.sin(), oh.cos()], 1)[None]


class HybridEncoder(nn.Module):
    def __init__(self, spec: ModelSpec):
        super().__init__()
        hid, L = spec.enc_hidden, spec.num_levels
        self.hidden = hid
        self.input_proj = nn.ModuleList(ConvNorm(c, hid, 1) for c in spec.in_channels)
        self.encoder = nn.ModuleList([_LayerStack(EncoderLayer(hid, spec.num_heads, spec.enc_ffn))])
        elan = lambda: RepNCSPELAN4(hid * 2, hid, hid * 2, spec.csp_c4, spec.csp_blocks)
        self.lateral_convs = nn.ModuleList(ConvNorm(hid, hid, 1) for _ in range(L - 1))
        self.fpn_blocks = nn.ModuleList(elan() for _ in range(L - 1))
        self.downsample_convs = nn.ModuleList(nn.Sequential(SCDown(hid, 3, 2)) for _ in range(L - 1))
        self.pan_blocks = nn.ModuleList(elan() for _ in range(L - 1))
        self._pos_cache = {}
        # the layers the reference builds as ConvNormLayer_fuse (re-parameterised by deploy()): everything in the FPN / PAN
        # except the RepVgg branches (folded as a pair by RepVgg.convert_to_deploy); the input projections stay conv + norm
        for group in (self.lateral_convs, self.fpn_blocks, self.downsample_convs, self.pan_blocks):
            for m in group.modules():
                if isinstance(m, ConvNorm):
                    m.deploy_fusable = True
        for m in self.modules():
            if isinstance(m, RepVgg):
                m.conv1.deploy_fusable = m.conv2.deploy_fusable = False

    def _pos(self, w, h, device):
        key = (w, h, str(device))
        if key not in self._pos_cache:  # the reference rebuilds this on the CPU every forward
            self._pos_cache[key] = sincos_position_embedding(w, h, self.hidden, device=device)
        return self._pos_cache[key]

    def forward(self, feats):
        """feats: backbone maps, fine to coarse -> (outs, inner_outs), both fine to coarse."""
        from . import ops

        # FPN / PAN concats without torch.cat (ops.ConcatBuffer): the upsampled map, the lateral input projection and the
        # PAN's downsampled map are WRITTEN into their halves of the block's input buffer by the kernels that produce them
        # (the stock cat of a channels_last and an NCHW-dense operand also cost a layout-conversion copy each way)
        buffered = self.training and ops.concat_buffer_supported(feats[-1], [self.hidden, self.hidden])
        if buffered:  # the finer levels' projections are computed later, straight into the FPN buffers
            proj = [None] * (len(feats) - 1) + [self.input_proj[-1](feats[-1])]
        else:
            proj = [p(f) for p, f in zip(self.input_proj, feats)]
        x = proj[-1]
        B, C, h, w = x.shape
        tokens = x.flatten(2).transpose(1, 2)  # a view for channel-last maps
        tokens = self.encoder[0].layers[0](tokens, self._pos(w, h, x.device).to(tokens.dtype))
        proj[-1] = tokens.transpose(1, 2).reshape(B, C, h, w).contiguous(memory_format=torch.channels_last)

        inner = [proj[-1]]
        for lvl in range(len(proj) - 1, 0, -1):
            j = len(proj) - 1 - lvl
            inner[0] = self.lateral_convs[j](inner[0])
            if buffered and ops.upsample_nearest2x_supported(inner[0]):
                Bn, _, hh, ww = inner[0].shape
                buf = ops.ConcatBuffer(Bn, 2 * hh, 2 * ww, 2 * self.hidden, torch.bfloat16, inner[0].device)
                up = ops.upsample_nearest2x(inner[0], into=buf)
                proj[lvl - 1] = self.input_proj[lvl - 1](feats[lvl - 1], into=buf)  # computed here, into the buffer's second half
                inner.insert(0, self.fpn_blocks[j](ops.concat_read(buf, [up, proj[lvl - 1]])))
                continue
            if proj[lvl - 1] is None:
                proj[lvl - 1] = self.input_proj[lvl - 1](feats[lvl - 1])
            up = F.interpolate(inner[0], scale_factor=2.0, mode="nearest")
            inner.insert(0, self.fpn_blocks[j](torch.cat([up, proj[lvl - 1]], 1)))
        outs = [inner[0]]
        for lvl in range(len(proj) - 1):
            if buffered:
                Bn, _, hh, ww = inner[lvl + 1].shape
                buf = ops.ConcatBuffer(Bn, hh, ww, 2 * self.hidden, torch.bfloat16, inner[lvl + 1].device)
                sc = self.downsample_convs[lvl][0]
                down = sc.cv2(sc.cv1(outs[-1]), into=buf)
                outs.append(self.pan_blocks[lvl](ops.concat_read(buf, [down, ops.concat_copy_in(inner[lvl + 1], buf)])))
                continue
            down = self.downsample_convs[lvl](outs[-1])
            outs.append(self.pan_blocks[lvl](torch.cat([down, inner[lvl + 1]], 1)))
        return outs, inner
