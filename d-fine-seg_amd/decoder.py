"""D-FINE transformer decoder: query selection, contrastive denoising queries, decoder layers
(self-attention, multi-scale deformable cross-attention, gate, FFN), the fine-grained
distribution refinement (FDR) heads and the query-based mask head.

Behaviour and parameter names after the reference's src/d_fine/arch/dfine_decoder.py;
what differs is how it runs on the device: the deformable gather is the HIP kernel behind
``ops.deformable_attention_core`` on the channel-last memory (no per-level permute/copy, no
grid_sample), anchors / W(n) / attention masks are cached on the device instead of being
rebuilt on the host every forward, and the denoising group is built densely from a padded
TargetBatch (no ``nonzero`` / per-image Python loop).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .configs import ModelSpec
from .layers import MLP, ConvNorm, LayerNorm, Linear, PackedSelfAttention, bias_for_prior, sync_group_of
from .targets import TargetBatch


def inverse_sigmoid(x, eps=1e-5):
    x = x.clamp(0.0, 1.0)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def weighting_function(reg_max, up, reg_scale):
    """W(n): reg_max+1 non-uniform bin positions (arch/utils.py:145-188), fp32 on up's device."""
    up0 = up.detach().float().reshape(-1)[0].abs()
    ub1 = up0 * reg_scale.detach().reshape(-1)[0].abs()
    ub2 = ub1 * 2
    step = (ub1 + 1) ** (2 / (reg_max - 2))
    left = [-(step**i) + 1 for i in range(reg_max // 2 - 1, 0, -1)]
    right = [step**i - 1 for i in range(1, reg_max // 2)]
    return torch.stack([-ub2] + left + [torch.zeros_like(ub1)] + right + [ub2]).float()


def distance2bbox(points, distance, reg_scale):
    """Edge distances (in units of the reference box / reg_scale) -> cxcywh (arch/utils.py:119-142)."""
    rs = float(reg_scale)
    half = 0.5 * rs
    wq, hq = points[..., 2] / rs, points[..., 3] / rs
    x1 = points[..., 0] - (half + distance[..., 0]) * wq
    y1 = points[..., 1] - (half + distance[..., 1]) * hq
    x2 = points[..., 0] + (half + distance[..., 2]) * wq
    y2 = points[..., 1] + (half + distance[..., 3]) * hq
    return torch.stack([(x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1], -1)


def mask_logits(embed, mask_feat):
    """Mask logits of queries: embed [B, Q, C] x mask_feat [B, C, H, W] -> [B, Q, H, W]
    (``_mask_logits_from_h``, dfine_decoder.py:937-940)."""
    return torch.einsum("bqc,bchw->bqhw", embed.to(mask_feat.dtype), mask_feat)


class MSDeformableAttention(nn.Module):
    """Offsets and weights from the query, locations around the 4-d reference box
    (dfine_decoder.py:148-166), then the HIP gather. No value/output projection."""

    def __init__(self, dim, n_head, num_points):
        super().__init__()
        self.n_head, self.num_points = n_head, list(num_points)
        P = sum(num_points)
        self.sampling_offsets = Linear(dim, n_head * P * 2)
        self.attention_weights = Linear(dim, n_head * P)
        self.register_buffer("num_points_scale", torch.tensor([1.0 / n for n in num_points for _ in range(n)]))
        self._reset()

    def _reset(self):  # reference init :100-117
        nn.init.zeros_(self.sampling_offsets.weight)
        theta = torch.arange(self.n_head, dtype=torch.float32) * (2.0 * math.pi / self.n_head)
        grid = torch.stack([theta.cos(), theta.sin()], -1)
        grid = grid / grid.abs().max(-1, keepdim=True).values
        grid = grid.view(self.n_head, 1, 2).repeat(1, sum(self.num_points), 1)
        grid = grid * torch.cat([torch.arange(1, n + 1) for n in self.num_points]).view(1, -1, 1)
        with torch.no_grad():
            self.sampling_offsets.bias.copy_(grid.flatten())
        nn.init.zeros_(self.attention_weights.weight)
        nn.init.zeros_(self.attention_weights.bias)

    def forward(self, query, ref_boxes, value, spatial_shapes, sink=None, finalize=True):
        """query [B,Lq,D]; ref_boxes [B,Lq,4] cxcywh; value [B,S,nH,c] channel-last memory."""
        B, Lq, _ = query.shape
        nH, P = self.n_head, sum(self.num_points)
        loc, weights = ops.msda_prepare(self.sampling_offsets(query), self.attention_weights(query), ref_boxes, self.num_points_scale, nH)
        return ops.deformable_attention_core(value, spatial_shapes, loc, weights, self.num_points, sink, finalize)


_FUSED_GATE = True  # tests flip it to compare with the op-by-op form


class Gate(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.gate = Linear(2 * dim, 2 * dim)
        nn.init.constant_(self.gate.bias, bias_for_prior(0.5))
        nn.init.zeros_(self.gate.weight)
        self.norm = LayerNorm(dim)

    def forward(self, a, b):
        pre = self.gate(torch.cat([a, b], -1))
        if a.is_cuda and pre.dtype == a.dtype == b.dtype and pre.dtype in (torch.float32, torch.bfloat16) and _FUSED_GATE:
            return self.norm(ops.gate_mix(pre, a, b))  # sigmoid, both products and their backward: one launch each way
        g1, g2 = torch.sigmoid(pre).chunk(2, -1)
        return self.norm(g1 * a, g2 * b)


class DecoderLayer(nn.Module):
    def __init__(self, dim, n_head, ffn, num_points):
        super().__init__()
        self.self_attn = PackedSelfAttention(dim, n_head)
        self.norm1 = LayerNorm(dim)
        self.cross_attn = MSDeformableAttention(dim, n_head, num_points)
        self.gateway = Gate(dim)
        self.linear1 = Linear(dim, ffn)
        self.linear2 = Linear(ffn, dim)
        self.norm3 = LayerNorm(dim)
        nn.init.xavier_uniform_(self.linear1.weight)
        nn.init.xavier_uniform_(self.linear2.weight)

    def forward(self, tgt, ref_boxes, value, spatial_shapes, blocked, pos, sink=None, finalize=True):
        tgt = self.norm1(tgt, self.self_attn(tgt + pos, tgt, blocked))
        cross = self.cross_attn(tgt + pos, ref_boxes, value, spatial_shapes, sink, finalize).to(tgt.dtype)
        tgt = self.gateway(tgt, cross)
        # bias + ReLU in linear1's GEMM epilogue; the ReLU's backward in the epilogue of linear2's input-gradient GEMM
        hidden = ops.linear(tgt, self.linear1.weight, self.linear1.bias, relu=True, mask_by_consumer=True)
        ffn = ops.linear(hidden, self.linear2.weight, self.linear2.bias, relu_in=True)
        return self.norm3(tgt, ffn, clamp=65504.0)


class LQE(nn.Module):
    """Location-quality estimate added to every class logit (:298-313)."""

    def __init__(self, k, hidden, reg_max):
        super().__init__()
        self.k, self.reg_max = k, reg_max
        self.reg_conf = MLP(4 * (k + 1), hidden, 1, 2)
        nn.init.zeros_(self.reg_conf.layers[-1].weight)
        nn.init.zeros_(self.reg_conf.layers[-1].bias)

    def forward(self, scores, corners):
        B, L, _ = corners.shape
        prob = torch.softmax(corners.float().view(B, L, 4, self.reg_max + 1), dim=-1)
        top = prob.topk(self.k, dim=-1).values
        stat = torch.cat([top, top.mean(-1, keepdim=True)], -1).view(B, L, -1)
        return scores + self.reg_conf(stat.to(scores.dtype))

    def from_stat(self, scores, stat):
        """``stat`` [B, L, 4*(k+1)]: the statistics ``forward`` derives from the corners, precomputed by
        ``ops.fdr_head`` together with the boxes."""
        return scores + self.reg_conf(stat.to(scores.dtype))


class _DecoderStack(nn.Module):  # key paths ``decoder.decoder.layers.{i}`` / ``.lqe_layers.{i}``
    def __init__(self, spec: ModelSpec, up, reg_scale):
        super().__init__()
        self.layers = nn.ModuleList(
            DecoderLayer(spec.hidden, spec.num_heads, spec.dec_ffn, spec.num_points) for _ in range(spec.num_layers))
        self.lqe_layers = nn.ModuleList(LQE(4, 64, spec.reg_max) for _ in range(spec.num_layers))
        # the reference registers the SAME two Parameters here and on the parent (:400,597-598),
        # so both ``decoder.up`` and ``decoder.decoder.up`` appear in its state_dict
        self.up, self.reg_scale = up, reg_scale


class MaskPixelDecoder(nn.Module):
    """FPN maps -> 1x1+BN each, bilinear-summed at stride 8, + 1x1+BN of the PAN stride-8 memory,
    then 2x2 stride-2 transposed conv + BN + ReLU: mask features at stride 4 (:316-376)."""

    def __init__(self, in_chs, out_ch):
        super().__init__()
        self.lateral = nn.ModuleList(nn.Conv2d(c, out_ch, 1, bias=False) for c in in_chs)
        self.bn = nn.ModuleList(nn.BatchNorm2d(out_ch) for _ in in_chs)
        self.enc_proj = nn.Conv2d(in_chs[0], out_ch, 1, bias=False)
        self.enc_bn = nn.BatchNorm2d(out_ch)
        self.upconv = nn.ConvTranspose2d(out_ch, out_ch, 2, stride=2, bias=False)
        self.bn1 = nn.BatchNorm2d(out_ch)

    def _conv_bn(self, conv, bn, x, act=None, addend=None, transposed=False):
        """conv -> BatchNorm (+addend) (+act) through the HIP kernels (implicit-GEMM conv with the statistics in its
        epilogue, fused norm); parameters stay in the reference's nn.Conv2d / nn.BatchNorm2d modules (state_dict keys)."""
        if not (x.is_cuda and ops.batch_norm_act_supported(x, conv.out_channels)):
            y = bn(conv(x))
            y = y + addend if addend is not None else y
            return F.relu(y) if act == "relu" else y
        bf16_mode = torch.is_autocast_enabled() or x.dtype == torch.bfloat16
        batch_stats = self.training and bn.training  # a norm in .eval() inside a training model keeps its running statistics
        ws = ops.bn_workspace(conv.out_channels, x.device) if (bf16_mode and batch_stats) else None
        stats = ws[: 32 * conv.out_channels] if ws is not None else None
        y = ops.conv_transpose2x2(x, conv.weight, stats) if transposed else ops.dense_conv(x, conv.weight, 1, 0, stats=stats)
        if batch_stats:
            bn.num_batches_tracked.add_(1)
        sync = sync_group_of(bn) if (batch_stats and isinstance(bn, nn.SyncBatchNorm)) else None
        return ops.batch_norm_act(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, act, addend=addend,
                                  training=batch_stats, momentum=bn.momentum or 0.1, eps=bn.eps, ws=ws, have_stats=ws is not None,
                                  sync=sync)

    def forward(self, feats, enc_s8):
        x = self._conv_bn(self.lateral[0], self.bn[0], feats[0])
        size = x.shape[-2:]
        for i in range(1, len(feats)):
            x = ops.bilinear_add(x, self._conv_bn(self.lateral[i], self.bn[i], feats[i]))  # x + F.interpolate(..., size, "bilinear")
        x = self._conv_bn(self.enc_proj, self.enc_bn, enc_s8, addend=x)  # same size: the reference's bilinear resize is the identity
        return self._conv_bn(self.upconv, self.bn1, x, act="relu", transposed=True)


class DFINETransformer(nn.Module):
    def __init__(self, spec: ModelSpec, num_classes=80, eval_spatial_size=None, enable_mask_head=False):
        super().__init__()
        self.spec, self.num_classes = spec, num_classes
        self.enable_mask_head = enable_mask_head
        hid = spec.hidden
        if spec.enc_hidden != hid:  # x: 384 -> 256 projection; Identity otherwise (:745-765)
            self.input_proj = nn.ModuleList(ConvNorm(spec.enc_hidden, hid, 1) for _ in range(spec.num_levels))
            for m in self.input_proj:
                nn.init.xavier_uniform_(m.conv.weight)
        else:
            self.input_proj = nn.ModuleList(nn.Identity() for _ in range(spec.num_levels))
        self.up = nn.Parameter(torch.tensor([0.5]), requires_grad=False)
        self.reg_scale = nn.Parameter(torch.tensor([spec.reg_scale]), requires_grad=False)  # int64, as shipped
        self.decoder = _DecoderStack(spec, self.up, self.reg_scale)
        self.denoising_class_embed = nn.Embedding(num_classes + 1, hid, padding_idx=num_classes)
        nn.init.normal_(self.denoising_class_embed.weight[:-1])
        if enable_mask_head:
            self.pixel_decoder = MaskPixelDecoder([spec.enc_hidden] * spec.num_levels, spec.mask_dim)
            self.mask_head = MLP(hid, hid, spec.mask_dim, 3)
        self.query_pos_head = MLP(4, 2 * hid, hid, 2)
        self.enc_output = nn.Sequential()
        self.enc_output.add_module("proj", Linear(hid, hid))
        self.enc_output.add_module("norm", LayerNorm(hid))
        self.enc_score_head = Linear(hid, num_classes)
        self.enc_bbox_head = MLP(hid, hid, 4, 3)
        self.dec_score_head = nn.ModuleList(Linear(hid, num_classes) for _ in range(spec.num_layers))
        self.pre_bbox_head = MLP(hid, hid, 4, 3)
        self.dec_bbox_head = nn.ModuleList(MLP(hid, hid, 4 * (spec.reg_max + 1), 3) for _ in range(spec.num_layers))
        if eval_spatial_size:  # buffers in the checkpoint layout (:700-706)
            h, w = eval_spatial_size
            anchors, valid = self.generate_anchors([(int(h / s), int(w / s)) for s in spec.strides])
            self.register_buffer("anchors", anchors)
            self.register_buffer("valid_mask", valid)
        self._reset()
        self._cache = {}

    def _reset(self):  # reference init :720-743
        prior = bias_for_prior(0.01)
        nn.init.constant_(self.enc_score_head.bias, prior)
        for head in (self.enc_bbox_head, self.pre_bbox_head, *self.dec_bbox_head):
            nn.init.zeros_(head.layers[-1].weight)
            nn.init.zeros_(head.layers[-1].bias)
        for cls in self.dec_score_head:
            nn.init.constant_(cls.bias, prior)
        nn.init.xavier_uniform_(self.enc_output.proj.weight)
        nn.init.xavier_uniform_(self.query_pos_head.layers[0].weight)
        nn.init.xavier_uniform_(self.query_pos_head.layers[1].weight)

    # ---------------------------------------------------------------- cached device constants
    def generate_anchors(self, spatial_shapes, device="cpu"):
        """(:815-838) logit-space anchors [1,S,4] (inf where invalid) and valid mask [1,S,1]."""
        spec, out = self.spec, []
        for lvl, (h, w) in enumerate(spatial_shapes):
            gy, gx = torch.meshgrid(torch.arange(h, device=device), torch.arange(w, device=device), indexing="ij")
            xy = torch.stack([(gx.float() + 0.5) / float(w), (gy.float() + 0.5) / float(h)], -1)  # no host tensor: capturable
            wh = torch.full_like(xy, spec.anchor_grid * (2.0**lvl))
            out.append(torch.cat([xy, wh], -1).view(1, h * w, 4))
        anchors = torch.cat(out, 1)
        valid = ((anchors > spec.anchor_eps) & (anchors < 1 - spec.anchor_eps)).all(-1, keepdim=True)
        anchors = torch.log(anchors / (1 - anchors))
        return torch.where(valid, anchors, torch.inf), valid

    def _cached(self, key, make):
        if key not in self._cache:
            self._cache[key] = make()
        return self._cache[key]

    def _dn_attention_mask(self, max_gt, groups, device):
        """bool [Nd+Q, Nd+Q], True = blocked (arch/utils.py:437-455)."""
        def make():  # built ON the device from index comparisons: a new (max_gt, groups) may first be seen inside a
            # graph capture, where a host-built mask would be a pageable host->device copy on the capturing stream
            Q, Nd = self.spec.num_queries, 2 * max_gt * groups
            idx = torch.arange(Nd + Q, device=device)
            is_dn = idx < Nd
            group = idx // (2 * max_gt)
            row_dn, col_dn = is_dn[:, None], is_dn[None, :]
            # matching queries cannot see denoising queries; a denoising group sees only itself (and the matching queries)
            blocked = (~row_dn & col_dn) | (row_dn & col_dn & (group[:, None] != group[None, :]))
            if blocked.is_cuda:
                blocked._attn_bits = ops.attention_mask_bits(blocked)  # packed once, next to the cached mask
            return blocked
        return self._cached(("dn_mask", max_gt, groups, str(device)), make)

    # ---------------------------------------------------------------- denoising queries
    def denoising_group(self, tb: TargetBatch, noise=None):
        """Contrastive denoising queries (arch/utils.py:357-467, invoked with box_noise_scale=1.0
        at dfine_decoder.py:949-959), dense over the padded TargetBatch. ``noise`` (dict with
        flip_u [B,N], new_label [B,N], sign01 [B,N,4], part_u [B,N,4]) can be injected for parity;
        otherwise it is drawn on the device. Returns (content, bbox_unact, blocked, dn_meta)."""
        spec, C = self.spec, self.num_classes
        if tb.total == 0 or spec.num_denoising <= 0:
            return None, None, None, {"dn_num_group": 0, "dn_num_split": [0, spec.num_queries], "dn_assign": None}
        G = max(tb.num_gts)
        groups = max(1, spec.num_denoising // G)
        B, N, dev = tb.labels.shape[0], 2 * G * groups, tb.labels.device
        cls = tb.labels[:, :G].repeat(1, 2 * groups)
        box = tb.boxes[:, :G].repeat(1, 2 * groups, 1)
        pad = tb.valid[:, :G].repeat(1, 2 * groups)
        slot = torch.arange(N, device=dev) % (2 * G)
        negative = (slot >= G).view(1, N, 1).float()
        if noise is None:
            noise = dict(flip_u=torch.rand(B, N, device=dev), new_label=torch.randint(0, C, (B, N), device=dev),
                         sign01=torch.randint(0, 2, (B, N, 4), device=dev).float(), part_u=torch.rand(B, N, 4, device=dev))
        flip = (noise["flip_u"] < spec.label_noise_ratio * 0.5) & pad
        cls = torch.where(flip, noise["new_label"].long(), cls)
        half_wh = box[..., 2:].clamp(min=0) * 0.5
        xyxy = torch.cat([box[..., :2] - half_wh, box[..., :2] + half_wh], -1)
        reach = (box[..., 2:] * 0.5).repeat(1, 1, 2)
        part = noise["part_u"] + negative  # negatives jitter by 1..2 box halves, positives by 0..1
        xyxy = (xyxy + (noise["sign01"] * 2.0 - 1.0) * part * reach).clamp(0.0, 1.0)
        box = torch.cat([(xyxy[..., :2] + xyxy[..., 2:]) / 2, xyxy[..., 2:] - xyxy[..., :2]], -1).abs()
        content = ops.embedding(cls, self.denoising_class_embed.weight, self.denoising_class_embed.padding_idx)  # HIP backward
        assign = torch.where((slot < G)[None, :] & pad, slot[None, :].expand(B, N), -1)  # GT index of positive slots
        meta = {"dn_num_group": groups, "dn_num_split": [N, spec.num_queries], "dn_assign": assign}
        return content, inverse_sigmoid(box), self._dn_attention_mask(G, groups, dev), meta

    # ---------------------------------------------------------------- forward
    def forward(self, feats, inner_feats, targets=None, inject=None):
        spec, inject = self.spec, inject or {}
        training = self.training
        Q, L = spec.num_queries, spec.num_layers
        proj = [p(f) for p, f in zip(self.input_proj, feats)]
        spatial_shapes = [tuple(f.shape[2:]) for f in proj]
        # channel-last maps flatten to [B, HW, C] views; the concat is the decoder memory AND the
        # deformable-attention value (value_op ignores value_proj, :416-426)
        memory = torch.cat([f.permute(0, 2, 3, 1).flatten(1, 2) for f in proj], 1)
        B, S, hid = memory.shape
        dev = memory.device

        tb = None
        if targets is not None:
            tb = TargetBatch.from_list(targets, self.num_classes, dev)
        do_masks = self.enable_mask_head and (tb is None or tb.masks is not None or tb.mask_tgt is not None)

        dn_content = dn_bbox = blocked = dn_meta = None
        if training:
            dn_content, dn_bbox, blocked, dn_meta = self.denoising_group(tb, inject.get("cdn_noise"))

        # ---- query selection (:840-922)
        key = ("anchors", tuple(spatial_shapes), str(dev))
        anchors, valid = self._cached(key, lambda: self.generate_anchors(spatial_shapes, dev))
        topk_ind = inject.get("topk_ind")
        sink = ops.ValueGradSink() if (training and torch.is_grad_enabled()) else None
        pick = lambda t: t.gather(1, topk_ind.unsqueeze(-1).expand(-1, -1, t.shape[-1]))
        if sink is not None:
            # training: all S tokens go through enc_output / the score head only to RANK them (no graph);
            # the differentiable branch is re-run on the Q selected rows, the only ones with a gradient
            if topk_ind is None:
                with torch.no_grad():
                    # the reference ranks enc_output(memory * valid_mask) (dfine_decoder.py:885-905): a token whose anchor is
                    # invalid contributes a ZERO row, and a zero row scores the same wherever it sits -- so the S tokens go
                    # through unmasked and the invalid ones take the score of ONE zero row pushed through the same two modules
                    # (row-wise ops: bit-equal), instead of a multiply over [B, S, D]
                    scores = ops.row_max(self.enc_score_head(self.enc_output(memory)))
                    zero_score = ops.row_max(self.enc_score_head(self.enc_output(memory.new_zeros(1, 1, hid))))
                    scores = torch.where(valid.view(1, S), scores, zero_score.view(1, 1))
                    topk_ind = torch.topk(scores, Q, dim=-1).indices
            topk_mem = self.enc_output(ops.select_rows(memory, topk_ind, valid, sink))
            topk_logits = self.enc_score_head(topk_mem)
        else:
            enc_mem = self.enc_output(memory * valid.to(memory.dtype))
            enc_logits = self.enc_score_head(enc_mem)
            if topk_ind is None:
                topk_ind = torch.topk(enc_logits.max(-1).values, Q, dim=-1).indices
            topk_mem, topk_logits = pick(enc_mem), pick(enc_logits)
        enc_bbox_unact = self.enc_bbox_head(topk_mem).float() + pick(anchors.expand(B, -1, -1))
        content, ref_unact = topk_mem.detach(), enc_bbox_unact.detach()
        if dn_bbox is not None:
            ref_unact = torch.cat([dn_bbox, ref_unact], 1)
            content = torch.cat([dn_content.to(content.dtype), content], 1)

        # ---- decoder layers + FDR heads (:435-530)
        value = memory.view(B, S, spec.num_heads, hid // spec.num_heads)
        project = self._cached(("W", str(dev)), lambda: weighting_function(spec.reg_max, self.up, self.reg_scale).to(dev))
        reg_scale = float(spec.reg_scale)
        out, out_detach, corners_prev = content, 0, 0
        ref_detach = torch.sigmoid(ref_unact)
        boxes_l, logits_l, corners_l, hs = [], [], [], []
        for i, layer in enumerate(self.decoder.layers):
            # clamp(-10, 10) of the reference (dfine_decoder.py:472) as hardtanh: the same forward, and ONE backward kernel
            # where clamp's autograd runs five (ge, le, logical_and, where, fill); the two differ only in passing a gradient
            # to an element that sits EXACTLY on +-10 (clamp does, hardtanh does not)
            pos = F.hardtanh(self.query_pos_head(ref_detach), -10.0, 10.0)
            out = layer(out, ref_detach, value, spatial_shapes, blocked, pos, sink, finalize=False)  # select_rows finalizes
            hs.append(out)
            if i == 0:
                pre_boxes = torch.sigmoid(self.pre_bbox_head(out).float() + inverse_sigmoid(ref_detach))
                pre_logits = self.dec_score_head[0](out)
                ref_initial = pre_boxes.detach()
            corners = self.dec_bbox_head[i](out + out_detach).float() + corners_prev
            box, stat = ops.fdr_head(corners, ref_initial, project, reg_scale, self.decoder.lqe_layers[i].k)  # HIP, fused
            if training or i == L - 1:
                logits_l.append(self.decoder.lqe_layers[i].from_stat(self.dec_score_head[i](out), stat))
                boxes_l.append(box)
                corners_l.append(corners)
            corners_prev, ref_detach, out_detach = corners, box.detach(), out.detach()

        # ---- mask branch (:999-1022)
        if do_masks:
            mask_feat = self.pixel_decoder(inner_feats, proj[0])  # proj[0] IS memory[:, :h8*w8] as a map (no copy)
            mask_of = lambda h: mask_logits(self.mask_head(h), mask_feat)

        if not training:
            res = {"pred_logits": logits_l[-1], "pred_boxes": boxes_l[-1]}
            if do_masks:
                res["pred_masks"] = torch.sigmoid(mask_of(hs[-1]))
            return res

        Nd = dn_meta["dn_num_split"][0]
        parts = {}  # each output tensor is split ONCE (ops.split_rows: one autograd node, its backward one concatenation)

        def _split(t):
            if id(t) not in parts:
                parts[id(t)] = (t, ops.split_rows(t, Nd))  # (the tensor is kept so that its id stays unique)
            return parts[id(t)][1]

        dn_part = lambda t: _split(t)[0]
        reg_part = lambda t: _split(t)[1]

        def head(part, i, teacher, masks=None):
            d = {"pred_logits": part(logits_l[i]), "pred_boxes": part(boxes_l[i]), "pred_corners": part(corners_l[i]),
                 "ref_points": part(ref_initial), "teacher_corners": part(corners_l[-1]) if teacher else None,
                 "teacher_logits": part(logits_l[-1]) if teacher else None}
            if teacher and i == L - 1:
                d["self_distilled"] = True  # teacher == student: the reference finds out with torch.equal (a sync)
            if masks is not None:
                d["pred_mask_embed"] = masks[i]
            return d

        # Mask logits are NOT materialised for every query (the reference's einsum writes 4 x [B,492,160,160]
        # and the loss then reads the matched rows only, dfine_decoder.py:999-1022): heads carry the mask
        # embeddings [B, Lq, hidden] and the shared feature map; the criterion forms the logits of the
        # matched rows after matching. ``mask_logits(embed, feat)`` gives the reference's tensor on demand.
        embeds = [self.mask_head(h) for h in hs] if do_masks else None
        reg_masks = [reg_part(e) for e in embeds] if do_masks else None
        res = {k: v for k, v in head(reg_part, L - 1, False).items() if v is not None}
        res.update(up=self.up, reg_scale=self.reg_scale)
        res["fdr"] = {"W": project, "reg_scale": reg_scale}  # host-side copies so the criterion need not read the Parameters back
        if do_masks:
            res["pred_mask_embed"] = reg_masks[-1]
            res["mask_feat"] = mask_feat
        res["aux_outputs"] = [head(reg_part, i, True, reg_masks) for i in range(L - 1)]
        res["enc_aux_outputs"] = [{"pred_logits": topk_logits, "pred_boxes": torch.sigmoid(enc_bbox_unact)}]
        res["pre_outputs"] = {"pred_logits": reg_part(pre_logits), "pred_boxes": reg_part(pre_boxes)}
        res["enc_meta"] = {"class_agnostic": False}
        res["enc_topk_ind"] = topk_ind  # the query selection of this forward (int64 [B, Q]); not a key of the reference's dict
        dn_masks = [dn_part(e) for e in embeds] if do_masks else None
        # detection: every layer's dn head; with masks the reference's zip truncates to L-1 layers and
        # carries the last layer's dn masks separately (:1059-1070)
        n_dn = L - 1 if do_masks else L
        res["dn_outputs"] = [head(dn_part, i, True, dn_masks) for i in range(n_dn)]
        if dn_masks is not None:
            res["dn_pred_mask_embed"] = dn_masks[-1]
        res["dn_pre_outputs"] = {"pred_logits": dn_part(pre_logits), "pred_boxes": dn_part(pre_boxes)}
        res["dn_meta"] = dn_meta
        res["targets"] = tb  # the packed batch, so the criterion does not pack it again
        return res
