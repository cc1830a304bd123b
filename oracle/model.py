"""Functional fp32 restatement of the D-FINE forward pass (backbone -> hybrid encoder ->
decoder), operating on a flat ``state_dict`` with the reference's key layout
(SURVEY.md A.10). Autograd supplies the backward. TEST INFRASTRUCTURE ONLY (see
oracle/__init__.py).

    out = forward(sd, cfg, x, targets=None, training=True, enable_mask_head=False, inject=None)

``sd``: name -> tensor (parameters with requires_grad as wanted; BatchNorm running stats are
updated in place when training, like nn.BatchNorm2d). ``cfg``: oracle.configs.size_cfg(name).
``inject`` optionally supplies the data-dependent / random intermediates so stage-wise parity
is possible: {"cdn_noise": dict from ops.cdn_noise, "topk_ind": LongTensor [B, 300]}.
"""
import math

import torch
import torch.nn.functional as F

from . import configs as C
from . import ops


# ======================================================================================
# building blocks
# ======================================================================================
class _DenseGrad(torch.autograd.Function):
    """Identity whose backward hands a contiguous gradient upstream. torch 2.10 CPU
    batch_norm backward returns wrong values when grad_output has channels-last-like strides
    with a size-1 batch dimension, e.g. (256, 1, 5120, 256) as produced by the backward of
    flatten(2).permute(0, 2, 1) at B=1, while the input is NCHW-contiguous (found while pinning
    this oracle; the reference escapes it at most layers only because its SiLU is in-place)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.contiguous()


def _batch_norm(sd, p, x, training, frozen=False, eps=1e-5):
    w, b = sd[p + ".weight"], sd[p + ".bias"]
    rm, rv = sd[p + ".running_mean"], sd[p + ".running_var"]
    if frozen:  # FrozenBatchNorm2d, common.py:59-67
        scale = w * (rv + eps).rsqrt()
        return x * scale.view(1, -1, 1, 1) + (b - rm * scale).view(1, -1, 1, 1)
    if training and (p + ".num_batches_tracked") in sd:
        sd[p + ".num_batches_tracked"] += 1
    return _DenseGrad.apply(F.batch_norm(x, rm, rv, w, b, training=training, momentum=0.1, eps=eps))


def _hg_conv(sd, p, x, k, cfg, training, stride=1, groups=1, act=True):
    """ConvBNAct, hgnetv2.py:35-80: conv(no bias) -> BN -> ReLU -> LAB (LAB only with act)."""
    x = F.conv2d(x, sd[p + ".conv.weight"], None, stride, (k - 1) // 2, 1, groups)
    x = _batch_norm(sd, p + ".bn", x, training, frozen=cfg["frozen"])
    if act:
        x = F.relu(x)
        if cfg["use_lab"]:
            x = sd[p + ".lab.scale"] * x + sd[p + ".lab.bias"]
    return x


def _conv_norm(sd, p, x, training, k=1, stride=1, groups=1, act=None):
    """ConvNormLayer(_fuse), hybrid_encoder.py:21-45,82-93: conv(no bias) -> BN -> act."""
    x = F.conv2d(x, sd[p + ".conv.weight"], None, stride, (k - 1) // 2, 1, groups)
    x = _batch_norm(sd, p + ".norm", x, training)
    return F.silu(x) if act == "silu" else x


def _linear(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"])


def _mlp(sd, p, x, n):
    """MLP, dfine_decoder.py:33-46 (ReLU between layers)."""
    for i in range(n):
        x = _linear(sd, f"{p}.layers.{i}", x)
        if i < n - 1:
            x = F.relu(x)
    return x


def _layer_norm(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _mha(sd, p, q_in, k_in, v_in, n_head, attn_mask=None):
    """nn.MultiheadAttention(batch_first=True) forward (packed in_proj, dropout 0).
    attn_mask: bool [Lq, Lk], True = not allowed."""
    B, Lq, D = q_in.shape
    Lk = k_in.shape[1]
    W, bias = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    dh = D // n_head
    q = F.linear(q_in, W[:D], bias[:D]).view(B, Lq, n_head, dh).transpose(1, 2)
    k = F.linear(k_in, W[D : 2 * D], bias[D : 2 * D]).view(B, Lk, n_head, dh).transpose(1, 2)
    v = F.linear(v_in, W[2 * D :], bias[2 * D :]).view(B, Lk, n_head, dh).transpose(1, 2)
    scores = (q / math.sqrt(dh)) @ k.transpose(-1, -2)
    if attn_mask is not None:
        scores = scores.masked_fill(attn_mask, float("-inf"))
    out = torch.softmax(scores, dim=-1) @ v
    out = out.transpose(1, 2).reshape(B, Lq, D)
    return _linear(sd, p + ".out_proj", out)


# ======================================================================================
# HGNetv2 backbone -- hgnetv2.py
# ======================================================================================
def backbone(sd, cfg, x, training):
    arch = C.HGNET[cfg["backbone"]]
    kw = dict(cfg=cfg, training=training)
    p = "backbone.stem"
    # StemBlock.forward, hgnetv2.py:156-166
    x = _hg_conv(sd, p + ".stem1", x, 3, stride=2, **kw)
    x = F.pad(x, (0, 1, 0, 1))
    x2 = _hg_conv(sd, p + ".stem2a", x, 2, **kw)
    x2 = F.pad(x2, (0, 1, 0, 1))
    x2 = _hg_conv(sd, p + ".stem2b", x2, 2, **kw)
    x1 = F.max_pool2d(x, kernel_size=2, stride=1, ceil_mode=True)
    x = torch.cat([x1, x2], dim=1)
    x = _hg_conv(sd, p + ".stem3", x, 3, stride=2, **kw)
    x = _hg_conv(sd, p + ".stem4", x, 1, **kw)

    outs = []
    for si, (cin, mid, cout, n_blocks, down, light, k, n_layers) in enumerate(arch["stages"]):
        sp = f"backbone.stages.{si}"
        if down:  # depthwise 3x3 s2 + BN, no act (hgnetv2.py:295-304)
            x = _hg_conv(sd, sp + ".downsample", x, 3, stride=2, groups=cin, act=False, **kw)
        for bi in range(n_blocks):  # HG_Block.forward, hgnetv2.py:265-275
            bp = f"{sp}.blocks.{bi}"
            identity, feats = x, [x]
            for li in range(n_layers):
                lp = f"{bp}.layers.{li}"
                if light:  # LightConvBNAct, hgnetv2.py:83-112
                    x = _hg_conv(sd, lp + ".conv1", x, 1, act=False, **kw)
                    x = _hg_conv(sd, lp + ".conv2", x, k, groups=mid, **kw)
                else:
                    x = _hg_conv(sd, lp, x, k, **kw)
                feats.append(x)
            x = torch.cat(feats, dim=1)
            x = _hg_conv(sd, bp + ".aggregation.0", x, 1, **kw)  # agg="se": two 1x1 ConvBNAct
            x = _hg_conv(sd, bp + ".aggregation.1", x, 1, **kw)
            if bi > 0:
                x = x + identity
        if si in cfg["return_idx"]:
            outs.append(x)
    return outs


# ======================================================================================
# HybridEncoder -- hybrid_encoder.py
# ======================================================================================
def sincos_pos_embed(w, h, dim, temperature=10000.0):
    """build_2d_sincos_position_embedding, hybrid_encoder.py:425-441. Called as (w, h) with
    meshgrid(arange(w), arange(h), 'ij') flattened, while tokens are row-major (ih*W+iw):
    the axis swap is the reference's behaviour and is kept."""
    gw, gh = torch.meshgrid(torch.arange(int(w), dtype=torch.float32), torch.arange(int(h), dtype=torch.float32), indexing="ij")
    pos_dim = dim // 4
    omega = 1.0 / (temperature ** (torch.arange(pos_dim, dtype=torch.float32) / pos_dim))
    ow = gw.flatten()[:, None] @ omega[None]
    oh = gh.flatten()[:, None] @ omega[None]
    return torch.cat([ow.sin(), ow.cos(), oh.sin(), oh.cos()], dim=1)[None]


def _vgg_block(sd, p, x, training):
    """VGGBlock.forward (training form), hybrid_encoder.py:115-121."""
    return F.silu(_conv_norm(sd, p + ".conv1", x, training, k=3) + _conv_norm(sd, p + ".conv2", x, training, k=1))


def _csp_layer(sd, p, x, n, training):
    """CSPLayer.forward, hybrid_encoder.py:235-239 (expansion 1 => conv3 is Identity)."""
    a = _conv_norm(sd, p + ".conv1", x, training, act="silu")
    for j in range(n):
        a = _vgg_block(sd, f"{p}.bottlenecks.{j}", a, training)
    b = _conv_norm(sd, p + ".conv2", x, training, act="silu")
    return a + b


def _rep_ncsp_elan4(sd, p, x, cfg, training):
    """RepNCSPELAN4.forward, hybrid_encoder.py:203-206."""
    n = cfg["csp_blocks"]
    y = _conv_norm(sd, p + ".cv1", x, training, act="silu")
    half = y.shape[1] // 2
    y0, y1 = y[:, :half], y[:, half:]
    y2 = _conv_norm(sd, p + ".cv2.1", _csp_layer(sd, p + ".cv2.0", y1, n, training), training, k=3, act="silu")
    y3 = _conv_norm(sd, p + ".cv3.1", _csp_layer(sd, p + ".cv3.0", y2, n, training), training, k=3, act="silu")
    return _conv_norm(sd, p + ".cv4", torch.cat([y0, y1, y2, y3], 1), training, act="silu")


def encoder(sd, cfg, feats, training):
    """HybridEncoder.forward, hybrid_encoder.py:443-486. Returns (outs, inner_outs)."""
    hid = cfg["enc_hidden"]
    L = len(feats)
    proj = [_conv_norm(sd, f"encoder.input_proj.{i}", f, training) for i, f in enumerate(feats)]
    # AIFI on the last level (use_encoder_idx = [L-1]); post-norm layer, :274-290
    x = proj[-1]
    B, _, h, w = x.shape
    src = x.flatten(2).permute(0, 2, 1)
    pos = sincos_pos_embed(w, h, hid)
    lp = "encoder.encoder.0.layers.0"
    qk = src + pos
    src = _layer_norm(sd, lp + ".norm1", src + _mha(sd, lp + ".self_attn", qk, qk, src, C.NUM_HEADS))
    ffn = _linear(sd, lp + ".linear2", F.gelu(_linear(sd, lp + ".linear1", src)))
    src = _layer_norm(sd, lp + ".norm2", src + ffn)
    proj[-1] = src.permute(0, 2, 1).reshape(B, hid, h, w).contiguous()
    # top-down FPN, :465-476
    inner = [proj[-1]]
    for idx in range(L - 1, 0, -1):
        j = L - 1 - idx
        high = _conv_norm(sd, f"encoder.lateral_convs.{j}", inner[0], training)
        inner[0] = high
        up = F.interpolate(high, scale_factor=2.0, mode="nearest")
        inner.insert(0, _rep_ncsp_elan4(sd, f"encoder.fpn_blocks.{j}", torch.cat([up, proj[idx - 1]], 1), cfg, training))
    # bottom-up PAN, :478-484
    outs = [inner[0]]
    for idx in range(L - 1):
        dp = f"encoder.downsample_convs.{idx}.0"  # SCDown, :96-103
        d = _conv_norm(sd, dp + ".cv1", outs[-1], training)
        d = _conv_norm(sd, dp + ".cv2", d, training, k=3, stride=2, groups=hid)
        outs.append(_rep_ncsp_elan4(sd, f"encoder.pan_blocks.{idx}", torch.cat([d, inner[idx + 1]], 1), cfg, training))
    return outs, inner


# ======================================================================================
# DFINETransformer -- dfine_decoder.py
# ======================================================================================
def generate_anchors(spatial_shapes, grid_size=0.05, eps=C.ANCHOR_EPS):
    """_generate_anchors, dfine_decoder.py:815-838 -> (anchors [1,S,4] logit space, valid [1,S,1])."""
    anchors = []
    for lvl, (h, w) in enumerate(spatial_shapes):
        gy, gx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
        xy = (torch.stack([gx, gy], -1).unsqueeze(0) + 0.5) / torch.tensor([w, h], dtype=torch.float32)
        wh = torch.ones_like(xy) * grid_size * (2.0**lvl)
        anchors.append(torch.cat([xy, wh], -1).reshape(-1, h * w, 4))
    anchors = torch.cat(anchors, 1)
    valid = ((anchors > eps) & (anchors < 1 - eps)).all(-1, keepdim=True)
    anchors = torch.log(anchors / (1 - anchors))
    return torch.where(valid, anchors, torch.inf), valid


def _msda_layer(sd, p, query, ref_points, value, spatial_shapes, cfg):
    """MSDeformableAttention.forward with 4-d reference boxes, dfine_decoder.py:119-178."""
    B, Lq, _ = query.shape
    nH, P = C.NUM_HEADS, sum(cfg["points"])
    offsets = _linear(sd, p + ".sampling_offsets", query).reshape(B, Lq, nH, P, 2)
    weights = torch.softmax(_linear(sd, p + ".attention_weights", query).reshape(B, Lq, nH, P), dim=-1)
    nps = sd[p + ".num_points_scale"].to(query.dtype).unsqueeze(-1)  # [P,1]
    ref = ref_points[:, :, None, None, :]  # [B,Lq,1,1,4] (same box for every level, :471)
    loc = ref[..., :2] + offsets * nps * ref[..., 2:] * 0.5
    return ops.deformable_attention_core(value, spatial_shapes, loc, weights, list(cfg["points"]))


def _decoder_layer(sd, p, tgt, ref_points, value, spatial_shapes, attn_mask, pos, cfg):
    """TransformerDecoderLayer.forward, dfine_decoder.py:233-255, and Gate :266-271."""
    qk = tgt + pos
    tgt = _layer_norm(sd, p + ".norm1", tgt + _mha(sd, p + ".self_attn", qk, qk, tgt, C.NUM_HEADS, attn_mask))
    cross = _msda_layer(sd, p + ".cross_attn", tgt + pos, ref_points, value, spatial_shapes, cfg)
    gates = torch.sigmoid(_linear(sd, p + ".gateway.gate", torch.cat([tgt, cross], -1)))
    g1, g2 = gates.chunk(2, dim=-1)
    tgt = _layer_norm(sd, p + ".gateway.norm", g1 * tgt + g2 * cross)
    ffn = _linear(sd, p + ".linear2", F.relu(_linear(sd, p + ".linear1", tgt)))
    return _layer_norm(sd, p + ".norm3", (tgt + ffn).clamp(-65504, 65504))


def _lqe(sd, p, scores, corners):
    """LQE.forward, dfine_decoder.py:307-313."""
    B, L, _ = corners.shape
    prob = torch.softmax(corners.reshape(B, L, 4, C.REG_MAX + 1), dim=-1)
    top, _ = prob.topk(4, dim=-1)
    stat = torch.cat([top, top.mean(-1, keepdim=True)], -1).reshape(B, L, -1)
    return scores + _mlp(sd, p + ".reg_conf", stat, 2)


def _pixel_decoder(sd, p, feats, enc_s8, training):
    """MaskPixelDecoder.forward, dfine_decoder.py:355-376."""
    def proj(i):
        return _batch_norm(sd, f"{p}.bn.{i}", F.conv2d(feats[i], sd[f"{p}.lateral.{i}.weight"]), training)

    x = proj(0)
    size = x.shape[-2:]
    for i in range(1, len(feats)):
        x = x + F.interpolate(proj(i), size=size, mode="bilinear", align_corners=False)
    e = _batch_norm(sd, p + ".enc_bn", F.conv2d(enc_s8, sd[p + ".enc_proj.weight"]), training)
    x = x + F.interpolate(e, size=size, mode="bilinear", align_corners=False)
    x = F.conv_transpose2d(x, sd[p + ".upconv.weight"], None, stride=2)
    return F.relu(_batch_norm(sd, p + ".bn1", x, training))


def decoder(sd, cfg, feats, inner_feats, targets, training, enable_mask_head=False, inject=None, num_classes=80):
    """DFINETransformer.forward, dfine_decoder.py:942-1074 (training and eval forms)."""
    inject = inject or {}
    hid, n_layers, Q = cfg["hidden"], cfg["layers"], C.NUM_QUERIES
    # _get_encoder_input :790-813 (input_proj = Identity unless enc_hidden != hidden, x only)
    proj = []
    for i, f in enumerate(feats):
        if cfg["enc_hidden"] != hid:
            f = _batch_norm(sd, f"decoder.input_proj.{i}.norm", F.conv2d(f, sd[f"decoder.input_proj.{i}.conv.weight"]), training)
        proj.append(f)
    spatial_shapes = [tuple(f.shape[2:]) for f in proj]
    memory = torch.cat([f.flatten(2).permute(0, 2, 1) for f in proj], 1)  # [B,S,hid]
    B = memory.shape[0]

    do_masks = enable_mask_head and (
        targets is None or any(t.get("masks") is not None and t["masks"].numel() > 0 for t in targets)
    )  # _should_do_masks :924-935

    # CDN :949-959 (box_noise_scale literal 1.0)
    dn_content = dn_bbox = attn_mask = dn_meta = None
    if training:
        noise = inject.get("cdn_noise")
        if noise is None:
            noise = ops.cdn_noise([len(t["labels"]) for t in targets], num_classes, C.NUM_DENOISING)
        dn_content, dn_bbox, attn_mask, dn_meta = ops.cdn_group(
            targets, num_classes, Q, sd["decoder.denoising_class_embed.weight"], noise, C.NUM_DENOISING, 0.5, 1.0)

    # _get_decoder_input :840-885
    anchors, valid = generate_anchors(spatial_shapes)
    mem = valid.to(memory.dtype) * memory
    out_mem = _layer_norm(sd, "decoder.enc_output.norm", _linear(sd, "decoder.enc_output.proj", mem))
    enc_logits = _linear(sd, "decoder.enc_score_head", out_mem)
    topk_ind = inject.get("topk_ind")
    if topk_ind is None:
        topk_ind = torch.topk(enc_logits.max(-1).values, Q, dim=-1).indices  # _select_topk :894
    if inject.get("return_topk"):  # test hook: just the selected token indices
        return topk_ind
    gather = lambda t: t.gather(1, topk_ind.unsqueeze(-1).expand(-1, -1, t.shape[-1]))
    topk_mem, topk_logits = gather(out_mem), gather(enc_logits)
    topk_anchor = gather(anchors.expand(B, -1, -1))
    enc_bbox_unact = _mlp(sd, "decoder.enc_bbox_head", topk_mem, 3) + topk_anchor
    enc_boxes = torch.sigmoid(enc_bbox_unact)
    content, ref_unact = topk_mem.detach(), enc_bbox_unact.detach()
    if dn_bbox is not None:
        ref_unact = torch.cat([dn_bbox, ref_unact], 1)
        content = torch.cat([dn_content, content], 1)

    # TransformerDecoder.forward :435-530
    value = memory.reshape(B, memory.shape[1], C.NUM_HEADS, -1)  # value_op :416-426, channel-last view
    up, reg_scale = sd["decoder.up"], sd["decoder.reg_scale"]
    project = ops.weighting_function(C.REG_MAX, up, reg_scale)
    out, out_detach, corners_prev = content, 0, 0
    ref_detach = torch.sigmoid(ref_unact)
    boxes_l, logits_l, corners_l, refs_l, hs = [], [], [], [], []
    eval_idx = n_layers - 1
    for i in range(n_layers):
        pos = _mlp(sd, "decoder.query_pos_head", ref_detach, 2).clamp(-10, 10)
        out = _decoder_layer(sd, f"decoder.decoder.layers.{i}", out, ref_detach, value, spatial_shapes, attn_mask, pos, cfg)
        hs.append(out)
        if i == 0:
            pre_boxes = torch.sigmoid(_mlp(sd, "decoder.pre_bbox_head", out, 3) + ops.inverse_sigmoid(ref_detach))
            pre_logits = _linear(sd, "decoder.dec_score_head.0", out)
            ref_initial = pre_boxes.detach()
        corners = _mlp(sd, f"decoder.dec_bbox_head.{i}", out + out_detach, 3) + corners_prev
        box = ops.distance2bbox(ref_initial, ops.integral(corners, project), float(reg_scale[0]))
        if training or i == eval_idx:
            scores = _lqe(sd, f"decoder.decoder.lqe_layers.{i}", _linear(sd, f"decoder.dec_score_head.{i}", out), corners)
            logits_l.append(scores), boxes_l.append(box), corners_l.append(corners), refs_l.append(ref_initial)
        corners_prev, ref_detach, out_detach = corners, box.detach(), out.detach()

    # mask branch :999-1022
    masks_l = dn_masks_l = None
    if do_masks:
        h8, w8 = spatial_shapes[0]
        mem0 = memory[:, : h8 * w8].permute(0, 2, 1).reshape(B, hid, h8, w8)
        mask_feat = _pixel_decoder(sd, "decoder.pixel_decoder", inner_feats, mem0, training)
        mask_of = lambda h: torch.einsum("bqc,bchw->bqhw", _mlp(sd, "decoder.mask_head", h, 3), mask_feat)

    if not training:
        res = {"pred_logits": logits_l[-1], "pred_boxes": boxes_l[-1]}
        if do_masks:
            res["pred_masks"] = torch.sigmoid(mask_of(hs[-1]))
        return res

    Nd = dn_meta["dn_num_split"][0] if dn_meta is not None else 0
    split = lambda t: (t[:, :Nd], t[:, Nd:])
    dn, reg = {}, {}
    for name, lst in (("logits", logits_l), ("boxes", boxes_l), ("corners", corners_l), ("refs", refs_l), ("hs", hs)):
        pairs = [split(t) for t in lst]
        dn[name], reg[name] = [a for a, _ in pairs], [b for _, b in pairs]
    dn_pre_logits, pre_logits = split(pre_logits)
    dn_pre_boxes, pre_boxes = split(pre_boxes)
    if do_masks:
        masks_l = [mask_of(h) for h in reg["hs"]]
        if dn_meta is not None:
            dn_masks_l = [mask_of(h) for h in dn["hs"]]

    def layer_dict(src, i, teacher_c, teacher_l, masks=None):
        d = {"pred_logits": src["logits"][i], "pred_boxes": src["boxes"][i], "pred_corners": src["corners"][i],
             "ref_points": src["refs"][i], "teacher_corners": teacher_c, "teacher_logits": teacher_l}
        if masks is not None:
            d["pred_masks"] = masks[i]
        return d

    res = {"pred_logits": reg["logits"][-1], "pred_boxes": reg["boxes"][-1], "pred_corners": reg["corners"][-1],
           "ref_points": reg["refs"][-1], "up": up, "reg_scale": reg_scale}
    if do_masks:
        res["pred_masks"] = masks_l[-1]
    res["aux_outputs"] = [layer_dict(reg, i, reg["corners"][-1], reg["logits"][-1], masks_l) for i in range(n_layers - 1)]
    res["enc_aux_outputs"] = [{"pred_logits": topk_logits, "pred_boxes": enc_boxes}]
    res["pre_outputs"] = {"pred_logits": pre_logits, "pred_boxes": pre_boxes}
    res["enc_meta"] = {"class_agnostic": False}
    if dn_meta is not None:
        # all L layers for detection; with the mask head the reference's zip over aux masks
        # truncates to L-1 and carries the last layer's masks separately (:1059-1070)
        n_dn = n_layers - 1 if do_masks else n_layers
        res["dn_outputs"] = [layer_dict(dn, i, dn["corners"][-1], dn["logits"][-1], dn_masks_l) for i in range(n_dn)]
        if do_masks and dn_masks_l is not None:
            res["dn_pred_masks"] = dn_masks_l[-1]
        res["dn_pre_outputs"] = {"pred_logits": dn_pre_logits, "pred_boxes": dn_pre_boxes}
        res["dn_meta"] = dn_meta
    return res


def forward(sd, cfg, x, targets=None, training=True, enable_mask_head=False, inject=None, num_classes=80):
    """DFINE.forward, dfine.py:37-41."""
    feats = backbone(sd, cfg, x, training)
    outs, inner = encoder(sd, cfg, feats, training)
    return decoder(sd, cfg, outs, inner, targets, training, enable_mask_head, inject, num_classes)
