"""Deterministic, name-hashed weights for parity fixtures. TEST INFRASTRUCTURE ONLY.

The reference's default init is degenerate for goldens (zero-initialised box heads, LQE, gates
and attention weights: dfine_decoder.py:102,116-117,263-264,304-305,724-734 => pred_corners
== 0 and DDF takes its torch.equal shortcut), and pretrained weights are absent. So every
float tensor of a state_dict is overwritten by a function of (its name, its shape) only:
numpy PCG64 seeded with crc32(name). The GPU box regenerates identical weights from the
names/shapes of the product model's own state_dict; nothing has to be shipped.
"""
import zlib

import numpy as np
import torch

# derived buffers / constants that keep the value the constructor gave them
KEEP_SUFFIXES = ("num_batches_tracked", "num_points_scale")
# (up / reg_scale are ONE Parameter each, registered on DFINETransformer and again on its
# TransformerDecoder, dfine_decoder.py:597-598,400: both names appear in the state_dict)
KEEP_NAMES = ("decoder.up", "decoder.reg_scale", "decoder.decoder.up", "decoder.decoder.reg_scale",
              "decoder.anchors", "decoder.valid_mask")


def _rng(name):
    return np.random.Generator(np.random.PCG64(zlib.crc32(name.encode())))


def _value(name, shape):
    r = _rng(name)
    leaf = name.rsplit(".", 1)[-1]
    n = int(np.prod(shape)) if len(shape) else 1
    is_norm = any(t in name for t in (".bn", ".norm", "_bn", "bn1"))
    if leaf == "running_var":
        return r.uniform(0.5, 1.5, n)
    if leaf == "running_mean":
        return r.normal(0, 0.1, n)
    if name.endswith("lab.scale"):
        return r.uniform(0.8, 1.2, n)
    if name.endswith("lab.bias"):
        return r.normal(0, 0.1, n)
    if is_norm and leaf == "weight" and len(shape) == 1:
        return r.uniform(0.5, 1.5, n)
    if leaf == "bias" and "score_head" in name:  # low prior like the reference's -log(99) init
        return r.normal(-4.0, 0.3, n)
    if leaf == "bias" or leaf == "in_proj_bias":
        return r.normal(0, 0.1, n)
    if name.endswith("denoising_class_embed.weight"):
        v = r.normal(0, 1.0, shape)
        v[-1] = 0.0  # padding_idx row
        return v
    if len(shape) >= 2:  # conv [out,in/g,kh,kw] / linear [out,in] / conv-transpose [in,out,kh,kw]
        fan_in = int(np.prod(shape[1:]))
        # class / corner / mask-embedding heads: small, so sigmoids and the 33-bin softmax are
        # not saturated (the interesting regime for loss parity)
        gain = 0.25 if any(t in name for t in ("score_head", "bbox_head", "reg_conf", "mask_head")) else 1.0
        v = r.normal(0, gain / np.sqrt(fan_in), n).reshape(shape)
        if len(shape) == 4:
            # zero-mean filters: activations are all-positive (ReLU/SiLU + LAB), so a filter with a
            # non-zero weight sum puts a large common offset on its output; BatchNorm then divides
            # by a std much smaller than that offset and amplifies bf16 rounding ~50x (measured:
            # 0.9 % error into stem2a's BN, 42 % out). Trained filters feeding a BN do not do that.
            v = v - v.mean(axis=tuple(range(1, 4)), keepdims=True)
        return v
    return r.normal(0, 0.1, n)


def _value_wellcond(name, shape):
    """A WELL-CONDITIONED variant for low-precision parity (fixtures *_wellcond): the default weights above make every conv a
    difference of large positive terms (zero-mean filters on post-ReLU/SiLU activations), which amplifies any relative input
    error by a few percent PER LAYER -- 60 layers on, a correct bf16 pipeline is 30-80 % from fp32 (measured on the
    reference's own autocast and on this repo's). Here every convolution has non-negative weights that sum to
    ~1 over the fan-in and every BatchNorm (used in eval mode) is y = gamma x + beta with beta >= 0 and unit statistics: all
    activations stay positive and O(1), no layer cancels, and rounding errors add instead of multiplying. Linear layers
    (followed by LayerNorm / softmax) keep the default values."""
    r = _rng("wellcond/" + name)
    leaf = name.rsplit(".", 1)[-1]
    n = int(np.prod(shape)) if len(shape) else 1
    is_norm = any(t in name for t in (".bn", ".norm", "_bn", "bn1"))
    if leaf == "running_var":
        return np.ones(n)
    if leaf == "running_mean":
        return np.zeros(n)
    if name.endswith("lab.scale"):
        return r.uniform(0.8, 1.2, n)
    if name.endswith("lab.bias"):
        return r.uniform(0.0, 0.1, n)
    if is_norm and leaf == "weight" and len(shape) == 1:
        return r.uniform(0.8, 1.2, n)
    if is_norm and leaf == "bias" and len(shape) == 1:
        return r.uniform(0.0, 0.1, n)
    if len(shape) == 4:
        fan_in = shape[0] if "upconv" in name else int(np.prod(shape[1:]))  # ConvTranspose2d [in, out, 2, 2]: one tap per input channel
        return np.abs(r.normal(0, 1.0, n)).reshape(shape) / (0.798 * fan_in)
    return _value(name, shape)


def fill_state_dict(state_dict, conditioning="default"):
    """Returns a new dict name -> tensor with every float entry re-generated (same dtype/shape). ``conditioning="wellcond"``:
    see _value_wellcond."""
    out = {}
    gen = _value if conditioning == "default" else _value_wellcond
    for name, t in state_dict.items():
        if (not t.dtype.is_floating_point) or name in KEEP_NAMES or name.endswith(KEEP_SUFFIXES):
            out[name] = t.detach().clone()
            continue
        v = np.asarray(gen(name, tuple(t.shape)), dtype=np.float64).reshape(tuple(t.shape))
        out[name] = torch.from_numpy(v.astype(np.float32)).to(t.dtype)
    return out


def synthetic_images(batch, height, width, seed):
    """U[0,1) float32 [B,3,H,W] from a numpy PCG64 stream (regenerated on the GPU box)."""
    r = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(r.random((batch, 3, height, width), dtype=np.float32))


def synthetic_targets(num_gts, num_classes, seed, with_masks=False, height=None, width=None):
    """COCO-shaped targets: labels U{0..C-1}; boxes cxcywh with centre U[0.2,0.8],
    size U[0.05,0.3]; masks = filled box rectangles (uint8 [n,H,W])."""
    r = np.random.Generator(np.random.PCG64(seed))
    out = []
    for n in num_gts:
        labels = torch.from_numpy(r.integers(0, num_classes, n).astype(np.int64))
        boxes = np.concatenate([r.uniform(0.2, 0.8, (n, 2)), r.uniform(0.05, 0.3, (n, 2))], 1).astype(np.float32)
        t = {"labels": labels, "boxes": torch.from_numpy(boxes)}
        if with_masks:
            m = np.zeros((n, height, width), dtype=np.uint8)
            for k, (cx, cy, w, h) in enumerate(boxes):
                x0, x1 = int((cx - w / 2) * width), int(np.ceil((cx + w / 2) * width))
                y0, y1 = int((cy - h / 2) * height), int(np.ceil((cy + h / 2) * height))
                m[k, max(y0, 0) : y1, max(x0, 0) : x1] = 1
            t["masks"] = torch.from_numpy(m)
            t["orig_size"] = torch.tensor([height, width])
        out.append(t)
    return out
