"""Model / loss / optimizer builders with the reference's signatures
(src/d_fine/dfine.py:51-124), so ``src/dl/train``-style callers can switch imports:

    build_model(model_name, num_classes, enable_mask_head, device, img_size=None, pretrained_model_path=None)
    build_loss(model_name, num_classes, label_smoothing, enable_mask_head)
    build_optimizer(model, lr, backbone_lr, betas, weight_decay, base_lr)
"""
from pathlib import Path

import torch
import torch.nn as nn

from .backbone import HGNetv2
from .configs import MODELS, CriterionSpec
from .criterion import DFINECriterion
from .decoder import DFINETransformer
from .encoder import HybridEncoder


class DFINE(nn.Module):
    """backbone -> encoder -> decoder (dfine.py:19-41). Feature maps are channel-last."""

    def __init__(self, backbone, encoder, decoder):
        super().__init__()
        self.backbone = backbone
        self.decoder = decoder
        self.encoder = encoder

    def forward(self, x, targets=None, inject=None):
        x = x.contiguous(memory_format=torch.channels_last)
        outs, inner = self.encoder(self.backbone(x))
        return self.decoder(outs, inner, targets, inject)


def _deploy(self):
    """Inference form (reference DFINE.deploy, dfine.py:43-48): eval mode and every re-parameterisable block folded."""
    self.eval()
    for m in list(self.modules()):
        if hasattr(m, "convert_to_deploy"):
            m.convert_to_deploy()
    return self


DFINE.deploy = _deploy


def load_checkpoint_state(model, path):
    """Accepts the layouts the reference's loader accepts (src/d_fine/utils.py:156-181):
    {"ema": {"module": sd}}, {"model": sd} or a bare state_dict; tensors only
    (``weights_only=True``), non-strict like the reference."""
    state = torch.load(path, map_location="cpu", weights_only=True)
    if "ema" in state and isinstance(state["ema"], dict) and "module" in state["ema"]:
        state = state["ema"]["module"]
    elif "model" in state:
        state = state["model"]
    own = model.state_dict()
    usable = {k: v for k, v in state.items() if k in own and own[k].shape == v.shape}
    model.load_state_dict(usable, strict=False)
    return model


def build_model(model_name, num_classes, enable_mask_head, device, img_size=None, pretrained_model_path=None):
    spec = MODELS[model_name]
    model = DFINE(
        HGNetv2(spec),
        HybridEncoder(spec),
        DFINETransformer(spec, num_classes, eval_spatial_size=img_size, enable_mask_head=enable_mask_head),
    )
    if pretrained_model_path:
        if not Path(pretrained_model_path).exists():
            raise FileNotFoundError(f"{pretrained_model_path} does not exist")
        load_checkpoint_state(model, str(pretrained_model_path))
    return model.to(device).to(memory_format=torch.channels_last)


def build_loss(model_name, num_classes, label_smoothing, enable_mask_head):
    spec = CriterionSpec()
    losses = spec.losses + (("masks",) if enable_mask_head else ())
    return DFINECriterion(spec, num_classes=num_classes, losses=losses, label_smoothing=label_smoothing)


def build_optimizer(model, lr, backbone_lr, betas, weight_decay, base_lr):
    """AdamW with the reference's four groups (dfine.py:87-124): backbone weights / backbone
    norms (no decay) / encoder+decoder norms and biases (no decay) / the rest."""
    groups = {"backbone": [], "backbone_norm": [], "encdec_norm_bias": [], "rest": []}
    for name, p in model.named_parameters():
        is_norm = "norm" in name or "bn" in name
        if "backbone" in name:
            groups["backbone_norm" if is_norm else "backbone"].append(p)
        elif ("encoder" in name or "decoder" in name) and (is_norm or "bias" in name):
            groups["encdec_norm_bias"].append(p)
        else:
            groups["rest"].append(p)
    param_groups = [
        {"params": groups["backbone"], "lr": backbone_lr, "initial_lr": backbone_lr},
        {"params": groups["backbone_norm"], "lr": backbone_lr, "weight_decay": 0.0, "initial_lr": backbone_lr},
        {"params": groups["encdec_norm_bias"], "weight_decay": 0.0, "lr": base_lr, "initial_lr": base_lr},
        {"params": groups["rest"], "lr": base_lr, "initial_lr": base_lr},
    ]
    return torch.optim.AdamW(param_groups, lr=lr, betas=betas, weight_decay=weight_decay)
