"""One optimizer step of the training loop, mirroring the reference's hot loop
(src/dl/train.py:488-562: autocast forward, criterion outside autocast, backward, clip 0.1,
AdamW over four groups, OneCycle LR, EMA) without its per-iteration host work (``loss.item()``,
an ``nvidia-smi`` subprocess, per-tensor target copies).
"""
import math
from copy import deepcopy

import torch

from .dfine import build_loss, build_model, build_optimizer
from .targets import TargetBatch


class ModelEMA:
    """Exponential moving average of the weights (train.py:51-72): decay
    momentum*(1-exp(-iters/2000)) over every floating-point state_dict entry."""

    def __init__(self, model, momentum=0.9998):
        self.model = deepcopy(model).eval()
        for p in self.model.parameters():
            p.requires_grad_(False)
        self.momentum = momentum
        self._pairs = None

    def decay(self, iters):
        return self.momentum * (1 - math.exp(-iters / 2000))

    def update(self, iters, model):
        if self._pairs is None:
            src = model.state_dict()
            pairs = [(t, src[k]) for k, t in self.model.state_dict().items() if t.dtype.is_floating_point]
            self._pairs = ([a for a, _ in pairs], [b for _, b in pairs])
        d = self.decay(iters)
        ema, cur = self._pairs
        with torch.no_grad():
            torch._foreach_mul_(ema, d)
            torch._foreach_add_(ema, cur, alpha=1.0 - d)


class TrainStep:
    """model + criterion + optimizer + scheduler + EMA; ``step(images, targets)`` runs one
    iteration and returns the (device-resident) total loss. No host synchronisation inside."""

    def __init__(self, model_name="m", num_classes=80, enable_mask_head=False, img_size=(640, 640),
                 device="cuda", base_lr=1.5e-4, backbone_lr=2e-5, betas=(0.9, 0.999), weight_decay=1.25e-4,
                 clip_max_norm=0.1, use_ema=True, ema_momentum=0.9998, amp=True, total_steps=10000,
                 pct_start=0.1, grad_reducer=None):
        self.device = torch.device(device)
        self.num_classes = num_classes
        self.model = build_model(model_name, num_classes, enable_mask_head, self.device, img_size=list(img_size))
        self.criterion = build_loss(model_name, num_classes, 0.0, enable_mask_head)
        self.optimizer = build_optimizer(self.model, lr=base_lr, backbone_lr=backbone_lr, betas=betas,
                                         weight_decay=weight_decay, base_lr=base_lr)
        max_lr = base_lr * 2
        if model_name in ("l", "x"):
            max_lr = [backbone_lr * 2, backbone_lr * 2, base_lr * 2, base_lr * 2]
        self.scheduler = torch.optim.lr_scheduler.OneCycleLR(
            self.optimizer, max_lr=max_lr, total_steps=total_steps, pct_start=pct_start, cycle_momentum=False)
        self.ema = ModelEMA(self.model, ema_momentum) if use_ema else None
        self.clip_max_norm = clip_max_norm
        self.amp = amp and self.device.type == "cuda"
        self.iters = 0
        self.grad_reducer = grad_reducer(self.model) if grad_reducer is not None else None
        self.model.train()

    def step(self, images, targets):
        model = self.model
        tb = TargetBatch.from_list(targets, self.num_classes, self.device)
        images = images.to(self.device, non_blocking=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.amp):
            out = model(images, targets=tb)
        losses = self.criterion(out)  # fp32, outside autocast like train.py:548-549
        loss = sum(losses.values())
        loss.backward()
        if self.grad_reducer is not None:
            self.grad_reducer.finish()  # wait for the bucketed all-reduces issued during backward
        if self.clip_max_norm:
            torch.nn.utils.clip_grad_norm_(model.parameters(), self.clip_max_norm, foreach=True)
        self.optimizer.step()
        self.scheduler.step()
        if self.grad_reducer is not None:
            self.grad_reducer.zero_grad()  # grads are views into the reducer's flat buckets
        else:
            self.optimizer.zero_grad(set_to_none=True)
        self.iters += 1
        if self.ema is not None:
            self.ema.update(self.iters, model)
        return loss.detach()
