"""Padded, device-resident view of a batch's ground truth.

The reference hands the model and the criterion a Python list of per-image dicts
(src/dl/dataset.py:617-640) and re-reads it tensor by tensor, with ``.item()``/``nonzero``
host syncs along the way. Here the list is packed ONCE per step into fixed-shape tensors
(only the per-image counts stay on the host, where ``len()`` already knows them), so
everything downstream is dense, mask-driven and free of host round-trips.
"""
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class TargetBatch:
    labels: torch.Tensor  # int64 [B, G]   padded with num_classes
    boxes: torch.Tensor  # f32   [B, G, 4] cxcywh in [0,1], padded with 0
    valid: torch.Tensor  # bool  [B, G]
    num_gts: List[int]  # host
    counts: torch.Tensor = None  # int32 [B] on the device (same numbers as num_gts)
    masks: Optional[List[Optional[torch.Tensor]]] = None  # per image uint8 [n, H, W] (segment task)
    mask_tgt: Optional[torch.Tensor] = None  # f32 [B, G, Hm, Wm]: masks resized to the prediction size, see dense_mask_targets

    @property
    def max_gt(self):
        return self.labels.shape[1]

    @property
    def total(self):
        return sum(self.num_gts)

    @staticmethod
    def from_list(targets, num_classes, device):
        """targets: list of {"labels" [n], "boxes" [n,4], optional "masks" [n,H,W]} (CPU or GPU)."""
        if isinstance(targets, TargetBatch):
            return targets
        num_gts = [int(t["labels"].shape[0]) for t in targets]
        B, G = len(targets), max(max(num_gts, default=0), 1)
        labels = torch.full((B, G), num_classes, dtype=torch.int64)
        boxes = torch.zeros(B, G, 4, dtype=torch.float32)
        on_host = all(not t["labels"].is_cuda for t in targets)
        if on_host:  # one pinned staging buffer, one H2D copy per tensor kind
            for i, t in enumerate(targets):
                n = num_gts[i]
                if n:
                    labels[i, :n] = t["labels"]
                    boxes[i, :n] = t["boxes"]
            if torch.device(device).type == "cuda":
                labels, boxes = labels.pin_memory(), boxes.pin_memory()
            labels = labels.to(device, non_blocking=True)
            boxes = boxes.to(device, non_blocking=True)
        else:
            labels, boxes = labels.to(device), boxes.to(device)
            for i, t in enumerate(targets):
                n = num_gts[i]
                if n:
                    labels[i, :n] = t["labels"].to(device)
                    boxes[i, :n] = t["boxes"].to(device)
        counts = torch.tensor(num_gts, dtype=torch.int64)
        valid = (torch.arange(G)[None, :] < counts[:, None]).to(device, non_blocking=True)
        masks = None
        if any(t.get("masks") is not None and t["masks"].numel() > 0 for t in targets):
            masks = [t.get("masks") for t in targets]
        return TargetBatch(labels, boxes, valid, num_gts, counts.to(torch.int32).to(device, non_blocking=True), masks)

    def dense_mask_targets(self, out_h, out_w, device):
        """[B, G, out_h, out_w] fp32 targets, computed once per batch (``mask_tgt``); a captured step is
        handed the tensor ready-made (train.TrainStep) so the staging stays outside the graph. On the GPU the uint8
        masks of the whole batch cross PCIe once through a pinned double buffer and are resized by one kernel
        (``MaskStager``); ``mask_targets`` below is the per-image formulation it is tested against."""
        if self.mask_tgt is None or tuple(self.mask_tgt.shape[-2:]) != (out_h, out_w):
            staged = None
            if torch.device(device).type == "cuda":
                staged = MaskStager.for_device(device).stage(self.masks, self.labels.shape[1], out_h, out_w)
            self.mask_tgt = staged if staged is not None else self.mask_targets(out_h, out_w, device)[0]
        return self.mask_tgt

    def mask_targets(self, out_h, out_w, device):
        """GT masks nearest-resized to the prediction size (dfine_criterion.py:239-270), padded to
        [B, G, out_h, out_w] float32. Images without masks contribute zeros (and are excluded by
        ``has_masks``)."""
        B, G = self.labels.shape
        out = torch.zeros(B, G, out_h, out_w, dtype=torch.float32, device=device)
        has = torch.zeros(B, dtype=torch.bool)
        for i, m in enumerate(self.masks or []):
            if m is None or m.numel() == 0 or m.dim() != 3:
                continue
            m = m.to(device, non_blocking=True)
            out[i, : m.shape[0]] = F.interpolate(m[None].float(), size=(out_h, out_w), mode="nearest")[0].clamp_(0, 1)
            has[i] = True
        return out, has.to(device)


def nearest_source_index(n_in, n_via, n_out):
    """Source index of every output position of F.interpolate(mode="nearest") from n_in to n_out samples -- PyTorch's
    min(floor(dst * float32(in / out)), in - 1) in float32 -- optionally composed with a first nearest resize n_in -> n_via
    (the multiscale augmentation, src/dl/dataset.py:645-670; n_via = 0: none). int64 tensor [n_out]."""
    n_via = n_via or n_in
    i = np.minimum(np.floor(np.arange(n_out, dtype=np.float32) * (np.float32(n_via) / np.float32(n_out))).astype(np.int64), n_via - 1)
    if n_via != n_in:
        i = np.minimum(np.floor(i.astype(np.float32) * (np.float32(n_in) / np.float32(n_via))).astype(np.int64), n_in - 1)
    return torch.from_numpy(i)


class MaskStager:
    """Input staging of the ground-truth masks: the per-image uint8 masks [n_i, H, W] of a batch are
    packed into ONE pinned host buffer [B, G, H, W] (two buffers, alternating, each guarded by an event so the host never
    overwrites a buffer whose copy is still in flight), copied to the device in one async transfer and turned into the
    loss targets [B, G, Hm, Wm] fp32 by one HIP kernel (dfine_mask_targets_nearest, bit-exact with the reference's
    F.interpolate(nearest)). The reference moves every image's masks with its own ``.to(device)`` inside the criterion."""

    _instances = {}

    @classmethod
    def for_device(cls, device):
        key = str(torch.device(device))
        if key not in cls._instances:
            cls._instances[key] = cls(device)
        return cls._instances[key]

    def __init__(self, device):
        self.device = torch.device(device)
        self.host, self.dev, self.events, self.turn = [None, None], [None, None], [None, None], 0

    def stage(self, masks, G, out_h, out_w, via=None):
        """masks: list (per image) of uint8 [n_i, H, W] CPU tensors or None -> fp32 [B, G, out_h, out_w] on the device, or
        None when the masks are not host uint8 maps of one size (the caller then takes the generic path). ``via`` = (Hv, Wv):
        the batch size after the collate function's multiscale augmentation, which resizes the masks (nearest) once before
        the criterion resizes them again -- pass it INSTEAD of resizing the masks on the host."""
        from . import _lib

        shapes = {tuple(m.shape[-2:]) for m in masks if m is not None and m.numel() > 0}
        if len(shapes) != 1 or any(m is not None and (m.is_cuda or m.dtype != torch.uint8 or m.dim() != 3) for m in masks):
            return None
        (H, W), B = shapes.pop(), len(masks)
        slot = self.turn
        self.turn ^= 1
        hv, wv = (int(via[0]), int(via[1])) if via is not None else (0, 0)
        # The loss reads only out_h x out_w pixels of every mask (nearest): when that is fewer than the mask has, the host picks
        # exactly those pixels while it packs the batch -- PyTorch's nearest source index in float32, composed with the
        # multiscale resize like dfine_mask_targets_nearest does -- and 1/16 of the bytes (640^2 -> 160^2) cross PCIe; the
        # kernel then only converts. Otherwise the whole masks travel and the kernel does the selection.
        select = out_h * out_w < H * W
        if select:
            iy, ix = nearest_source_index(H, hv, out_h).numpy(), nearest_source_index(W, wv, out_w).numpy()
            H, W, hv, wv = out_h, out_w, 0, 0
        n = B * G * H * W
        if self.host[slot] is None or self.host[slot].numel() < n:
            self.host[slot] = torch.zeros(n, dtype=torch.uint8).pin_memory()
            self.dev[slot] = torch.zeros(n, dtype=torch.uint8, device=self.device)
        if self.events[slot] is not None:
            self.events[slot].synchronize()  # the transfer that last read this pinned buffer has finished
        host = self.host[slot][:n].view(B, G, H, W)
        host_np = host.numpy()  # the same pinned memory
        for i, m in enumerate(masks):
            k = 0 if m is None else min(int(m.shape[0]), G)
            if k:
                if select:
                    # rows first (contiguous 1-row copies), then columns: 0.3 ms per image; torch's index_select takes 50
                    np.take(np.take(m[:k].numpy(), iy, axis=1), ix, axis=2, out=host_np[i, :k])
                else:
                    host[i, :k].copy_(m[:k])
            host[i, k:].zero_()
        dev = self.dev[slot][:n]
        dev.copy_(self.host[slot][:n], non_blocking=True)
        ev = self.events[slot] or torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.events[slot] = ev
        out = torch.empty(B, G, out_h, out_w, dtype=torch.float32, device=self.device)
        lib = _lib.load()
        rc = lib.dfine_mask_targets_nearest(_lib.ptr(dev), _lib.ptr(out), B * G, H, W, hv, wv, out_h, out_w, _lib.stream_ptr())
        _lib.check(rc, "dfine_mask_targets_nearest")
        return out


class ImageStager:
    """Input staging of the images. The reference's loader normalises every image on the host
    (A.Normalize + ToTensorV2, src/dl/dataset.py:128-131: uint8 HWC -> float32 CHW), stacks them, sometimes resizes the whole
    float batch (multiscale, dataset.py:645-670) and then copies 4 bytes per value to the device. Here the decoded uint8 pixels go
    into ONE pinned buffer [B, H, W, 3] (two, alternating, event-guarded like ``MaskStager``), cross PCIe at 1 byte per value,
    and ONE kernel (dfine_stage_images) normalises, optionally resizes (PyTorch's bilinear rule) and writes the channels_last
    tensor the stem reads."""

    _instances = {}

    @classmethod
    def for_device(cls, device):
        key = str(torch.device(device))
        if key not in cls._instances:
            cls._instances[key] = cls(device)
        return cls._instances[key]

    def __init__(self, device):
        self.device = torch.device(device)
        self.host, self.dev, self.events, self.turn = [None, None], [None, None], [None, None], 0

    def stage(self, images, size=None, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0), dtype=torch.float32):
        """images: list of uint8 [H, W, 3] CPU tensors / numpy arrays of ONE size (or one [B, H, W, 3] tensor) ->
        [B, 3, OH, OW] ``dtype`` (float32 or bfloat16) in channels_last memory on the device; ``size`` = (OH, OW) resizes."""
        import ctypes

        from . import _lib

        if not isinstance(images, torch.Tensor):
            images = [torch.as_tensor(im) for im in images]
            if len({tuple(im.shape) for im in images}) != 1:
                raise ValueError("ImageStager: the images of a batch must have one size (the loader letterboxes them)")
            B, (H, W, C) = len(images), images[0].shape
        else:
            B, H, W, C = images.shape
        if C != 3 or (images[0] if not isinstance(images, torch.Tensor) else images).dtype != torch.uint8:
            raise ValueError("ImageStager: uint8 HWC images with 3 channels expected")
        OH, OW = (int(size[0]), int(size[1])) if size is not None else (H, W)
        slot = self.turn
        self.turn ^= 1
        n = B * H * W * 3
        if self.host[slot] is None or self.host[slot].numel() < n:
            self.host[slot] = torch.empty(n, dtype=torch.uint8).pin_memory()
            self.dev[slot] = torch.empty(n, dtype=torch.uint8, device=self.device)
        if self.events[slot] is not None:
            self.events[slot].synchronize()  # the transfer that last read this pinned buffer has finished
        host = self.host[slot][:n].view(B, H, W, 3)
        if isinstance(images, torch.Tensor):
            host.copy_(images)
        else:
            for i, im in enumerate(images):
                host[i].copy_(im)
        dev = self.dev[slot][:n]
        dev.copy_(self.host[slot][:n], non_blocking=True)
        ev = self.events[slot] or torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.events[slot] = ev
        out = torch.empty(B, OH, OW, 3, dtype=dtype, device=self.device)
        lib = _lib.load()
        m3, s3 = (ctypes.c_float * 3)(*[float(v) for v in mean]), (ctypes.c_float * 3)(*[float(v) for v in std])
        rc = lib.dfine_stage_images(_lib.ptr(dev), _lib.ptr(out), _lib.dtype_code(out), B, H, W, OH, OW, m3, s3, _lib.stream_ptr())
        _lib.check(rc, "dfine_stage_images")
        return out.permute(0, 3, 1, 2)
