"""Evaluation post-processing on the device: the reference's ``Trainer.preds_postprocess``
(src/dl/train.py:227-319, ``use_focal_loss=True`` branch) with the same signature and the same result
records, computed by two HIP kernels (csrc/postprocess.hip) instead of sigmoid + top-k + per-image numpy
box loops + two half-precision ``F.interpolate`` calls per image + host-side mask clean-up.

    results = preds_postprocess(inputs, outputs, orig_sizes, num_labels, keep_ratio, conf_thresh)
    results[b] = {"labels", "boxes", "scores", "all_boxes", "all_scores", "all_labels"[, "masks"]}

Records stay on the device unless ``to_cpu=True`` (the reference's records live on the CPU).
"""
from typing import Dict, List

import torch

from . import _lib


@torch.no_grad()
def preds_postprocess(inputs: torch.Tensor, outputs: Dict[str, torch.Tensor], orig_sizes: torch.Tensor, num_labels: int,
                      keep_ratio: bool, conf_thresh: float, num_top_queries: int = 300, use_focal_loss: bool = True,
                      to_cpu: bool = False) -> List[Dict[str, torch.Tensor]]:
    if not use_focal_loss:
        raise NotImplementedError("only the focal-loss (sigmoid) head of D-FINE is on this path (reference default)")
    logits, boxes = outputs["pred_logits"], outputs["pred_boxes"]
    _lib.require_gpu(logits, boxes)
    lib = _lib.load()
    pred_masks = outputs.get("pred_masks")
    B, Q, C = logits.shape
    if C != num_labels:
        raise ValueError(f"pred_logits has {C} classes, num_labels={num_labels}")
    proc_h, proc_w = int(inputs.shape[2]), int(inputs.shape[3])
    K = min(num_top_queries, Q * C)
    dev = logits.device
    if logits.dtype not in (torch.float32, torch.bfloat16):
        logits = logits.float()
    logits = logits.contiguous()
    boxes = boxes.float().contiguous()
    sizes_host = orig_sizes.to("cpu", torch.int64)
    sizes = sizes_host.to(dev).contiguous()
    scores = torch.empty(B, K, dtype=torch.float32, device=dev)
    labels = torch.empty(B, K, dtype=torch.int64, device=dev)
    qidx = torch.empty(B, K, dtype=torch.int64, device=dev)
    all_boxes = torch.empty(B, K, 4, dtype=torch.float32, device=dev)
    rc = lib.dfine_postprocess_topk(_lib.ptr(logits), _lib.dtype_code(logits), _lib.ptr(boxes), _lib.ptr(sizes), _lib.ptr(scores),
                                    _lib.ptr(labels), _lib.ptr(qidx), _lib.ptr(all_boxes), B, Q, C, K, proc_h, proc_w, int(bool(keep_ratio)),
                                    _lib.stream_ptr())
    _lib.check(rc, "dfine_postprocess_topk")
    keep_counts = (scores >= conf_thresh).sum(1).tolist()  # scores are sorted: the kept records are a prefix (one sync)
    if pred_masks is not None:
        if pred_masks.dtype not in (torch.float32, torch.bfloat16):
            pred_masks = pred_masks.float()
        pred_masks = pred_masks.contiguous()
    move = (lambda t: t.cpu()) if to_cpu else (lambda t: t)
    results = []
    for b in range(B):
        n = int(keep_counts[b])
        out = {"labels": move(labels[b, :n]), "boxes": move(all_boxes[b, :n]), "scores": move(scores[b, :n]),
               "all_boxes": move(all_boxes[b]), "all_scores": move(scores[b]), "all_labels": move(labels[b])}
        if pred_masks is not None and n > 0:
            H0, W0 = int(sizes_host[b, 0]), int(sizes_host[b, 1])
            Hm, Wm = pred_masks.shape[-2:]
            masks = torch.empty(n, H0, W0, dtype=torch.uint8, device=dev)
            rc = lib.dfine_postprocess_masks(_lib.ptr(pred_masks[b]), _lib.dtype_code(pred_masks), _lib.ptr(qidx[b]), _lib.ptr(all_boxes[b]),
                                             _lib.ptr(masks), n, Hm, Wm, proc_h, proc_w, H0, W0, int(bool(keep_ratio)), float(conf_thresh),
                                             _lib.stream_ptr())
            _lib.check(rc, "dfine_postprocess_masks")
            out["masks"] = move(masks)
        results.append(out)
    return results
