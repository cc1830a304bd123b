"""Synthetic COCO-shaped batches: the reference only reads YOLO-format datasets
from disk (src/dl/dataset.py); the benchmark needs the collate layout of dataset.py:617-640
(images f32 [B,3,H,W] in [0,1), a list of per-image {"labels","boxes"[,"masks","orig_size"]})
without any I/O."""
import torch


def synthetic_batch(batch, height, width, num_classes=80, gts_per_image=8, with_masks=False,
                    seed=42, device="cpu", random_counts=False):
    g = torch.Generator().manual_seed(seed)
    images = torch.rand(batch, 3, height, width, generator=g)
    targets = []
    for _ in range(batch):
        n = int(torch.randint(1, 21, (1,), generator=g)) if random_counts else gts_per_image
        labels = torch.randint(0, num_classes, (n,), generator=g)
        boxes = torch.cat([torch.rand(n, 2, generator=g) * 0.6 + 0.2, torch.rand(n, 2, generator=g) * 0.25 + 0.05], 1)
        t = {"labels": labels, "boxes": boxes}
        if with_masks:  # filled box rectangles
            ys = torch.arange(height).view(1, height, 1)
            xs = torch.arange(width).view(1, 1, width)
            x0, x1 = ((boxes[:, 0] - boxes[:, 2] / 2) * width).view(n, 1, 1), ((boxes[:, 0] + boxes[:, 2] / 2) * width).view(n, 1, 1)
            y0, y1 = ((boxes[:, 1] - boxes[:, 3] / 2) * height).view(n, 1, 1), ((boxes[:, 1] + boxes[:, 3] / 2) * height).view(n, 1, 1)
            t["masks"] = ((xs >= x0) & (xs < x1) & (ys >= y0) & (ys < y1)).to(torch.uint8)
            t["orig_size"] = torch.tensor([height, width])
        targets.append(t)
    if torch.device(device).type == "cuda":
        images = images.pin_memory().to(device, non_blocking=True)
    return images, targets
