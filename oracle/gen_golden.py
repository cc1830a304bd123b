"""Generates tests/golden/*.npz by running the REFERENCE itself (imported read-only from
/root/reference) on seeded inputs. Runs only in the build container: /root/reference does
not exist on the GPU box, so tests consume the committed .npz files, never this script.

Fixtures hold data only (inputs + the reference's outputs), never reference source.

    python -m oracle.gen_golden            # all fixtures
    python -m oracle.gen_golden msda ...   # selected groups
"""
import os
import sys
import types

import numpy as np
import torch

REFERENCE_ROOT = "/root/reference"
GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def import_reference():
    """Make `src.d_fine` importable: two absent third-party modules are stubbed in-process
    (loguru: logging only; torchvision.ops.boxes.box_area: (x2-x1)*(y2-y1)), SURVEY.md App. B."""
    if "src.d_fine" in sys.modules:
        return
    loguru = types.ModuleType("loguru")

    class _Logger:
        def __getattr__(self, name):
            return lambda *a, **k: None

    loguru.logger = _Logger()
    tv, tv_ops, tv_boxes = (types.ModuleType(n) for n in ("torchvision", "torchvision.ops", "torchvision.ops.boxes"))
    tv_boxes.box_area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    tv_ops.boxes, tv.ops = tv_boxes, tv_ops
    sys.modules.update({"loguru": loguru, "torchvision": tv, "torchvision.ops": tv_ops, "torchvision.ops.boxes": tv_boxes})
    sys.path.insert(0, REFERENCE_ROOT)
    import src.d_fine  # noqa: F401


def save(name, **arrays):
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    path = os.path.join(GOLDEN_DIR, name + ".npz")
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


# ------------------------------------------------------------------------------------------
def gen_msda():
    """deformable_attention_core_func_v2 forward + autograd grads (arch/utils.py:191-264)."""
    import_reference()
    from src.d_fine.arch.utils import deformable_attention_core_func_v2

    cases = {
        # name: (B, Lq, nH, c, level shapes, points per level)
        "small": (2, 16, 2, 16, [(8, 8), (4, 4)], [3, 2]),
        "rect3": (2, 10, 8, 32, [(6, 10), (3, 5), (2, 3)], [3, 6, 3]),
    }
    out = {}
    for name, (B, Lq, nH, c, shapes, npts) in cases.items():
        g = torch.Generator().manual_seed(1234 + len(name))
        S, P = sum(h * w for h, w in shapes), sum(npts)
        value = torch.randn(B, S, nH, c, generator=g)  # channel-last, the build's layout
        # locations deliberately overshoot [0,1] so zero padding and partly-outside corners occur
        loc = torch.rand(B, Lq, nH, P, 2, generator=g) * 1.5 - 0.25
        attn = torch.softmax(torch.randn(B, Lq, nH, P, generator=g), -1)
        grad_out = torch.randn(B, Lq, nH * c, generator=g)
        value.requires_grad_(True), loc.requires_grad_(True), attn.requires_grad_(True)
        # reference layout: tuple over levels of (B, nH, c, H_l*W_l)  (dfine_decoder.py:416-426)
        value_ref = value.permute(0, 2, 3, 1).split([h * w for h, w in shapes], dim=-1)
        y = deformable_attention_core_func_v2(value_ref, [list(s) for s in shapes], loc, attn, npts)
        y.backward(grad_out)
        out.update({
            f"{name}.shapes": np.array(shapes), f"{name}.num_points": np.array(npts),
            f"{name}.value": value, f"{name}.loc": loc, f"{name}.attn": attn,
            f"{name}.grad_out": grad_out, f"{name}.out": y,
            f"{name}.grad_value": value.grad, f"{name}.grad_loc": loc.grad, f"{name}.grad_attn": attn.grad,
        })
    save("msda_core", **out)


GROUPS = {"msda": gen_msda}


def main(argv):
    torch.set_num_threads(4)
    names = argv or list(GROUPS)
    for n in names:
        GROUPS[n]()


if __name__ == "__main__":
    main(sys.argv[1:])
