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
def gen_msda(names=None):
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


# ------------------------------------------------------------------------------------------
def _sample(t, n=512):
    """Evenly strided sample of a tensor (keeps fixtures small) + its mean |value|."""
    f = t.detach().float().flatten()
    step = max(1, f.numel() // n)
    return f[::step][:n].clone(), f.abs().mean()


def _flat_indices(indices):
    """list over images of (q_idx, gt_idx) -> int64 [M,3] rows (image, query, gt)."""
    rows = [torch.stack([torch.full_like(q, b), q, g], 1) for b, (q, g) in enumerate(indices)]
    return torch.cat(rows, 0) if rows else torch.zeros(0, 3, dtype=torch.int64)


MODEL_CASES = {
    # fixture name: (size, img, n_gts per image, segment?)   -- BASELINE configs 0 and 1-3 shapes
    "model_n320": ("n", 320, [3, 5], False),
    "model_n320_empty": ("n", 320, [0, 4], False),  # ragged: an image without GT
    # B=2 everywhere: at B=1 torch 2.10's CPU batch_norm backward mis-reads channels-last-like
    # gradients with a size-1 batch stride (see oracle/model.py:_DenseGrad), which would bake a
    # PyTorch bug into the reference's own gradients.
    "model_s640": ("s", 640, [6, 2], False),
    "model_m640": ("m", 640, [4, 9], False),
    "model_m640_seg": ("m", 640, [4, 3], True),
    "model_x640": ("x", 640, [2, 5], False),  # x architecture (frozen BN, 384->256 proj) at a small image
}
SEED_WEIGHTS_IMG, SEED_TARGETS, SEED_CDN = 1, 2, 123
BF16_CALIBRATION = ("model_m640",)


def gen_model(names=None):
    """End-to-end train-mode forward + criterion + backward of the reference model with the
    name-hashed weights (oracle/weights.py), plus an eval-mode forward."""
    import_reference()
    from src.d_fine import configs as ref_configs
    from src.d_fine.dfine import build_loss, build_model

    from . import weights

    for fname, (size, img, n_gts, seg) in MODEL_CASES.items():
        if names and fname not in names:
            continue
        torch.manual_seed(0)
        model = build_model(size, 80, seg, "cpu", img_size=[img, img], pretrained_model_path=None)
        model.load_state_dict(weights.fill_state_dict(model.state_dict()), strict=True)
        x = weights.synthetic_images(len(n_gts), img, img, SEED_WEIGHTS_IMG)
        targets = weights.synthetic_targets(n_gts, 80, SEED_TARGETS, with_masks=seg, height=img, width=img)
        out = {"n_gts": np.array(n_gts), "img": np.array(img), "seg": np.array(int(seg))}
        # the state_dict layout is itself a contract: names, shapes, dtypes, trainability
        sd0 = model.state_dict()
        out["sd.names"] = np.array(list(sd0.keys()))
        out["sd.shapes"] = np.array([list(v.shape) + [-1] * (4 - v.dim()) for v in sd0.values()])
        out["sd.dtypes"] = np.array([str(v.dtype).replace("torch.", "") for v in sd0.values()])
        out["sd.trainable"] = np.array([n for n, p in model.named_parameters() if p.requires_grad])

        # eval-mode forward first: BatchNorm running stats are still the initial ones
        model.eval()
        with torch.no_grad():
            ev = model(x)
        out["eval.pred_logits"], out["eval.pred_boxes"] = ev["pred_logits"], ev["pred_boxes"]
        if seg:
            out["eval.pred_masks.sample"], out["eval.pred_masks.absmean"] = _sample(ev["pred_masks"], 4096)

        # capture intermediates with hooks (no reference code is modified)
        grabbed = {}
        hooks = [
            model.backbone.register_forward_hook(lambda m, i, o: grabbed.__setitem__("backbone", o)),
            model.encoder.register_forward_hook(lambda m, i, o: grabbed.__setitem__("encoder", o)),
            model.decoder.enc_score_head.register_forward_hook(lambda m, i, o: grabbed.__setitem__("enc_logits", o)),
        ]
        model.train()
        torch.manual_seed(SEED_CDN)
        res = model(x, targets=targets)
        for lvl, f in enumerate(grabbed["backbone"]):
            out[f"backbone.{lvl}.sample"], out[f"backbone.{lvl}.absmean"] = _sample(f)
        for lvl, f in enumerate(grabbed["encoder"][0]):
            out[f"encoder.outs.{lvl}.sample"], out[f"encoder.outs.{lvl}.absmean"] = _sample(f)
        for lvl, f in enumerate(grabbed["encoder"][1]):
            out[f"encoder.inner.{lvl}.sample"], out[f"encoder.inner.{lvl}.absmean"] = _sample(f)
        out["topk_ind"] = torch.topk(grabbed["enc_logits"].max(-1).values, 300, dim=-1).indices

        for k in ("pred_logits", "pred_boxes", "pred_corners", "ref_points"):
            out[f"train.{k}"] = res[k]
        if seg:
            out["train.pred_masks.sample"], out["train.pred_masks.absmean"] = _sample(res["pred_masks"], 4096)
        heads = {f"aux{i}": d for i, d in enumerate(res["aux_outputs"])}
        heads.update(pre=res["pre_outputs"], enc0=res["enc_aux_outputs"][0])
        if "dn_outputs" in res:
            heads.update({f"dn{i}": d for i, d in enumerate(res["dn_outputs"])})
            heads["dn_pre"] = res["dn_pre_outputs"]
            out["dn_num_group"] = np.array(res["dn_meta"]["dn_num_group"])
            out["dn_num_split"] = np.array(res["dn_meta"]["dn_num_split"])
        for hname, d in heads.items():
            out[f"train.{hname}.pred_boxes"] = d["pred_boxes"]
            out[f"train.{hname}.pred_logits.sample"], out[f"train.{hname}.pred_logits.absmean"] = _sample(d["pred_logits"], 2048)

        if fname in BF16_CALIBRATION:
            # the reference's own bf16-autocast error (its AMP path, CPU flavour) on quantities that do
            # not depend on query selection: calibrates the tolerance of the product's bf16 run
            import copy

            m16 = copy.deepcopy(model)
            m16.load_state_dict(weights.fill_state_dict(model.state_dict()), strict=True)  # fresh BN stats
            m16.train()
            g16 = {}
            hk = [m16.encoder.register_forward_hook(lambda m, i, o: g16.__setitem__("encoder", o)),
                  m16.decoder.enc_score_head.register_forward_hook(lambda m, i, o: g16.__setitem__("enc_logits", o))]
            torch.manual_seed(SEED_CDN)
            with torch.autocast("cpu", dtype=torch.bfloat16):
                res16 = m16(x, targets=targets)
            ref_configs.models[size]["DFINECriterion"]["losses"] = ["vfl", "boxes", "local"]
            loss16 = build_loss(size, 80, 0.0, seg)(res16, targets)  # loss outside autocast, as train.py:548
            out["bf16ref.loss.total"] = torch.stack([v.detach().float().reshape(()) for v in loss16.values()]).sum()
            for h in hk:
                h.remove()
            for lvl, f in enumerate(g16["encoder"][0]):
                out[f"bf16ref.encoder.outs.{lvl}.sample"], _ = _sample(f)
            idx = out["topk_ind"].unsqueeze(-1).expand(-1, -1, 80)
            out["bf16ref.enc0.pred_logits.sample"], _ = _sample(g16["enc_logits"].float().gather(1, idx), 2048)
            del m16

        ref_configs.models[size]["DFINECriterion"]["losses"] = ["vfl", "boxes", "local"]  # build_loss appends in place
        crit = build_loss(size, 80, 0.0, seg)
        matcher = crit.matcher
        out["match.main"] = _flat_indices(matcher({"pred_logits": res["pred_logits"], "pred_boxes": res["pred_boxes"]}, targets)["indices"])
        out["match.enc0"] = _flat_indices(matcher(res["enc_aux_outputs"][0], targets)["indices"])
        losses = crit(res, targets)
        out["loss.names"] = np.array(list(losses.keys()))
        out["loss.values"] = torch.stack([v.detach().float().reshape(()) for v in losses.values()])
        sum(losses.values()).backward()
        names_, norms = [], []
        for n_, p in model.named_parameters():
            names_.append(n_)
            norms.append(p.grad.float().norm() if p.grad is not None else torch.tensor(-1.0))
        out["grad.names"] = np.array(names_)
        out["grad.norms"] = torch.stack(norms)
        bn = {k: v for k, v in model.state_dict().items() if k.endswith("running_var")}
        out["bn.running_var.absmean"] = torch.stack([v.abs().mean() for v in bn.values()])

        for h in hooks:
            h.remove()
        save(fname, **out)


WELLCOND_CASES = {
    # fixture name: (size, img, n_gts per image, segment?) -- the benchmarked architecture with well-conditioned weights
    "model_m640_wellcond": ("m", 640, [4, 9], False),
    "model_m640_seg_wellcond": ("m", 640, [4, 3], True),
}
GRAD_SAMPLES = 256


def grad_sample_index(numel):
    """Evenly strided element indices (<= GRAD_SAMPLES) of a flattened gradient: regenerated by the tests."""
    step = max(1, numel // GRAD_SAMPLES)
    return torch.arange(0, numel, step)[:GRAD_SAMPLES]


def gen_wellcond(names=None):
    """A WELL-CONDITIONED training-mode fixture for the bf16 path. The name-hashed default weights make every layer a
    difference of large terms, so ANY bf16 pipeline drifts 30-80 % from fp32 by the encoder (the reference's own autocast
    included: bf16ref.* in model_m640.npz) and percent-level end-to-end bounds are impossible with them. Here the reference
    model runs in ``train()`` mode with every ``nn.BatchNorm2d`` in ``.eval()`` (PyTorch's frozen-statistics fine-tuning
    mode) on oracle/weights.py's "wellcond" weights: non-negative convolutions, unit statistics, positive activations --
    rounding errors add instead of multiplying. Recorded: stage outputs, head outputs, the loss dict and, per parameter
    tensor, the gradient's norm and a strided sample of it (src/d_fine/dfine.py:37-41, dfine_criterion.py:410-578)."""
    import_reference()
    from src.d_fine import configs as ref_configs
    from src.d_fine.dfine import build_loss, build_model

    from . import weights

    for fname, (size, img, n_gts, seg) in WELLCOND_CASES.items():
        if names and fname not in names:
            continue
        torch.manual_seed(0)
        model = build_model(size, 80, seg, "cpu", img_size=[img, img], pretrained_model_path=None)
        model.load_state_dict(weights.fill_state_dict(model.state_dict(), "wellcond"), strict=True)
        x = weights.synthetic_images(len(n_gts), img, img, SEED_WEIGHTS_IMG)
        targets = weights.synthetic_targets(n_gts, 80, SEED_TARGETS, with_masks=seg, height=img, width=img)
        out = {"n_gts": np.array(n_gts), "img": np.array(img), "seg": np.array(int(seg))}
        model.train()
        bns = [(n, m) for n, m in model.named_modules() if isinstance(m, torch.nn.BatchNorm2d)]
        for _, m in bns:
            m.eval()
        # calibration pass: every norm's running_var := (per-channel mean of ITS input)^2 with running_mean = 0, set just
        # before the norm runs -- y = gamma x / mean(x) + beta is O(1) after every layer although residual sums and concats
        # grow the raw magnitudes, and nothing is subtracted. The calibrated buffers travel in the fixture.
        def calibrate(m, args):
            with torch.no_grad():
                m.running_mean.zero_()
                m.running_var.copy_((args[0].mean((0, 2, 3)) ** 2 - m.eps).clamp(min=1e-6))
        pre = [m.register_forward_pre_hook(calibrate) for _, m in bns]
        with torch.no_grad():
            torch.manual_seed(SEED_CDN)
            model(x, targets=targets)
        for h in pre:
            h.remove()
        out["bn.names"] = np.array([n for n, _ in bns])
        out["bn.channels"] = np.array([m.num_features for _, m in bns])
        out["bn.running_var"] = torch.cat([m.running_var for _, m in bns])
        grabbed = {}
        hooks = [
            model.backbone.register_forward_hook(lambda m, i, o: grabbed.__setitem__("backbone", o)),
            model.encoder.register_forward_hook(lambda m, i, o: grabbed.__setitem__("encoder", o)),
            model.decoder.enc_score_head.register_forward_hook(lambda m, i, o: grabbed.__setitem__("enc_logits", o)),
        ]
        torch.manual_seed(SEED_CDN)
        res = model(x, targets=targets)
        for lvl, f in enumerate(grabbed["backbone"]):
            out[f"backbone.{lvl}.sample"], out[f"backbone.{lvl}.absmean"] = _sample(f, 4096)
            out[f"backbone.{lvl}.std"] = f.detach().float().std()
        for lvl, f in enumerate(grabbed["encoder"][0]):
            out[f"encoder.outs.{lvl}.sample"], out[f"encoder.outs.{lvl}.absmean"] = _sample(f, 4096)
            out[f"encoder.outs.{lvl}.std"] = f.detach().float().std()
        out["topk_ind"] = torch.topk(grabbed["enc_logits"].max(-1).values, 300, dim=-1).indices
        for k in ("pred_logits", "pred_boxes", "pred_corners"):
            out[f"train.{k}"] = res[k]
        if seg:
            out["train.pred_masks.sample"], out["train.pred_masks.absmean"] = _sample(res["pred_masks"], 4096)
        heads = {f"aux{i}": d for i, d in enumerate(res["aux_outputs"])}
        heads.update(pre=res["pre_outputs"], enc0=res["enc_aux_outputs"][0])
        heads.update({f"dn{i}": d for i, d in enumerate(res["dn_outputs"])})
        for hname, d in heads.items():
            out[f"train.{hname}.pred_boxes"] = d["pred_boxes"]
        ref_configs.models[size]["DFINECriterion"]["losses"] = ["vfl", "boxes", "local"]  # build_loss appends in place
        crit = build_loss(size, 80, 0.0, seg)
        losses = crit(res, targets)
        out["loss.names"] = np.array(list(losses.keys()))
        out["loss.values"] = torch.stack([v.detach().float().reshape(()) for v in losses.values()])
        sum(losses.values()).backward()
        names_, norms, samples, counts = [], [], [], []
        for n_, p in model.named_parameters():
            if p.grad is None:
                continue
            g_ = p.grad.float().flatten()
            idx = grad_sample_index(g_.numel())
            names_.append(n_)
            norms.append(g_.norm())
            samples.append(g_[idx])
            counts.append(idx.numel())
        out["grad.names"] = np.array(names_)
        out["grad.norms"] = torch.stack(norms)
        out["grad.samples"] = torch.cat(samples)
        out["grad.counts"] = np.array(counts)
        for h in hooks:
            h.remove()
        save(fname, **out)


def import_reference_dl():
    """Make ``src.dl.train`` / ``src.dl.utils`` importable for the two pieces of it that are on the path and need none
    of its absent third-party imports: ``Trainer.preds_postprocess`` (+ process_boxes / process_masks / cleanup_masks)
    and the optimizer-step recipe. hydra, omegaconf, wandb, cv2, albumentations, faster_coco_eval and torchmetrics are
    not installed here; they are replaced by INERT placeholder modules (every attribute is a do-nothing object) whose
    only job is to let the ``import`` statements at the top of those files succeed -- nothing that is executed for a
    fixture touches them."""
    import importlib.abc
    import importlib.machinery

    import_reference()
    if "src.dl.train" in sys.modules:
        return
    absent = ("hydra", "omegaconf", "wandb", "cv2", "albumentations", "faster_coco_eval", "torchmetrics", "pycocotools")

    class _Inert:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return a[0] if (len(a) == 1 and callable(a[0]) and not k) else _Inert()  # also usable as a decorator

        def __getattr__(self, n):
            return _Inert()

        def __mro_entries__(self, bases):
            return (object,)

    class _InertModule(types.ModuleType):
        def __getattr__(self, n):
            if n.startswith("__"):
                raise AttributeError(n)
            return _Inert()

    class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        def find_spec(self, name, path, target=None):
            if name.split(".")[0] in absent:
                return importlib.machinery.ModuleSpec(name, self, is_package=True)

        def create_module(self, spec):
            m = _InertModule(spec.name)
            m.__path__ = []
            return m

        def exec_module(self, m):
            pass

    sys.meta_path.insert(0, _Finder())
    sys.modules["torchvision.ops"].box_iou = None  # imported by name in validator.py, never called here
    sys.modules["torchvision.ops"].nms = None
    import src.dl.train  # noqa: F401


def gen_postprocess(names=None):
    """Trainer.preds_postprocess (src/dl/train.py:227-319) on seeded predictions: detection only and with masks, with and
    without letterbox (keep_ratio), non-square original sizes."""
    import_reference_dl()
    from src.dl.train import Trainer

    g = torch.Generator().manual_seed(11)
    B, Q, C, Hm, Wm, proc = 2, 60, 7, 40, 40, (160, 160)
    logits = torch.randn(B, Q, C, generator=g) * 2.0 - 1.0
    cxcy = torch.rand(B, Q, 2, generator=g) * 0.7 + 0.15
    wh = torch.rand(B, Q, 2, generator=g) * 0.3 + 0.05
    boxes = torch.cat([cxcy, wh], -1)
    # smooth blobs as mask probabilities (so the bilinear resizes are exercised away from 0/1 plateaus)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, Hm), torch.linspace(0, 1, Wm), indexing="ij")
    d2 = (xx[None, None] - cxcy[..., 0, None, None]) ** 2 / (wh[..., 0, None, None] ** 2) + (yy[None, None] - cxcy[..., 1, None, None]) ** 2 / (wh[..., 1, None, None] ** 2)
    masks = torch.sigmoid(4.0 * (0.3 - d2) + 0.3 * torch.randn(B, Q, Hm, Wm, generator=g))
    orig_sizes = torch.tensor([[120, 90], [64, 100]])
    inputs = torch.zeros(B, 3, *proc)
    out = {"logits": logits, "boxes": boxes, "masks": masks, "orig_sizes": orig_sizes, "proc": np.array(proc), "num_top_queries": np.array(25),
           "conf_thresh": np.array(0.5)}
    for keep_ratio in (False, True):
        for with_masks in (False, True):
            outputs = {"pred_logits": logits.clone(), "pred_boxes": boxes.clone()}
            if with_masks:
                outputs["pred_masks"] = masks.clone()
            res = Trainer.preds_postprocess(inputs, outputs, orig_sizes, C, keep_ratio, 0.5, num_top_queries=25)
            tag = f"kr{int(keep_ratio)}.m{int(with_masks)}"
            for b, r in enumerate(res):
                for k, v in r.items():
                    out[f"{tag}.{b}.{k}"] = v
    save("postprocess", **out)


STEP_CASE = dict(size="n", img=320, n_gts=[3, 5], base_lr=0.0008, backbone_lr=0.0004, betas=(0.9, 0.999), weight_decay=0.000125,
                 clip=0.1, ema_momentum=0.9998, total_steps=10, pct_start=0.3, steps=2)


def gen_step(names=None):
    """Two optimizer steps exactly as the reference's hot loop runs them in fp32 (src/dl/train.py:488-511,545-560 with
    amp off: forward, criterion, backward, clip_grad_norm_(0.1), AdamW over build_optimizer's four groups, OneCycleLR
    with the scalar max_lr = 2*base_lr the n/s/m sizes get (:193-208), zero_grad, ModelEMA.update (:61-72))."""
    import_reference_dl()
    from copy import deepcopy

    from src.d_fine import configs as ref_configs
    from src.d_fine.dfine import build_loss, build_model, build_optimizer
    from src.dl.train import ModelEMA
    from torch.optim.lr_scheduler import OneCycleLR

    from . import weights

    c = STEP_CASE
    torch.manual_seed(0)
    model = build_model(c["size"], 80, False, "cpu", img_size=[c["img"], c["img"]], pretrained_model_path=None)
    model.load_state_dict(weights.fill_state_dict(model.state_dict()), strict=True)
    x = weights.synthetic_images(len(c["n_gts"]), c["img"], c["img"], SEED_WEIGHTS_IMG)
    targets = weights.synthetic_targets(c["n_gts"], 80, SEED_TARGETS, with_masks=False, height=c["img"], width=c["img"])
    ref_configs.models[c["size"]]["DFINECriterion"]["losses"] = ["vfl", "boxes", "local"]
    crit = build_loss(c["size"], 80, 0.0, False)
    opt = build_optimizer(model, lr=c["base_lr"], backbone_lr=c["backbone_lr"], betas=c["betas"], weight_decay=c["weight_decay"], base_lr=c["base_lr"])
    sched = OneCycleLR(opt, max_lr=c["base_lr"] * 2, epochs=1, steps_per_epoch=c["total_steps"], pct_start=c["pct_start"], cycle_momentum=False)
    ema = ModelEMA(model, c["ema_momentum"])
    grabbed = {}
    model.decoder.enc_score_head.register_forward_hook(lambda m, i, o: grabbed.__setitem__("enc_logits", o))
    model.train()
    p0 = {n: p.detach().clone() for n, p in model.named_parameters()}
    out = {k: np.array(v) for k, v in c.items() if k not in ("size",)}
    out["size"] = np.array(c["size"])
    g = torch.Generator().manual_seed(5)
    names_ = [n for n, p in model.named_parameters() if p.requires_grad]
    sample_idx = {n: torch.randint(0, p0[n].numel(), (min(64, p0[n].numel()),), generator=g) for n in names_}
    out["param.names"] = np.array(names_)
    for s in range(c["steps"]):
        torch.manual_seed(SEED_CDN + s)
        res = model(x, targets=targets)
        out[f"step{s}.topk_ind"] = torch.topk(grabbed["enc_logits"].max(-1).values, 300, dim=-1).indices
        losses = crit(res, targets)
        out[f"step{s}.loss.names"] = np.array(list(losses.keys()))
        out[f"step{s}.loss.values"] = torch.stack([v.detach().float().reshape(()) for v in losses.values()])
        sum(losses.values()).backward()
        out[f"step{s}.lrs"] = np.array([g_["lr"] for g_ in opt.param_groups])
        out[f"step{s}.grad_norm"] = torch.nn.utils.clip_grad_norm_(model.parameters(), c["clip"])
        opt.step()
        sched.step()
        opt.zero_grad()
        ema.update(s + 1, model)
        cur = dict(model.named_parameters())
        out[f"step{s}.param.norm"] = torch.stack([cur[n].detach().double().norm() for n in names_])
        out[f"step{s}.delta.sample"] = torch.cat([(cur[n].detach() - p0[n]).flatten()[sample_idx[n]] for n in names_])
    out["sample.index"] = torch.cat([sample_idx[n] for n in names_])
    out["sample.count"] = np.array([sample_idx[n].numel() for n in names_])
    ema_p = dict(ema.model.named_parameters())
    out["ema.param.norm"] = torch.stack([ema_p[n].detach().double().norm() for n in names_])
    out["ema.delta.sample"] = torch.cat([(ema_p[n].detach() - p0[n]).flatten()[sample_idx[n]] for n in names_])
    bufs = {k: v for k, v in model.state_dict().items() if k.endswith("running_mean") or k.endswith("running_var")}
    out["buffer.names"] = np.array(list(bufs.keys()))
    out["buffer.norm"] = torch.stack([v.double().norm() for v in bufs.values()])
    ema_sd = ema.model.state_dict()
    out["ema.buffer.norm"] = torch.stack([ema_sd[k].double().norm() for k in bufs])
    save("step_n320", **out)


GROUPS = {"msda": gen_msda, "model": gen_model, "postprocess": gen_postprocess, "step": gen_step, "wellcond": gen_wellcond}


def main(argv):
    torch.set_num_threads(4)
    names = argv or list(GROUPS)
    for n in names:
        group, _, sub = n.partition(":")
        GROUPS[group](sub.split(",")) if sub else GROUPS[group]()


if __name__ == "__main__":
    main(sys.argv[1:])
