"""Fused per-head criterion kernel (vfl + L1/GIoU + FGL + DDF with gradients, one launch per head)
against the dense PyTorch formulation of the same losses in d-fine-seg_amd/criterion.py -- which is
itself pinned to the reference's loss values by tests/test_product_model_gpu.py. Synthetic heads:
random logits/boxes/corners, ragged ground truth including an image with none."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _outputs(B, Q, Nd, C, G_counts, L, seed, dtype=torch.float32):
    from dfine_seg_amd.decoder import weighting_function
    from dfine_seg_amd.targets import TargetBatch

    g = torch.Generator().manual_seed(seed)
    dev = "cuda"
    leaves = []

    def leaf(*shape, scale=1.0, shift=0.0, dt=torch.float32):
        t = (torch.randn(*shape, generator=g) * scale + shift).to(dt).to(dev).requires_grad_(True)
        leaves.append(t)
        return t

    def boxes(n):
        t = torch.cat([torch.rand(B, n, 2, generator=g) * 0.6 + 0.2, torch.rand(B, n, 2, generator=g) * 0.3 + 0.02], -1).to(dev).requires_grad_(True)
        leaves.append(t)
        return t

    def head(n, corners=True):
        d = {"pred_logits": leaf(B, n, C, scale=2.0, shift=-2.0, dt=dtype), "pred_boxes": boxes(n)}
        if corners:
            d["pred_corners"] = leaf(B, n, 4 * 33, scale=1.5)
        return d

    targets = []
    for n in G_counts:
        targets.append({"labels": torch.randint(0, C, (n,), generator=g),
                        "boxes": torch.cat([torch.rand(n, 2, generator=g) * 0.6 + 0.2, torch.rand(n, 2, generator=g) * 0.25 + 0.05], -1)})
    tb = TargetBatch.from_list(targets, C, dev)
    ref = torch.cat([torch.rand(B, Q + Nd, 2, generator=g) * 0.6 + 0.2, torch.rand(B, Q + Nd, 2, generator=g) * 0.3 + 0.05], -1).to(dev)
    up, reg_scale = torch.tensor([0.5], device=dev), torch.tensor([4], device=dev)
    layers = [head(Q) for _ in range(L)]
    dn_layers = [head(Nd) for _ in range(L)]
    for group, r in ((layers, ref[:, Nd:]), (dn_layers, ref[:, :Nd])):
        for i, d in enumerate(group):
            d["ref_points"] = r
            d["teacher_corners"], d["teacher_logits"] = group[-1]["pred_corners"], group[-1]["pred_logits"]
            if i == L - 1:
                d["self_distilled"] = True
    main = {k: v for k, v in layers[-1].items() if not k.startswith("teacher") and k != "self_distilled"}
    out = dict(main)
    out.update(up=up, reg_scale=reg_scale, fdr={"W": weighting_function(32, up, reg_scale).to(dev), "reg_scale": 4.0})
    out["aux_outputs"] = layers[:-1]
    out["pre_outputs"] = head(Q, corners=False)
    out["enc_aux_outputs"] = [head(Q, corners=False)]
    G = max(G_counts)
    groups = max(1, 100 // G)
    assert Nd == 2 * G * groups
    slot = torch.arange(Nd, device=dev) % (2 * G)
    pad = tb.valid[:, :G].repeat(1, 2 * groups)
    out["dn_outputs"] = dn_layers
    out["dn_pre_outputs"] = head(Nd, corners=False)
    out["dn_meta"] = {"dn_num_group": groups, "dn_num_split": [Nd, Q],
                      "dn_assign": torch.where((slot < G)[None] & pad, slot[None].expand(B, Nd), -1)}
    out["targets"] = tb
    return out, leaves


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("counts", [[3, 0, 7, 1], [20, 20], [1]])
def test_fused_heads_match_dense_formulation(hip_lib, counts, dtype):
    from dfine_seg_amd.configs import CriterionSpec
    from dfine_seg_amd.criterion import DFINECriterion

    B, Q, C, L = len(counts), 300, 80, 3
    G = max(counts)
    Nd = 2 * G * max(1, 100 // G)
    results = {}
    for fused in (False, True):
        out, leaves = _outputs(B, Q, Nd, C, counts, L, seed=5, dtype=dtype)
        crit = DFINECriterion(CriterionSpec(), num_classes=C, fused=fused)
        losses = crit(out)
        torch.stack(list(losses.values())).sum().backward()
        results[fused] = ({k: float(v) for k, v in losses.items()}, [t.grad.float().cpu() if t.grad is not None else None for t in leaves])
    dense, fused = results[False], results[True]
    assert list(dense[0].keys()) == list(fused[0].keys())
    for k in dense[0]:
        a, b = dense[0][k], fused[0][k]
        assert abs(a - b) <= 2e-4 * abs(a) + 2e-5, (k, a, b)
    assert any(k.startswith("loss_ddf") and v > 0 for k, v in fused[0].items())
    # logits in bf16: d_logits is rounded once to bf16 by both paths (different op order before that)
    tol = dict(rtol=2e-3, atol=2e-6) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-5)
    for i, (ga, gb) in enumerate(zip(dense[1], fused[1])):
        assert (ga is None) == (gb is None), i
        if ga is not None:
            torch.testing.assert_close(gb, ga, **tol, msg=lambda m: f"leaf {i} {tuple(ga.shape)}: {m}")


@pytest.mark.gpu
def test_head_reads_slices_of_a_wider_decoder_output_in_place(hip_lib):
    """The decoder's outputs are [B, Nd + Q, .]; a head gets the [:, :Nd] or [:, Nd:] part as a strided view
    (dfine_criterion_head batch_strides) instead of a contiguous copy: loss terms and gradients bit-equal to the dense call."""
    from dfine_seg_amd import ops

    torch.manual_seed(3)
    B, Nd, Q, C, G, NB = 3, 40, 60, 80, 7, 33
    dev = "cuda"
    full_logits = torch.randn(B, Nd + Q, C, device=dev).bfloat16()
    full_boxes = torch.rand(B, Nd + Q, 4, device=dev) * 0.5 + 0.2
    full_corners = torch.randn(B, Nd + Q, 4 * NB, device=dev)
    full_ref = torch.rand(B, Nd + Q, 4, device=dev) * 0.5 + 0.2
    t_corners, t_logits = torch.randn_like(full_corners), torch.randn_like(full_logits)
    labels = torch.randint(0, C, (B, G), device=dev)
    tboxes = torch.rand(B, G, 4, device=dev) * 0.4 + 0.3
    Wn = torch.linspace(-2, 2, NB, device=dev)
    scal = torch.tensor([0.1, 0.1, 0.01, 0.02, 1.0, 5.0, 2.0, 0.15, 1.5], device=dev)
    for lo, hi in ((0, Nd), (Nd, Nd + Q)):
        n = hi - lo
        own = torch.randint(-1, G, (B, n), device=dev, dtype=torch.int32)
        outs = []
        for dense in (True, False):
            sl = lambda t: (t[:, lo:hi].contiguous() if dense else t[:, lo:hi]).detach().requires_grad_(t.is_floating_point())
            lg, bx, cr = sl(full_logits), sl(full_boxes), sl(full_corners)
            terms = ops.criterion_head(lg, bx, cr, full_ref[:, lo:hi] if not dense else full_ref[:, lo:hi].contiguous(),
                                       t_corners[:, lo:hi] if not dense else t_corners[:, lo:hi].contiguous(),
                                       t_logits[:, lo:hi] if not dense else t_logits[:, lo:hi].contiguous(),
                                       own, own, labels, tboxes, Wn, scal)
            torch.stack(terms).sum().backward()
            outs.append((torch.stack(terms).detach(), lg.grad, bx.grad, cr.grad))
        for a, b in zip(*outs):
            assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_one_launch_for_all_heads_equals_one_launch_per_head(hip_lib, dtype):
    """ops.CriterionBatch / dfine_criterion_heads: the heads of a criterion call computed by ONE launch (blockIdx.y = head, heads
    of 300 and of 200 queries, with and without corners / teacher) against one launch per head -- the same kernel body: every loss
    term and every gradient bit for bit (the five sums of a head are float atomics into 64 replicas in both: compared to 1e-6)."""
    from dfine_seg_amd import ops
    from dfine_seg_amd.configs import CriterionSpec
    from dfine_seg_amd.criterion import DFINECriterion

    counts = [3, 0, 7, 1]
    B, Q, C, L = len(counts), 300, 80, 3
    Nd = 2 * 7 * (100 // 7)
    results = {}
    old = ops.CRITERION_BATCH
    try:
        for batched in (False, True):
            ops.CRITERION_BATCH = batched
            out, leaves = _outputs(B, Q, Nd, C, counts, L, seed=9, dtype=dtype)
            losses = DFINECriterion(CriterionSpec(), num_classes=C, fused=True)(out)
            torch.stack(list(losses.values())).sum().backward()
            results[batched] = ({k: float(v) for k, v in losses.items()}, [t.grad.clone() if t.grad is not None else None for t in leaves])
    finally:
        ops.CRITERION_BATCH = old
    one, all_ = results[False], results[True]
    assert list(one[0].keys()) == list(all_[0].keys())
    for k in one[0]:
        assert abs(one[0][k] - all_[0][k]) <= 1e-6 * abs(one[0][k]) + 1e-9, (k, one[0][k], all_[0][k])
    for i, (ga, gb) in enumerate(zip(one[1], all_[1])):
        assert (ga is None) == (gb is None), i
        if ga is not None:
            assert torch.equal(ga, gb), i
