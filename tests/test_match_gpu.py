"""Device-side Hungarian matching (cost + LSAP + GO union) through the C ABI against
the oracle (criterion.match_cost + the scipy-equivalent C LSAP) and scipy itself."""
import numpy as np
import pytest
import torch
from scipy.optimize import linear_sum_assignment as scipy_lsap

from oracle import criterion as oc

pytestmark = pytest.mark.gpu


def _problem(H, B, Q, C, num_gts, seed, tie_boxes=False):
    g = torch.Generator().manual_seed(seed)
    G = max(max(num_gts), 1)
    logits = torch.randn(H, B, Q, C, generator=g) * 2 - 2
    boxes = torch.cat([torch.rand(H, B, Q, 2, generator=g), torch.rand(H, B, Q, 2, generator=g) * 0.4 + 0.01], -1)
    if tie_boxes:  # many identical predictions -> exact cost ties
        boxes = boxes[:, :, :1].expand(-1, -1, Q, -1).contiguous()
        logits = logits[:, :, :1].expand(-1, -1, Q, -1).contiguous()
    labels = torch.full((B, G), C, dtype=torch.int64)
    tboxes = torch.zeros(B, G, 4)
    for b, n in enumerate(num_gts):
        labels[b, :n] = torch.randint(0, C, (n,), generator=g)
        tboxes[b, :n] = torch.cat([torch.rand(n, 2, generator=g) * 0.6 + 0.2, torch.rand(n, 2, generator=g) * 0.25 + 0.05], -1)
    return logits, boxes, labels, tboxes


@pytest.mark.parametrize("num_gts", [[3, 5], [0, 4, 1], [20, 1, 7, 12], [40] * 3])
def test_match_equals_scipy_on_oracle_cost(hip_lib, num_gts):
    from dfine_seg_amd import ops

    H, B, Q, C = 3, len(num_gts), 300, 80
    logits, boxes, labels, tboxes = _problem(H, B, Q, C, num_gts, seed=sum(num_gts))
    counts = torch.tensor(num_gts, dtype=torch.int32)
    gt_of_q, q_of_gt = ops.hungarian_match(logits.cuda(), boxes.cuda(), labels.cuda(), tboxes.cuda(), counts.cuda())
    gt_of_q, q_of_gt = gt_of_q.cpu(), q_of_gt.cpu()
    for h in range(H):
        for b, n in enumerate(num_gts):
            if n == 0:
                assert (gt_of_q[h, b] == -1).all()
                continue
            cost = oc.match_cost(logits[h, b : b + 1], boxes[h, b : b + 1], labels[b, :n], tboxes[b, :n])  # [Q, n]
            rows, cols = scipy_lsap(cost.numpy())
            got_rows = q_of_gt[h, b, :n].numpy()
            # same optimum: compare the assignment itself and, robustly, its total cost
            ref_cost = cost.numpy()[rows, cols].sum()
            got_cost = cost.numpy()[got_rows, np.arange(n)].sum()
            assert abs(ref_cost - got_cost) <= 1e-4 * max(1.0, abs(ref_cost))
            assert sorted(got_rows.tolist()) == sorted(set(got_rows.tolist()))  # one query per GT
            order = np.argsort(got_rows)
            assert np.array_equal(got_rows[order], rows) and np.array_equal(np.arange(n)[order], cols)
            inv = gt_of_q[h, b].numpy()
            assert (inv[got_rows] == np.arange(n)).all() and (inv >= 0).sum() == n


def test_match_tie_breaking_follows_scipy(hip_lib):
    """All queries identical -> every column ties; scipy's scan order decides."""
    from dfine_seg_amd import ops

    num_gts = [4, 2]
    logits, boxes, labels, tboxes = _problem(1, 2, 300, 80, num_gts, seed=5, tie_boxes=True)
    counts = torch.tensor(num_gts, dtype=torch.int32)
    _, q_of_gt = ops.hungarian_match(logits.cuda(), boxes.cuda(), labels.cuda(), tboxes.cuda(), counts.cuda())
    for b, n in enumerate(num_gts):
        cost = oc.match_cost(logits[0, b : b + 1], boxes[0, b : b + 1], labels[b, :n], tboxes[b, :n])
        rows, cols = scipy_lsap(cost.numpy())
        got = q_of_gt[0, b, :n].cpu().numpy()
        order = np.argsort(got)
        assert np.array_equal(got[order], rows) and np.array_equal(np.arange(n)[order], cols)


def test_go_union_matches_oracle(hip_lib):
    from dfine_seg_amd import ops

    H, B, Q, G = 6, 3, 300, 9
    g = torch.Generator().manual_seed(0)
    gt_of_q = torch.full((H, B, Q), -1, dtype=torch.int32)
    lists = []
    for h in range(H):
        per_img = []
        for b in range(B):
            qs = torch.randperm(40, generator=g)[:G]  # crowded on purpose: heads disagree
            gt_of_q[h, b, qs] = torch.arange(G, dtype=torch.int32)
            per_img.append((qs.long(), torch.arange(G)))
        lists.append(per_img)
    go = ops.go_union(gt_of_q.cuda()).cpu()
    ref = oc.go_indices(lists[0], lists[1:])
    for b, (qs, gs) in enumerate(ref):
        assert (go[b] >= 0).sum() == len(qs)
        # compare where the most frequent GT is unique (the reference leaves ties to sort order)
        for q, gt in zip(qs.tolist(), gs.tolist()):
            votes = [int(gt_of_q[h, b, q]) for h in range(H) if gt_of_q[h, b, q] >= 0]
            counts = {v: votes.count(v) for v in votes}
            top = max(counts.values())
            winners = [v for v, c in counts.items() if c == top]
            if len(winners) == 1:
                assert int(go[b, q]) == gt
            else:
                assert int(go[b, q]) == min(winners)
