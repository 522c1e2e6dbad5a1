"""Host-side pieces of forward_train that were rewritten without host read-backs, against the
reference's formulation (softgroup/model/softgroup.py:152-222) on CPU tensors."""
import torch
import torch.nn.functional as F

from softgroup_amd.model.softgroup import _assign_proposals, _cross_entropy


def test_cross_entropy_equals_torch():
    """log-softmax + gather + masked mean == F.cross_entropy(weight=, ignore_index=) (reference
    softgroup.py:159-160), value and gradient"""
    torch.manual_seed(0)
    for n, c in ((1000, 13), (257, 20), (5, 3)):
        s = torch.randn(n, c, requires_grad=True)
        y = torch.randint(0, c, (n, ))
        y[::7] = -100
        for w in (None, torch.rand(c) + 0.1):
            want = F.cross_entropy(s, y, weight=w, ignore_index=-100)
            got = _cross_entropy(s, y, w, -100)
            gw, = torch.autograd.grad(want, s)
            gg, = torch.autograd.grad(got, s)
            assert abs(float(want) - float(got)) <= 1e-6 * max(abs(float(want)), 1.0)
            assert float((gw - gg).abs().max()) <= 1e-7


def _reference_labels(ious, cls, thr, mlq, min_pos, background):
    """softgroup.py:196-222 as written: boolean indexing, the loop over the GTs"""
    fg = cls != -100
    fg_cls = cls[fg]
    f = ious[:, fg]
    n_prop, n_gt = f.shape
    assigned = f.new_full((n_prop, ), -1, dtype=torch.long)
    max_iou, argmax_iou = f.max(1)
    pos = max_iou >= thr
    assigned[pos] = argmax_iou[pos]
    if mlq:
        gt_max, gt_arg = f.max(0)
        for i in range(n_gt):
            if gt_max[i] >= min_pos:
                assigned[gt_arg[i]] = i
    labels = fg_cls.new_full((n_prop, ), background)
    pos = assigned >= 0
    labels[pos] = fg_cls[assigned[pos]]
    return labels


def test_proposal_assignment_equals_the_reference_loop():
    g = torch.Generator().manual_seed(1)
    checked = 0
    for trial in range(400):
        n_prop = int(torch.randint(1, 14, (1, ), generator=g))
        n_gt = int(torch.randint(1, 10, (1, ), generator=g))
        ious = torch.rand(n_prop, n_gt, generator=g)
        ious[torch.rand(n_prop, n_gt, generator=g) < 0.35] = 0          # ties at zero, empty columns
        if trial % 5 == 0:
            ious[:, 0] = ious[:, -1]                                     # duplicated columns: argmax ties
        cls = torch.randint(0, 6, (n_gt, ), generator=g)
        cls[torch.rand(n_gt, generator=g) < 0.3] = -100                  # background GTs
        if int((cls != -100).sum()) == 0:
            continue                                                     # (instance_loss returns early)
        for mlq in (False, True):
            for min_pos in (0.0, 0.1):
                want = _reference_labels(ious, cls, 0.5, mlq, min_pos, 18)
                got = _assign_proposals(ious, cls, cls != -100, 0.5, mlq, min_pos, 18)
                assert torch.equal(want, got), (trial, mlq, min_pos)
                checked += 1
    assert checked > 1000
