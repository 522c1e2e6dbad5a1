"""Training path (BASELINE config 3 shape, small): sparse-conv gradients against PyTorch autograd on
the dense equivalents, and one optimisation step of ``forward_train`` through the HIP operators
(voxelize_bp, conv dgrad/wgrad, global_avg_pool_bp, mask IoU/label kernels)."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import softgroup_amd.spconv.pytorch as spconv
from softgroup_amd import synthetic
from softgroup_amd.model import SoftGroup

pytestmark = pytest.mark.gpu
DEV = 'cuda'
TOL = dict(atol=2e-4, rtol=1e-4)


def _grid(rng, D=9, B=2, p=0.3):
    occ = rng.random((B, D, D, D)) < p
    idx = np.argwhere(occ).astype(np.int32)
    return idx[rng.permutation(len(idx))]


def test_conv_gradients_match_dense_autograd():
    rng = np.random.default_rng(5)
    D, B, Cin, Cmid = 9, 2, 32, 64
    idx = _grid(rng, D, B)
    M = len(idx)
    ti = torch.from_numpy(idx).to(DEV)
    li = ti.long()
    feats = torch.randn(M, Cin, device=DEV, requires_grad=True)
    subm = spconv.SubMConv3d(Cin, Cmid, 3, padding=1, bias=False, indice_key='subm1').to(DEV)
    down = spconv.SparseConv3d(Cmid, Cmid, kernel_size=2, stride=2, bias=False, indice_key='sp1').to(DEV)
    inv = spconv.SparseInverseConv3d(Cmid, Cin, kernel_size=2, bias=False, indice_key='sp1').to(DEV)
    x = spconv.SparseConvTensor(feats, ti, [D] * 3, B)
    y = subm(x)
    d = down(y)
    u = inv(d)
    r1, r2, r3 = (torch.randn_like(t.features) for t in (y, d, u))
    loss = (y.features * r1).sum() + (d.features * r2).sum() + (u.features * r3).sum()
    loss.backward()
    got = [feats.grad.clone(), subm.weight.grad.clone(), down.weight.grad.clone(), inv.weight.grad.clone()]

    # dense reference with the same weights
    f2 = feats.detach().clone().requires_grad_(True)
    ws = [w.detach().clone().requires_grad_(True) for w in (subm.weight, down.weight, inv.weight)]
    dense = torch.zeros(B, D, D, D, Cin, device=DEV)
    dense = dense.index_put((li[:, 0], li[:, 1], li[:, 2], li[:, 3]), f2).permute(0, 4, 1, 2, 3)
    yd = F.conv3d(dense, ws[0].permute(0, 4, 1, 2, 3), padding=1)
    active = torch.zeros(B, 1, D, D, D, device=DEV)
    active[li[:, 0], 0, li[:, 1], li[:, 2], li[:, 3]] = 1
    yd = yd * active                                            # submanifold: only active sites exist
    dd = F.conv3d(yd, ws[1].permute(0, 4, 1, 2, 3), stride=2)
    oi = d.indices.long()
    amask = torch.zeros(B, 1, D // 2, D // 2, D // 2, device=DEV)
    amask[oi[:, 0], 0, oi[:, 1], oi[:, 2], oi[:, 3]] = 1
    dd = dd * amask
    ud = F.conv_transpose3d(dd, ws[2].permute(4, 0, 1, 2, 3), stride=2)
    ud = F.pad(ud, (0, 1, 0, 1, 0, 1))                           # back to extent 9 (last plane = 0)
    ys = yd[li[:, 0], :, li[:, 1], li[:, 2], li[:, 3]]
    ds = dd[oi[:, 0], :, oi[:, 1], oi[:, 2], oi[:, 3]]
    us = ud[li[:, 0], :, li[:, 1], li[:, 2], li[:, 3]]
    np.testing.assert_allclose(y.features.detach().cpu().numpy(), ys.detach().cpu().numpy(), **TOL)
    np.testing.assert_allclose(u.features.detach().cpu().numpy(), us.detach().cpu().numpy(), **TOL)
    ((ys * r1).sum() + (ds * r2).sum() + (us * r3).sum()).backward()
    ref = [f2.grad, ws[0].grad, ws[1].grad, ws[2].grad]
    for a, b, name in zip(got, ref, ['d_feats', 'd_W_subm', 'd_W_down', 'd_W_inverse']):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=2e-3, rtol=1e-3, err_msg=name)


def _train_batch(n=30000):
    xyz, rgb, inst = synthetic.scene_s2(seed=21, n=n, room_scale=0.45)
    return synthetic.make_batch(xyz, rgb, instance_labels=inst)


@pytest.mark.parametrize('frozen', [True, False])
def test_forward_train_step(frozen):
    cfg = copy.deepcopy(synthetic.SCANNET_MODEL_CFG)
    if not frozen:
        cfg['fixed_modules'] = []
        cfg['channels'] = 16
        cfg['num_blocks'] = 3
    torch.manual_seed(0)
    model = SoftGroup(**cfg).to(DEV)
    with torch.no_grad():   # peaky semantic head so that proposals exist on untrained weights
        model.semantic_linear[-1].weight.normal_(0, 20.0)
    model.train()
    if frozen:   # frozen backbone keeps its BatchNorm in eval mode (softgroup.py:98-104)
        assert not model.unet.blocks.block0.conv_branch[0].training
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-3)
    batch = _train_batch()
    # ground truth that overlaps the (untrained) model's own proposals, so that the IoU / mask-label
    # kernels produce positives and every loss term is active
    model.eval()
    with torch.no_grad():
        b = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
        from softgroup_amd import ops
        vf = ops.voxelization(torch.cat((b['feats'], b['coords_float']), 1), b['p2v_map'])
        x = spconv.SparseConvTensor(vf, b['voxel_coords'].int(), b['spatial_shape'], 1)
        sem, off, _ = model.forward_backbone(x, b['v2p_map'])
        pidx, poff = model.forward_grouping(sem, off, b['batch_idxs'], b['coords_float'])
    assert poff.numel() > 3
    inst = torch.full((batch['coords_float'].shape[0], ), -100, dtype=torch.long)
    inst[pidx[:, 1].long().cpu()] = pidx[:, 0].long().cpu()
    ids = torch.unique(inst[inst >= 0])
    remap = torch.full((int(ids.max()) + 1, ), -100, dtype=torch.long)
    remap[ids] = torch.arange(ids.numel())
    inst = torch.where(inst >= 0, remap[inst.clamp(min=0)], inst)
    batch['instance_labels'] = inst
    batch['instance_pointnum'] = torch.bincount(inst[inst >= 0], minlength=ids.numel()).int()
    batch['instance_cls'] = (torch.arange(ids.numel()) % 18).long()
    batch['semantic_labels'] = torch.where(inst >= 0, 2 + batch['instance_cls'][inst.clamp(min=0)],
                                           torch.zeros_like(inst))
    model.train()
    loss, log_vars = model(batch, return_loss=True)
    assert torch.isfinite(loss) and {'semantic_loss', 'offset_loss', 'cls_loss', 'mask_loss',
                                     'iou_score_loss', 'loss'} <= set(log_vars)
    opt.zero_grad()
    loss.backward()
    grads = {n: p.grad for n, p in model.named_parameters() if p.requires_grad}
    assert all(g is not None and torch.isfinite(g).all() for g in grads.values())
    if frozen:    # GT was built from this (eval-BN) model's proposals: every instance term is live
        assert log_vars['cls_loss'] > 0 and log_vars['mask_loss'] > 0
        assert grads['tiny_unet.blocks.block0.conv_branch.2.weight'].abs().sum() > 0
        assert grads['mask_linear.2.weight'].abs().sum() > 0
        assert grads['iou_score_linear.weight'].abs().sum() > 0
    if not frozen:
        assert grads['input_conv.0.weight'].abs().sum() > 0
        assert grads['unet.u.conv.2.weight'].abs().sum() > 0 and grads['unet.deconv.2.weight'].abs().sum() > 0
    watched = model.tiny_unet.blocks.block0.conv_branch[2].weight if frozen else model.input_conv[0].weight
    before = watched.detach().clone()
    opt.step()
    assert not torch.equal(before, watched)
    loss2, _ = model(batch, return_loss=True)
    assert torch.isfinite(loss2)


def test_ddp_bf16_autocast_step():
    """BASELINE config 3 shape at world_size 1: DistributedDataParallel over RCCL ('nccl' backend on
    ROCm), bf16 autocast around the forward (tools/train.py:47 uses fp16 autocast + GradScaler; bf16
    needs no scaler), frozen backbone as in softgroup_s3dis_fold5.yaml."""
    import os
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        cfg = copy.deepcopy(synthetic.S3DIS_MODEL_CFG)
        cfg['test_cfg']['x4_split'] = False
        torch.manual_seed(0)
        model = SoftGroup(**cfg).to(DEV)
        with torch.no_grad():
            model.semantic_linear[-1].weight.normal_(0, 20.0)
        model.train()
        ddp = DDP(model, device_ids=[0], find_unused_parameters=True)
        opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-3)
        batch = _train_batch(20000)
        batch['semantic_labels'] = batch['semantic_labels'].clamp(max=12)
        batch['instance_cls'] = batch['instance_cls'].clamp(max=12)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            loss, log_vars = ddp(batch, return_loss=True)
        assert torch.isfinite(loss)
        opt.zero_grad()
        loss.backward()
        g = model.cls_linear.weight.grad
        assert g is not None and torch.isfinite(g).all()
        opt.step()
    finally:
        dist.destroy_process_group()
