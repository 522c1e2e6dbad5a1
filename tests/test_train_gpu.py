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


def _bf16_ulp_close(got, ref, what):
    """bf16 result vs the same sum in fp32: half a bf16 ulp of rounding plus the fp32 summation-order
    noise of a few hundred terms -> 2^-7 relative on the row scale"""
    got, ref = got.float().cpu().numpy(), ref.float().cpu().numpy()
    scale = np.abs(ref).max(axis=1, keepdims=True) + 1e-6
    err = np.abs(got - ref) / scale
    assert err.max() <= 2.0 ** -7, (what, float(err.max()))


@pytest.mark.parametrize('cin,cout,D,B', [(32, 32, 10, 2), (64, 96, 10, 2), (6, 32, 10, 2), (160, 224, 7, 1),
                                          (48, 40, 9, 1), (32, 64, 48, 1)])
def test_bf16_conv_matches_fp32_math_on_rounded_operands(cin, cout, D, B):
    """sg_spconv_gather_conv_bf16 (MFMA bf16, fp32 accumulation) against the fp32 kernel fed with the
    same bf16-rounded features and weights: SubM k3, strided k2 and its inverse; small grids take
    the offset-split path, the 48^3 grid (~39 k rows, 1.2 k tiles) the direct one; Cin = 6 / 48 the
    unaligned gather."""
    rng = np.random.default_rng(cin * 1000 + cout)
    idx = _grid(rng, D, B, p=0.35)
    ti = torch.from_numpy(idx).to(DEV)
    M = len(idx)
    torch.manual_seed(cin + cout)
    feats = torch.randn(M, cin, device=DEV).bfloat16()
    subm = spconv.SubMConv3d(cin, cout, 3, padding=1, bias=False, indice_key='s').to(DEV)
    down = spconv.SparseConv3d(cout, cout, kernel_size=2, stride=2, bias=False, indice_key='d').to(DEV)
    inv = spconv.SparseInverseConv3d(cout, cin, kernel_size=2, bias=False, indice_key='d').to(DEV)
    ref_mods = [copy.deepcopy(m) for m in (subm, down, inv)]
    for m in ref_mods:
        with torch.no_grad():
            m.weight.copy_(m.weight.bfloat16().float())
    with torch.no_grad():
        x = spconv.SparseConvTensor(feats, ti, [D] * 3, B)
        y = subm(x)
        d = down(y)
        u = inv(d)
        assert y.features.dtype == d.features.dtype == u.features.dtype == torch.bfloat16
        xr = spconv.SparseConvTensor(feats.float(), ti, [D] * 3, B)
        yr = ref_mods[0](xr)
        # the next layer sees the bf16-rounded output of the previous one in both paths
        dr = ref_mods[1](yr.replace_feature(y.features.float()))
        ur = ref_mods[2](dr.replace_feature(d.features.float()))
    _bf16_ulp_close(y.features, yr.features, 'subm')
    _bf16_ulp_close(d.features, dr.features, 'down')
    _bf16_ulp_close(u.features, ur.features, 'inverse')
    assert torch.equal(d.indices, dr.indices)


def test_bf16_autocast_gradients_and_deterministic_wgrad():
    """Under bf16 autocast the convs run forward, dgrad and wgrad on the bf16 kernels: gradients
    against fp32 dense autograd at bf16 tolerance; two backward passes give bit-identical weight
    gradients (fixed-order chunk sums, no atomics)."""
    rng = np.random.default_rng(9)
    D, B, Cin, Cmid = 11, 2, 32, 64
    idx = _grid(rng, D, B, p=0.4)
    ti = torch.from_numpy(idx).to(DEV)
    li = ti.long()
    M = len(idx)
    torch.manual_seed(3)
    feats = torch.randn(M, Cin, device=DEV, requires_grad=True)
    subm = spconv.SubMConv3d(Cin, Cmid, 3, padding=1, bias=False, indice_key='subm1').to(DEV)
    subm2 = spconv.SubMConv3d(Cmid, Cmid, 3, padding=1, bias=False, indice_key='subm1').to(DEV)
    r = torch.randn(M, Cmid, device=DEV)

    def run():
        for t in (feats, subm.weight, subm2.weight):
            t.grad = None
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y = subm2(subm(spconv.SparseConvTensor(feats, ti, [D] * 3, B)))
        assert y.features.dtype == torch.bfloat16
        (y.features.float() * r).sum().backward()
        return [feats.grad.clone(), subm.weight.grad.clone(), subm2.weight.grad.clone()]

    got = run()
    again = run()
    for a, b in zip(got, again):
        assert torch.equal(a, b)
    assert got[1].dtype == torch.float32 and got[0].dtype == torch.float32

    f2 = feats.detach().clone().requires_grad_(True)
    ws = [w.detach().clone().requires_grad_(True) for w in (subm.weight, subm2.weight)]
    dense = torch.zeros(B, D, D, D, Cin, device=DEV)
    dense = dense.index_put((li[:, 0], li[:, 1], li[:, 2], li[:, 3]), f2).permute(0, 4, 1, 2, 3)
    active = torch.zeros(B, 1, D, D, D, device=DEV)
    active[li[:, 0], 0, li[:, 1], li[:, 2], li[:, 3]] = 1
    y1 = F.conv3d(dense, ws[0].permute(0, 4, 1, 2, 3), padding=1) * active
    y2 = F.conv3d(y1, ws[1].permute(0, 4, 1, 2, 3), padding=1) * active
    (y2[li[:, 0], :, li[:, 1], li[:, 2], li[:, 3]] * r).sum().backward()
    for a, b, name in zip(got, [f2.grad, ws[0].grad, ws[1].grad], ['d_feats', 'd_W1', 'd_W2']):
        a, b = a.cpu().numpy(), b.cpu().numpy()
        rel = np.abs(a - b).max() / np.abs(b).max()
        assert rel < 3e-2, (name, rel)
        # and no systematic loss: the gradients correlate almost perfectly
        assert np.corrcoef(a.ravel(), b.ravel())[0, 1] > 0.9995, name


def test_wgrad_bf16_and_fp32_operands_agree_and_channel_shapes():
    """sg_spconv_wgrad over the channel shapes of the model (vector widths 1 and 2, Cin = 6, odd
    multiples of 32, concatenated 2C inputs) against a dense gather + matmul in fp64."""
    from softgroup_amd.spconv import core
    rng = np.random.default_rng(13)
    idx = _grid(rng, 12, 2, p=0.3)
    ti = torch.from_numpy(idx).to(DEV)
    M = len(idx)
    rule = core.SubMRule(ti, [12] * 3)
    nbr = rule.plan.nbr.long()
    for cin, cout in [(6, 32), (32, 32), (64, 32), (96, 96), (192, 96), (160, 224), (40, 72)]:
        torch.manual_seed(cin)
        x = torch.randn(M, cin, device=DEV)
        g = torch.randn(M, cout, device=DEV)
        xz = torch.cat([x, torch.zeros(1, cin, device=DEV)]).double()
        ref = torch.stack([xz[nbr[:, k]].T @ g.double() for k in range(27)])          # [K, Cin, Cout]
        dw = core.conv_wgrad(x, g, rule.plan, cin, cout)
        np.testing.assert_allclose(dw.cpu().numpy(), ref.cpu().numpy(), atol=2e-3, rtol=1e-4)
        assert torch.equal(dw, core.conv_wgrad(x, g, rule.plan, cin, cout))
        xb, gb = x.bfloat16(), g.bfloat16()
        refb = torch.stack([torch.cat([xb.double(), torch.zeros(1, cin, device=DEV).double()])[nbr[:, k]].T
                            @ gb.double() for k in range(27)])
        for a, b in ((xb, gb), (xb, gb.float()), (xb.float(), gb)):
            dwb = core.conv_wgrad(a.contiguous(), b.contiguous(), rule.plan, cin, cout)
            np.testing.assert_allclose(dwb.cpu().numpy(), refb.cpu().numpy(), atol=2e-3, rtol=1e-4)


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


# ------------------------------------------------------------------------------------------------
# forward_train against the REFERENCE'S OWN forward_train (tests/golden/ref_train_*.npz, generated
# by tests/golden/make_ref_train.py: the reference's Python executed as written on the CPU oracle)
import json  # noqa: E402
import os  # noqa: E402
import sys  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import make_ref_forward as G  # noqa: E402
import make_ref_train as GT  # noqa: E402

LOSS_RTOL = 1e-4        # north_star: floats within 1e-4


def _train_case(case):
    g = np.load(os.path.join(HERE, 'golden', f'ref_train_{case}.npz'))
    fwd = GT.CASES[case]['forward']
    cfg = GT.case_cfg(case)
    batch, xyz = GT.case_batch(case)
    assert abs(np.abs(xyz.astype(np.float64)).sum() - float(g['xyz_checksum'])) < 1e-6, 'scene drifted'
    GT.apply_gt(batch, {k: g[k] for k in ('instance_labels', 'semantic_labels', 'instance_pointnum',
                                          'instance_cls', 'pt_offset_labels')})
    model = synthetic.build_model(cfg, seed=0)
    if G.CASES[fwd].get('force_lvl2'):
        model.get_level = G.lvl2
    ref = dict(zip([str(k) for k in g['log_keys']], g['log_vals'].tolist()))
    return model, batch, ref, int(g['seed'])


@pytest.mark.parametrize('case', sorted(GT.CASES))
def test_forward_train_losses_match_reference(case):
    """every entry of log_vars (semantic / offset / cls / mask / iou_score loss, num_pos, num_neg,
    total) of softgroup_amd's forward_train in fp32 == the reference's forward_train on the same
    batch, weights and seed, within 1e-4 relative (counts exact).  Reference lines:
    softgroup/model/softgroup.py:113-298."""
    model, batch, ref, seed = _train_case(case)
    model.train()
    torch.manual_seed(seed)
    loss, log_vars = model(batch, return_loss=True)
    print(case, {k: (round(log_vars[k], 6), round(ref[k], 6)) for k in ref})
    assert list(log_vars) == list(ref)
    for k, want in ref.items():
        got = log_vars[k]
        if k.startswith('num_'):
            assert got == want, (k, got, want)
        else:
            assert abs(got - want) <= LOSS_RTOL * max(abs(want), 1e-3), (k, got, want)
    assert abs(float(loss) - ref['loss']) <= LOSS_RTOL * abs(ref['loss'])
    if case != 'scannet_full':
        loss.backward()      # the same graph trains: gradients reach the refinement heads
        assert model.cls_linear.weight.grad.abs().sum() > 0


@pytest.mark.parametrize('case', sorted(GT.GRAD_CASES))
def test_forward_train_gradients_match_reference(case):
    """d loss / d parameter of softgroup_amd's forward_train + backward on the GPU (training executor:
    sg_unet_train_forward / _backward; the operators' own backward kernels) against autograd through the
    REFERENCE'S forward_train (softgroup/model/softgroup.py:113-298 executed as written over the
    differentiable oracle stand-ins, tests/golden/make_ref_train.py `reference_gradients`).
    Tolerance per tensor: 1e-4 of the tensor's largest reference entry, plus twice the `slack` the
    generator measured for that tensor by inverting the branch of the ReLU inputs that lie within
    5e-6 rms of zero (a handful of ~10^6: units that another fp32 summation order may flip; 0 for the
    tensors behind the last ReLU).  Not a comparison with the module path."""
    g = np.load(os.path.join(HERE, 'golden', f'ref_train_{case}.npz'))
    names = [str(n) for n in g['grad_names']]
    model, batch, ref, seed = _train_case(case)
    model.train()
    torch.manual_seed(seed)
    loss, _ = model(batch, return_loss=True)
    loss.backward()
    params = dict(model.named_parameters())
    worst, strict, need_slack = [], 0, []
    # (a parameter whose true gradient is zero -- the bias of a Linear in front of a training-mode
    # BatchNorm -- carries only rounding noise (sums of ~10^4 cancelling terms): floor of 1e-5 of the largest
    # gradient entry of the case)
    floor = 1e-5 * max(float(np.abs(g[f'grad_{i:03d}']).max()) for i in range(len(names)))
    for i, n in enumerate(names):
        want = torch.from_numpy(g[f'grad_{i:03d}'])
        p = params[n]
        got = torch.zeros_like(want) if p.grad is None else p.grad.detach().float().cpu()
        assert got.shape == want.shape, n
        scale = float(want.abs().max())
        slack = float(g['grad_slack'][i])
        tol = 1e-4 * scale + 2.0 * slack + floor
        err = float((got - want).abs().max())
        worst.append((err / max(scale, 1e-30), n, err, tol))
        strict += slack == 0.0
        if err > 1e-4 * scale + floor:      # passes only thanks to the tensor's ReLU-flip slack: say so
            need_slack.append((n, round(err / max(scale, 1e-30), 6)))
        assert err <= tol, f'{n}: max |d| {err:.3e} > {tol:.3e} (scale {scale:.3e}, slack {slack:.3e})'
    worst.sort(reverse=True)
    print(case, f'{len(names)} tensors ({strict} with zero slack), ambiguous ReLU inputs '
                f'{int(g["grad_relu_ambiguous"])} of {int(g["grad_relu_inputs"])}; worst relative errors:',
          [(round(w[0], 7), w[1]) for w in worst[:3]],
          f'; tensors above 1e-4 of their scale, inside their slack only: {len(need_slack)}', need_slack[:6])


@pytest.mark.parametrize('case,rel,abs_tol,min_agree', [('scannet_frozen', 0.02, 5e-3, 0.9), ('scannet_full', 0.05, 2e-2, 0.0)])
def test_bf16_autocast_losses_explained(case, rel, abs_tol, min_agree):
    """bf16 autocast vs fp32 on the same batch, weights and seed (BASELINE config 3 precision), with the
    backbone frozen (fine-tune configs) and with NOTHING frozen (`scannet_full`: training-mode BatchNorm through
    the whole backbone; profiles/r05_train_step.txt shows total losses 12.5 fp32 vs 9.5 bf16 on such a step --
    this is the decomposition of that gap: with the fp32 run's proposals every term agrees within 5 %, the
    bf16 run's OWN proposals are a different sample altogether -- a random-init backbone in train() mode puts
    most softmax scores near the 0.2 threshold -- so no overlap bound is asserted there, it is printed).
    The two runs differ for two reasons that this test separates:
      (1) rounding: with the PROPOSALS of the fp32 run handed to the bf16 run, every loss term
          agrees within bf16 accuracy through ~60 layers (2 % of the term, 5e-3 absolute);
      (2) proposal flips: soft grouping thresholds softmax scores at 0.2 and a bf16 backbone moves
          borderline points across it, so the bf16 run's OWN proposals differ in a few points or
          clusters -- a different (equally valid) sample of proposals, which moves the instance
          losses by more than rounding does.  Reported, and bounded by the share of proposal
          points both runs agree on."""
    model, batch, ref, seed = _train_case(case)
    model.train()
    model.use_native_scan = False      # the proposals are intercepted at forward_grouping (the native
    #                                    driver returns the same ones, tests/test_native_scan_gpu.py)
    torch.manual_seed(seed)
    _, fp32 = model(batch, return_loss=True)
    keep = {}
    orig = model.forward_grouping

    def record(*a, **k):
        keep['p'] = orig(*a, **k)
        return keep['p']

    model.forward_grouping = record
    torch.manual_seed(seed)
    model(batch, return_loss=True)
    p32 = keep['p']
    torch.manual_seed(seed)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        _, own = model(batch, return_loss=True)
    p16 = keep['p']
    model.forward_grouping = lambda *a, **k: p32
    torch.manual_seed(seed)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        _, same = model(batch, return_loss=True)
    a = set(map(tuple, p32[0].cpu().numpy()[:, 1:2].tolist()))
    b = set(map(tuple, p16[0].cpu().numpy()[:, 1:2].tolist()))
    agree = len(a & b) / max(len(a | b), 1)
    print(case, 'fp32            ', {k: round(v, 5) for k, v in fp32.items()})
    print(case, 'bf16, fp32 props', {k: round(v, 5) for k, v in same.items()})
    print(case, 'bf16, own props ', {k: round(v, 5) for k, v in own.items()},
          f'proposal points shared {agree:.4f}; proposals {p32[1].numel() - 1} vs {p16[1].numel() - 1}')
    for k, want in fp32.items():
        if k.startswith('num_'):
            assert same[k] == want
        else:
            assert abs(same[k] - want) <= rel * abs(want) + abs_tol, (k, same[k], want)
    assert agree >= min_agree


def test_bench_ddp_leg_runs_on_rccl():
    """bench.py's N > 1 leg (gradient all-reduce of the trainable heads on the job's RCCL group + a DDP
    training step per rank, softgroup_s3dis_fold5.yaml shapes) executed on the 'nccl' backend at world
    size 1 -- the GPU box exposes one device; the world-2 path of the same function runs on gloo in
    tests/test_bench_launch.py."""
    import argparse
    import os
    import sys
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29541')
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        rec = bench.ddp_leg(argparse.Namespace(points=30000), 0, 1, 0, False)
    finally:
        dist.destroy_process_group()
    assert rec['backend'] == 'nccl' and rec['allreduce_bytes'] == 2919208
    assert rec['trainable_bytes'] == 2919208            # what DDP all-reduces: the heads, not the frozen backbone
    assert 0 < rec['allreduce_ms_min'] and 0 < rec['train_ms_per_step_min'] < 1000


def test_bf16_autocast_full_model_follows_fp32_from_the_same_weights():
    """The full-model lines of tools/train_step_bench.py up to round 5 (profiles/r05_train_step.txt: total loss 12.51
    in fp32, 9.50 under bf16 autocast) were NOT a precision gap: the tool timed 13 optimisation steps in fp32 and then
    13 MORE under autocast on the same model and optimiser and printed each block's last loss -- the bf16 figure is the
    loss after 26 steps.  From the SAME weights (S3DIS model section, nothing frozen, one S2 scene, semantic head
    re-drawn with std 20) the two precisions follow each other: every loss term of every one of 4 steps within 5 %."""
    import copy
    from softgroup_amd.model import SoftGroup
    cfg = copy.deepcopy(synthetic.S3DIS_MODEL_CFG)
    cfg['test_cfg']['x4_split'] = False
    cfg['fixed_modules'] = []
    xyz, rgb, inst = synthetic.scene_s2(seed=21, n=60000, room_scale=0.65)
    batch = synthetic.make_batch(xyz, rgb, instance_labels=inst)
    batch['semantic_labels'] = batch['semantic_labels'].clamp(max=12)
    batch['instance_cls'] = batch['instance_cls'].clamp(max=12)
    runs = {}
    for name in ('fp32', 'bf16'):
        torch.manual_seed(0)
        model = SoftGroup(**cfg).cuda()
        with torch.no_grad():
            model.semantic_linear[-1].weight.normal_(0, 20.0)
        model.train()
        opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4)
        ctx = torch.autocast('cuda', dtype=torch.bfloat16) if name == 'bf16' else torch.autocast('cuda', enabled=False)
        logs = []
        for it in range(4):
            torch.manual_seed(100 + it)
            with ctx:
                loss, log = model(batch, return_loss=True)
            opt.zero_grad()
            loss.backward()
            opt.step()
            logs.append(log)
        runs[name] = logs
    for it, (a, b) in enumerate(zip(runs['fp32'], runs['bf16'])):
        print(f'step {it}: fp32', {k: round(v, 4) for k, v in a.items()}, '| bf16', {k: round(v, 4) for k, v in b.items()})
        for k in ('semantic_loss', 'offset_loss', 'loss'):
            assert abs(a[k] - b[k]) <= 0.05 * abs(a[k]) + 2e-2, (it, k, a[k], b[k])
