"""Generates golden vectors of the REFERENCE'S OWN ``forward_train`` (authoring container only:
needs /root/reference).

The reference's ``SoftGroup`` class (softgroup/model/softgroup.py, imported from where it lies) is
built from the ``model:`` section of its own YAML, loaded with our seeded synthetic weights, put in
training mode by its own ``train()`` override (:98-104: frozen modules keep BatchNorm in eval, the
rest uses batch statistics) and ``model(batch, return_loss=True)`` (:113-155 ``forward_train``,
:157-175 ``point_wise_loss``, :177-262 ``instance_loss``, :264-298 ``parse_losses``) is executed AS
WRITTEN -- on the CPU, over the C-oracle-backed stand-ins of oracle/facade.py for spconv and
softgroup.ops.  ``torch.manual_seed(SEED)`` right before the call fixes the two ``torch.rand(3)``
draws of ``clusters_voxelization(rand_quantize=True)`` (:690-694; CPU generator in the reference,
hence also in softgroup_amd).

Ground truth of a case: the instances are the model's own eval-mode proposals (first the grouping
runs through the oracle restatement, then every proposal becomes a GT instance of class
``id % instance_classes``), so that every loss term is live -- positives, negatives, mask labels,
IoU regression.  The GT arrays are stored next to the losses; tests only read the file.

Output (committed): tests/golden/ref_train_<case>.npz -- GT arrays + every entry of ``log_vars``.
tests/test_train_gpu.py::test_forward_train_losses_match_reference compares softgroup_amd's
``forward_train`` on the GPU with it (<= 1e-4 relative per term, counts exact).

Usage:  python tests/golden/make_ref_train.py
"""
import copy
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_ref_forward as G  # noqa: E402
from oracle import facade  # noqa: E402
from oracle.model import OracleSoftGroup  # noqa: E402
from softgroup_amd import synthetic  # noqa: E402

SEED = 1234

# case -> forward case (scene + yaml) and overrides of the model section
CASES = {
    'scannet_frozen': dict(forward='scannet'),                         # fine-tune stage: frozen backbone
    'scannet_full': dict(forward='scannet', fixed_modules=[]),         # every BatchNorm on batch statistics
    'stpls3d_pp': dict(forward='stpls3d_pp'),                          # semantic_weight, match_low_quality,
    #                                                                    octree + pyramid grouping in training
    's3dis_fold5': dict(forward='s3dis', x4_split=False, scene=dict(seed=3, n=50000, room_scale=0.55)),
    #                                                                    BASELINE config 3's own YAML: 13 classes,
    #                                                                    sem2ins_classes, its fixed_modules; training
    #                                                                    batches are whole crops (x4_split is test-only)
}


def case_batch(case):
    """the deterministic scene + batch dict of a training case (generator and tests)"""
    c = CASES[case]
    if 'x4_split' not in c:
        return G.make_case_batch(c['forward'])
    fc = G.CASES[c['forward']]
    xyz, rgb, inst = synthetic.scene_s2(**c.get('scene', fc['scene']))      # (a scene big enough for the
    #                                 S3DIS class means: 0.05 x 1 724 .. 12 210 points per kept cluster)
    xyz = (xyz * np.float32(fc['xyz_scale'])).astype(np.float32)
    if fc.get('one_channel'):
        rgb = rgb[:, :1].copy()
    return synthetic.make_batch(xyz, rgb, scale=fc['vox_scale'], instance_labels=inst,
                                x4_split=c['x4_split']), xyz


def case_cfg(case):
    configs = json.load(open(os.path.join(HERE, 'ref_configs.json')))
    c = CASES[case]
    cfg = copy.deepcopy(configs[G.CASES[c['forward']]['yaml']])
    if 'fixed_modules' in c:
        cfg['fixed_modules'] = c['fixed_modules']
    return cfg


def gt_from_proposals(pidx, n, n_inst_cls, sem_shift, xyz):
    """every proposal becomes a GT instance (a point listed by several proposals goes to the last)"""
    inst = np.full(n, -100, np.int64)
    inst[pidx[:, 1]] = pidx[:, 0]
    ids = np.unique(inst[inst >= 0])
    remap = np.full(int(ids.max()) + 1, -100, np.int64)
    remap[ids] = np.arange(len(ids))
    inst = np.where(inst >= 0, remap[np.clip(inst, 0, None)], inst)
    cls = np.arange(len(ids)) % n_inst_cls
    sem = np.where(inst >= 0, sem_shift + cls[np.clip(inst, 0, None)], 0).astype(np.int64)
    pointnum = np.bincount(inst[inst >= 0], minlength=len(ids)).astype(np.int32)
    off = np.zeros((n, 3), np.float32)
    for i in range(len(ids)):
        m = inst == i
        off[m] = xyz[m].mean(0) - xyz[m]
    return dict(instance_labels=inst, semantic_labels=sem, instance_pointnum=pointnum,
                instance_cls=cls.astype(np.int64), pt_offset_labels=off)


def apply_gt(batch, gt):
    for k, v in gt.items():
        batch[k] = torch.from_numpy(np.asarray(v))
    return batch


# ---------------------------------------------------------------------------------------------------
# GRADIENT goldens (round 5): d loss / d parameter of the reference's forward_train, by autograd through
# the reference's own Python over the differentiable stand-ins of oracle/facade.py (conv: the gradients
# of out[j] = sum_k W_k . in[nbr[j, k]] in float64; voxelization / ROI pool: the C restatements of the
# reference's own backward kernels).
#
# A ReLU whose input lies within rounding distance of zero makes the gradient discontinuous: an
# implementation that differs in the 7th digit of that pre-activation may take the other branch, and
# the gradient of everything upstream moves by that unit's whole contribution.  With ~10^6 ReLU inputs
# per case a handful always sit that close, so a tolerance alone cannot separate "wrong" from "flipped".
# The generator therefore measures it: every ReLU records its inputs with |x| < GRAD_MARGIN * rms(x)
# ("ambiguous"), the backward runs a second time with exactly those units' branches inverted, and the
# per-tensor difference of the two gradients is stored as `slack`: the test allows
# 1e-4 * max|g_ref| + 2 * slack, i.e. a strict tolerance plus only what the measured ambiguous units
# can explain.
GRAD_MARGIN = 5e-6
GRAD_CASES = {
    # case -> which gradients are stored (None: every trainable parameter)
    'scannet_frozen': None,
    's3dis_fold5': None,
    'scannet_full': ('tiny_unet', 'cls_linear', 'mask_linear', 'iou_score_linear', 'semantic_linear',
                     'offset_linear', 'input_conv.0.weight', 'unet.blocks.block0.conv_branch.2.weight',
                     'unet.blocks.block1.conv_branch.5.weight', 'unet.conv.2.weight', 'unet.deconv.2.weight',
                     'unet.u.u.blocks.block0.conv_branch.2.weight', 'unet.blocks_tail.block0.i_branch.0.weight',
                     'unet.blocks.block0.conv_branch.0', 'unet.u.conv.0', 'unet.u.u.u.deconv.0', 'output_layer.0'),
}


class _ProbeReLUFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, probe):
        ctx.save_for_backward(x)
        ctx.probe = probe
        return x.clamp(min=0)

    @staticmethod
    def backward(ctx, g):
        (x, ) = ctx.saved_tensors
        rms = float(x.detach().double().pow(2).mean().sqrt())
        close = x.abs() < GRAD_MARGIN * max(rms, 1e-12)
        st = ctx.probe.stats
        st['inputs'] += x.numel()
        st['ambiguous'] += int(close.sum())
        mask = x > 0
        if ctx.probe.stats['flip']:
            mask = mask ^ close
        return g * mask.to(g.dtype), None


class ProbeReLU(torch.nn.Module):
    """nn.ReLU with the same forward; backward counts the ambiguous inputs and can invert their branch"""
    stats = dict(inputs=0, ambiguous=0, flip=False)

    def forward(self, x):
        return _ProbeReLUFn.apply(x, self) if torch.is_grad_enabled() and x.requires_grad else x.clamp(min=0)


def _swap_relus(model):
    for parent in model.modules():
        for name, child in list(parent._modules.items()):
            if isinstance(child, torch.nn.ReLU):
                assert not child.inplace
                parent._modules[name] = ProbeReLU()


def reference_gradients(cfg, sd, batch, mod_lvl2, keep):
    """-> (names, grads, slack, stats): gradients of the reference's forward_train loss"""
    out = []
    for flip in (False, True):
        ref, mod = facade.reference_model(cfg, sd)
        if mod_lvl2:
            ref.get_level = G.lvl2
        ref.train()
        _swap_relus(ref)
        ProbeReLU.stats.update(inputs=0, ambiguous=0, flip=flip)
        torch.manual_seed(SEED)
        with facade.cpu_only(mod):
            loss, _ = ref(batch, return_loss=True)
            loss.backward()
        grads = {n: p.grad.detach().clone() for n, p in ref.named_parameters()
                 if p.requires_grad and p.grad is not None and
                 (keep is None or any(n == k or n.startswith(k + '.') for k in keep))}
        out.append((grads, dict(ProbeReLU.stats), float(loss.detach())))
    (g0, st, loss0), (g1, _, _) = out
    names = sorted(g0)
    slack = np.array([float((g0[n] - g1[n]).abs().max()) for n in names], np.float64)
    return names, [g0[n].numpy().astype(np.float32) for n in names], slack, st, loss0


def main():
    only = sys.argv[1:]
    for case, c in CASES.items():
        if only and case not in only:
            continue
        fc = G.CASES[c['forward']]
        cfg = case_cfg(case)
        batch, xyz = case_batch(case)
        sd = synthetic.build_model(cfg, seed=0, device='cpu').state_dict()
        ora = OracleSoftGroup(sd, cfg)
        if fc.get('force_lvl2'):
            ora.get_level = G.lvl2
        sem, off, _ = ora.point_wise(batch)
        pidx, poff = ora.grouping(sem, off, batch['batch_idxs'], batch['coords_float'])
        assert len(poff) - 1 > 3, 'too few proposals for a meaningful GT'
        gt = gt_from_proposals(pidx, xyz.shape[0], cfg['instance_classes'],
                               cfg['semantic_classes'] - cfg['instance_classes'], xyz)
        apply_gt(batch, gt)

        ref, mod = facade.reference_model(cfg, sd)
        if fc.get('force_lvl2'):
            ref.get_level = G.lvl2
        ref.train()                                    # the reference's override (softgroup.py:98-104)
        torch.manual_seed(SEED)
        with facade.cpu_only(mod), torch.no_grad():    # loss VALUES only: the stand-in convs have no autograd
            loss, log_vars = ref(batch, return_loss=True)
        rec = dict(gt, yaml=np.array(fc['yaml']), seed=np.int64(SEED), n_points=np.int64(xyz.shape[0]),
                   xyz_checksum=np.float64(np.abs(xyz.astype(np.float64)).sum()),
                   fixed_modules=np.array(json.dumps(cfg['fixed_modules'])),
                   n_proposals_eval=np.int64(len(poff) - 1),
                   log_keys=np.array(list(log_vars.keys())),
                   log_vals=np.array([float(v) for v in log_vars.values()], np.float64))
        if case in GRAD_CASES:
            names, grads, slack, st, loss_g = reference_gradients(cfg, sd, batch, fc.get('force_lvl2'), GRAD_CASES[case])
            assert abs(loss_g - float(log_vars['loss'])) <= 1e-5 * abs(float(log_vars['loss'])), (loss_g, log_vars['loss'])
            rec.update(grad_names=np.array(names), grad_slack=slack, grad_margin=np.float64(GRAD_MARGIN),
                       grad_relu_inputs=np.int64(st['inputs']), grad_relu_ambiguous=np.int64(st['ambiguous']))
            for i, g in enumerate(grads):
                rec[f'grad_{i:03d}'] = g
            worst = max((float(sl) / max(float(np.abs(g).max()), 1e-30) for sl, g in zip(slack, grads)), default=0.0)
            print(f'  gradients: {len(names)} tensors, {sum(g.size for g in grads)} values; ReLU inputs {st["inputs"]}, '
                  f'ambiguous {st["ambiguous"]}; largest slack / max|g| = {worst:.2e}')
        path = os.path.join(HERE, f'ref_train_{case}.npz')
        np.savez_compressed(path, **rec)
        print(case, 'points', xyz.shape[0], 'GT instances', len(gt['instance_pointnum']),
              {k: round(float(v), 6) for k, v in log_vars.items()}, os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
