"""Deterministic synthetic inputs of the shapes BASELINE.json names (no datasets offline).

S1  BASELINE config 1: 20k uniform points in a 2.56 m cube, 0.02 m voxels
S2  BASELINE config 2 proxy: 150k-point "room shell" (SURVEY App. D), ScanNet-like
G1  grouping-head input: 40 Gaussian blobs + uniform noise, 50k points

``make_batch`` assembles the batch dictionary exactly as the reference's ``collate_fn`` does
(softgroup/data/custom.py:196-256), including the CPU ``voxelization_idx`` call.
"""
import numpy as np
import torch

from . import ops

SCANNET_MODEL_CFG = dict(
    channels=32, num_blocks=7, semantic_classes=20, instance_classes=18, sem2ins_classes=[],
    semantic_only=False, ignore_label=-100,
    grouping_cfg=dict(score_thr=0.2, radius=0.04, mean_active=300,
                      class_numpoint_mean=[-1., -1., 3917., 12056., 2303., 8331., 3948., 3166.,
                                           5629., 11719., 1003., 3317., 4912., 10221., 3889., 4136.,
                                           2120., 945., 3967., 2589.],
                      npoint_thr=0.05, ignore_classes=[0, 1]),
    instance_voxel_cfg=dict(scale=50, spatial_shape=20),
    train_cfg=dict(max_proposal_num=200, pos_iou_thr=0.5),
    test_cfg=dict(x4_split=False, cls_score_thr=0.001, mask_score_thr=-0.5, min_npoint=100,
                  eval_tasks=['semantic', 'instance']),
    fixed_modules=['input_conv', 'unet', 'output_layer', 'semantic_linear', 'offset_linear'])
"""configs/softgroup/softgroup_scannet.yaml `model:` section (reference), verbatim values."""


STPLS3D_PP_MODEL_CFG = dict(
    channels=16, num_blocks=7, semantic_classes=15, instance_classes=14, sem2ins_classes=[],
    semantic_only=False,
    semantic_weight=[1.0, 1.0, 44.0, 21.9, 1.8, 25.1, 31.5, 21.8, 24.0, 54.4, 114.4, 81.2, 43.6, 9.7,
                     22.4],
    ignore_label=-100, with_coords=False,
    grouping_cfg=dict(with_pyramid=True, pyramid_base_size=0.3333, with_octree=True, score_thr=0.2,
                      radius=0.9, mean_active=3,
                      class_numpoint_mean=[-1., 10408., 58., 124., 1351., 162., 430., 1090., 451., 26.,
                                           43., 61., 39., 109., 1239],
                      npoint_thr=0.01, ignore_classes=[0]),
    instance_voxel_cfg=dict(scale=3, spatial_shape=20),
    train_cfg=dict(lvl_fusion=True, max_proposal_num=300, pos_iou_thr=0.5, match_low_quality=True,
                   min_pos_thr=0.1),
    test_cfg=dict(x4_split=False, cls_score_thr=0.001, mask_score_thr=-0.5, min_npoint=10,
                  eval_tasks=['semantic', 'instance']),
    fixed_modules=[])
"""configs/softgroup++/softgroup++_stpls3d.yaml `model:` section (reference), verbatim values."""

KITTI_MODEL_CFG = dict(
    in_channels=1, channels=32, num_blocks=7, semantic_classes=19, instance_classes=8,
    sem2ins_classes=[], semantic_only=False, ignore_label=-100, with_coords=False,
    grouping_cfg=dict(score_thr=0.2, radius=0.1, mean_active=300, class_numpoint_mean=[-1.] * 19,
                      npoint_thr=5, ignore_classes=list(range(11))),
    instance_voxel_cfg=dict(scale=20, spatial_shape=20),
    train_cfg=dict(max_proposal_num=200, pos_iou_thr=0.5),
    test_cfg=dict(x4_split=False, cls_score_thr=0.1, mask_score_thr=-0.5, min_npoint=25,
                  eval_tasks=['panoptic'], panoptic_skip_iou=0.5),
    fixed_modules=[])
"""configs/softgroup/softgroup_kitti.yaml `model:` section (reference), verbatim values."""

S3DIS_MODEL_CFG = dict(
    channels=32, num_blocks=7, semantic_classes=13, instance_classes=13, sem2ins_classes=[0, 1],
    semantic_only=False, ignore_label=-100,
    grouping_cfg=dict(score_thr=0.2, radius=0.04, mean_active=300,
                      class_numpoint_mean=[34229, 39796, 12210, 7457, 5439, 10225, 6016, 1724, 5092,
                                           7424, 5279, 6189, 1823],
                      npoint_thr=0.05, ignore_classes=[0, 1]),
    instance_voxel_cfg=dict(scale=50, spatial_shape=20),
    train_cfg=dict(max_proposal_num=200, pos_iou_thr=0.5),
    test_cfg=dict(x4_split=True, cls_score_thr=0.001, mask_score_thr=-0.5, min_npoint=100,
                  eval_tasks=['semantic', 'instance']),
    fixed_modules=['input_conv', 'unet', 'output_layer', 'semantic_linear', 'offset_linear'])
"""configs/softgroup/softgroup_s3dis_fold5.yaml `model:` section (reference), verbatim values."""


def _shell(n, size, rng):
    """n points on the 6 faces of an axis-aligned box"""
    size = np.array(size, dtype=np.float64)
    area = np.array([size[1] * size[2]] * 2 + [size[0] * size[2]] * 2 + [size[0] * size[1]] * 2)
    face = rng.choice(6, size=n, p=area / area.sum())
    p = rng.random((n, 3)) * size
    p[np.arange(n), face // 2] = (face % 2) * size[face // 2]
    return p


def scene_s1(seed=0, n=20000):
    rng = np.random.default_rng(seed)
    xyz = (rng.random((n, 3)) * 2.56).astype(np.float32)
    rgb = rng.standard_normal((n, 3)).astype(np.float32)
    return xyz, rgb


def scene_s2(seed=1, n=150000, room_scale=1.0, class_colour=False):
    """room shell + 12 box-shaped objects; returns xyz, rgb and a synthetic instance labelling.
    ``class_colour``: instead of noise, points carry the colour of their semantic class
    (2 + instance % 18; room = class 0) plus noise, so that a head fitted by
    ``fit_point_heads`` can read the class off the backbone features.
    ``room_scale`` shrinks the room so that smaller test scenes keep the ScanNet-like surface
    density (~900 pts/m^2 at n=150k, scale 1) that makes the 4 cm neighbour graph connected."""
    rng = np.random.default_rng(seed)
    n_room = int(n * 0.6)
    per_obj = (n - n_room) // 12
    room = np.array([6, 5, 2.7]) * room_scale
    parts = [_shell(n_room, room, rng)]
    inst = [np.full(n_room, -100, np.int64)]
    for i in range(12):
        sz = rng.uniform(0.4, 1.5, 3) * room_scale
        sz[2] = rng.uniform(0.4, 1.0) * room_scale
        origin = np.array([rng.uniform(0, room[0] - sz[0]), rng.uniform(0, room[1] - sz[1]), 0])
        parts.append(_shell(per_obj, sz, rng) + origin)
        inst.append(np.full(per_obj, i, np.int64))
    xyz = np.concatenate(parts)
    xyz = (xyz + rng.normal(0, 0.003, xyz.shape)).astype(np.float32)
    rgb = rng.uniform(-1, 1, xyz.shape).astype(np.float32)
    inst = np.concatenate(inst)
    if class_colour:
        palette = np.random.default_rng(12345).uniform(-1, 1, (20, 3))
        cls = np.where(inst >= 0, 2 + inst % 18, 0)
        rgb = (palette[cls] + 0.15 * rgb).astype(np.float32)
    return xyz, rgb, inst


def scene_g1(seed=2, blobs=40, per_blob=1000, noise=10000, sigma=0.03):
    rng = np.random.default_rng(seed)
    ext = np.array([6, 5, 2.7])
    ctr = rng.random((blobs, 3)) * ext
    pts = [ctr[i] + rng.normal(0, sigma, (per_blob, 3)) for i in range(blobs)]
    pts.append(rng.random((noise, 3)) * ext)
    xyz = np.concatenate(pts).astype(np.float32)
    return xyz[rng.permutation(len(xyz))]


def scene_lidar(seed=3, n=120000, beams=64, n_boxes=24):
    """BASELINE config 5 proxy: one LiDAR-like sweep -- `beams` rings (elevation -24.8
    .. +2 degrees, HDL-64 like) x n/beams azimuth steps, rays cast against a ground plane 1.73 m
    below the sensor and `n_boxes` car-sized boxes 4-18 m away; range noise 1 cm.  Returns xyz
    (sensor frame), a 1-channel intensity feature and instance labels (box id, -100 = ground)."""
    if n > 0:        # rays that hit nothing are dropped: cast n / hit-rate rays (hit rate of a first pass)
        got = scene_lidar(seed, -n, beams, n_boxes)[0].shape[0]
        n = -int(n * n / got)
    rng = np.random.default_rng(seed)
    per = -n // beams
    el = np.deg2rad(np.linspace(-24.8, 2.0, beams))[:, None]
    az = (np.arange(per) / per * 2 * np.pi)[None, :] + rng.uniform(0, 2 * np.pi / per, (beams, 1))
    d = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.broadcast_to(np.sin(el), az.shape)], -1)
    d = d.reshape(-1, 3)
    n = len(d)
    t_hit = np.full(n, 60.0)
    inst = np.full(n, -100, np.int64)
    down = d[:, 2] < -1e-3
    t_ground = np.where(down, -1.73 / np.where(down, d[:, 2], -1.0), np.inf)
    t_hit = np.minimum(t_hit, t_ground)
    ang = rng.uniform(0, 2 * np.pi, n_boxes)
    rad = rng.uniform(4.0, 18.0, n_boxes)
    for b in range(n_boxes):
        size = np.array([rng.uniform(3.5, 4.8), rng.uniform(1.6, 2.0), rng.uniform(1.4, 1.9)])
        if rng.random() < 0.5:
            size[[0, 1]] = size[[1, 0]]
        lo = np.array([rad[b] * np.cos(ang[b]), rad[b] * np.sin(ang[b]), -1.73]) - size * [0.5, 0.5, 0]
        hi = lo + size
        with np.errstate(divide='ignore', invalid='ignore'):
            t1, t2 = (lo - 0.0) / d, (hi - 0.0) / d
        tn = np.nanmax(np.minimum(t1, t2), 1)
        tf = np.nanmin(np.maximum(t1, t2), 1)
        hit = (tn <= tf) & (tn > 0.5) & (tn < t_hit)
        t_hit = np.where(hit, tn, t_hit)
        inst = np.where(hit, b, inst)
    keep = t_hit < 59.0
    r = t_hit[keep] + rng.normal(0, 0.01, int(keep.sum()))
    xyz = (d[keep] * r[:, None]).astype(np.float32)
    inst = inst[keep]
    ids = np.unique(inst[inst >= 0])                 # dense ids (a box may be fully occluded)
    remap = np.full(n_boxes, -100, np.int64)
    remap[ids] = np.arange(len(ids))
    inst = np.where(inst >= 0, remap[np.clip(inst, 0, None)], -100)
    intensity = rng.uniform(0, 1, (len(xyz), 1)).astype(np.float32)
    return xyz, intensity, inst


def make_batch(xyz, rgb, scale=50, min_spatial=128, instance_labels=None, semantic_labels=None,
               scan_id='synthetic_0000', x4_split=False):
    """One-scene batch dict with the keys/dtypes of collate_fn (data/custom.py:240-256).
    ``x4_split``: S3DIS test layout (data/s3dis.py:46-115): the scene is cut into 4 interleaved
    sub-clouds (points i, i+4, ...) that become batch items 0..3, stored part-major."""
    n = xyz.shape[0]
    if x4_split:
        part = np.concatenate([np.arange(i, n, 4) for i in range(4)])
        bidx = np.concatenate([np.full(len(range(i, n, 4)), i) for i in range(4)])
        xyz, rgb = xyz[part], rgb[part]
        if instance_labels is not None:
            instance_labels = instance_labels[part]
        if semantic_labels is not None:
            semantic_labels = semantic_labels[part]
    else:
        bidx = np.zeros(n, np.int64)
    xyz64 = xyz.astype(np.float64)
    coord = torch.from_numpy(np.floor((xyz64 - xyz64.min(0)) * scale).astype(np.int64))
    coords = torch.cat([torch.from_numpy(bidx.astype(np.int64))[:, None], coord], 1)   # [N,4]
    spatial_shape = np.clip((coords.max(0)[0][1:] + 1).numpy(), min_spatial, None)
    batch_size = 4 if x4_split else 1
    voxel_coords, v2p_map, p2v_map = ops.voxelization_idx(coords.contiguous(), batch_size)
    if instance_labels is None:
        instance_labels = np.full(n, -100, np.int64)
    if semantic_labels is None:
        semantic_labels = np.where(instance_labels >= 0, 2 + instance_labels % 18, 0).astype(np.int64)
    inst_ids = np.unique(instance_labels[instance_labels >= 0])
    pointnum = np.array([(instance_labels == i).sum() for i in inst_ids], np.int32)
    inst_cls = np.array([semantic_labels[instance_labels == i][0] - 2 for i in inst_ids], np.int64)
    center = np.zeros((n, 3), np.float32)
    for i in inst_ids:
        m = instance_labels == i
        center[m] = xyz[m].mean(0)
    pt_offset_labels = np.where((instance_labels >= 0)[:, None], center - xyz, 0).astype(np.float32)
    return dict(
        scan_ids=[scan_id], coords=coords, batch_idxs=coords[:, 0].int(), voxel_coords=voxel_coords,
        p2v_map=p2v_map, v2p_map=v2p_map, coords_float=torch.from_numpy(xyz),
        feats=torch.from_numpy(rgb), semantic_labels=torch.from_numpy(semantic_labels),
        instance_labels=torch.from_numpy(instance_labels),
        instance_pointnum=torch.from_numpy(pointnum), instance_cls=torch.from_numpy(inst_cls),
        pt_offset_labels=torch.from_numpy(pt_offset_labels), spatial_shape=spatial_shape,
        batch_size=batch_size)


def build_model(cfg=None, seed=0, device='cuda', head_std=20.0):
    """Random-init SoftGroup (reference init scheme) with non-trivial eval-mode BatchNorm
    statistics (running_mean ~ N(0,0.1), running_var ~ U(0.5,1.5)).

    The reference initialises the last layer of the semantic head with std 0.01
    (blocks.py:26), which on untrained weights gives a uniform softmax (1/20 < score_thr = 0.2)
    and therefore NO grouping work.  ``head_std`` re-draws that one layer with std 20 (logit std ~8) so the
    untrained network produces peaky, spatially varying scores and the grouping head, proposal
    voxelisation and tiny U-Net see a realistic load.  Pass ``head_std=None`` for the untouched
    reference init."""
    from .model import SoftGroup
    torch.manual_seed(seed)
    cfg = dict(SCANNET_MODEL_CFG if cfg is None else cfg)
    model = SoftGroup(**cfg)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
        if head_std is not None:
            w = model.semantic_linear[-1].weight
            w.copy_(torch.randn(w.shape, generator=g) * head_std)
    return model.to(device).eval()


def passthrough_init(model, damp=0.2):
    """Part of the stand-in for a trained checkpoint (`fit_model_to_scenes`): re-shape the random
    backbone so that the input colour survives to the output features.
      * `input_conv`: output channels 0..2c-1 carry +-(input channel) of the voxel itself (centre
        tap), the other channels keep their random taps;
      * every residual branch and every decoder (deconv) branch is damped by `damp`, the 1x1
        shortcut of every `blocks_tail.block0` starts as [identity | 0]: the skip path dominates and
        the deep levels act as a perturbation (they still run, at full cost).
    A random U-Net with calibrated BatchNorms scrambles the colour (linear R^2 of rgb from the
    features 0.03-0.16, measured), so no read-out of it finds the synthetic classes."""
    from .model.blocks import ResidualBlock, UBlock
    with torch.no_grad():
        w = model.input_conv[0].weight                     # [Cout, 3, 3, 3, Cin]
        cin = min(w.shape[-1], 3)
        w[:2 * cin] = 0
        for c in range(cin):
            w[2 * c, 1, 1, 1, c] = 1.0
            w[2 * c + 1, 1, 1, 1, c] = -1.0
        for m in model.unet.modules():
            if isinstance(m, ResidualBlock):
                m.conv_branch[5].weight.mul_(damp)
                if not isinstance(m.i_branch[0], torch.nn.Identity):
                    iw = m.i_branch[0].weight             # [C, 1, 1, 1, 2C]
                    iw.mul_(damp)
                    c = iw.shape[0]
                    iw[torch.arange(c), 0, 0, 0, torch.arange(c)] = 1.0
            elif isinstance(m, UBlock) and hasattr(m, 'deconv'):
                m.deconv[2].weight.mul_(damp)
    return model


def _lstsq_head(linear, a, target, ridge=1e-4):
    """linear.weight / bias := ridge least-squares solution of [a, 1] w = target"""
    a1 = torch.cat([a, torch.ones_like(a[:, :1])], 1).double()
    g = a1.t() @ a1
    g += ridge * g.diagonal().mean() * torch.eye(g.shape[0], dtype=g.dtype, device=g.device)
    w = torch.linalg.solve(g, a1.t() @ target.double()).float()
    linear.weight.copy_(w[:-1].t())
    linear.bias.copy_(w[-1])


def fit_point_heads(model, feats, semantic_labels, instance_labels, pt_offset_labels, steps=300, lr=0.02,
                    fit_offsets=False):
    """A few hundred Adam steps on the point-wise heads (both layers, BatchNorm in eval mode) over
    fixed backbone features: cross-entropy on the semantic labels and, with ``fit_offsets``, L1 on
    the offsets of instance points.  By default the offset head is set to predict ZERO instead: the
    centre of a box-shaped object is not a function of local colour / geometry, a fitted head is
    off by ~8 cm (twice the grouping radius) and tears the surfaces apart, whereas the synthetic
    objects are connected surfaces of one class each and cluster on their raw coordinates.
    Works on any device; returns the final loss."""
    heads = [model.semantic_linear, model.offset_linear]
    params = [p for h in heads for p in h.parameters()]
    was = [p.requires_grad for p in params]
    for h in heads:
        h.eval()
    feats = feats.detach().float()
    valid = semantic_labels >= 0
    pos = instance_labels >= 0
    with torch.enable_grad():
        for p in params:
            p.requires_grad_(True)
        opt = torch.optim.Adam(params, lr=lr)
        for _ in range(steps):
            loss = torch.nn.functional.cross_entropy(model.semantic_linear(feats)[valid], semantic_labels[valid])
            if fit_offsets and pos.any():
                loss = loss + (model.offset_linear(feats)[pos] - pt_offset_labels[pos]).abs().mean()
            opt.zero_grad()
            loss.backward()
            opt.step()
        for p, w in zip(params, was):
            p.requires_grad_(w)
            p.grad = None
    if not fit_offsets:
        with torch.no_grad():
            model.offset_linear[-1].weight.zero_()
            model.offset_linear[-1].bias.zero_()
    return float(loss.detach())


def fit_model_to_scenes(model, batches, logit_scale=8.0):
    """Stand-in for a trained checkpoint (none is available offline), GPU only.  Over the given
    labelled scenes (batch dicts of ``make_batch`` / ``collate_device``):

      1. every BatchNorm's running statistics := the statistics of its input (cumulative average in
         train mode) -- as in a trained network, so the activations stay standardised and the input
         signal survives the random backbone;
      0. ``passthrough_init``: the input colour survives the random backbone;
      2. ``semantic_linear`` / ``offset_linear`` are fitted to the semantic and offset labels
         (``fit_point_heads``: a few hundred Adam steps on frozen features);
      3. the tiny U-Net's BatchNorms are calibrated on the resulting proposals, ``cls_linear`` is
         fitted to the majority class of each proposal (background when < 50 % of its points carry
         an instance label) and ``iou_score_linear`` predicts 1.

    Returns dict(sem_acc=..., proposals=..., cls_acc=...)."""
    from torch import nn
    from . import ops as _ops
    from .spconv import pytorch as spconv

    def bns(mods):
        return [m for mod in mods for m in mod.modules() if isinstance(m, nn.BatchNorm1d)]

    def calibrate(mods, run):
        layers = bns(mods)
        for m in layers:
            m.reset_running_stats()
            m.momentum = None
            m.train()
        for b in batches:
            run(b)
        for m in layers:
            m.momentum = 0.1
            m.eval()

    def cuda(b):
        return {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}

    def backbone(b):
        feats = torch.cat((b['feats'], b['coords_float']), 1) if model.with_coords else b['feats']
        x = spconv.SparseConvTensor(_ops.voxelization(feats, b['p2v_map']), b['voxel_coords'].int(),
                                    b['spatial_shape'], b['batch_size'])
        return model.forward_backbone(x, b['v2p_map'])

    batches = [cuda(b) for b in batches]
    info = {}
    passthrough_init(model)
    model.invalidate_caches()
    with torch.no_grad():
        model.eval()
        point_mods = [model.input_conv, model.unet, model.output_layer]
        calibrate(point_mods, backbone)
        # hidden layer of the heads: calibrate its BatchNorm too, then solve the last layer
        feats = torch.cat([backbone(b)[2] for b in batches])
        for head in (model.semantic_linear, model.offset_linear):
            for m in bns([head]):
                m.reset_running_stats()
                m.momentum = None
                m.train()
            head(feats)
            for m in bns([head]):
                m.momentum = 0.1
                m.eval()
        sem = torch.cat([b['semantic_labels'] for b in batches]).long()
        ins = torch.cat([b['instance_labels'] for b in batches])
        off = torch.cat([b['pt_offset_labels'] for b in batches]).float()

        valid = sem >= 0
        fit_point_heads(model, feats, sem, ins, off)
        info['sem_acc'] = (model.semantic_linear(feats).argmax(1)[valid] == sem[valid]).float().mean().item()
        model.invalidate_caches()
        if model.semantic_only:
            return info

        # ---- refinement stage
        pooled_all, label_all = [], []

        def refine(b, collect=False):
            s, o, f = backbone(b)
            pidx, poff = model.forward_grouping(s, o, b['batch_idxs'], b['coords_float'])
            if pidx.shape[0] == 0:
                return
            t, inp_map = model.clusters_voxelization(pidx, poff, f, b['coords_float'],
                                                     **model.instance_voxel_cfg)
            y = model.tiny_unet_outputlayer(model.tiny_unet(t))
            if collect:
                pooled_all.append(model.global_pool(y))
                # majority class of each proposal
                prop, pt = pidx[:, 0].long(), pidx[:, 1].long()
                cls = torch.where(b['instance_labels'][pt] >= 0,
                                  b['semantic_labels'][pt] - (model.semantic_classes - model.instance_classes),
                                  torch.full_like(prop, model.instance_classes))
                votes = torch.zeros((poff.numel() - 1, model.instance_classes + 1), device=prop.device)
                votes.index_put_((prop, cls.clamp(min=0)), torch.ones_like(prop, dtype=votes.dtype),
                                 accumulate=True)
                label_all.append(votes.argmax(1))

        calibrate([model.tiny_unet, model.tiny_unet_outputlayer], refine)
        model.invalidate_caches()
        for b in batches:
            refine(b, collect=True)
        info['proposals'] = int(sum(p.shape[0] for p in pooled_all))
        if pooled_all:
            pooled, labels = torch.cat(pooled_all), torch.cat(label_all)
            target = torch.full((pooled.shape[0], model.instance_classes + 1), -logit_scale / 2,
                                device=pooled.device)
            target[torch.arange(pooled.shape[0]), labels] = logit_scale / 2
            _lstsq_head(model.cls_linear, pooled, target, ridge=1e-3)
            info['cls_acc'] = (model.cls_linear(pooled).argmax(1) == labels).float().mean().item()
        model.iou_score_linear.weight.zero_()
        model.iou_score_linear.bias.fill_(1.0)
        model.invalidate_caches()
    return info
