"""TEST INFRASTRUCTURE: full-model parity report, HIP-hosted SoftGroup vs the CPU restatement of the
reference model (oracle/model.py), on one scene.  Used by tests/test_parity_at_size.py and by
bench.py's cpu_baseline leg (which already runs the oracle's forward on the bench scene: its
outputs are compared here instead of being thrown away).

Two views of the same run:
  * stage-wise (each stage fed the GPU's previous-stage output on both sides, as
    tests/test_model_gpu.py does): float stages within 1e-4, integer stages bit-exact;
  * end to end (the oracle's own forward_test from the raw batch): how far the final instances of
    a GPU run drift from those of an oracle run (a 1e-6 difference in a softmax score can move a
    point across score_thr, so this view reports counts and mask IoU instead of asserting).
Follows /root/reference/softgroup/model/softgroup.py:299-361 (forward_test)."""
import time

import numpy as np
import torch

from .model import OracleSoftGroup, SparseT, rle_decode

ATOL = RTOL = 1e-4      # north-star tolerance for float features


def gpu_stages(model, batch):
    """the stages of forward_test on the GPU, every intermediate kept"""
    from softgroup_amd import ops
    import softgroup_amd.spconv.pytorch as spconv
    with torch.no_grad():
        b = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
        feats = torch.cat((b['feats'], b['coords_float']), 1) if model.with_coords else b['feats']
        vf = ops.voxelization(feats, b['p2v_map'])
        x = spconv.SparseConvTensor(vf, b['voxel_coords'].int(), b['spatial_shape'], b['batch_size'])
        sem, off, out_feats = model.forward_backbone(x, b['v2p_map'])
        pidx, poff = model.forward_grouping(sem, off, b['batch_idxs'], b['coords_float'],
                                            model.grouping_cfg)
        inst, inst_map = model.clusters_voxelization(pidx, poff, out_feats, b['coords_float'],
                                                     **model.instance_voxel_cfg)
        _, cls_s, iou_s, mask_s = model.forward_instance(inst, inst_map)
        preds = model.get_instances(batch['scan_ids'][0], pidx, sem, cls_s, iou_s, mask_s)
    return dict(sem=sem, off=off, feats=out_feats, pidx=pidx, poff=poff, inst=inst,
                inst_map=inst_map, cls=cls_s, iou=iou_s, mask=mask_s, preds=preds)


def _close(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.shape != b.shape:
        return False, float('inf')
    if a.size == 0:
        return True, 0.0
    d = np.abs(a - b)
    return bool((d <= ATOL + RTOL * np.abs(b)).all()), float(d.max())


def _instances_equal(got, ref):
    if len(got) != len(ref):
        return False
    return all(a['label_id'] == b['label_id'] and a['pred_mask'] == b['pred_mask'] and
               abs(float(a['conf']) - float(b['conf'])) < 1e-6 for a, b in zip(got, ref))


def _mean_best_iou(got, ref):
    """for every oracle instance the best IoU with a GPU instance of the same class"""
    if not ref:
        return 1.0 if not got else 0.0
    gm = {}
    for g in got:
        gm.setdefault(int(g['label_id']), []).append(rle_decode(g['pred_mask']).astype(bool))
    tot = 0.0
    for r in ref:
        m = rle_decode(r['pred_mask']).astype(bool)
        best = 0.0
        for c in gm.get(int(r['label_id']), []):
            inter = np.count_nonzero(m & c)
            if inter:
                best = max(best, inter / np.count_nonzero(m | c))
        tot += best
    return tot / len(ref)


def parity_report(model, batch, cfg, end_to_end=True, timed_path=False):
    """-> dict of parity figures for one scene (all python scalars / bools, JSON-ready).
    ``timed_path``: additionally run ``model(batch)`` -- the call bench.py times: native scan driver +
    U-Net executor -- and compare ITS results (not only the operator path's) with the oracle."""
    n = lambda t: t.detach().cpu().numpy()  # noqa: E731
    g = gpu_stages(model, batch)
    native = None
    if timed_path:
        with torch.no_grad():
            b = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
            native = dict(model(b))
    ora = OracleSoftGroup(model.state_dict(), cfg)
    rep = {'points': int(batch['coords_float'].shape[0]), 'tolerance': ATOL}

    # (1) backbone + point-wise heads.  The end-to-end oracle forward is the expensive CPU leg
    #     (and bench.py's cpu_baseline): run it once, reuse its stage-1 outputs.
    t0 = time.perf_counter()
    if end_to_end:
        e2e = ora.forward_test(batch)
        sem, off, feats = e2e['semantic_scores'], e2e['pt_offsets'], e2e['output_feats']
    else:
        e2e = None
        sem, off, feats = ora.point_wise(batch)
    rep['oracle_forward_s'] = round(time.perf_counter() - t0, 2)
    ok_f, rep['max_abs_feat'] = _close(n(g['feats']), feats)
    ok_s, rep['max_abs_semantic_scores'] = _close(n(g['sem']), sem)
    ok_o, rep['max_abs_offsets'] = _close(n(g['off']), off)

    # (2) grouping from the GPU's scores/offsets: bit-exact proposals (membership and order)
    pidx, poff = ora.grouping(n(g['sem']), n(g['off']), batch['batch_idxs'], batch['coords_float'])
    rep['proposals'] = int(max(len(poff) - 1, 0))
    rep['proposal_points'] = int(pidx.shape[0])
    rep['proposals_equal'] = bool(np.array_equal(n(g['pidx']), pidx) and np.array_equal(n(g['poff']), poff))

    # (3) proposal voxelisation, (4) tiny U-Net + heads, (5) instances + RLE, each from the GPU's
    #     previous stage
    ok_v = ok_h = True
    rep['instances_equal'] = True
    if pidx.shape[0]:
        inst, inst_map = ora.clusters_voxelization(pidx, poff, n(g['feats']), batch['coords_float'])
        idx_eq = bool(np.array_equal(n(g['inst'].indices), inst.indices) and
                      np.array_equal(n(g['inst_map']), inst_map))
        ok_v, rep['max_abs_proposal_voxel_feat'] = _close(n(g['inst'].features), inst.features)
        rep['proposal_voxel_index_equal'] = idx_eq
        ok_v = ok_v and idx_eq
        inst_same = SparseT(n(g['inst'].features), inst.indices, inst.spatial_shape, inst.batch_size)
        cls_s, iou_s, mask_s = ora.instance_heads(inst_same, inst_map)
        oks = [_close(n(g[k]), r) for k, r in (('cls', cls_s), ('iou', iou_s), ('mask', mask_s))]
        ok_h = all(o for o, _ in oks)
        rep['max_abs_instance_heads'] = max(d for _, d in oks)
        ref = ora.get_instances(batch['scan_ids'][0], pidx, n(g['sem']), n(g['cls']), n(g['iou']),
                                n(g['mask']))
        rep['instances'] = len(ref)
        rep['instances_equal'] = _instances_equal(g['preds'], ref)
        if native is not None:
            rep['timed_path_instances_equal_oracle'] = _instances_equal(native['pred_instances'], ref)
    rep['float_stages_within_tol'] = bool(ok_f and ok_s and ok_o and ok_v and ok_h)

    # end-to-end drift of a pure GPU run against a pure oracle run
    if e2e is not None:
        rep['e2e_proposals_equal'] = bool(np.array_equal(n(g['pidx']), e2e['proposals_idx']) and
                                          np.array_equal(n(g['poff']), e2e['proposals_offset']))
        rep['e2e_instances_gpu'] = len(g['preds'])
        rep['e2e_instances_oracle'] = len(e2e['pred_instances'])
        rep['e2e_mean_best_mask_iou'] = round(_mean_best_iou(g['preds'], e2e['pred_instances']), 6)
    rep['ok'] = bool(rep['float_stages_within_tol'] and rep['proposals_equal'] and rep['instances_equal'])
    if native is not None:
        # the timed call against (a) the operator path it is claimed equal to, (b) the oracle
        rep['checked_call'] = 'model(batch): sg_scan_forward, one C call per scan (the timed region\'s call)'
        # labels and RLE strings identical; confidences within 1e-6 (the timed call's class / IoU heads are
        # fp32 FMA chains, sg_linear_rows, the operator path's a GEMM library: rounding order only)
        rep['timed_path_instances_equal_operator_path'] = _instances_equal(native['pred_instances'], g['preds'])
        if len(native['pred_instances']) == len(g['preds']) and len(g['preds']):
            rep['timed_path_max_abs_conf_vs_operator_path'] = float(max(
                abs(float(a['conf']) - float(c['conf'])) for a, c in zip(native['pred_instances'], g['preds'])))
        rep['timed_path_semantic_preds_equal_operator_path'] = bool(
            np.array_equal(native['semantic_preds'], n(g['sem']).argmax(1)))
        ok_no, rep['timed_path_max_abs_offsets_vs_oracle'] = _close(native['offset_preds'], off)
        rep['timed_path_semantic_preds_vs_oracle_mismatches'] = int(
            np.count_nonzero(native['semantic_preds'] != np.asarray(sem).argmax(1)))
        if e2e is not None:
            rep['timed_path_e2e_instances_equal_oracle'] = _instances_equal(native['pred_instances'],
                                                                             e2e['pred_instances'])
        rep['ok'] = bool(rep['ok'] and ok_no and rep['timed_path_instances_equal_operator_path'] and
                         rep.get('timed_path_instances_equal_oracle', True))
    return rep
