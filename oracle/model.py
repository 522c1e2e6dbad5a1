"""CPU restatement of the reference model's inference forward -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference/softgroup/model/softgroup.py (forward_test :299-361, forward_backbone
:363-378, forward_grouping :411-480, clusters_voxelization :655-709, forward_instance :509-522,
global_pool :718-731, get_instances :537-604) and blocks.py (:9-143) stage by stage, with the
reference's control flow kept as is (per-class Python loop, CPU BFS, dense int masks + numpy RLE).
Sparse ops come from the C oracle (oracle/sg_oracle*.c); dense fp32 math (Linear, BatchNorm,
softmax) is plain PyTorch on the CPU.  Weights arrive as a ``state_dict`` of the reference's key
names, so the same dict drives this and the HIP-hosted model.

Used by tests (stage-wise parity), by __graft_entry__.smoke() and as bench.py's cpu_baseline.
"""
import numpy as np
import torch
import torch.nn.functional as F

import oracle as O

EPS = 1e-4  # norm_fn = BatchNorm1d(eps=1e-4), softgroup.py:54


class SparseT:
    """minimal stand-in for spconv.SparseConvTensor on the CPU"""

    def __init__(self, features, indices, spatial_shape, batch_size, rules=None):
        self.features = np.ascontiguousarray(features, dtype=np.float32)
        self.indices = np.ascontiguousarray(indices, dtype=np.int32)
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.rules = {} if rules is None else rules

    def replace(self, features):
        return SparseT(features, self.indices, self.spatial_shape, self.batch_size, self.rules)


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


class OracleSoftGroup:

    def __init__(self, state_dict, cfg):
        self.sd = {k: _np(v).astype(np.float32) if _np(v).dtype.kind == 'f' else _np(v)
                   for k, v in state_dict.items()}
        self.cfg = cfg
        self.channels = cfg['channels']
        self.num_blocks = cfg['num_blocks']

    # ---------------------------------------------------------------- dense pieces
    def bn(self, x, name, relu=True):
        sd = self.sd
        y = F.batch_norm(torch.from_numpy(x), torch.from_numpy(sd[name + '.running_mean']),
                         torch.from_numpy(sd[name + '.running_var']),
                         torch.from_numpy(sd[name + '.weight']), torch.from_numpy(sd[name + '.bias']),
                         False, 0.1, EPS)
        return (F.relu(y) if relu else y).numpy()

    def linear(self, x, name):
        return F.linear(torch.from_numpy(x), torch.from_numpy(self.sd[name + '.weight']),
                        torch.from_numpy(self.sd[name + '.bias'])).numpy()

    def mlp(self, x, name, norm=True):
        """blocks.py:9-19: Linear -> [BN] -> ReLU -> Linear"""
        h = self.linear(x, name + '.0')
        if norm:
            h = self.bn(h, name + '.1', relu=True)
            return self.linear(h, name + '.3')
        return self.linear(np.maximum(h, 0), name + '.2')

    # ---------------------------------------------------------------- sparse pieces
    def subm(self, x, wname, key):
        if key not in x.rules:
            x.rules[key] = O.subm_rulebook(x.indices, x.spatial_shape)
        return x.replace(O.subm_conv3d(x.features, x.rules[key], self.sd[wname]))

    def residual_block(self, x, prefix, key):
        """blocks.py:44-79"""
        sd = self.sd
        w1x1 = prefix + '.i_branch.0.weight'
        if w1x1 in sd:
            w = sd[w1x1]
            identity = x.features @ w.reshape(w.shape[0], w.shape[-1]).T
        else:
            identity = x.features
        h = x.replace(self.bn(x.features, prefix + '.conv_branch.0'))
        h = self.subm(h, prefix + '.conv_branch.2.weight', key)
        h = h.replace(self.bn(h.features, prefix + '.conv_branch.3'))
        h = self.subm(h, prefix + '.conv_branch.5.weight', key)
        return h.replace(h.features + identity)

    def ublock(self, x, prefix, n_planes, key_id):
        """blocks.py:131-143"""
        for i in range(2):
            x = self.residual_block(x, f'{prefix}.blocks.block{i}', f'subm{key_id}')
        if len(n_planes) == 1:
            return x
        identity = x.features
        h = self.bn(x.features, prefix + '.conv.0')
        out_idx, in2out, child, oshape = O.down_rulebook(x.indices, x.spatial_shape)
        d = SparseT(O.sparse_conv3d_k2s2(h, child, self.sd[prefix + '.conv.2.weight']), out_idx,
                    oshape, x.batch_size, x.rules)
        d = self.ublock(d, prefix + '.u', n_planes[1:], key_id + 1)
        h = self.bn(d.features, prefix + '.deconv.0')
        up = O.inverse_conv3d_k2(h, x.indices, in2out, self.sd[prefix + '.deconv.2.weight'])
        x = x.replace(np.concatenate([identity, up], 1))
        for i in range(2):
            x = self.residual_block(x, f'{prefix}.blocks_tail.block{i}', f'subm{key_id}')
        return x

    def backbone(self, voxel_feats, voxel_coords, spatial_shape, batch_size):
        x = SparseT(voxel_feats, voxel_coords, spatial_shape, batch_size)
        x = self.subm(x, 'input_conv.0.weight', 'subm1')
        x = self.ublock(x, 'unet', [self.channels * (i + 1) for i in range(self.num_blocks)], 1)
        return self.bn(x.features, 'output_layer.0')

    # ---------------------------------------------------------------- forward_test stages
    def _unet(self, voxel_feats, voxel_coords, spatial_shape, batch_size):
        return self.backbone(voxel_feats, voxel_coords, spatial_shape, batch_size)

    def point_wise(self, batch, x4_split=False, lvl_fusion=False):
        """softgroup.py:303-307,363-409 -> semantic_scores, pt_offsets, output_feats (per point, or
        per voxel with lvl_fusion)"""
        feats = _np(batch['feats']).astype(np.float32)
        if self.cfg.get('with_coords', True):
            feats = np.concatenate([feats, _np(batch['coords_float'])], 1).astype(np.float32)
        voxel_feats = O.voxelization(feats, _np(batch['p2v_map']))
        coords = _np(batch['voxel_coords']).astype(np.int32)
        v2p = _np(batch['v2p_map']).astype(np.int64)
        if x4_split:                                   # forward_4_parts + merge_4_parts
            outs = []
            for i in range(4):
                inds = coords[:, 0] == i
                c = coords[inds].copy()
                c[:, 0] = 0
                outs.append(self._unet(voxel_feats[inds], c, batch['spatial_shape'], 1))
            output_feats = merge_4_parts(np.concatenate(outs, 0)[v2p])
        else:
            vfeat = self._unet(voxel_feats, coords, batch['spatial_shape'], batch['batch_size'])
            output_feats = vfeat if lvl_fusion else vfeat[v2p]
        return (self.mlp(output_feats, 'semantic_linear'), self.mlp(output_feats, 'offset_linear'),
                output_feats)

    def get_level(self, num_points):
        """softgroup.py:482-489"""
        if num_points > 1000000:
            return 3
        return 2 if num_points > 100000 else 1

    def pyramid_map(self, coords_float, pt_offsets, batch_idxs, level, base_size):
        """softgroup.py:491-498"""
        coords = (torch.from_numpy(coords_float) / (base_size * level)).long()
        coords = torch.cat([torch.from_numpy(batch_idxs.astype(np.int64))[:, None], coords], 1).numpy()
        vcoords, l2p_map, p2l_map = O.voxelization_idx(coords, int(batch_idxs[-1]) + 1)
        return (O.voxelization(coords_float, p2l_map), O.voxelization(pt_offsets, p2l_map),
                vcoords[:, 0].astype(np.int32), l2p_map)

    def pyramid_inverse_map(self, proposals_idx, proposals_offset, num_points, l2p_map):
        """softgroup.py:500-507 (dense [nProposal, n] int matrix)"""
        proposals = np.zeros((proposals_offset.shape[0] - 1, num_points), np.int32)
        proposals[proposals_idx[:, 0].astype(np.int64), proposals_idx[:, 1].astype(np.int64)] = 1
        proposals = proposals[:, l2p_map.astype(np.int64)]
        pidx = np.stack(np.nonzero(proposals), 1).astype(np.int32)
        poff = np.concatenate([[0], np.cumsum(proposals.sum(1))]).astype(np.int32)
        return pidx, poff

    def grouping(self, semantic_scores, pt_offsets, batch_idxs, coords_float, lvl_fusion=False):
        """softgroup.py:411-480: per-class loop (optional pyramid / octree), merged proposals"""
        g = self.cfg['grouping_cfg']
        tcfg = self.cfg['test_cfg']
        batch_idxs = _np(batch_idxs).astype(np.int32)
        coords_float = _np(coords_float).astype(np.float32)
        batch_size = int(batch_idxs.max()) + 1
        scores = F.softmax(torch.from_numpy(semantic_scores), dim=-1).numpy()
        class_mean = np.asarray(g['class_numpoint_mean'], np.float32)
        with_pyramid = g.get('with_pyramid', False)
        with_octree = g.get('with_octree', False)
        base_size = g.get('pyramid_base_size', 0.02)
        idx_list, off_list = [], []
        for class_id in range(self.cfg['semantic_classes']):
            if class_id in g['ignore_classes']:
                continue
            object_idxs = np.nonzero(scores[:, class_id] > np.float32(g['score_thr']))[0]
            if object_idxs.shape[0] < tcfg['min_npoint']:
                continue
            b_ = batch_idxs[object_idxs]
            c_ = coords_float[object_idxs]
            o_ = pt_offsets[object_idxs]
            radius, level, l2p_map = g['radius'], 1, None
            if with_pyramid:
                level = self.get_level(c_.shape[0])
                radius = g['radius'] * level
                if level > 1 or not lvl_fusion:
                    c_, o_, b_, l2p_map = self.pyramid_map(c_, o_, b_, level, base_size)
            if with_octree:
                nbr, start_len = O.octree_ball_query(c_ + o_, g['mean_active'], radius)
            else:
                offs = np.zeros(batch_size + 1, np.int32)
                for i in range(batch_size):
                    offs[i + 1] = offs[i] + (b_ == i).sum()
                nbr, start_len = O.ballquery_batch_p(c_ + o_, b_, offs, radius, g['mean_active'])
            pidx, poff = O.bfs_cluster(class_mean, nbr, start_len, g['npoint_thr'], class_id)
            if l2p_map is not None:
                pidx, poff = self.pyramid_inverse_map(pidx, poff, c_.shape[0], l2p_map)
            pidx[:, 1] = object_idxs[pidx[:, 1].astype(np.int64)].astype(np.int32)
            if len(off_list) > 0:
                pidx[:, 0] += sum(x.shape[0] for x in off_list) - 1
                poff = poff + off_list[-1][-1]
                poff = poff[1:]
            if pidx.shape[0] > 0:
                idx_list.append(pidx)
                off_list.append(poff)
        if idx_list:
            return np.concatenate(idx_list, 0), np.concatenate(off_list).astype(np.int32)
        return np.zeros((0, 2), np.int32), np.zeros((0, ), np.int32)

    def clusters_voxelization(self, clusters_idx, clusters_offset, feats, coords):
        """softgroup.py:655-709 (rand_quantize=False)"""
        scale = self.cfg['instance_voxel_cfg']['scale']
        ss = self.cfg['instance_voxel_cfg']['spatial_shape']
        batch_idx = clusters_idx[:, 0].astype(np.int64)
        c_idxs = clusters_idx[:, 1].astype(np.int64)
        feats = feats[c_idxs]
        coords = torch.from_numpy(_np(coords).astype(np.float32)[c_idxs])
        cmin = torch.from_numpy(O.sec_min(coords.numpy(), clusters_offset))
        cmax = torch.from_numpy(O.sec_max(coords.numpy(), clusters_offset))
        cscale = 1 / ((cmax - cmin) / ss).max(1)[0] - 0.01
        cscale = torch.clamp(cscale, min=None, max=scale)
        cmin = cmin * cscale[:, None]
        cscale_pt = cscale[torch.from_numpy(batch_idx)]
        coords = coords * cscale_pt[:, None]
        coords -= cmin[torch.from_numpy(batch_idx)]
        assert coords.numel() == int(((coords >= 0) * (coords < ss)).sum())
        vox = torch.cat([torch.from_numpy(batch_idx).view(-1, 1), coords.long()], 1).numpy()
        n_prop = int(clusters_idx[-1, 0]) + 1
        out_coords, inp_map, out_map = O.voxelization_idx(vox, n_prop)
        out_feats = O.voxelization(feats, out_map)
        return SparseT(out_feats, out_coords.astype(np.int32), [ss] * 3, n_prop), inp_map

    def instance_heads(self, inst, inst_map):
        """softgroup.py:509-522, 718-731"""
        x = self.ublock(inst, 'tiny_unet', [self.channels, 2 * self.channels], 11)
        feats = self.bn(x.features, 'tiny_unet_outputlayer.0')
        mask_scores = self.mlp(feats, 'mask_linear', norm=False)[inst_map.astype(np.int64)]
        counts = np.bincount(x.indices[:, 0])
        offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        pooled = O.global_avg_pool(feats, offsets)
        return self.linear(pooled, 'cls_linear'), self.linear(pooled, 'iou_score_linear'), mask_scores

    def get_instances(self, scan_id, proposals_idx, semantic_scores, cls_scores, iou_scores,
                      mask_scores, v2p_map=None, lvl_fusion=False):
        """softgroup.py:537-604 with dense [nProposal, N] masks and the reference's numpy RLE"""
        if proposals_idx.shape[0] == 0:
            return []
        tcfg = self.cfg['test_cfg']
        n_inst, n_pts = cls_scores.shape[0], semantic_scores.shape[0]
        cls_prob = F.softmax(torch.from_numpy(cls_scores), 1).numpy()
        sem_pred = semantic_scores.argmax(1)
        if v2p_map is not None:
            v2p_map = _np(v2p_map).astype(np.int64)
        cls_l, score_l, mask_l = [], [], []
        for i in range(self.cfg['instance_classes']):
            if i in self.cfg.get('sem2ins_classes', []):
                cls_pred = np.array([i + 1], np.int64)
                score = np.array([1.], np.float32)
                mask = (sem_pred == i)[None, :].astype(np.int32)
                if lvl_fusion:
                    mask = mask[:, v2p_map]
            else:
                score = cls_prob[:, i] * np.clip(iou_scores[:, i], 0, 1)
                mask = np.zeros((n_inst, n_pts), np.int32)
                on = mask_scores[:, i] > tcfg['mask_score_thr']
                cur = proposals_idx[on].astype(np.int64)
                mask[cur[:, 0], cur[:, 1]] = 1
                inds = cls_prob[:, i] > tcfg['cls_score_thr']
                cls_pred = np.full(n_inst, i + 1, np.int64)[inds]
                score, mask = score[inds], mask[inds]
                if lvl_fusion:
                    mask = mask[:, v2p_map]
                inds = mask.sum(1) >= tcfg['min_npoint']
                cls_pred, score, mask = cls_pred[inds], score[inds], mask[inds]
            cls_l.append(cls_pred)
            score_l.append(score)
            mask_l.append(mask)
        cls_pred, score_pred, mask_pred = np.concatenate(cls_l), np.concatenate(score_l), np.concatenate(mask_l)
        return [dict(scan_id=scan_id, label_id=cls_pred[i], conf=score_pred[i],
                     pred_mask=rle_encode(mask_pred[i])) for i in range(cls_pred.shape[0])]

    def panoptic_fusion(self, semantic_preds, instance_preds):
        """softgroup.py:606-639"""
        sc, ic = self.cfg['semantic_classes'], self.cfg['instance_classes']
        cls_offset = sc - ic - 1
        panoptic_cls = semantic_preds.copy().astype(np.uint32)
        panoptic_ids = np.zeros_like(semantic_preds).astype(np.uint32)
        scores = [x['conf'] for x in instance_preds]
        score_inds = np.argsort(scores)[::-1]
        prev_paste = np.zeros_like(semantic_preds, dtype=bool)
        panoptic_id = 1
        for i in score_inds:
            instance = instance_preds[i]
            mask = rle_decode(instance['pred_mask']).astype(bool)
            intersect = (mask * prev_paste).sum()
            if intersect / (mask.sum() + 1e-5) > self.cfg['test_cfg']['panoptic_skip_iou']:
                continue
            paste = mask * (~prev_paste)
            panoptic_cls[paste] = instance['label_id'] + cls_offset
            panoptic_ids[paste] = panoptic_id
            prev_paste[paste] = 1
            panoptic_id += 1
        ignore_inds = (panoptic_cls >= 11) & (panoptic_ids == 0)
        panoptic_preds = (panoptic_cls & 0xFFFF) | (panoptic_ids << 16)
        panoptic_preds[ignore_inds] = sc
        return panoptic_preds.astype(np.uint32)

    def forward_test(self, batch):
        """softgroup.py:299-361 (without the label bookkeeping)"""
        tcfg = self.cfg['test_cfg']
        lvl_fusion = tcfg.get('lvl_fusion', False)
        x4 = tcfg.get('x4_split', False)
        sem, off, feats = self.point_wise(batch, x4, lvl_fusion)
        coords_float = _np(batch['coords_float']).astype(np.float32)
        batch_idxs = _np(batch['batch_idxs'])
        if x4:
            coords_float = merge_4_parts(coords_float)
        if lvl_fusion:
            batch_idxs = _np(batch['voxel_coords'])[:, 0].astype(np.int32)
            coords_float = O.voxelization(coords_float, _np(batch['p2v_map']))
        pidx, poff = self.grouping(sem, off, batch_idxs, coords_float, lvl_fusion)
        ret = dict(semantic_scores=sem, pt_offsets=off, output_feats=feats, proposals_idx=pidx,
                   proposals_offset=poff, pred_instances=[])
        if pidx.shape[0] == 0:
            return ret
        inst, inst_map = self.clusters_voxelization(pidx, poff, feats, coords_float)
        cls_scores, iou_scores, mask_scores = self.instance_heads(inst, inst_map)
        preds = self.get_instances(batch['scan_ids'][0], pidx, sem, cls_scores, iou_scores, mask_scores,
                                   batch['v2p_map'], lvl_fusion)
        ret.update(cls_scores=cls_scores, iou_scores=iou_scores, mask_scores=mask_scores,
                   pred_instances=preds)
        if 'panoptic' in tcfg['eval_tasks']:
            ret['panoptic_preds'] = self.panoptic_fusion(sem.argmax(1), preds)
        return ret


def merge_4_parts(x):
    """softgroup.py:397-409"""
    inds = np.arange(x.shape[0])
    ps = [inds[i::4] for i in range(4)]
    out = np.zeros_like(x)
    start = 0
    for p in ps:
        out[p] = x[start:start + len(p)]
        start += len(p)
    return out


def rle_decode(rle):
    """util/rle.py:22-41"""
    s = rle['counts'].split()
    starts = np.asarray(s[0::2], dtype=np.int32) - 1
    nums = np.asarray(s[1::2], dtype=np.int32)
    mask = np.zeros(rle['length'], dtype=np.uint8)
    for lo, hi in zip(starts, starts + nums):
        mask[lo:hi] = 1
    return mask


def rle_encode(mask):
    """util/rle.py:5-19"""
    length = mask.shape[0]
    m = np.concatenate([[0], mask, [0]])
    runs = np.where(m[1:] != m[:-1])[0] + 1
    runs[1::2] -= runs[::2]
    return dict(length=length, counts=' '.join(str(x) for x in runs))
