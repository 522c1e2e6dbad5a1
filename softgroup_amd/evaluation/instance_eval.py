"""Instance-segmentation evaluation (AP / AP50 / AP25 / recall), the step right AFTER the hot path
(SURVEY 8f-3).  Same interface and results as the reference's ``ScanNetEval``
(softgroup/evaluation/instance_eval.py: ``evaluate(pred_list, gt_list)`` -> dict of averages,
``print_results``, ``write_result_file``), which follows the ScanNet benchmark's
evaluate_semantic_instance.py.

What differs is how a scan's predictions are associated with its ground truth.  The reference
decodes every RLE mask to a dense 0/1 array and, for every (prediction, GT instance) pair of equal
class, counts ``logical_and`` over all N points, in a multiprocessing pool (:228-309, 375-385).
Here the masks stay runs: all runs of a scan go to the GPU once and ``sg_eval_intersections``
visits every mask point exactly once, producing the whole prediction x GT count matrix (plus the
"void" column) -- O(total mask points) instead of O(nPred * nGT * N).  Without a GPU the same
matrix comes from ``numpy.add.at`` (the evaluator is also used in CPU-only tooling).

The matching / precision-recall logic is kept operation for operation (greedy assignment in
prediction order, the duplicate-match rule, ignore proportions from void and small instances,
unique-threshold PR curve with the [-0.5, 0, 0.5] step kernel), so the averages equal the
reference's to the last bit; tests/golden/eval_golden.json pins that against the reference's own
evaluator.
"""
import numpy as np


def _runs_of(pred_mask, n_points):
    """0-based (starts, lens) of a mask given as RLE dict or array"""
    if isinstance(pred_mask, dict):
        assert int(pred_mask['length']) == n_points
        flat = np.array(pred_mask['counts'].split(), dtype=np.int64) if pred_mask['counts'] else \
            np.zeros(0, np.int64)
        return flat[0::2] - 1, flat[1::2]
    m = np.not_equal(np.asarray(pred_mask), 0)
    assert m.shape[0] == n_points
    edges = np.flatnonzero(np.diff(np.concatenate([[0], m.astype(np.int8), [0]])))
    return edges[0::2], edges[1::2] - edges[0::2]


class ScanNetEval(object):

    def __init__(self, class_labels, min_npoint=None, iou_type=None, use_label=True, device=None):
        self.valid_class_labels = class_labels
        self.valid_class_ids = np.arange(len(class_labels)) + 1
        self.id2label = {int(i): n for i, n in zip(self.valid_class_ids, class_labels)}
        self.label2id = {n: int(i) for i, n in zip(self.valid_class_ids, class_labels)}
        self.ious = np.append(np.arange(0.5, 0.95, 0.05), 0.25)
        self.min_region_sizes = np.array([min_npoint if min_npoint else 100])
        self.distance_threshes = np.array([float('inf')])
        self.distance_confs = np.array([-float('inf')])
        self.iou_type = iou_type
        self.use_label = use_label
        self.eval_class_labels = self.valid_class_labels if use_label else ['class_agnostic']
        self.device = device      # None: GPU when available

    # ------------------------------------------------------------------ association (per scan)
    def _count_matrix(self, starts, lens, run_pred, n_pred, gt_slot, n_slots):
        """counts[p, s] = points of prediction p in GT slot s"""
        import torch
        use_gpu = torch.cuda.is_available() if self.device is None else str(self.device).startswith('cuda')
        if not use_gpu or len(starts) == 0:
            counts = np.zeros((n_pred, n_slots), np.int64)
            for s, n, p in zip(starts.tolist(), lens.tolist(), run_pred.tolist()):
                np.add.at(counts[p], gt_slot[s:s + n], 1)
            return counts
        from .. import _lib as L
        dev = torch.device('cuda' if self.device is None else self.device)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        d_start = torch.from_numpy(starts.astype(np.int32)).to(dev)
        d_off = torch.from_numpy(off).to(dev)
        d_pred = torch.from_numpy(run_pred.astype(np.int32)).to(dev)
        d_slot = torch.from_numpy(gt_slot.astype(np.int32)).to(dev)
        counts = torch.empty((n_pred, n_slots), dtype=torch.int32, device=dev)
        L.check(L.lib().sg_eval_intersections(L.ptr(d_start), L.ptr(d_off), L.ptr(d_pred), len(starts),
                                              int(off[-1]), L.ptr(d_slot), n_pred, n_slots, L.ptr(counts),
                                              L.stream()), 'sg_eval_intersections')
        return counts.cpu().numpy().astype(np.int64)

    def assign_instances_for_scan(self, preds, gts):
        """-> (gt2pred, pred2gt) with the reference's structure: per evaluated label a list of GT
        instance dicts (each with 'matched_pred') and a list of prediction dicts (each with
        'matched_gt'), only the fields the matching reads."""
        gts = np.asarray(gts)
        n_points = gts.shape[0]
        ids, inverse, vert = np.unique(gts, return_inverse=True, return_counts=True)
        lab = ids // 1000
        is_inst = (ids != 0) & np.isin(lab, self.valid_class_ids)
        inst_rows = np.flatnonzero(is_inst)                    # ascending instance id (np.unique order)
        n_gt = len(inst_rows)
        slot_of_id = np.full(len(ids), n_gt, np.int64)         # slot n_gt: not an evaluated instance
        slot_of_id[inst_rows] = np.arange(n_gt)
        gt_slot = slot_of_id[inverse.reshape(-1)]
        # void = points whose CLASS is not evaluated (instance_eval.py:258); points of valid classes
        # with instance id 0 cannot exist (id 0 has class 0)
        gt_label = lab[inst_rows]
        gt_vert = vert[inst_rows]
        gt_ids = ids[inst_rows]

        # predictions that take part (valid label, >= min region size), in list order
        runs_s, runs_l, runs_p, kept = [], [], [], []
        for pred in preds:
            if self.use_label:
                if pred['label_id'] not in self.id2label:
                    continue
            s, n = _runs_of(pred['pred_mask'], n_points)
            num = int(n.sum())
            if num < self.min_region_sizes[0]:
                continue
            runs_s.append(s)
            runs_l.append(n)
            runs_p.append(np.full(len(s), len(kept), np.int64))
            kept.append((pred, num))
        n_pred = len(kept)
        cat = (lambda xs: np.concatenate(xs) if xs else np.zeros(0, np.int64))  # noqa: E731
        counts = self._count_matrix(cat(runs_s), cat(runs_l), cat(runs_p), n_pred, gt_slot, n_gt + 1)

        gt2pred = {label: [] for label in self.eval_class_labels}
        gt_entry = []                                         # per GT slot: its dict
        for g in range(n_gt):
            label = self.id2label[int(gt_label[g])] if self.use_label else self.eval_class_labels[0]
            d = dict(instance_id=int(gt_ids[g]), label_id=int(gt_label[g]), vert_count=int(gt_vert[g]),
                     med_dist=-1, dist_conf=0.0, matched_pred=[])
            gt_entry.append(d)
        if self.use_label:
            for g in range(n_gt):
                gt2pred[self.id2label[int(gt_label[g])]].append(gt_entry[g])
        else:
            # class agnostic: the reference concatenates the per-label lists in label order
            for label in self.valid_class_labels:
                for g in range(n_gt):
                    if self.id2label[int(gt_label[g])] == label:
                        gt2pred[self.eval_class_labels[0]].append(gt_entry[g])
        pred2gt = {label: [] for label in self.eval_class_labels}
        for k, (pred, num) in enumerate(kept):
            label_id = pred['label_id'] if self.use_label else None
            label = self.id2label[label_id] if self.use_label else self.eval_class_labels[0]
            pi = dict(filename='{}_{}'.format(pred['scan_id'], k), pred_id=k, label_id=label_id,
                      vert_count=num, confidence=pred['conf'], void_intersection=int(counts[k, n_gt]))
            matched_gt = []
            for gt in gt2pred[label]:                          # list order, as the reference loops
                g = int(np.searchsorted(gt_ids, gt['instance_id']))
                inter = int(counts[k, g])
                if inter > 0:
                    iou = float(inter) / (gt['vert_count'] + num - inter)
                    gc = {kk: v for kk, v in gt.items() if kk != 'matched_pred'}
                    gc.update(intersection=inter, iou=iou)
                    pc = dict(pi, intersection=inter, iou=iou)
                    matched_gt.append(gc)
                    gt['matched_pred'].append(pc)
            pi['matched_gt'] = matched_gt
            pred2gt[label].append(pi)
        return gt2pred, pred2gt

    # ------------------------------------------------------------------ AP over all scans
    def evaluate_matches(self, matches):
        ious = self.ious
        min_region_size = self.min_region_sizes[0]
        n_lab = len(self.eval_class_labels)
        ap = np.zeros((1, n_lab, len(ious)), float)
        rc = np.zeros((1, n_lab, len(ious)), float)
        for oi, iou_th in enumerate(ious):
            visited = set()                                    # predictions already assigned to a GT
            for li, label_name in enumerate(self.eval_class_labels):
                y_true, y_score = np.empty(0), np.empty(0)
                hard_fn = 0
                has_gt = has_pred = False
                for m in matches:
                    preds = matches[m]['pred'][label_name]
                    gts = [g for g in matches[m]['gt'][label_name]
                           if g['instance_id'] >= 1000 and g['vert_count'] >= min_region_size]
                    has_gt = has_gt or bool(gts)
                    has_pred = has_pred or bool(preds)
                    cur_true = np.ones(len(gts))
                    cur_score = np.ones(len(gts)) * (-float('inf'))
                    cur_match = np.zeros(len(gts), dtype=bool)
                    for gi, gt in enumerate(gts):
                        found = False
                        for pred in gt['matched_pred']:
                            if pred['filename'] in visited:
                                continue
                            if pred['iou'] > iou_th:
                                conf = pred['confidence']
                                if cur_match[gi]:
                                    # a second prediction on a matched GT: the lower score is a false
                                    # positive (and this prediction stays unassigned, as in the reference)
                                    hi, lo = max(cur_score[gi], conf), min(cur_score[gi], conf)
                                    cur_score[gi] = hi
                                    cur_true = np.append(cur_true, 0)
                                    cur_score = np.append(cur_score, lo)
                                    cur_match = np.append(cur_match, True)
                                else:
                                    found = True
                                    cur_match[gi] = True
                                    cur_score[gi] = conf
                                    visited.add(pred['filename'])
                        if not found:
                            hard_fn += 1
                    cur_true = cur_true[cur_match]
                    cur_score = cur_score[cur_match]
                    for pred in preds:                         # unmatched predictions: false positives
                        if any(gt['iou'] > iou_th for gt in pred['matched_gt']):
                            continue
                        ignore = pred['void_intersection']
                        for gt in pred['matched_gt']:
                            if gt['instance_id'] < 1000:
                                ignore += gt['intersection']
                            if gt['vert_count'] < min_region_size:
                                ignore += gt['intersection']
                        if float(ignore) / pred['vert_count'] <= iou_th:
                            cur_true = np.append(cur_true, 0)
                            cur_score = np.append(cur_score, pred['confidence'])
                    y_true = np.append(y_true, cur_true)
                    y_score = np.append(y_score, cur_score)
                if has_gt and has_pred:
                    order = np.argsort(y_score)
                    s_sorted, t_sorted = y_score[order], y_true[order]
                    cum = np.cumsum(t_sorted)
                    _, first = np.unique(s_sorted, return_index=True)
                    n_pr = len(first) + 1
                    n_ex = len(s_sorted)
                    n_true = cum[-1]
                    precision, recall = np.zeros(n_pr), np.zeros(n_pr)
                    cum = np.append(cum, 0)                    # cum[-1] = 0 for the first threshold
                    for i, idx in enumerate(first):
                        below = cum[idx - 1]
                        tp = n_true - below
                        fp = n_ex - idx - tp
                        fn = below + hard_fn
                        precision[i] = float(tp) / (tp + fp)
                        recall[i] = float(tp) / (tp + fn)
                    rc_cur = recall[0]
                    precision[-1], recall[-1] = 1., 0.
                    r = np.append(recall[0], recall)
                    r = np.append(r, 0.)
                    ap_cur = np.dot(precision, np.convolve(r, [-0.5, 0, 0.5], 'valid'))
                elif has_gt:
                    ap_cur = rc_cur = 0.0
                else:
                    ap_cur = rc_cur = float('nan')
                ap[0, li, oi] = ap_cur
                rc[0, li, oi] = rc_cur
        return ap, rc

    def compute_averages(self, aps, rcs):
        o50 = np.where(np.isclose(self.ious, 0.5))
        o25 = np.where(np.isclose(self.ious, 0.25))
        rest = np.where(np.logical_not(np.isclose(self.ious, 0.25)))
        avg = {'all_ap': np.nanmean(aps[0, :, rest]), 'all_ap_50%': np.nanmean(aps[0, :, o50]),
               'all_ap_25%': np.nanmean(aps[0, :, o25]), 'all_rc': np.nanmean(rcs[0, :, rest]),
               'all_rc_50%': np.nanmean(rcs[0, :, o50]), 'all_rc_25%': np.nanmean(rcs[0, :, o25]),
               'classes': {}}
        for li, name in enumerate(self.eval_class_labels):
            avg['classes'][name] = {
                'ap': np.average(aps[0, li, rest]), 'ap50%': np.average(aps[0, li, o50]),
                'ap25%': np.average(aps[0, li, o25]), 'rc': np.average(rcs[0, li, rest]),
                'rc50%': np.average(rcs[0, li, o50]), 'rc25%': np.average(rcs[0, li, o25])}
        return avg

    def evaluate(self, pred_list, gt_list, verbose=True):
        """pred_list: per scan a list of dict(scan_id, label_id, conf, pred_mask (RLE dict or array));
        gt_list: per scan an array of class_id * 1000 + instance_id per point (0 = unannotated)."""
        matches = {}
        for i, (preds, gts) in enumerate(zip(pred_list, gt_list)):
            gt2pred, pred2gt = self.assign_instances_for_scan(preds, gts)
            matches[f'gt_{i}'] = {'gt': gt2pred, 'pred': pred2gt}
        avgs = self.compute_averages(*self.evaluate_matches(matches))
        if verbose:
            self.print_results(avgs)
        return avgs

    def print_results(self, avgs):
        width = 64
        cols = ('AP', 'AP_50%', 'AP_25%', 'AR', 'RC_50%', 'RC_25%')
        print()
        print('#' * width)
        print('{:<15}'.format('what') + ':' + ''.join('{:>8}'.format(c) for c in cols))
        print('#' * width)
        keys = ('ap', 'ap50%', 'ap25%', 'rc', 'rc50%', 'rc25%')
        for name in self.eval_class_labels:
            c = avgs['classes'][name]
            print('{:<15}'.format(name) + ':' + ''.join('{:>8.3f}'.format(c[k]) for k in keys))
        print('-' * width)
        alls = ('all_ap', 'all_ap_50%', 'all_ap_25%', 'all_rc', 'all_rc_50%', 'all_rc_25%')
        print('{:<15}'.format('average') + ':' + ''.join('{:>8.3f}'.format(avgs[k]) for k in alls))
        print('#' * width)
        print()

    def write_result_file(self, avgs, filename):
        with open(filename, 'w') as f:
            f.write(','.join(['class', 'class id', 'ap', 'ap50', 'ap25']) + '\n')
            for name in self.eval_class_labels:
                c = avgs['classes'][name]
                f.write(','.join(str(x) for x in [name, c['ap'], c['ap50%'], c['ap25%']]) + '\n')
