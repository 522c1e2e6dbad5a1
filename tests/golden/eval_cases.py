"""Deterministic evaluation inputs shared by tests/golden/make_eval_golden.py (which runs the
REFERENCE's ScanNetEval on them) and tests/test_eval.py (which runs ours): two synthetic scans with
GT instance ids (class*1000 + instance) and predictions that hit every branch of the matcher --
several predictions per GT with IoUs across the 0.25..0.95 thresholds, duplicates on one GT, false
positives on unannotated points, a GT below the minimum region size, predictions with an invalid
label or too few points, confidence ties."""
import numpy as np

CLASSES = ('cabinet', 'bed', 'chair', 'sofa', 'table', 'door', 'window', 'bookshelf', 'picture',
           'counter', 'desk', 'curtain', 'refrigerator', 'shower curtain', 'toilet', 'sink', 'bathtub',
           'otherfurniture')


def _rle(mask):
    m = np.concatenate([[0], mask.astype(np.int8), [0]])
    runs = np.flatnonzero(m[1:] != m[:-1]) + 1
    runs[1::2] -= runs[::2]
    return dict(length=int(mask.shape[0]), counts=' '.join(str(x) for x in runs))


def make_case(seed, n_points=60000, n_inst=14, as_rle=True):
    rng = np.random.default_rng(seed)
    gts = np.zeros(n_points, np.int64)
    # instances = contiguous-ish blocks with holes (point order of a scan is spatially coherent)
    cuts = np.sort(rng.choice(np.arange(2000, n_points - 2000), n_inst * 2, replace=False))
    inst_masks = []
    for i in range(n_inst):
        lo, hi = cuts[2 * i], cuts[2 * i + 1]
        if i == 3:
            hi = lo + 60                                  # a GT below min_region_size (100)
        m = np.zeros(n_points, bool)
        m[lo:hi] = rng.random(hi - lo) < 0.9
        cls = int(rng.integers(1, len(CLASSES) + 1)) if i != 5 else 25     # one GT of a non-evaluated class
        gts[m] = cls * 1000 + i + 1
        inst_masks.append((m, cls))
    preds = []

    def add(mask, label, conf):
        preds.append(dict(scan_id=f'scan{seed}', label_id=int(label), conf=np.float32(conf),
                          pred_mask=_rle(mask) if as_rle else mask.astype(np.int32)))

    for i, (m, cls) in enumerate(inst_masks):
        idx = np.flatnonzero(m)
        for _ in range(int(rng.integers(0, 4))):
            keep = rng.random(len(idx)) < rng.uniform(0.3, 1.0)
            pm = np.zeros(n_points, bool)
            pm[idx[keep]] = True
            extra = rng.integers(0, n_points, int(rng.uniform(0, 0.6) * len(idx)))
            pm[extra] = True
            label = cls if rng.random() < 0.85 else int(rng.integers(1, len(CLASSES) + 1))
            add(pm, label, rng.choice([0.3, 0.5, 0.5, 0.7, 0.9, rng.random()]))
    for _ in range(6):                                     # false positives, partly on unannotated points
        pm = np.zeros(n_points, bool)
        a = int(rng.integers(0, n_points - 3000))
        pm[a:a + int(rng.integers(150, 3000))] = True
        add(pm, rng.integers(1, len(CLASSES) + 1), rng.random())
    pm = np.zeros(n_points, bool)
    pm[:50] = True
    add(pm, 3, 0.99)                                       # too small: skipped
    pm = np.zeros(n_points, bool)
    pm[1000:5000] = True
    add(pm, 40, 0.8)                                       # label not evaluated: skipped
    order = rng.permutation(len(preds))
    return [preds[i] for i in order], gts


def cases(as_rle=True):
    pl, gl = [], []
    for seed in (101, 202):
        p, g = make_case(seed, as_rle=as_rle)
        pl.append(p)
        gl.append(g)
    return pl, gl
