"""Developer tool: per-wave phase stamps of the sparse-conv kernel (SG_CONV_TRACE build path).
Usage (GPU box): python tools/conv_trace.py [Cin Cout M_out]   -> phase statistics for that layer"""
import os
import sys

import numpy as np

TRACE = '/tmp/sg_conv_trace.bin'
os.environ['SG_CONV_TRACE'] = TRACE
if os.path.exists(TRACE):
    os.remove(TRACE)
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softgroup_amd import synthetic  # noqa: E402


def main():
    want = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (64, 64, 76839)
    xyz, rgb, inst = synthetic.scene_s2(seed=1, n=150000)
    batch = synthetic.make_batch(xyz, rgb, instance_labels=inst)
    batch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    model = synthetic.build_model(seed=0)
    model.use_executor = False
    with torch.no_grad():
        model(batch)
        torch.cuda.synchronize()
        os.remove(TRACE)
        model(batch)
        torch.cuda.synchronize()
    raw = np.fromfile(TRACE, dtype=np.uint64)
    pos = 0
    done = 0
    while pos < len(raw):
        hdr = raw[pos:pos + 8].astype(np.int64)
        M, K, Cin, Cout, units, col_units, ksplit = [int(x) for x in hdr[:7]]
        n = units * 4 * 8
        rec = raw[pos + 8:pos + 8 + n].reshape(units, 4, 8)
        pos += 8 + n
        if (Cin, Cout, M) != want or K != 27:
            continue
        done += 1
        if done != 2:
            continue
        # the split-precision path stores its waves-per-unit where the fp32 path stores ksplit; units of
        # fewer than four waves leave the other slots zero: keep the waves that stamped something
        live = int((rec[:, :, 0] != 0).any(0).sum()) if rec.size else 4
        if 0 < live < 4:
            rec = rec[:, :live]
        t = rec[:, :, :6].astype(np.int64)
        wgid = rec[:, 0, 6].astype(np.int64)
        meta = rec[:, :, 7]
        n_iter = (meta >> np.uint64(48)).astype(np.int64)
        xcc = ((meta >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64)
        hw = (meta & np.uint64(0xffffffff)).astype(np.int64)
        cu = xcc * 256 + ((hw >> 8) & 0xff)
        # s_memtime bases differ across the chip: align every CU to its own first stamp
        for x in np.unique(cu):
            sel = cu[:, 0] == x
            t[sel] -= t[sel].min()
        span = t.max()
        G = int(hdr[7])
        print(f'layer K={K} Cin={Cin} Cout={Cout} M={M} units={units} col_units={col_units} ksplit={ksplit} grid={G}')
        print(f'span {span} ticks; distinct CUs {len(np.unique(cu))}')
        names = ['matrix loop(t1-t0)', 'barrier A(t2-t1)', 'red+publish+barrier B(t3-t2)', 'setup next(t4-t3)', 'epilogue(t5-t4)', 'unit total(t5-t0)']
        d = [t[:, :, 1] - t[:, :, 0], t[:, :, 2] - t[:, :, 1], t[:, :, 3] - t[:, :, 2], t[:, :, 4] - t[:, :, 3],
             t[:, :, 5] - t[:, :, 4], t[:, :, 5] - t[:, :, 0]]
        for nm, x in zip(names, d):
            print(f'  {nm:30s} mean {x.mean():9.0f}  p10 {np.percentile(x, 10):8.0f} p50 {np.percentile(x, 50):8.0f} p90 {np.percentile(x, 90):8.0f} max {x.max():8.0f}')
        it = n_iter.astype(np.float64)
        loop = (t[:, :, 1] - t[:, :, 0]).astype(np.float64)
        ok = it > 0
        print(f'  slices/wave mean {it.mean():.1f} (min {it.min():.0f} max {it.max():.0f}); ticks per slice {loop[ok].sum() / it[ok].sum():.0f}')
        mfma = it.sum() * 8 * 64
        ncu = len(np.unique(cu))
        print(f'  MFMA cycles issued {mfma:.3e}; per SIMD-span utilisation {mfma / (span * ncu * 4):.3f}')
        # gaps between consecutive units of one persistent workgroup
        o = np.lexsort((t[:, 0, 0], wgid))
        same = wgid[o][1:] == wgid[o][:-1]
        gap = (t[o][1:, 0, 0] - t[o][:-1, 0, 5])[same]
        print(f'  gap end-of-unit -> next loop start (same workgroup): mean {gap.mean():.0f} p90 {np.percentile(gap, 90):.0f}')
        first = t[o][np.r_[True, ~same], 0, 0]
        print(f'  first loop start per workgroup: mean {first.mean():.0f} p90 {np.percentile(first, 90):.0f}')
        last = np.array([t[wgid == w, :, 5].max() for w in np.unique(wgid)])
        print(f'  workgroup end: p10 {np.percentile(last, 10):.0f} p50 {np.percentile(last, 50):.0f} p90 {np.percentile(last, 90):.0f} max {last.max()}')
        # per workgroup: work (slices of its 4 waves' max per unit) vs finishing time
        wgs = np.unique(wgid)
        work = np.array([n_iter[wgid == w].max(1).sum() for w in wgs])
        nun = np.array([(wgid == w).sum() for w in wgs])
        print(f'  per workgroup: units {nun.min()}..{nun.max()}, slices mean {work.mean():.1f} min {work.min()} max {work.max()}; '
              f'corr(work, end) = {np.corrcoef(work, last)[0, 1]:.2f}')
        speed = last / np.maximum(work, 1)
        wxcc = np.array([xcc[wgid == w][0, 0] for w in wgs])
        for x in np.unique(wxcc):
            sel = wxcc == x
            print(f'    xcc {x}: workgroups {sel.sum():4d} end mean {last[sel].mean():9.0f} max {last[sel].max():9.0f} '
                  f'ticks/slice {speed[sel].mean():7.0f} work mean {work[sel].mean():.1f}')
        wcu = np.array([cu[wgid == w][0, 0] for w in wgs])
        per_cu_end = np.array([last[wcu == c].max() for c in np.unique(wcu)])
        per_cu_n = np.array([(wcu == c).sum() for c in np.unique(wcu)])
        print(f'  per CU: resident workgroups {per_cu_n.min()}..{per_cu_n.max()}; last end p10 {np.percentile(per_cu_end, 10):.0f} '
              f'p50 {np.percentile(per_cu_end, 50):.0f} p90 {np.percentile(per_cu_end, 90):.0f}; '
              f'corr(n resident, end) = {np.corrcoef(per_cu_n, per_cu_end)[0, 1]:.2f}')
        w0 = np.unique(wgid)[:3]
        for w in w0:
            sel = np.where(wgid == w)[0]
            sel = sel[np.argsort(t[sel, 0, 0])]
            print(f'  workgroup {w}:')
            for i in sel:
                print(f'    unit {i:6d} start {t[i, 0, 0]:8d} loop_end {t[i, :, 1].tolist()} B {t[i, 0, 3]:8d} end {t[i, 0, 5]:8d} slices {n_iter[i].tolist()}')
        return
    print('layer not found')


if __name__ == '__main__':
    main()
