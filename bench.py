"""Benchmark of the SoftGroup hot path on MI355X: ScanNet-v2-like inference (BASELINE config 2).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full ``forward_test`` (voxel pooling -> sparse U-Net -> point heads -> soft
grouping -> proposal voxelisation -> tiny U-Net -> instance heads -> masks/RLE) over one synthetic
150k-point scene whose tensors are already resident in HBM.  Scenes shard one per GPU (no
data-path collective): weak scaling, value = scans/s summed over ranks.

Besides the headline line the JSON carries
  roofline      the dominant kernel (sparse-conv gather implicit GEMM): algorithmic gather/scatter
                bytes (SURVEY 8d B_gs) / HIP-event kernel time vs the 8 TB/s HBM roof, and its
                fp32 MFMA rate vs 157.3 TFLOP/s
  cpu_baseline  the CPU restatement of the reference model (oracle/, kind "port") timed on the
                host cores of the same box on one scan of the same scene (rank 0, N=1 only)
  stages        per-stage milliseconds of the GPU path (HIP events), for orientation
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP32_MFMA_PEAK_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32 dense peak
REF_MS_PER_SCAN = 288.0       # reference README.md:22, 1x Titan X, real ScanNet v2


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--points', type=int, default=150000)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    return ap.parse_args()


def stage_times(model, batch, reps=5):
    """HIP-event timing of the forward's stages on the current stream (orientation only)."""
    from softgroup_amd import ops
    import softgroup_amd.spconv.pytorch as spconv
    names = ['voxelize+backbone+heads', 'grouping', 'proposal_voxelization', 'tiny_unet+heads',
             'instances+rle']
    acc = [0.0] * len(names)
    b = batch
    for _ in range(reps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
        with torch.no_grad():
            ev[0].record()
            feats = torch.cat((b['feats'], b['coords_float']), 1)
            vf = ops.voxelization(feats, b['p2v_map'])
            x = spconv.SparseConvTensor(vf, b['voxel_coords'].int(), b['spatial_shape'], 1)
            sem, off, out_feats = model.forward_backbone(x, b['v2p_map'])
            ev[1].record()
            pidx, poff = model.forward_grouping(sem, off, b['batch_idxs'], b['coords_float'])
            ev[2].record()
            inst, inst_map = model.clusters_voxelization(pidx, poff, out_feats, b['coords_float'],
                                                         **model.instance_voxel_cfg)
            ev[3].record()
            _, cls_s, iou_s, mask_s = model.forward_instance(inst, inst_map)
            ev[4].record()
            preds = model.get_instances('s', pidx, sem, cls_s, iou_s, mask_s)
            ev[5].record()
        torch.cuda.synchronize()
        for i in range(len(names)):
            acc[i] += ev[i].elapsed_time(ev[i + 1]) / reps
    info = dict(points=int(b['coords_float'].shape[0]), voxels=int(b['voxel_coords'].shape[0]),
                grouped_points=None, proposals=int(max(poff.numel() - 1, 0)),
                proposal_points=int(pidx.shape[0]), instances=len(preds))
    return {n: round(t, 3) for n, t in zip(names, acc)}, info


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist_on = world > 1
    import torch.distributed as dist
    torch.cuda.set_device(local_rank if torch.cuda.device_count() > local_rank else 0)
    if dist_on:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl')       # RCCL on ROCm
    from softgroup_amd import _lib, synthetic
    from softgroup_amd.spconv import core as spcore
    assert os.path.exists(_lib.LIB_PATH), 'libsoftgroup_hip.so missing: run __graft_entry__.build()'

    # one scene per rank (different seed per rank), weights identical on all ranks
    xyz, rgb, inst = synthetic.scene_s2(seed=1 + rank, n=args.points)
    batch = synthetic.make_batch(xyz, rgb, instance_labels=inst, scan_id=f'synthetic_{rank:04d}')
    batch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    model = synthetic.build_model(seed=0)

    def sync_all():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    # forward_test returns as soon as the scan's GPU work is enqueued; turning its results into host
    # objects (numpy arrays, RLE mask strings) runs on the model's results thread and overlaps the
    # next scan.  Every one of the K result dicts is fully materialised inside the timed region.
    with torch.no_grad():
        for _ in range(args.warmup):
            model(batch).resolve()
        sync_all()
        t0 = time.perf_counter()
        rets = [model(batch) for _ in range(args.steps)]
        for r in rets:
            r.resolve()
        sync_all()
        elapsed = time.perf_counter() - t0
    assert all('pred_instances' in r and 'semantic_preds' in r for r in rets)
    del rets
    if dist_on:
        t = torch.tensor([elapsed], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1000.0
    value = world * args.steps / elapsed

    out = {
        'metric': 'scans/sec ScanNet-v2-like inference (softgroup_scannet.yaml, full forward_test)',
        'value': round(value, 3), 'unit': 'scans/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': round(value / world / (1000.0 / REF_MS_PER_SCAN), 3),
        'dtype': 'f32', 'data': 'synthetic',
        'config': {
            'workload': f'S2 synthetic room scene, {args.points} pts/scene, 0.02 m voxels, one scene '
                        f'per GPU; softgroup_scannet.yaml model section; random-init weights '
                        f'(30.84 M params), BN running stats randomised, semantic head last layer '
                        f're-drawn with std 20 so grouping/refinement run on a realistic load',
            'baseline_note': 'vs_baseline = per-GPU scans/s / (1000/288): 288 ms/scan is the '
                             'reference README number on 1x Titan X with real ScanNet v2 data',
            'parallelism': f'scenes sharded one per GPU x{world}, no data-path collective',
        },
    }

    if rank == 0:
        # the same K scans with result formatting done in line (no overlap with the next scan), for
        # reference next to the pipelined figure above
        model.async_results = False
        with torch.no_grad():
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(min(args.steps, 10)):
                model(batch)
            torch.cuda.synchronize()
            out['ms_per_step_unpipelined'] = round((time.perf_counter() - t1) / min(args.steps, 10) * 1e3, 3)
        model.async_results = True
        stages, info = stage_times(model, batch)
        out['stages_ms'] = stages
        out['scene'] = info
        name = (_lib.C.c_char * 128)()
        cu, clk = _lib.C.c_int(0), _lib.C.c_int(0)
        _lib.lib().sg_device_info(name, 128, _lib.C.byref(cu), _lib.C.byref(clk))
        out['device'] = {'name': name.value.decode(), 'cus': cu.value, 'clock_khz': clk.value}

    if rank == 0 and not args.no_roofline:
        # dominant kernel: gather_conv_persistent_kernel (sparse-conv implicit GEMM)
        # (a) algorithmic bytes / flops per launch: one forward on the module path, which makes one
        #     Python call per conv launch (same kernels as the native executor of the timed region);
        # (b) kernel time: the library brackets every conv launch with a HIP event pair on its launch
        #     stream (sg_spconv_profile) while the SAME path as the timed region runs n_pass scans.
        prof = spcore.ConvProfiler()
        spcore.PROFILER = prof
        model.use_executor = False
        with torch.no_grad():
            model(batch).resolve()
        s = prof.summary()
        spcore.PROFILER = None
        model.use_executor = True
        n_pass = min(args.steps, 5)
        lib = _lib.lib()
        _lib.check(lib.sg_spconv_profile(1), 'sg_spconv_profile')
        with torch.no_grad():
            for _ in range(n_pass):
                model(batch).resolve()
        torch.cuda.synchronize()
        ms, nl = _lib.C.c_double(0), _lib.C.c_int(0)
        _lib.check(lib.sg_spconv_profile_read(_lib.C.byref(ms), _lib.C.byref(nl)), 'sg_spconv_profile_read')
        _lib.check(lib.sg_spconv_profile(0), 'sg_spconv_profile')
        assert nl.value == s['launches'] * n_pass, (nl.value, s['launches'])
        s = dict(launches=nl.value, ms=ms.value, bytes=s['bytes'] * n_pass, flops=s['flops'] * n_pass)
        gbps = s['bytes'] / (s['ms'] * 1e-3) / 1e9
        tflops = s['flops'] / (s['ms'] * 1e-3) / 1e12
        launches = max(s['launches'], 1)
        # HBM traffic of the same kernel from rocprofv3 PMC (FETCH_SIZE / WRITE_SIZE in KiB, separate
        # passes, collected by the recipe in profiles/README.md on this workload and committed as
        # profiles/r01_conv_pmc.json).  FETCH_SIZE x2 is the guide's gfx950 correction for 16-B/lane
        # reads (MI355X_MICROARCH.md "HBM"); null when the file is not there.
        traffic = None
        pmc_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r01_conv_pmc.json')
        try:
            pmc = json.load(open(pmc_path))['counters']
            traffic = int((2 * pmc['FETCH_SIZE']['per_dispatch'] + pmc['WRITE_SIZE']['per_dispatch']) * 1024)
        except (OSError, ValueError, KeyError):
            pass
        hbm = {'achieved': round(gbps, 1), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
               'frac': round(gbps / HBM_PEAK_GBPS, 4)}
        mfma = {'achieved': round(tflops, 2), 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(tflops / FP32_MFMA_PEAK_TFLOPS, 4)}
        # the roof that actually binds is the one with the larger fraction (fp32 MFMA for this
        # kernel: AI = 2*Cout/4.25 flop/B of gathered data); the other one is kept alongside
        bound = 'mfma' if mfma['frac'] >= hbm['frac'] else 'hbm'
        out['roofline'] = {
            'kernel': 'gather_conv_persistent_kernel (SubM/strided/inverse sparse conv, fp32 MFMA)',
            'bound': bound, **(mfma if bound == 'mfma' else hbm),
            'traffic': traffic,
            'launches_per_scan': s['launches'] // n_pass,
            'kernel_ms_per_scan': round(s['ms'] / n_pass, 3),
            'avg_launch_us': round(s['ms'] * 1e3 / launches, 2),
            'algorithmic_bytes_per_launch': s['bytes'] // launches,
            'algorithmic_flops_per_launch': s['flops'] // launches,
            'algorithmic_bytes_per_scan': s['bytes'] // n_pass,
            'flops_per_scan': s['flops'] // n_pass,
            'hbm': hbm, 'mfma': mfma,
        }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle  # checker only: CPU restatement of the reference model
        from oracle import parity
        oracle.build()
        cpu_batch = synthetic.make_batch(xyz, rgb, instance_labels=inst)
        # one oracle forward of the SAME scene: timed as the CPU baseline, and its outputs are the
        # full-size parity check of the GPU path (stage-wise: floats <= 1e-4, proposals / instance
        # labels / RLE strings identical; end to end: instance drift)
        rep = parity.parity_report(model, cpu_batch, synthetic.SCANNET_MODEL_CFG)
        cpu_s = rep.pop('oracle_forward_s')
        out['parity_at_bench'] = rep
        out['cpu_baseline'] = {
            'value': round(1.0 / cpu_s, 4), 'unit': 'scans/s', 'cores': os.cpu_count(),
            'kind': 'port',
            'sample': f'1 scan of the same S2 scene ({args.points} pts): C/OpenMP sparse conv on all '
                      f'cores, single-thread brute-force ball query + BFS like the reference CPU ops; '
                      f'{cpu_s:.1f} s',
        }

    if rank == 0:
        print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
