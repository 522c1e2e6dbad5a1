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
  parity_at_bench  the oracle's outputs on this very scene compared with the GPU's (not discarded)
  legs          (config_legs) BASELINE configs 3 / 4 / 5 at their own sizes: S3DIS-shaped training
                step (batch 4, frozen backbone, fp32 and bf16 autocast), SoftGroup++/STPLS3D and
                SemanticKITTI inference ms/scan with stage split; and
                SURVEY 8(d)'s other measurement legs (rank 0, N=1): G1 grouping-head input with
                roofline entries for the ball query and the BFS clustering, a 2x denser 300k-point
                scene, the S1 config-1 backbone on the CPU (all cores and OMP_NUM_THREADS=1), and
                the host-to-device inclusive rate of the S2 scan (device-side collate)

The timed region keeps ``--contexts`` scans in flight (model.scan_contexts worker threads, one HIP
stream each): a single scan has ~7 host<->device round trips and several latency-bound kernels
(one workgroup replaying a BFS, 18-voxel U-Net levels), so the GPU is kept busy by overlapping
scans, as a serving deployment would.  ``ms_per_step_unpipelined`` is one scan at a time with all
result formatting in line (the latency figure).
"""
import argparse
import ctypes
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
BF16_MFMA_PEAK_TFLOPS = 2500.0  # v_mfma_f32_32x32x16_bf16 dense peak (MI355X_MICROARCH.md)
REF_MS_PER_SCAN = 288.0       # reference README.md:22, 1x Titan X, real ScanNet v2


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=160)
    ap.add_argument('--warmup', type=int, default=8)
    ap.add_argument('--points', type=int, default=150000)
    ap.add_argument('--scenes', type=int, default=4,
                    help='distinct pre-built scenes the timed region cycles through')
    ap.add_argument('--contexts', type=int, default=0,
                    help='scans in flight in the timed region; 0 = 10 (with one stream less per scan worker and two '
                         'backbone permits, 8-12 in flight beat 3-5 in the 20-step region and in the 160-step one: '
                         'profiles/r06_token_matrix.txt, end of round 6)')
    ap.add_argument('--switch-interval-us', type=int, default=0,
                    help='sys.setswitchinterval for the process (0 = leave CPython\'s 5 ms)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-legs', action='store_true')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help="'gloo' + --stub: the launch / sharding / timing plumbing on CPU (tests)")
    ap.add_argument('--stub', action='store_true',
                    help='replace the scan by a fixed host-side delay (plumbing test, no GPU)')
    ap.add_argument('--stub-ddp-hang-rank', type=int, default=-1,
                    help='with --stub: this rank never joins the DDP leg (tests the deadline that protects the line)')
    ap.add_argument('--stub-fail-rank', type=int, default=-1,
                    help='(tests) this rank raises before the first barrier: the job must exit non-zero')
    return ap.parse_args()


def host_thread_plan(contexts, world):
    """Host threads of one rank: `contexts` scan threads + the results thread + the main thread, no
    intra-op pools (OMP_NUM_THREADS=1, torch.set_num_threads(1)).  All ranks of the node together
    stay within half the host cores: contexts is lowered (never below 1) if 8 ranks x (contexts + 2)
    would not fit -- the line reports what was used."""
    cores = os.cpu_count() or 1
    per_rank_budget = max(3, (cores // 2) // max(world, 1))
    used = max(1, min(contexts, per_rank_budget - 2))
    # page-locked result staging of one rank: the region submits all its scans up front, so a consumer that
    # lags keeps as many results alive as the cap allows -- SG_PINNED_RESULTS_MB (256) of ~12 MB blocks (16 MB
    # in the host allocator) plus the one being filled; beyond the cap results are copied to pageable memory
    # (util/cast.py).  Measured: 5 blocks page-locked inside a 160-scan region when the consumer keeps up, 17
    # when results complete out of order (profiles/r05_inflight_modes.txt).
    pinned_mb = int(os.environ.get('SG_PINNED_RESULTS_MB', '256')) + 16
    return used, {'scan_threads': used, 'results_thread': 1, 'main_thread': 1, 'per_rank': used + 2,
                  'all_ranks': world * (used + 2), 'host_cores': cores, 'intra_op_threads': 1,
                  'pinned_staging_mb_per_rank_bound': pinned_mb}


def finish_n_gt_1(out, args, rank, world, local_rank, stub):
    """N > 1: the DDP leg, the JSON line and the common exit, with the line GUARANTEED.  `out` is complete
    when this is called (the headline was measured and max-reduced over the ranks); the DDP leg only adds
    `ddp` and `rank_core_slices`.  A collective of the leg that never completes (a rank lost, ranks taking
    different branches) must not cost the job its line: a watchdog thread -- it runs while the main thread
    sits in a collective, which releases the interpreter lock -- prints the line with the reason after
    SG_BENCH_DDP_DEADLINE_S (180) seconds and ends the process; the other ranks end the same way."""
    import threading
    import torch.distributed as dist
    deadline = float(os.environ.get('SG_BENCH_DDP_DEADLINE_S', '180'))
    done = threading.Event()
    printed = threading.Lock()

    def emit():
        if rank == 0 and printed.acquire(blocking=False):
            out['legs'] = 'skipped (N>1)'           # nothing is skipped silently: no single-GPU legs with N > 1
            out.setdefault('cpu_baseline', 'skipped (N>1)')
            print(json.dumps(out), flush=True)

    def watchdog():
        if not done.wait(deadline):
            out.setdefault('ddp', {'error': f'the DDP leg / the common exit did not finish within {deadline:.0f} s '
                                            f'(rank {rank}); headline unaffected'})
            emit()
            sys.stderr.write(f'[bench] rank {rank}: DDP leg deadline passed, leaving\n')
            sys.stderr.flush()
            os._exit(0)

    threading.Thread(target=watchdog, daemon=True).start()
    try:
        ddp = ddp_leg(args, rank, world, local_rank, stub)
        cores = [None] * world
        dist.all_gather_object(cores, args._core_slice)
        out['ddp'] = ddp
        out['rank_core_slices'] = cores
    except Exception as e:      # (the same error on every rank: the line still goes out)
        out['ddp'] = {'error': f'{type(e).__name__}: {e}'}
    emit()
    dist.barrier()              # leave together
    done.set()
    dist.destroy_process_group()


def bind_rank_to_cores(local_rank, world):
    """One contiguous slice of the host cores per rank (rank r of N: cores [r * C / N, (r + 1) * C / N) of
    the cores this process may run on -- on a two-socket box with the GPUs split between the sockets
    that is the NUMA-local half).  Returns the slice as [first, last] or None where the platform has
    no sched_setaffinity or the launcher already pinned the rank to fewer cores than its share."""
    if world <= 1 or not hasattr(os, 'sched_setaffinity'):
        return None
    try:
        allowed = sorted(os.sched_getaffinity(0))
        share = len(allowed) // world
        if share < 1:
            return None
        mine = allowed[local_rank * share:(local_rank + 1) * share]
        os.sched_setaffinity(0, mine)
        return [mine[0], mine[-1]]
    except OSError:
        return None


def ddp_leg(args, rank, world, local_rank, stub):
    """N > 1 only: what the training side adds per step on this node -- the gradient all-reduce of the
    trainable heads (2 919 208 bytes with the fine-tune configs' frozen backbone: DistributedDataParallel
    puts them in one bucket; reference tools/train.py:172-174) -- measured on the job's own process
    group (RCCL over xGMI on the GPU node, gloo in the CPU plumbing test), and a short DDP training
    leg on every rank: softgroup_s3dis_fold5.yaml shapes, one 150 k-point scene per rank, frozen
    backbone, forward_train + backward + Adam under bf16 autocast.  The stub leg trains a parameter
    vector of the same byte size on the host.  -> dict for the JSON line (rank 0), per-rank step times
    gathered from all ranks."""
    import numpy as np
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    n_bytes = 2919208
    dev = torch.device('cpu') if stub else torch.device('cuda', local_rank)
    bucket = torch.ones(n_bytes // 4, dtype=torch.float32, device=dev)

    def sync():
        if not stub:
            torch.cuda.synchronize()

    for _ in range(3):
        dist.all_reduce(bucket)
    sync()
    ts = []
    for _ in range(20):
        bucket.fill_(1.0)
        sync()
        dist.barrier()
        t0 = time.perf_counter()
        dist.all_reduce(bucket)
        sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    assert float(bucket[0]) == float(world), 'all-reduce(SUM) of ones must give the world size'
    ts.sort()
    rec = {'allreduce_bytes': n_bytes, 'allreduce_ms_median': round(ts[len(ts) // 2], 4),
           'allreduce_ms_min': round(ts[0], 4), 'backend': dist.get_backend(),
           'algbw_GBps': round(n_bytes / (ts[len(ts) // 2] * 1e-3) / 1e9, 3)}
    # ---- DDP training steps
    steps = 6
    if stub and rank == getattr(args, 'stub_ddp_hang_rank', -1):      # plumbing test: this rank never joins the leg
        time.sleep(3600)
    if stub:
        torch.manual_seed(0)
        net = torch.nn.Linear(n_bytes // 4 - 1, 1)          # weight + bias = n_bytes / 4 parameters
        model = DDP(net)
        opt = torch.optim.Adam(net.parameters(), lr=1e-4)
        x = torch.randn(4, n_bytes // 4 - 1)
        step_ms = []
        for it in range(steps + 2):
            t0 = time.perf_counter()
            loss = model(x).square().mean()
            opt.zero_grad()
            loss.backward()
            opt.step()
            step_ms.append((time.perf_counter() - t0) * 1e3)
        trainable = sum(p.numel() * p.element_size() for p in net.parameters())
    else:
        import copy
        from softgroup_amd import synthetic
        from softgroup_amd.data import collate_device, make_item
        from softgroup_amd.model import SoftGroup
        cfg = copy.deepcopy(synthetic.S3DIS_MODEL_CFG)
        torch.manual_seed(0)
        net = SoftGroup(**cfg).cuda()
        with torch.no_grad():
            net.semantic_linear[-1].weight.normal_(0, 20.0)
        net.train()
        model = DDP(net, device_ids=[local_rank], find_unused_parameters=True)
        params = [p for p in net.parameters() if p.requires_grad]
        opt = torch.optim.Adam(params, lr=1e-4)
        x, c, ins = synthetic.scene_s2(seed=31 + rank, n=args.points)
        sem = np.where(ins >= 0, 2 + ins % 11, 0).astype(np.int64)
        batch = collate_device([make_item(x, c, 50, sem, ins, f'crop_{rank}')])
        batch['instance_cls'] = batch['instance_cls'].clamp(min=0)
        step_ms = []
        for it in range(steps + 2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.autocast('cuda', dtype=torch.bfloat16):
                loss, _ = model(batch, return_loss=True)
            opt.zero_grad()
            loss.backward()
            opt.step()
            torch.cuda.synchronize()
            step_ms.append((time.perf_counter() - t0) * 1e3)
        trainable = sum(p.numel() * p.element_size() for p in params)
    mine = sorted(step_ms[2:])[len(step_ms[2:]) // 2]
    per_rank = [None] * world
    dist.all_gather_object(per_rank, round(mine, 3))
    rec.update({'train_steps_timed': steps, 'trainable_bytes': int(trainable),
                'train_ms_per_step_per_rank': per_rank, 'train_ms_per_step_min': min(per_rank),
                'train_ms_per_step_max': max(per_rank),
                'workload': ('stub: Linear of the same parameter bytes on the host' if stub else
                             'softgroup_s3dis_fold5.yaml shapes, one 150 k-point scene per rank, frozen backbone, '
                             'bf16 autocast, DistributedDataParallel')})
    return rec


def spawn_ranks(args):
    """`python bench.py --gpus N` outside torchrun: launch N ranks of this script, one process per
    GPU, the way the reference's tools/dist_test.sh:7 does (torchrun, OMP_NUM_THREADS=1), and hand
    its exit code back.  Rank 0 of the child job prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS='1', MASTER_ADDR='127.0.0.1')
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def device_identity(stub, rank):
    """something that differs between two physical GPUs: uuid, else PCI address, else the index"""
    if stub:
        return f'cpu:{rank}'
    p = torch.cuda.get_device_properties(torch.cuda.current_device())
    for attr in ('uuid', 'pci_bus_id'):
        v = getattr(p, attr, None)
        if v not in (None, ''):
            extra = f':{getattr(p, "pci_domain_id", 0)}:{getattr(p, "pci_device_id", 0)}' if attr == 'pci_bus_id' else ''
            return f'{attr}={v}{extra}'
    return f'index={torch.cuda.current_device()}'


def stage_times(model, batch, reps=5, warm=1):
    """HIP-event timing of the forward's stages on the current stream (orientation only); `warm`
    untimed passes first (allocator, plan caches), then the MEDIAN of `reps` per stage (a shared box
    throws an occasional 50-100 ms hiccup into one repetition)."""
    from softgroup_amd import ops
    import softgroup_amd.spconv.pytorch as spconv
    names = ['voxelize+backbone+heads', 'grouping', 'proposal_voxelization', 'tiny_unet+heads',
             'instances+rle']
    acc = [[] for _ in names]
    b = batch
    for it in range(warm + reps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
        with torch.no_grad():
            ev[0].record()
            feats = torch.cat((b['feats'], b['coords_float']), 1) if model.with_coords else b['feats']
            vf = ops.voxelization(feats, b['p2v_map'])
            x = spconv.SparseConvTensor(vf, b['voxel_coords'].int(), b['spatial_shape'], b['batch_size'])
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
        if it < warm:
            continue
        for i in range(len(names)):
            acc[i].append(ev[i].elapsed_time(ev[i + 1]))
    info = dict(points=int(b['coords_float'].shape[0]), voxels=int(b['voxel_coords'].shape[0]),
                grouped_points=None, proposals=int(max(poff.numel() - 1, 0)),
                proposal_points=int(pidx.shape[0]), instances=len(preds))
    return {n: round(sorted(t)[len(t) // 2], 3) for n, t in zip(names, acc)}, info



def _events_ms(fn, reps=5, warm=2):
    """mean milliseconds of fn() on the current stream (HIP events)"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def config_legs(args):
    """BASELINE configs 3 / 4 / 5 at SURVEY 8(d)'s sizes (they are parity cases in tests/, here their
    measured cost; bounded to a few seconds each):
      train_step_s3dis  softgroup_s3dis_fold5.yaml shapes: batch 4 scenes x 150 k points per rank,
                        frozen backbone, forward_train + backward + Adam, fp32 and bf16 autocast;
                        gradient bytes = what DDP all-reduces per step
      stpls3d_pp        softgroup++_stpls3d.yaml inference on the S2 coordinates x 40 tile
                        (octree ball query, pyramid level 2 on the 149 k-point class)
      kitti             softgroup_kitti.yaml panoptic inference on one ~120 k-point LiDAR-like sweep
    """
    import copy
    import numpy as np
    from softgroup_amd import synthetic
    from softgroup_amd.data import collate_device, make_item
    from softgroup_amd.model import SoftGroup
    legs = {}

    def cuda(b):
        return {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}

    def infer_leg(cfg, batch):
        model = synthetic.build_model(cfg, seed=0)
        model.async_results = False
        with torch.no_grad():
            for _ in range(2):
                model(batch)
            ts = []
            for _ in range(7):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                model(batch)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            ts.sort()
            stages, info = stage_times(model, batch, reps=5)
        return {'ms_per_scan_unpipelined': round(ts[len(ts) // 2], 3), 'ms_per_scan_min_max': [round(ts[0], 3), round(ts[-1], 3)],
                'stages_ms': stages, 'scene': info}

    # ---- config 2 again, but with a stand-in for a TRAINED checkpoint (synthetic.fit_model_to_scenes:
    #      colour pass-through backbone, calibrated BatchNorms, heads fitted to the synthetic labels of
    #      4 class-coloured scenes): the proposals are the 12 furniture-sized objects of a scene, and
    #      the predictions can be scored against the synthetic ground truth (AP of the evaluator that
    #      equals the reference's ScanNetEval, tests/test_eval.py) -- "at reference AP" has no
    #      checkpoint or dataset to stand on offline; this is its stand-in
    from softgroup_amd.evaluation import ScanNetEval
    fit_batches, eval_batches = [], []
    for i in range(6):
        x, c, ins = synthetic.scene_s2(seed=200 + i, n=args.points, class_colour=True)
        (fit_batches if i < 4 else eval_batches).append(cuda(synthetic.make_batch(x, c, instance_labels=ins)))
    fmodel = synthetic.build_model(seed=0, head_std=None)
    fit = synthetic.fit_model_to_scenes(fmodel, fit_batches)
    fmodel.async_results = False
    with torch.no_grad():
        res = [fmodel(b) for b in eval_batches]        # held-out scenes
        fts = []
        for _ in range(3):
            for b in eval_batches:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fmodel(b)
                torch.cuda.synchronize()
                fts.append((time.perf_counter() - t0) * 1e3)
        fitted_ms = sorted(fts)[len(fts) // 2]
        fstages, finfo = stage_times(fmodel, eval_batches[0], reps=5)
    classes = ['c%d' % i for i in range(18)]
    avgs = ScanNetEval(classes).evaluate([r['pred_instances'] for r in res], [r['gt_instances'] for r in res],
                                         verbose=False)
    legs['fitted_checkpoint'] = {
        'ms_per_scan_unpipelined': round(fitted_ms, 3), 'stages_ms': fstages, 'scene': finfo, 'fit': fit,
        'held_out_scenes': len(eval_batches),
        'AP': round(float(avgs['all_ap']), 4), 'AP50': round(float(avgs['all_ap_50%']), 4),
        'AP25': round(float(avgs['all_ap_25%']), 4),
        'note': 'synthetic class-coloured S2 scenes, stand-in checkpoint fitted on 4 other scenes'}
    del fmodel, fit_batches, eval_batches, res

    # ---- config 4
    xyz, rgb, inst = synthetic.scene_s2(seed=1, n=args.points)
    xyz = (xyz * np.float32(40)).astype(np.float32)
    legs['stpls3d_pp'] = infer_leg(copy.deepcopy(synthetic.STPLS3D_PP_MODEL_CFG),
                                   cuda(synthetic.make_batch(xyz, rgb, scale=3, instance_labels=inst)))
    # ---- config 5
    xyz, intensity, inst = synthetic.scene_lidar(seed=3, n=120000)
    legs['kitti'] = infer_leg(copy.deepcopy(synthetic.KITTI_MODEL_CFG),
                              cuda(synthetic.make_batch(xyz, intensity, scale=20, instance_labels=inst)))
    # ---- config 3
    cfg = copy.deepcopy(synthetic.S3DIS_MODEL_CFG)
    torch.manual_seed(0)
    model = SoftGroup(**cfg).cuda()
    with torch.no_grad():
        model.semantic_linear[-1].weight.normal_(0, 20.0)
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-4)
    items = []
    for i in range(4):
        x, c, ins = synthetic.scene_s2(seed=31 + i, n=args.points)
        sem = np.where(ins >= 0, 2 + ins % 11, 0).astype(np.int64)          # 13 S3DIS classes
        items.append(make_item(x, c, 50, sem, ins, f'crop_{i}'))
    batch = collate_device(items)
    batch['instance_cls'] = batch['instance_cls'].clamp(min=0)
    rec = {'scenes_per_step': 4, 'points_per_step': int(batch['coords_float'].shape[0]),
           'voxels_per_step': int(batch['voxel_coords'].shape[0]),
           'trainable_params': int(sum(p.numel() for p in params)),
           'gradient_allreduce_bytes': int(sum(p.numel() * p.element_size() for p in params))}
    for name, ctx in (('fp32', lambda: torch.autocast('cuda', enabled=False)),
                      ('bf16_autocast', lambda: torch.autocast('cuda', dtype=torch.bfloat16))):
        ts, fw = [], []
        for it in range(17):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with ctx():
                loss, _ = model(batch, return_loss=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            opt.zero_grad()
            loss.backward()
            opt.step()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
            fw.append((t1 - t0) * 1e3)
        ts, fw = sorted(ts[2:]), sorted(fw[2:])
        rec[name] = {'ms_per_step': round(ts[len(ts) // 2], 2), 'forward_ms': round(fw[len(fw) // 2], 2),
                     'loss': round(float(loss.detach()), 4)}
    # where a step goes (bf16 autocast; a device synchronisation on either side of every stage, so
    # the sum exceeds the unsynchronised step above)
    acc, names = {}, ('forward_backbone', 'point_wise_loss', '_native_proposals', 'clusters_voxelization',
                      'forward_instance', 'instance_loss', 'parse_losses')

    def timed(fn, label):
        def wrap(*a, **k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn(*a, **k)
            torch.cuda.synchronize()
            acc[label] = acc.get(label, 0.0) + (time.perf_counter() - t0) * 1e3
            return r
        return wrap
    for nm in names:
        setattr(model, nm, timed(getattr(model, nm), nm))
    reps = 5
    for it in range(reps):
        with torch.autocast('cuda', dtype=torch.bfloat16):
            loss, _ = model(batch, return_loss=True)
        opt.zero_grad()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss.backward()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        opt.step()
        torch.cuda.synchronize()
        acc['backward'] = acc.get('backward', 0.0) + (t1 - t0) * 1e3
        acc['optimizer'] = acc.get('optimizer', 0.0) + (time.perf_counter() - t1) * 1e3
    for nm in names:
        delattr(model, nm)
    rec['stages_ms_bf16_autocast'] = {k: round(v / reps, 3) for k, v in acc.items()}
    # conv launches of a training step (forward + input gradient; fp32 arithmetic): algorithmic bytes /
    # flops from one step on the module path (one Python call per launch, SURVEY 8(d) formulas),
    # time from the in-library HIP events over 3 steps on the executors
    from softgroup_amd import _lib
    from softgroup_amd.spconv import core as spcore
    try:
        prof = spcore.ConvProfiler()
        spcore.PROFILER = prof
        model.use_executor = model.use_train_executor = False
        loss, _ = model(batch, return_loss=True)
        opt.zero_grad()
        loss.backward()
        s = prof.summary()
    finally:
        spcore.PROFILER = None
        model.use_executor = model.use_train_executor = True
    lib = _lib.lib()
    _lib.check(lib.sg_spconv_profile(1), 'sg_spconv_profile')
    for it in range(3):
        loss, _ = model(batch, return_loss=True)
        opt.zero_grad()
        loss.backward()
    torch.cuda.synchronize()
    ms, nl = _lib.C.c_double(0), _lib.C.c_int(0)
    _lib.check(lib.sg_spconv_profile_read(_lib.C.byref(ms), _lib.C.byref(nl)), 'sg_spconv_profile_read')
    _lib.check(lib.sg_spconv_profile(0), 'sg_spconv_profile')
    conv_ms = ms.value / 3
    rec['conv_forward_and_input_gradient_fp32'] = {
        'kernel': 'gather_conv_persistent_kernel: frozen backbone forward + tiny U-Net forward and input gradients',
        'launches_per_step': nl.value // 3, 'launches_on_the_module_path': s['launches'],
        'ms_per_step': round(conv_ms, 3), 'algorithmic_bytes_per_step': int(s['bytes']),
        'flops_per_step': int(s['flops']), 'bound': 'hbm',
        'achieved': round(s['bytes'] / (conv_ms * 1e-3) / 1e9, 1), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
        'frac': round(s['bytes'] / (conv_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
        'note': 'weight gradients (sg_spconv_wgrad) are not in this entry: tools/train_conv_bench.py, '
                'profiles/r04_train_conv.txt'}
    rec['executors'] = {'backbone': 'sg_unet_forward (frozen: inference executor, bf16 operands under autocast)',
                        'tiny_unet': 'sg_unet_train_forward / sg_unet_train_backward',
                        'grouping': 'sg_scan_grouping (proposals only)'}
    legs['train_step_s3dis'] = rec
    return legs


def measurement_legs(args, model, batch, xyz, rgb, inst):
    """SURVEY 8(d)'s other legs.  Bounded: a few seconds of GPU work, ~20 s of CPU work."""
    import numpy as np
    from softgroup_amd import ops, synthetic
    legs = {}

    # ---- G1: the grouping head on its own input (40 blobs x 1000 pts + 10 000 noise points,
    #      r = 0.04, ~6.9 M neighbour pairs).  Algorithmic bytes of SURVEY 8(d):
    #      ball query >= n*20 + nActive*4, BFS clustering >= nActive*12 + n*16 + S*8.
    g1 = torch.from_numpy(synthetic.scene_g1(seed=2)).cuda()
    n = g1.shape[0]
    bi = torch.zeros(n, dtype=torch.int32, device='cuda')
    bo = torch.tensor([0, n], dtype=torch.int32, device='cuda')
    mean = torch.tensor([-1.0])
    with torch.no_grad():
        idx, sl = ops.ballquery_batch_p(g1, bi, bo, 0.04, 300)
        ci, co = ops.bfs_cluster(mean, idx, sl, 100.0, 0)
        n_active, S = int(idx.numel()), int(ci.shape[0])
        t_bq = _events_ms(lambda: ops.ballquery_batch_p(g1, bi, bo, 0.04, 300))
        t_bfs = _events_ms(lambda: ops.bfs_cluster(mean, idx, sl, 100.0, 0))
    b_bq = n * 20 + n_active * 4
    b_bfs = n_active * 12 + n * 16 + S * 8
    legs['G1_grouping'] = {
        'points': n, 'neighbour_pairs': n_active, 'clusters': int(co.numel() - 1), 'cluster_points': S,
        'ball_query_ms': round(t_bq, 3), 'bfs_cluster_ms': round(t_bfs, 3),
        'roofline': [
            {'kernel': 'bq_* (hashed-grid ball query: grid build + count + fill)', 'bound': 'hbm',
             'achieved': round(b_bq / t_bq / 1e6, 1), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
             'frac': round(b_bq / t_bq / 1e6 / HBM_PEAK_GBPS, 4), 'algorithmic_bytes': b_bq},
            {'kernel': 'bfs_* (union-find labelling + ordered emission)', 'bound': 'hbm',
             'achieved': round(b_bfs / t_bfs / 1e6, 1), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
             'frac': round(b_bfs / t_bfs / 1e6 / HBM_PEAK_GBPS, 4), 'algorithmic_bytes': b_bfs,
             'note': 'the ordered emission replays the BFS level by level: bound by the number of '
                     'levels of the largest cluster (latency), not by bytes'}],
    }

    # ---- a scene twice as dense (300k points in the same room): giant clusters, 5x the proposal
    #      points -- the load real rooms with wall/floor-sized clusters put on the grouping head
    dx, dr, di = synthetic.scene_s2(seed=1, n=2 * args.points)
    dbatch = synthetic.make_batch(dx, dr, instance_labels=di)
    dbatch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in dbatch.items()}
    model.async_results = False
    with torch.no_grad():
        for _ in range(2):
            model(dbatch)
        ts = []
        for _ in range(7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model(dbatch)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        dense_ms = ts[len(ts) // 2]          # median, like the config legs
        dstages, dinfo = stage_times(model, dbatch, reps=5)
    model.async_results = True
    legs['dense_scene'] = {'ms_per_scan_unpipelined': round(dense_ms, 3),
                           'ms_per_scan_min_max': [round(ts[0], 3), round(ts[-1], 3)], 'stages_ms': dstages,
                           'scene': dinfo}

    # ---- host-to-device inclusive rate of the bench scan: raw points arrive in pinned host memory,
    #      the voxel index is built on the device (data side of the reference: collate_fn's CPU
    #      voxelization_idx + cuda_cast, data/custom.py:196-256, util/utils.py:157-173)
    from softgroup_amd.data import collate_device, make_item
    sample = make_item(xyz, rgb, 50, None, inst, 'synthetic_0000')
    with torch.no_grad():
        for _ in range(2):
            model(collate_device([sample])).resolve()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rets = [model(collate_device([sample])) for _ in range(10)]
        for r in rets:
            r.resolve()
        torch.cuda.synchronize()
        ms_h2d = (time.perf_counter() - t0) / 10 * 1e3
        del rets
        # where the extra time goes: the collate alone (host copies into pinned staging + async H2D + device
        # voxel index, synchronised), its host part alone (the staging copies)
        ts = []
        for _ in range(7):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            b_ = collate_device([sample])
            t2 = time.perf_counter()
            torch.cuda.synchronize()
            ts.append(((t2 - t1) * 1e3, (time.perf_counter() - t1) * 1e3))
            del b_
        collate_host_ms = sorted(t[0] for t in ts)[3]
        collate_ms = sorted(t[1] for t in ts)[3]
        # the same loop with the NEXT scan's collate on a loader thread and stream (data.prefetch_device: what
        # DataLoader workers do for the reference's test loop, tools/test.py:145)
        from softgroup_amd.data import prefetch_device
        for b_ in prefetch_device([[sample]] * 2):
            model(b_).resolve()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rets = [model(b_) for b_ in prefetch_device([[sample]] * 10)]
        for r in rets:
            r.resolve()
        torch.cuda.synchronize()
        ms_h2d_prefetch = (time.perf_counter() - t0) / 10 * 1e3
        del rets
        # ... and the headline's shape -- scans in flight on the worker streams -- fed from pinned HOST scenes by the
        # loader thread instead of from HBM-resident batches: what a serving loop over a DataLoader sees
        in_flight, _ = host_thread_plan(max(1, args.contexts), 1)
        n_loaders = int(os.environ.get('SG_BENCH_LOADERS', '1'))
        model.scan_contexts = in_flight
        try:
            for r in [model(b_) for b_ in prefetch_device([[sample]] * (2 * in_flight), depth=2, workers=n_loaders)]:
                r.resolve()
            torch.cuda.synchronize()
            n_fed = 160      # (the headline region's length: a 40-scan region carried ~0.4 ms/scan of pipeline fill)
            t0 = time.perf_counter()
            # results are taken and released as a serving loop would, at most 2 x in_flight scans behind the
            # submission (keeping all 160 alive put ~140 of them into pageable copies beyond SG_PINNED_RESULTS_MB
            # and made every scan of the region pay a 12 MB allocation + copy: 3.6-4.1 ms/scan until round 6)
            import collections
            pending = collections.deque()
            for b_ in prefetch_device([[sample]] * n_fed, depth=2, workers=n_loaders):
                pending.append(model(b_))
                if len(pending) > 2 * in_flight:
                    pending.popleft().resolve()
            while pending:
                pending.popleft().resolve()
            torch.cuda.synchronize()
            ms_fed = (time.perf_counter() - t0) / n_fed * 1e3
        finally:
            model.scan_contexts = 1
        # the host link of this box, for reading the figure: pinned -> device copy rate
        hbuf = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
        dbuf = torch.empty(64 << 20, dtype=torch.uint8, device='cuda')
        link_ms = _events_ms(lambda: dbuf.copy_(hbuf, non_blocking=True), reps=5, warm=1)
        scan_bytes = sum(v.numel() * v.element_size() for v in sample if isinstance(v, torch.Tensor))
        legs['with_h2d'] = {'ms_per_step_with_h2d': round(ms_h2d, 3),
                            'ms_per_step_with_h2d_next_scan_prefetched': round(ms_h2d_prefetch, 3),
                            'ms_per_scan_scans_in_flight_fed_from_pinned_host': round(ms_fed, 3),
                            'scans_in_flight': in_flight, 'loader_threads': n_loaders,
                            'collate_ms': round(collate_ms, 3), 'collate_host_part_ms': round(collate_host_ms, 3),
                            'host_bytes_per_scan': int(scan_bytes),
                            'pinned_h2d_GBps_on_this_box': round((64 << 20) / link_ms / 1e6, 2),
                            'note': 'one scan at a time; raw points pinned on the host -> async H2D -> '
                                    'device voxel index (ops.voxelization_idx CUDA path) -> forward_test'}

    # ---- S1 (BASELINE config 1): 20k-point cloud, 0.02 m voxels, backbone-only forward on the CPU
    #      (the oracle's C/OpenMP sparse conv = the "port" of the spconv CPU path), all cores and
    #      OMP_NUM_THREADS=1 (what tools/dist_test.sh:7 runs the reference with)
    if not args.no_cpu_baseline:
        import ctypes
        import oracle
        from oracle.model import OracleSoftGroup
        oracle.build()
        sx, sr = synthetic.scene_s1(seed=0)
        sbatch = synthetic.make_batch(sx, sr)
        cfg = dict(synthetic.SCANNET_MODEL_CFG, semantic_only=True)
        ora = OracleSoftGroup(model.state_dict(), cfg)
        # GPU first: the OpenMP team of the CPU runs keeps spinning on the host cores for a while
        sb = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in sbatch.items()}
        import softgroup_amd.spconv.pytorch as spconv
        with torch.no_grad():
            vf = ops.voxelization(torch.cat((sb['feats'], sb['coords_float']), 1), sb['p2v_map'])
            x = spconv.SparseConvTensor(vf, sb['voxel_coords'].int(), sb['spatial_shape'], 1)
            gpu_ms = _events_ms(lambda: model.forward_backbone(x, sb['v2p_map']))
        t0 = time.perf_counter()
        ora.point_wise(sbatch)
        all_cores = time.perf_counter() - t0
        one = None
        try:
            gomp = ctypes.CDLL('libgomp.so.1')
            gomp.omp_set_num_threads(1)
            t0 = time.perf_counter()
            ora.point_wise(sbatch)
            one = time.perf_counter() - t0
            gomp.omp_set_num_threads(os.cpu_count())
        except OSError:
            pass
        legs['S1_backbone'] = {'points': int(sx.shape[0]), 'voxels': int(sb['voxel_coords'].shape[0]),
                               'cpu_all_cores_s': round(all_cores, 3), 'cores': os.cpu_count(),
                               'cpu_1_thread_s': None if one is None else round(one, 3),
                               'gpu_ms': round(gpu_ms, 3), 'kind': 'port'}
    return legs


def reference_cpu_ops_leg(model, batch):
    """the reference's own CPU operators (voxelize_idx, bfs_cluster,
    build_and_export_octree -- oracle/_ref/sg_ref_ops.so, compiled from /root/reference by
    oracle/build_ref.py and shipped like our own .so) timed on this box's host on the bench scan:
    the voxel index of the whole scan (what the DataLoader workers run, data/custom.py:239), the BFS
    clustering of every grouped class on the neighbour lists the GPU ball query produced (what
    forward_grouping runs on the host per class, softgroup.py:459-461) and the octree export of the
    scan's shifted coordinates.  Single thread, like the reference.  None when the library is absent."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    try:
        import build_ref
        ref = build_ref.load()
    except Exception:          # noqa: BLE001  (checker infrastructure: absence is reported, not fatal)
        ref = None
    if ref is None:
        return None
    from softgroup_amd import ops
    rec = {'kind': 'reference', 'cores': 1}
    # ---- voxelize_idx over the scan's voxel coordinates (mode 4, as the dataset calls it)
    pts = (batch['coords_float'] * 50).long().cpu()
    coords = torch.cat([torch.zeros(pts.shape[0], 1, dtype=torch.long), pts - pts.min(0)[0]], 1).contiguous()
    oc, im, om = coords.new(), torch.IntTensor(coords.shape[0]).zero_(), torch.IntTensor()
    t0 = time.perf_counter()
    ref.voxelize_idx(coords, oc, im, om, 1, 4)
    rec['voxelize_idx_s'] = round(time.perf_counter() - t0, 4)
    rec['voxelize_idx_points'] = int(coords.shape[0])
    # ---- bfs_cluster per grouped class, on the GPU's neighbour lists
    g = model.grouping_cfg
    with torch.no_grad():
        b = batch
        feats = torch.cat((b['feats'], b['coords_float']), 1) if model.with_coords else b['feats']
        import softgroup_amd.spconv.pytorch as spconv
        vf = ops.voxelization(feats, b['p2v_map'])
        x = spconv.SparseConvTensor(vf, b['voxel_coords'].int(), b['spatial_shape'], b['batch_size'])
        sem, off, _ = model.forward_backbone(x, b['v2p_map'])
        scores = sem.softmax(-1)
        mean = torch.tensor(g['class_numpoint_mean'], dtype=torch.float32)
        bfs_s, edges, members, shifted = 0.0, 0, 0, None
        for cls in range(model.semantic_classes):
            if cls in g['ignore_classes']:
                continue
            obj = (scores[:, cls] > g['score_thr']).nonzero().view(-1)
            if obj.numel() < model.test_cfg['min_npoint']:
                continue
            c = (b['coords_float'][obj] + off[obj]).contiguous()
            shifted = c if shifted is None or c.shape[0] > shifted.shape[0] else shifted
            bi = torch.zeros(obj.numel(), dtype=torch.int32, device=c.device)
            bo = torch.tensor([0, obj.numel()], dtype=torch.int32, device=c.device)
            idx, sl = ops.ballquery_batch_p(c, bi, bo, g['radius'], g['mean_active'])
            idx_h, sl_h = idx.cpu(), sl.cpu()
            ci, co = torch.IntTensor(), torch.IntTensor()
            t0 = time.perf_counter()
            ref.bfs_cluster(mean, idx_h, sl_h, ci, co, sl_h.shape[0], float(g['npoint_thr']), cls)
            bfs_s += time.perf_counter() - t0
            edges += int(idx_h.numel())
            members += int(sl_h.shape[0])
    rec.update(bfs_cluster_s=round(bfs_s, 4), bfs_cluster_points=members, bfs_cluster_neighbour_pairs=edges)
    # ---- octree export (SoftGroup++ grouping, functions.py:29) of the largest class's shifted points
    if shifted is not None:
        p = shifted.cpu().float().contiguous()
        mx, mn = p.max(0)[0], p.min(0)[0]
        xyzwhl = torch.cat([(mx + mn) / 2, mx - mn]).contiguous()
        boxes, pt_inds = torch.zeros((585, 6)), torch.zeros(p.shape[0], dtype=torch.int32)
        psl = torch.zeros((512, 2), dtype=torch.int32)
        t0 = time.perf_counter()
        ref.build_and_export_octree(p, xyzwhl, boxes, pt_inds, psl, 3)
        rec.update(octree_build_s=round(time.perf_counter() - t0, 4), octree_points=int(p.shape[0]))
    return rec


def timed_steps(step, resolve, steps, sync_all):
    """the contract's timed region: EXACTLY `steps` steps between two barrier+synchronize brackets.
    Also returns the ms/step of each third of the region (host time stamps taken as results of
    the window's last step are resolved) -- the rate of a run with several scans in flight wanders
    between windows, the median says how representative the whole-region figure is."""
    sync_all()
    diag = [] if os.environ.get('SG_BENCH_DIAG') else None      # developer: completion time of every step
    if diag is not None and torch.cuda.is_available():           # ... and what the allocators did in the region
        d0 = (torch.cuda.memory_stats().get('num_device_alloc', 0), torch._C._cuda_hostMemoryStats()['num_host_alloc'])
    t0 = time.perf_counter()
    rets = [step() for _ in range(steps)]
    marks, cuts = [], [steps * (i + 1) // 3 for i in range(3)]
    if diag is not None:
        diag.append(round((time.perf_counter() - t0) * 1e3, 1))   # submission done
    for i in range(steps):
        # every result is fully materialised, checked and then RELEASED, as a consumer would: its
        # host arrays live in pinned staging blocks that the caching host allocator hands to a later
        # scan; holding all K results alive would make every scan of the region allocate (and
        # page-lock) a fresh 12 MB block
        rets[i] = resolve(rets[i])
        if diag is not None:
            diag.append(round((time.perf_counter() - t0) * 1e3, 1))
        if i + 1 in cuts:
            marks.append(time.perf_counter())
    sync_all()
    elapsed = time.perf_counter() - t0
    if diag is not None:
        sys.stderr.write(f'[bench diag] submitted at {diag[0]} ms, resolved at {diag[1:]}\n')
        if torch.cuda.is_available():
            from softgroup_amd.util.cast import pinned_result_bytes
            sys.stderr.write(f'[bench diag] in the region: device allocations '
                             f'{torch.cuda.memory_stats().get("num_device_alloc", 0) - d0[0]}, pinned host allocations '
                             f'{torch._C._cuda_hostMemoryStats()["num_host_alloc"] - d0[1]}, pinned result bytes alive '
                             f'{pinned_result_bytes()}\n')
    windows, prev_t, prev_n = [], t0, 0
    for m, c in zip(marks, cuts):
        if c > prev_n:
            windows.append((m - prev_t) / (c - prev_n) * 1e3)
        prev_t, prev_n = m, c
    return elapsed, rets, windows


def stub_main(args, rank, world, devices):
    """plumbing only (tests/test_bench_launch.py): same launch, barrier, MAX-over-ranks and JSON
    code as the real run, the scan replaced by a 2 ms host delay"""
    import torch.distributed as dist

    def sync_all():
        if world > 1:
            dist.barrier()

    if rank == args.stub_fail_rank:
        raise RuntimeError(f'rank {rank}: injected failure before the first barrier (--stub-fail-rank)')
    for _ in range(args.warmup):
        time.sleep(0.002)
    elapsed, _, windows = timed_steps(lambda: time.sleep(0.002), lambda r: None, args.steps, sync_all)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    out = {'metric': 'stub steps/s (plumbing test)', 'value': round(world * args.steps / elapsed, 3),
           'unit': 'steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
           'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True,
           'scaling': 'weak', 'vs_baseline': None, 'dtype': 'none', 'data': 'stub',
           'config': {'workload': 'stub'}, 'ranks_seen': world, 'devices': devices,
           'host_threads': host_thread_plan(args.contexts, world)[1],
           'rank_core_slices': [None] * world, 'ddp': None, 'legs': 'skipped (stub)'}
    if world > 1:
        del out['ddp']
        return finish_n_gt_1(out, args, rank, world, int(os.environ.get('LOCAL_RANK', '0')), True)
    if rank == 0:
        print(json.dumps(out))


def main():
    args = parse()
    if args.contexts <= 0:
        args.contexts = 10
    if args.switch_interval_us > 0:
        sys.setswitchinterval(args.switch_interval_us * 1e-6)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == max(args.gpus, 1), f'--gpus {args.gpus} but the launcher started {world} ranks'
    dist_on = world > 1
    import torch.distributed as dist
    if dist_on:
        # one process per GPU next to N-1 others on the same host: no intra-op thread pools
        # (a single torch CPU op otherwise wakes one thread per host core)
        torch.set_num_threads(1)
    if not args.stub:
        n_dev = torch.cuda.device_count()
        assert n_dev > local_rank, (f'rank {rank}: LOCAL_RANK {local_rank} but only {n_dev} GPU(s) visible -- '
                                    f'refusing to share a device between ranks')
        torch.cuda.set_device(local_rank)
    # every rank on its own slice of the host cores (its scan / results threads inherit it)
    args._core_slice = bind_rank_to_cores(local_rank, world)
    if dist_on:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(args.backend)       # 'nccl' is RCCL on ROCm
    # every rank's device: all distinct, or the figure would be N ranks time-sharing one GPU
    ident = device_identity(args.stub, rank)
    devices = [ident]
    if dist_on:
        devices = [None] * world
        dist.all_gather_object(devices, ident)
        assert len(set(devices)) == world, f'ranks share a device: {devices}'
    if args.stub:
        return stub_main(args, rank, world, devices)
    from softgroup_amd import _lib, synthetic
    from softgroup_amd.spconv import core as spcore
    assert os.path.exists(_lib.LIB_PATH), 'libsoftgroup_hip.so missing: run __graft_entry__.build()'

    # every rank gets its own scenes (different seeds per rank), weights identical on all ranks.  The
    # timed region cycles through `--scenes` DISTINCT pre-built scenes resident in HBM, as a dataloader
    # would hand them over (reference loop: tools/test.py:145-150); scene 0 is also the scene of the
    # cpu_baseline / parity_at_bench legs
    from softgroup_amd.util.digest import result_digest
    n_scenes = max(1, args.scenes)
    batches = []
    for i in range(n_scenes):
        xyz_i, rgb_i, inst_i = synthetic.scene_s2(seed=1 + rank + 97 * i, n=args.points)
        if i == 0:
            xyz, rgb, inst = xyz_i, rgb_i, inst_i
        b = synthetic.make_batch(xyz_i, rgb_i, instance_labels=inst_i, scan_id=f'synthetic_{rank:04d}_{i}')
        batches.append({k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()})
    batch = batches[0]
    model = synthetic.build_model(seed=0)

    def sync_all():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    # forward_test returns as soon as the scan's GPU work is enqueued; turning its results into host
    # objects (numpy arrays, RLE mask strings) runs on the model's results thread and overlaps the
    # next scan.  Every one of the K result dicts is fully materialised inside the timed region.
    contexts, host_threads = host_thread_plan(max(1, args.contexts), world)
    with torch.no_grad():
        # what every scene returns when it runs ALONE (one scan at a time, the reference's loop): the
        # digest covers semantic_preds, offset_preds and every instance's label / confidence / RLE string
        model.scan_contexts = 1
        alone = []
        for b in batches:
            r = dict(model(b))
            assert len(r['semantic_preds']) == args.points and 'pred_instances' in r
            alone.append(result_digest(r))
            del r
        model.scan_contexts = contexts
        # set-up of the scan pool (not a warm-up step of the workload): every worker thread creates its
        # stream, its executor / driver arenas (3.3 GB + 160 MB through hipMalloc) and its allocator
        # pools on its first scans -- two per worker, so that the `--warmup` steps and the timed region
        # meet workers that exist
        for _ in range(2):
            for r in [model(batches[i % n_scenes]) for i in range(contexts)]:
                r.resolve()
        # ... and the pinned staging blocks of the results (12 MB each, torch's caching host allocator hands them
        # out): as many as can be alive at once in the region -- the scans in flight plus finished results the
        # consumer has not released yet -- exist before it.  A page-locking allocation inside the region stalls the
        # streams for milliseconds, and a 20-step region saw 2-3 of them (profiles/r06_driver_shape.txt)
        held = [model(batches[i % n_scenes]) for i in range(2 * contexts + 4)]
        for r in held:
            r.resolve()
        del held, r
        # the digest of every result is computed and compared by the consumer inside the timed region
        # (SG_BENCH_DIGEST=worker: the digest computed by the scan worker through model.scan_result_hook instead
        # of by the consumer -- measured within the run-to-run noise of the consumer-side check at 20 steps and
        # worse at 160 steps with 5 workers, profiles/README.md)
        digest_on_worker = os.environ.get('SG_BENCH_DIGEST', 'main') == 'worker'
        if digest_on_worker and not os.environ.get('SG_BENCH_SKIP_DIGEST'):
            model.scan_result_hook = lambda res: res.__setitem__('result_digest', result_digest(res))
        # the interpreter's cyclic garbage collector pauses every thread of the process while it walks
        # the heap (scenes, modules, the oracle's tables: millions of objects by now): collect once and
        # park what is alive, so that no full collection falls into the timed region.  BEFORE the warm-up
        # steps (SG_BENCH_GC_AFTER_WARMUP=1: after them, as until round 5): the collection takes a few hundred
        # milliseconds of host time during which the GPU idles, and the timed region should follow the warm-up
        # steps directly, as the contract describes it
        import gc
        gc_after = os.environ.get('SG_BENCH_GC_AFTER_WARMUP') == '1'
        if not gc_after:
            gc.collect()
            gc.freeze()
        for r in [model(batches[i % n_scenes]) for i in range(max(args.warmup, 1))]:
            r.resolve()
        if gc_after:
            gc.collect()
            gc.freeze()
        issued = iter(range(args.steps))
        checked = iter(range(args.steps))

        def step():
            return model(batches[next(issued) % n_scenes])

        def consume(r):
            # (inside the timed region) the result is fully materialised and compared with the scene's
            # one-at-a-time digest: a scan in flight next to others must return the same bits
            r.resolve()
            if os.environ.get('SG_BENCH_SKIP_DIGEST'):      # developer diagnosis only: what the check costs
                return next(checked) >= 0
            return (r['result_digest'] if digest_on_worker else result_digest(r)) == alone[next(checked) % n_scenes]

        elapsed, rets, windows = timed_steps(step, consume, args.steps, sync_all)
    model.scan_result_hook = None
    identical = all(rets) and len(rets) == args.steps
    assert identical, f'results of the timed region differ from the one-at-a-time results: {rets}'
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
        'scaling': 'weak', 'vs_baseline': None,      # filled below from the single-scan latency
        'vs_baseline_throughput': round(value / world / (1000.0 / REF_MS_PER_SCAN), 3),
        'dtype': 'f32', 'data': 'synthetic',
        'config': {
            'workload': f'S2 synthetic room scene, {args.points} pts/scene, 0.02 m voxels, one scene '
                        f'per GPU; softgroup_scannet.yaml model section; random-init weights '
                        f'(30.84 M params), BN running stats randomised, semantic head last layer '
                        f're-drawn with std 20 so grouping/refinement run on a realistic load',
            'baseline_note': 'vs_baseline = 288 / latency_ms (one scan at a time, like the reference '
                             'figure); vs_baseline_throughput = per-GPU scans/s / (1000/288) with '
                             'scans_in_flight scans overlapping; 288 ms/scan is the reference README '
                             'number on 1x Titan X with real ScanNet v2 data',
            'parallelism': f'scenes sharded one per GPU x{world}, no data-path collective',
            'scans_in_flight': model.scan_contexts,
            'arithmetic': 'fp32 in, fp32 out, fp32 accumulation everywhere; the sparse-conv products run on '
                          'the bf16 matrix pipe as six v_mfma_f32_32x32x16_bf16 per 16-channel slice over '
                          '3-way split operands (x = h + m + l, bf16 each: 24+ significant bits; dropped '
                          'terms <= 2^-24 |ab|) -- measured against the fp32-MFMA kernel and the oracle '
                          'in parity_at_bench; SG_CONV_SPLIT=0 selects the fp32-MFMA kernel',
        },
        'ranks_seen': world, 'devices': devices, 'host_threads': host_threads,
        # every result of the timed region was compared (inside the region) with the digest of the same
        # scene run one scan at a time: semantic_preds, offset_preds, all instances' label/conf/RLE
        'timed_results_identical': bool(identical), 'distinct_scenes_in_timed_region': n_scenes,
        # ms/scan of each third of the timed region on rank 0, and their median
        'ms_per_step_windows': [round(w, 3) for w in windows],
        'ms_per_step_window_median': round(sorted(windows)[len(windows) // 2], 3) if windows else None,
    }
    model.scan_contexts = 1

    if rank == 0:
        # one scan at a time: (a) with the result formatting of scan i overlapping scan i+1 on the
        # results thread, (b) with everything in line -- the latency of a single scan
        with torch.no_grad():
            # (this thread's stream has not run a scan yet -- the timed region ran on the workers'
            # streams: arenas, per-stream executor state and the copy stream are created by two
            # untimed scans first)
            for _ in range(2):
                model(batch).resolve()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            rets = [model(batch) for _ in range(min(args.steps, 10))]
            for r in rets:
                r.resolve()
            torch.cuda.synchronize()
            out['ms_per_step_one_scan_at_a_time'] = round((time.perf_counter() - t1) / len(rets) * 1e3, 3)
            del rets
        model.async_results = False
        with torch.no_grad():
            model(batch)
            ts = []
            for _ in range(min(args.steps, 10)):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                model(batch)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t1) * 1e3)
            out['ms_per_step_unpipelined'] = round(sum(ts) / len(ts), 3)
            out['latency_ms_min_median_max'] = [round(min(ts), 3), round(sorted(ts)[len(ts) // 2], 3), round(max(ts), 3)]
        out['latency_ms'] = out['ms_per_step_unpipelined']        # one scan, everything in line
        out['vs_baseline'] = round(REF_MS_PER_SCAN / out['latency_ms'], 3)
        model.async_results = True
        stages, info = stage_times(model, batch)
        out['stages_ms'] = stages
        out['scene'] = info
        name = (_lib.C.c_char * 128)()
        cu, clk = _lib.C.c_int(0), _lib.C.c_int(0)
        _lib.lib().sg_device_info(name, 128, _lib.C.byref(cu), _lib.C.byref(clk))
        out['device'] = {'name': name.value.decode(), 'cus': cu.value, 'clock_khz': clk.value}

    if rank == 0 and world == 1 and not args.no_legs:
        # before anything touches the CPU oracle: its OpenMP team keeps spinning on the host cores
        # for a while and slows every host-side step of these legs (octree build, numpy) many-fold
        out['legs'] = config_legs(args)

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
        # one event pair per LAUNCH: a layer launched by itself (dims = M_out, K, Cin, Cout, input rows) or a
        # multi-layer launch of the deep levels / the tiny U-Net (conv_chain_kernel; dims[0] = -(conv layers
        # with K > 1 it carried), its 1x1 convs and skip concats are inside its time as well)
        import numpy as np
        d_ms = np.zeros(max(nl.value, 1), np.float32)
        d_dims = np.zeros((max(nl.value, 1), 5), np.int32)
        calls = _lib.C.c_int(0)
        _lib.check(lib.sg_spconv_profile_detail(d_ms.ctypes.data, d_dims.ctypes.data, int(nl.value), _lib.C.byref(calls)),
                   'sg_spconv_profile_detail')
        _lib.check(lib.sg_spconv_profile(0), 'sg_spconv_profile')
        chained = d_dims[:nl.value, 0] < 0
        layers = int(np.where(chained, -d_dims[:nl.value, 0], 1).sum())
        assert layers == s['launches'] * n_pass, (layers, nl.value, s['launches'])
        chain_info = {'launches_per_scan': int(chained.sum()) // n_pass,
                      'layers_per_scan': int(-d_dims[:nl.value, 0][chained].sum()) // n_pass,
                      'ms_per_scan': round(float(d_ms[:nl.value][chained].sum()) / n_pass, 3)}
        s = dict(launches=nl.value, ms=ms.value, bytes=s['bytes'] * n_pass, flops=s['flops'] * n_pass, layers=layers)
        gbps = s['bytes'] / (s['ms'] * 1e-3) / 1e9
        tflops = s['flops'] / (s['ms'] * 1e-3) / 1e12
        launches = max(s['launches'], 1)
        # HBM traffic of the same kernel from rocprofv3 PMC (FETCH_SIZE / WRITE_SIZE in KiB, separate
        # passes, collected by tools/profile_round.sh on this workload and committed as
        # profiles/rNN_conv_pmc.json, newest round first).  The FETCH_SIZE factor is the one stored
        # in the file: calibrated on this kernel's access pattern (profiles/r02_calib.txt, x1.97;
        # the guide's gfx950 correction is x2); null when no file is there.
        # NOT measured in this run (PMC needs rocprofv3 around the process): `traffic_source` says
        # which committed counter file the figure comes from and which FETCH_SIZE factor was applied.
        traffic, traffic_source = None, None
        here = os.path.dirname(os.path.abspath(__file__))
        for fn in ('r06_conv_pmc.json', 'r05_conv_pmc.json', 'r04_conv_pmc.json', 'r03_conv_pmc.json', 'r02_conv_pmc.json', 'r01_conv_pmc.json'):
            try:
                rec = json.load(open(os.path.join(here, 'profiles', fn)))
                pmc = rec['counters']
                factor = float(rec.get('fetch_size_factor', 2.0))
                traffic = int((factor * pmc['FETCH_SIZE']['per_dispatch'] +
                               pmc['WRITE_SIZE']['per_dispatch']) * 1024)
                traffic_source = (f'profiles/{fn} (separate rocprofv3 --pmc passes; FETCH_SIZE x {factor:g} + '
                                  f'WRITE_SIZE per dispatch of the conv kernel, over {rec["counters"]["FETCH_SIZE"]["dispatches"]} '
                                  f'dispatches; ' + rec.get('note', 'same workload') + ')')
                break
            except (OSError, ValueError, KeyError):
                continue
        hbm = {'achieved': round(gbps, 1), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
               'frac': round(gbps / HBM_PEAK_GBPS, 4)}
        mfma = {'achieved': round(tflops, 2), 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(tflops / FP32_MFMA_PEAK_TFLOPS, 4)}
        # Which roof binds: the kernel is priced against the pipe it ISSUES on.  With split-precision
        # products (the default) every algorithmic fp32 flop is six bf16 MFMA flops, so the matrix
        # roof is 2500 / 6 = 416.7 TFLOP/s-equivalent and its time floor (flops / that) is below the
        # HBM floor (B_gs / 8 TB/s): the binding roof is HBM.  With SG_CONV_SPLIT=0 the kernel issues
        # fp32 MFMAs and is priced against 157.3 TFLOP/s.  The fp32-MFMA-equivalent rate stays in the
        # line as `mfma_fp32_equivalent` (the series of rounds 1-3).
        split_on = os.environ.get('SG_CONV_SPLIT', '1') != '0'
        issued = ({'achieved': round(6 * tflops, 1), 'peak': BF16_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                   'frac': round(6 * tflops / BF16_MFMA_PEAK_TFLOPS, 4),
                   'note': 'six bf16 MFMA flops are issued per algorithmic fp32 flop'} if split_on else mfma)
        bound = 'mfma' if issued['frac'] >= hbm['frac'] else 'hbm'
        bound_rec = hbm if bound == 'hbm' else {k: issued[k] for k in ('achieved', 'peak', 'unit', 'frac')}
        out['roofline'] = {
            'kernel': 'gather_conv_persistent_kernel + conv_chain_kernel (the same layer body: SubM/strided/inverse '
                      'sparse conv, one launch per layer on the big levels, one launch per <= 22 layers on the deep '
                      'levels and the tiny U-Net; fp32 products as six bf16 MFMAs on split operands, fp32 accumulate; '
                      'bound = the larger of B_gs / 8 TB/s and issued MFMA flops / the peak of the pipe they issue on; '
                      'achieved = algorithmic bytes of ALL conv layers / summed duration of ALL conv launches)',
            'mfma_issued': issued,
            'mfma_fp32_equivalent': mfma,
            'bound': bound, **bound_rec,
            'traffic': traffic, 'traffic_source': traffic_source,
            'launches_per_scan': s['launches'] // n_pass,
            'layers_per_scan': s['layers'] // n_pass,
            'multi_layer_launches': chain_info,
            'kernel_ms_per_scan': round(s['ms'] / n_pass, 3),
            'avg_launch_us': round(s['ms'] * 1e3 / launches, 2),
            'algorithmic_bytes_per_launch': s['bytes'] // launches,
            'algorithmic_flops_per_launch': s['flops'] // launches,
            'algorithmic_bytes_per_scan': s['bytes'] // n_pass,
            'flops_per_scan': s['flops'] // n_pass,
            'hbm': hbm,
        }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle  # checker only: CPU restatement of the reference model
        from oracle import parity
        oracle.build()
        cpu_batch = synthetic.make_batch(xyz, rgb, instance_labels=inst)
        # one oracle forward of the SAME scene: timed as the CPU baseline, and its outputs are the
        # full-size parity check of the GPU path (stage-wise: floats <= 1e-4, proposals / instance
        # labels / RLE strings identical; end to end: instance drift)
        # `cores` = the threads really used: the port is run on ONE thread.  (Its sparse conv has an
        # OpenMP loop, but the scan's CPU time is in single-threaded parts -- hash-based rulebooks,
        # brute-force ball query and BFS as in the reference's CPU ops: round 4 reported 256 cores for a
        # forward that took 1.99 s on all cores and 1.90 s on one, legs.S1_backbone.)
        threads = 1
        try:
            ctypes.CDLL('libgomp.so.1').omp_set_num_threads(1)
        except OSError:
            threads = os.cpu_count()
        rep = parity.parity_report(model, cpu_batch, synthetic.SCANNET_MODEL_CFG, timed_path=True)
        cpu_s = rep.pop('oracle_forward_s')
        out['parity_at_bench'] = rep
        ref_ops = reference_cpu_ops_leg(model, batch)
        out['cpu_baseline'] = {
            'value': round(1.0 / cpu_s, 4), 'unit': 'scans/s', 'cores': threads,
            'kind': 'port',
            'sample': f'1 scan of the same S2 scene ({args.points} pts) on {threads} thread(s): C sparse conv, '
                      f'brute-force ball query + BFS like the reference CPU ops; '
                      f'{cpu_s:.1f} s.  reference_ops: the reference\'s OWN CPU ops (oracle/_ref/sg_ref_ops.so = '
                      f'softgroup/ops/src compiled unmodified, 1 thread) on the same scan'
                      + ('' if ref_ops else ' -- library not on this box, not timed'),
            'reference_ops': ref_ops,
        }

    if rank == 0 and world == 1 and not args.no_legs:
        out.setdefault('legs', {}).update(measurement_legs(args, model, batch, xyz, rgb, inst))

    if dist_on:
        # N > 1: the training side's per-step exchange on this node -- gradient all-reduce of the
        # trainable heads over the job's RCCL group -- and a short DDP training leg on every rank; then
        # the line and the common exit (the line does not depend on the leg finishing)
        model.scan_contexts = 1
        return finish_n_gt_1(out, args, rank, world, local_rank, False)
    if rank == 0:
        print(json.dumps(out))


if __name__ == '__main__':
    main()
