import sys, time, torch
sys.path.insert(0, '/root/repo')
from softgroup_amd import synthetic
model = synthetic.build_model(seed=0)
model.scan_contexts = 4
with torch.no_grad():
    for seed in range(1, 9):
        xyz, rgb, inst = synthetic.scene_s2(seed=seed, n=150000)
        b = synthetic.make_batch(xyz, rgb, instance_labels=inst, scan_id=f's{seed}')
        b = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
        rs = [model(b) for _ in range(6)]
        for r in rs: r.resolve()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rs = [model(b) for _ in range(20)]
        for r in rs: r.resolve()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20 * 1e3
        n = [len(r['pred_instances']) for r in rs]
        assert len(set(n)) == 1, n
        print(f'seed {seed}: voxels {b["voxel_coords"].shape[0]} instances {n[0]} {dt:.2f} ms/scan', flush=True)
