"""Developer experiment: is the scans-in-flight mode bound by the GPU or by one process's host side?
Runs P processes side by side on the same GPU, each with C scans in flight; prints combined ms/scan.
Usage: python tools/two_procs.py P C"""
import os
import subprocess
import sys
import time

WORKER = r'''
import sys, time, torch
sys.path.insert(0, %r)
from softgroup_amd import synthetic
C, N = int(sys.argv[1]), int(sys.argv[2])
model = synthetic.build_model(seed=0)
model.scan_contexts = C
xyz, rgb, inst = synthetic.scene_s2(seed=1, n=150000)
b = synthetic.make_batch(xyz, rgb, instance_labels=inst)
b = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
with torch.no_grad():
    for r in [model(b) for _ in range(8)]: r.resolve()
    torch.cuda.synchronize()
    print('READY', flush=True)
    sys.stdin.readline()
    t0 = time.perf_counter()
    rs = [model(b) for _ in range(N)]
    for r in rs: r.resolve()
    torch.cuda.synchronize()
    print('DONE', time.perf_counter() - t0, flush=True)
'''


def main():
    P, C = int(sys.argv[1]), int(sys.argv[2])
    N = 120
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = [subprocess.Popen([sys.executable, '-c', WORKER % root, str(C), str(N)], stdin=subprocess.PIPE,
                              stdout=subprocess.PIPE, text=True) for _ in range(P)]
    for p in procs:
        while 'READY' not in p.stdout.readline():
            pass
    t0 = time.perf_counter()
    for p in procs:
        p.stdin.write('go\n')
        p.stdin.flush()
    ts = []
    for p in procs:
        line = p.stdout.readline()
        ts.append(float(line.split()[1]))
    wall = time.perf_counter() - t0
    print(f'{P} process(es) x {C} scans in flight: per-process {[round(t / N * 1e3, 2) for t in ts]} ms/scan, '
          f'combined {wall / (P * N) * 1e3:.2f} ms/scan')
    for p in procs:
        p.wait()


if __name__ == '__main__':
    main()
