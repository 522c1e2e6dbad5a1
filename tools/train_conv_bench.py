"""Training-side sparse conv on the S2 scene's U-Net levels: forward / input gradient in fp32 and in
bf16 (sg_spconv_gather_conv_bf16), weight gradient (sg_spconv_wgrad) with fp32 and bf16 operands.
Per layer shape: time (HIP events, median of 20), TFLOP/s on algorithmic flops 2*P*Cin*Cout and GB/s on
the gather/scatter bytes B_gs = P*Cin*s + M*Cout*s + 8*P (s = element size).
Usage (GPU box): python tools/train_conv_bench.py [points]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softgroup_amd import synthetic  # noqa: E402
from softgroup_amd.spconv import core  # noqa: E402


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
    xyz, rgb, inst = synthetic.scene_s2(seed=1, n=n)
    batch = synthetic.make_batch(xyz, rgb, instance_labels=inst)
    idx = batch['voxel_coords'].int().cuda()
    shape = list(batch['spatial_shape'])
    print(f'{"level":>5} {"M":>7} {"P":>8} {"C":>4} | {"fwd32 us":>9} {"TF/s":>6} | {"fwd16 us":>9} {"TF/s":>6} '
          f'{"GB/s":>6} | {"wgrad32":>8} {"TF/s":>6} | {"wgrad16":>8} {"TF/s":>6}')
    tot = dict(f32=0.0, f16=0.0, w32=0.0, w16=0.0)
    for lvl, C in enumerate([32, 64, 96, 128, 160, 192, 224]):
        rule = core.SubMRule(idx, shape)
        plan = rule.plan
        M = plan.num_out
        P = int((plan.nbr >= 0).sum())
        torch.manual_seed(lvl)
        w = torch.randn(C, 27, C, device='cuda') * 0.05
        x = torch.randn(M, C, device='cuda')
        g = torch.randn(M, C, device='cuda')
        xb, gb = x.bfloat16(), g.bfloat16()
        w32 = core.pack_weight(w, C, 27, C, False)
        w16 = core.pack_weight_bf16(w, C, 27, C, False)
        t32 = timed(lambda: core.gather_conv(x, plan, w32, C))
        t16 = timed(lambda: core.gather_conv_bf16(xb, plan, w16, C))
        tw32 = timed(lambda: core.conv_wgrad(x, g, plan, C, C))
        tw16 = timed(lambda: core.conv_wgrad(xb, gb, plan, C, C))
        fl = 2.0 * P * C * C
        b16 = P * C * 2 + M * C * 2 + 8 * P
        print(f'{lvl:>5} {M:>7} {P:>8} {C:>4} | {t32 * 1e3:>9.1f} {fl / t32 / 1e9:>6.1f} | {t16 * 1e3:>9.1f} '
              f'{fl / t16 / 1e9:>6.1f} {b16 / t16 / 1e6:>6.0f} | {tw32 * 1e3:>8.1f} {fl / tw32 / 1e9:>6.1f} | '
              f'{tw16 * 1e3:>8.1f} {fl / tw16 / 1e9:>6.1f}')
        for k, v in zip(tot, (t32, t16, tw32, tw16)):
            tot[k] += v
        if lvl < 6:
            d = core.DownRule(idx, shape, 1)
            idx, shape = d.out_indices, d.out_spatial_shape
    print('sum over the 7 levels (one SubM layer each), ms:', {k: round(v, 3) for k, v in tot.items()})


if __name__ == '__main__':
    main()
