"""Developer diagnostic (GPU): how much class signal do the backbone features of the synthetic
class-coloured scenes carry, with and without BatchNorm calibration?  Prints feature statistics,
the linear R^2 of colour / position from the features and the accuracy of three read-outs."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softgroup_amd import synthetic  # noqa: E402
from softgroup_amd import ops  # noqa: E402
import softgroup_amd.spconv.pytorch as spconv  # noqa: E402


def feats_of(model, b):
    with torch.no_grad():
        f = torch.cat((b['feats'], b['coords_float']), 1)
        x = spconv.SparseConvTensor(ops.voxelization(f, b['p2v_map']), b['voxel_coords'].int(),
                                    b['spatial_shape'], b['batch_size'])
        return model.forward_backbone(x, b['v2p_map'])[2]


def r2(a, t):
    a1 = torch.cat([a, torch.ones_like(a[:, :1])], 1).double()
    w = torch.linalg.lstsq(a1, t.double()).solution
    r = t.double() - a1 @ w
    return (1 - r.var(0) / t.double().var(0)).cpu().numpy().round(3)


def main():
    batches = []
    for i in range(4):
        xyz, rgb, inst = synthetic.scene_s2(seed=100 + i, n=24000, room_scale=0.4, class_colour=True)
        b = synthetic.make_batch(xyz, rgb, instance_labels=inst)
        batches.append({k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()})
    for calibrated in (False, True):
        model = synthetic.build_model(seed=0, head_std=None)
        if calibrated:
            info = synthetic.fit_model_to_scenes(model, batches)
            print('fit_model_to_scenes ->', info)
        f = torch.cat([feats_of(model, b) for b in batches])
        sem = torch.cat([b['semantic_labels'] for b in batches]).long()
        rgb = torch.cat([b['feats'] for b in batches])
        xyz = torch.cat([b['coords_float'] for b in batches])
        print(f'calibrated={calibrated}: feats finite {torch.isfinite(f).all().item()} mean|f| {f.abs().mean():.4f} '
              f'std {f.std(0).mean():.4f} frac>0 {(f > 0).float().mean():.3f} dead channels {(f.std(0) < 1e-6).sum().item()}')
        print('  R2 rgb', r2(f, rgb), 'R2 xyz', r2(f, xyz))
        bn = model.output_layer[0]
        print('  output_layer BN running_mean[:4]', bn.running_mean[:4].tolist(), 'var[:4]', bn.running_var[:4].tolist())
        # read-outs: multinomial logistic regression on standardised feats; 2-layer MLP
        X = (f - f.mean(0)) / (f.std(0) + 1e-6)
        for name, net in (('logreg', torch.nn.Linear(32, 20)),
                          ('mlp', torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.ReLU(), torch.nn.Linear(64, 20)))):
            net = net.cuda()
            opt = torch.optim.Adam(net.parameters(), lr=0.02)
            for _ in range(400):
                loss = torch.nn.functional.cross_entropy(net(X), sem)
                opt.zero_grad()
                loss.backward()
                opt.step()
            print(f'  {name}: acc {(net(X).argmax(1) == sem).float().mean():.4f} loss {loss.item():.4f}')


if __name__ == '__main__':
    main()
