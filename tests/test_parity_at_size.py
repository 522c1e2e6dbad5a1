"""Parity at BASELINE.json's own sizes (the other model tests run 8-40 k points).

  * S2: the exact bench scene (150 000 points, softgroup_scannet.yaml, seed 1): every stage of the
    HIP forward against the CPU restatement of the reference model -- float stages <= 1e-4
    (tolerance of north_star), proposals (membership and order), proposal voxel index, instance
    labels and RLE strings identical -- plus the end-to-end drift of a pure GPU run against a pure
    oracle run.  bench.py prints the same report as "parity_at_bench".
  * G1: the grouping-head input of SURVEY 8(d) (40 blobs x 1000 pts + 10 000 noise points,
    r = 0.04: ~6.9 M neighbour pairs): ball-query CSR and BFS clusters bit-exact.
"""
import numpy as np
import pytest
import torch

import oracle
from oracle import parity
from softgroup_amd import ops, synthetic

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('seed', [1, 7])      # 1 = the bench scene of rank 0, 7 = rank 6's scene at N=8
def test_s2_150k_full_model_parity(seed):
    xyz, rgb, inst = synthetic.scene_s2(seed=seed, n=150000)
    batch = synthetic.make_batch(xyz, rgb, instance_labels=inst)
    model = synthetic.build_model(seed=0)
    rep = parity.parity_report(model, batch, synthetic.SCANNET_MODEL_CFG)
    print(rep)
    assert rep['points'] == 150000 and rep['proposal_points'] > 10000 and rep['instances'] > 0
    assert rep['max_abs_feat'] <= 1e-3 and rep['float_stages_within_tol'], rep
    assert rep['proposals_equal'] and rep['proposal_voxel_index_equal'] and rep['instances_equal'], rep
    # end to end: threshold flips may move single points; the instances must still be the same objects
    assert rep['e2e_instances_oracle'] > 0
    assert abs(rep['e2e_instances_gpu'] - rep['e2e_instances_oracle']) <= max(2, rep['e2e_instances_oracle'] // 50), rep
    assert rep['e2e_mean_best_mask_iou'] >= 0.98, rep


def test_g1_grouping_input_bit_exact():
    xyz = synthetic.scene_g1(seed=2)
    n = len(xyz)
    assert n == 50000
    bi = np.zeros(n, np.int32)
    bo = np.array([0, n], np.int32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    idx, sl = ops.ballquery_batch_p(t(xyz), t(bi), t(bo), 0.04, 300)
    ridx, rsl = oracle.ballquery_batch_p(xyz, bi, bo, 0.04, 300)
    assert 6_000_000 < len(ridx) < 8_000_000          # nActive = 6 922 156
    assert np.array_equal(sl.cpu().numpy(), rsl) and np.array_equal(idx.cpu().numpy(), ridx)
    mean = torch.tensor([-1.0])
    ci, co = ops.bfs_cluster(mean, idx, sl, 100.0, 0)
    rci, rco = oracle.bfs_cluster(mean.numpy(), ridx, rsl, 100.0, 0)
    assert len(rco) - 1 >= 38                          # 40 components >= 100 pts
    assert np.array_equal(co.cpu().numpy(), rco) and np.array_equal(ci.cpu().numpy(), rci)


def _assert_report(rep, min_props):
    print(rep)
    assert rep['proposals'] >= min_props and rep['instances'] > 0
    assert rep['float_stages_within_tol'], rep
    assert rep['proposals_equal'] and rep['proposal_voxel_index_equal'] and rep['instances_equal'], rep
    assert rep['e2e_instances_oracle'] > 0
    assert abs(rep['e2e_instances_gpu'] - rep['e2e_instances_oracle']) <= max(2, rep['e2e_instances_oracle'] // 50), rep
    assert rep['e2e_mean_best_mask_iou'] >= 0.98, rep


def test_config4_stpls3d_pp_at_size():
    """BASELINE config 4 at SURVEY 8(d)'s size: softgroup++_stpls3d.yaml (channels 16, 0.33 m voxels,
    octree ball query, pyramid levels with the model's own get_level -- the 149 k-point class takes
    level 2 naturally) on the S2 coordinates x 40 (a ~240 m tile, 150 000 points).  Same stage-wise
    bar as the S2 test: floats <= 1e-4, proposals / proposal voxel index / instances + RLE
    identical."""
    import copy
    xyz, rgb, inst = synthetic.scene_s2(seed=1, n=150000)
    xyz = (xyz * np.float32(40)).astype(np.float32)
    batch = synthetic.make_batch(xyz, rgb, scale=3, instance_labels=inst)
    cfg = copy.deepcopy(synthetic.STPLS3D_PP_MODEL_CFG)
    model = synthetic.build_model(cfg, seed=0)
    rep = parity.parity_report(model, batch, cfg)
    assert rep['points'] == 150000
    _assert_report(rep, min_props=1000)


def test_config5_kitti_sweep_at_size():
    """BASELINE config 5 at SURVEY 8(d)'s size: softgroup_kitti.yaml (1 input channel, no coords,
    5 cm voxels, radius 0.1, npoint_thr 5 absolute, panoptic) on one ~120 k-point LiDAR-like sweep
    (64 rings against a ground plane and car-sized boxes).  Stage-wise bar as above, plus the
    panoptic fusion of the GPU's instances == the oracle's fusion of the same instances."""
    import copy
    from oracle.model import OracleSoftGroup
    xyz, intensity, inst = synthetic.scene_lidar(seed=3, n=120000)
    assert 119000 <= xyz.shape[0] <= 121000
    batch = synthetic.make_batch(xyz, intensity, scale=20, instance_labels=inst)
    cfg = copy.deepcopy(synthetic.KITTI_MODEL_CFG)
    model = synthetic.build_model(cfg, seed=0)
    rep = parity.parity_report(model, batch, cfg)
    _assert_report(rep, min_props=100)
    with torch.no_grad():
        full = model(batch)
    assert full['panoptic_preds'].dtype == np.uint32 and full['panoptic_preds'].shape[0] == xyz.shape[0]
    g = parity.gpu_stages(model, batch)
    sem_pred = g['sem'].argmax(1).cpu().numpy()
    ora = OracleSoftGroup(model.state_dict(), cfg)
    assert np.array_equal(model.panoptic_fusion(sem_pred, g['preds']), ora.panoptic_fusion(sem_pred, g['preds']))
