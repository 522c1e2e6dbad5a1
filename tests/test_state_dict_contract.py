"""Checkpoint contract: our SoftGroup must expose exactly the parameter/buffer names and shapes
of the REFERENCE's own SoftGroup class (recorded by tests/golden/make_golden.py by instantiating
/root/reference/softgroup/model/softgroup.py on top of our spconv/ops shims), and -- since the
construction order and init scheme are the same -- the same values under the same seed."""
import json
import os

import torch

from softgroup_amd.model import SoftGroup

HERE = os.path.dirname(os.path.abspath(__file__))


def test_state_dict_matches_reference_class():
    contract = json.load(open(os.path.join(HERE, 'golden', 'state_dict_contract.json')))
    assert set(contract) == {'scannet', 'semantic_only_kitti_like', 'stpls3d_like'}
    for name, rec in contract.items():
        torch.manual_seed(0)
        model = SoftGroup(**rec['cfg'])
        sd = model.state_dict()
        assert [[k, list(v.shape)] for k, v in sd.items()] == rec['keys'], name
        checksum = float(sum(v.double().abs().sum() for v in sd.values()))
        assert abs(checksum - rec['checksum']) <= 1e-6 * max(1.0, abs(rec['checksum'])), name


def test_reference_style_checkpoint_roundtrip(tmp_path):
    """ckpt['net'] written the reference's way (util/utils.py:88-108) loads strictly"""
    contract = json.load(open(os.path.join(HERE, 'golden', 'state_dict_contract.json')))
    cfg = contract['scannet']['cfg']
    a, b = SoftGroup(**cfg), SoftGroup(**cfg)
    path = tmp_path / 'epoch_1.pth'
    torch.save({'net': {k: v.cpu() for k, v in a.state_dict().items()}, 'epoch': 1}, path)
    missing, unexpected = b.load_state_dict(torch.load(path)['net'], strict=True)
    assert not missing and not unexpected
    assert all(torch.equal(x, y) for x, y in zip(a.state_dict().values(), b.state_dict().values()))
    assert [n for n, p in b.named_parameters() if not p.requires_grad][:1] == ['input_conv.0.weight']
