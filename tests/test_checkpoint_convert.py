"""HAIS / spconv-1 checkpoint conversion (softgroup_amd.util.checkpoint, tools/convert_checkpoint.py)
against the reference's own script run as a subprocess on the same file (when /root/reference is
present), and as a round trip: SoftGroup weights written the spconv-1 way load back bit-identical."""
import os
import subprocess
import sys

import pytest
import torch

from softgroup_amd import synthetic
from softgroup_amd.model import SoftGroup
from softgroup_amd.util.checkpoint import convert_checkpoint_file, convert_spconv1_state_dict

REF_TOOL = '/root/reference/tools/convert_checkpoint.py'


def _spconv1_style(sd):
    """what a HAIS checkpoint looks like: KKKIO conv weights, the old module names"""
    out = {}
    for k, v in sd.items():
        if 'weight' in k and v.dim() == 5:
            v = v.permute(1, 2, 3, 4, 0).contiguous()
        k = k.replace('tiny_unet_outputlayer', 'intra_ins_outputlayer').replace('tiny_unet', 'intra_ins_unet')
        k = k.replace('iou_score_linear', 'score_linear')
        out[k] = v
    return out


def test_round_trip_loads_strictly():
    torch.manual_seed(0)
    model = SoftGroup(**synthetic.SCANNET_MODEL_CFG)
    sd = model.state_dict()
    old = _spconv1_style(sd)
    assert any('intra_ins_unet' in k for k in old) and 'score_linear.weight' in old
    new = convert_spconv1_state_dict(old)
    assert list(new) == list(sd)
    assert all(torch.equal(new[k], sd[k]) for k in sd)
    other = SoftGroup(**synthetic.SCANNET_MODEL_CFG)
    missing, unexpected = other.load_state_dict(new, strict=True)
    assert not missing and not unexpected


@pytest.mark.skipif(not os.path.exists(REF_TOOL), reason='/root/reference not present')
def test_same_file_as_the_reference_script(tmp_path):
    torch.manual_seed(1)
    sd = SoftGroup(**dict(synthetic.SCANNET_MODEL_CFG, channels=16)).state_dict()
    ckpt = {'net': _spconv1_style(sd), 'epoch': 7}
    a, b = tmp_path / 'a.pth', tmp_path / 'b.pth'
    torch.save(ckpt, a)
    torch.save(ckpt, b)
    subprocess.check_call([sys.executable, REF_TOOL, str(a)])
    ours = convert_checkpoint_file(str(b))
    ref = torch.load(str(a).replace('.pth', '_spconv2.pth'))
    got = torch.load(ours)
    assert got['epoch'] == ref['epoch'] == 7 and list(got['net']) == list(ref['net'])
    assert all(torch.equal(got['net'][k], ref['net'][k]) for k in ref['net'])
