"""Checkpoint key/layout conversion for HAIS / spconv-1 era weights.

The reference ships ``tools/convert_checkpoint.py:1-29`` for this: spconv 1 stores conv weights
as [kD, kH, kW, Cin, Cout] ("KKKIO"), spconv 2 -- the layout our layers keep as the parameter --
as [Cout, kD, kH, kW, Cin] ("OKKKI"), and three HAIS module names were renamed in SoftGroup
(``intra_ins_unet`` -> ``tiny_unet``, ``intra_ins_outputlayer`` -> ``tiny_unet_outputlayer``,
``score_linear`` -> ``iou_score_linear``)."""
from collections import OrderedDict

_RENAMES = (('intra_ins_unet', 'tiny_unet'), ('score_linear', 'iou_score_linear'),
            ('intra_ins_outputlayer', 'tiny_unet_outputlayer'))


def convert_spconv1_state_dict(state_dict):
    """spconv-1 / HAIS ``net`` state dict -> the key names and OKKKI conv layout SoftGroup loads.
    Same precedence as the reference script: the first matching rename wins (its if/elif chain)."""
    out = OrderedDict()
    for key, value in state_dict.items():
        if 'weight' in key and value.dim() == 5:
            value = value.permute(4, 0, 1, 2, 3)          # KKKIO -> OKKKI
        for old, new in _RENAMES:
            if old in key:
                key = key.replace(old, new)
                break
        out[key] = value
    return out


def convert_checkpoint_file(path, out_path=None):
    """ckpt['net'] converted in place of the reference CLI: writes <name>_spconv2.pth"""
    import torch
    ckpt = torch.load(path, map_location='cpu')
    ckpt['net'] = convert_spconv1_state_dict(ckpt['net'])
    out_path = out_path or path.replace('.pth', '_spconv2.pth')
    torch.save(ckpt, out_path)
    return out_path
