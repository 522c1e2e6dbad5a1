"""softgroup_amd -- MI355X-native SoftGroup hot path (sparse 3D U-Net backbone + grouping head).

``softgroup_amd.ops``     mirrors the reference's ``softgroup.ops`` function surface
``softgroup_amd.spconv``  mirrors the subset of ``spconv.pytorch`` the reference model uses
``softgroup_amd.model``   the reference-shaped ``SoftGroup`` nn.Module hosting both
All compute goes through the C ABI of ``lib/libsoftgroup_hip.so`` (include/softgroup_hip.h).
"""
__version__ = '0.1.0'
