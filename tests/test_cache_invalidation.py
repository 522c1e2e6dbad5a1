"""Derived-tensor caches (packed conv weights, BatchNorm affines, executor descriptors):
invalidation hooks and copy/pickle safety.  CPU only: the caches are exercised
through their keys, no kernel runs."""
import copy
import pickle

import torch

from softgroup_amd import synthetic
from softgroup_amd.model import SoftGroup
from softgroup_amd.spconv import core


def _model():
    return SoftGroup(**synthetic.SCANNET_MODEL_CFG)


def test_data_writes_need_and_get_explicit_invalidation():
    m = _model().eval()
    bn = m.output_layer[0]
    s0, b0 = core._bn_affine(bn)
    bn.running_mean.data.add_(1.0)                    # .data write: version counter unchanged
    s1, b1 = core._bn_affine(bn)
    assert b1 is b0, 'a .data write is invisible to the version-keyed cache (documented)'
    m.invalidate_caches()
    s2, b2 = core._bn_affine(bn)
    assert not torch.equal(b2, b0)
    bn.weight.data.mul_(2.0)
    core.invalidate_caches()                          # module-level spelling
    assert not torch.equal(core._bn_affine(bn)[0], s2)


def test_train_eval_load_and_apply_invalidate():
    m = _model().eval()
    bn = m.output_layer[0]
    e0 = core.cache_epoch()
    m.train()
    m.eval()
    assert core.cache_epoch() > e0
    e1 = core.cache_epoch()
    m.load_state_dict(m.state_dict())
    assert core.cache_epoch() > e1
    e2 = core.cache_epoch()
    m.float()
    assert core.cache_epoch() > e2
    b = core._bn_affine(bn)[1]
    bn.running_mean.data.add_(1.0)
    m.train(False)                                    # back to eval after a (frozen-BN) train phase
    assert not torch.equal(core._bn_affine(bn)[1], b)


def test_model_with_runtime_state_is_deepcopyable_and_picklable():
    m = _model().eval()
    # what a forward leaves behind (executor with ctypes structures, a stream, device constants)
    import ctypes
    class Holder:                                      # noqa: E306
        def __init__(self):
            self.p = ctypes.pointer(ctypes.c_int(3))   # "ctypes objects containing pointers cannot be pickled"
    m.__dict__['_backbone_exec'] = Holder()
    m.__dict__['_tiny_exec'] = Holder()
    m.__dict__['_grouping_const'] = {'k': Holder()}
    m.input_conv[0]._kio_cache = ('key', Holder())
    c = copy.deepcopy(m)
    assert '_backbone_exec' not in c.__dict__ and c.input_conv[0]._kio_cache is None
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), c.state_dict().values()))
    d = pickle.loads(pickle.dumps(m))
    assert '_tiny_exec' not in d.__dict__
    assert '_backbone_exec' in m.__dict__              # the original keeps its runtime state


def test_executor_declines_channel_counts_that_are_not_multiples_of_4():
    from softgroup_amd.spconv.unet_exec import UNetExecutor
    ok = SoftGroup(**dict(synthetic.SCANNET_MODEL_CFG, channels=16))
    assert UNetExecutor(ok.unet, ok.input_conv, ok.output_layer)._supported()
    odd = SoftGroup(**dict(synthetic.SCANNET_MODEL_CFG, channels=6))
    assert not UNetExecutor(odd.unet, odd.input_conv, odd.output_layer)._supported()


def test_executor_key_sees_a_submodule_apply_that_moves_storage():
    """`param.data = x` (nn.Module._apply on a SUBMODULE: .double().float(),
    .cuda()) keeps the Parameter's id and version counter; the storage address in the key catches it."""
    from softgroup_amd.spconv.unet_exec import UNetExecutor
    m = _model().eval()
    ex = UNetExecutor(m.unet, m.input_conv, m.output_layer)
    k0, ts0 = ex._state_key()
    k1, _ = ex._state_key()
    assert k0 == k1
    e0 = core.cache_epoch()
    bn = m.output_layer[0]
    ids, vers = id(bn.weight), bn.weight._version
    bn.double().float()                                # submodule only: SoftGroup._apply is not involved
    assert core.cache_epoch() == e0 and id(bn.weight) == ids and bn.weight._version == vers
    k2, _ = ex._state_key()
    assert k2 != k0, 'a storage move under an unchanged Parameter object must change the key'
    del ts0
