"""CPU tests of the plug-in point itself: `install_into(pyHGT.conv)` makes the reference's own
GeneralConv / model.GNN build this implementation's layers, whole-module pickles (the OAG scripts `torch.save(model)`,
/root/reference/OAG/train_paper_field.py:279) survive in both directions, and the parameter caches are invalidated by
everything that can change a weight.  No compute calls (there is no GPU here)."""
import io
import pickle

import pytest
import torch

import pyhgt_amd
from pyhgt_amd import conv as C
from oracle.reference_loader import reference_available, load_reference_conv, load_reference_model

needs_ref = pytest.mark.skipif(not reference_available(), reason="reference tree not present (GPU box)")

RUNTIME_KEYS = ("keep_att", "precision", "kernel_flags", "att", "_packed", "_packed_key", "_prepared", "_prepared_tag",
                "_prepared_valid")


class _Installed:
    """install_into() for the duration of a test; the reference module is put back afterwards (other tests compare against
    the reference's own classes)."""

    def __enter__(self):
        self.mod = load_reference_conv()
        self.saved = (self.mod.HGTConv, self.mod.DenseHGTConv)
        pyhgt_amd.install_into(self.mod)
        return self.mod

    def __exit__(self, *exc):
        self.mod.HGTConv, self.mod.DenseHGTConv = self.saved


@needs_ref
def test_install_into_makes_the_reference_gnn_build_our_layers():
    ref_model = load_reference_model()
    torch.manual_seed(0)
    pristine = ref_model.GNN(conv_name='hgt', in_dim=33, n_hid=64, n_heads=4, n_layers=2, dropout=0.2, num_types=3,
                             num_relations=5)
    names = {k: tuple(v.shape) for k, v in pristine.state_dict().items()}
    with _Installed() as mod:
        assert mod.HGTConv is C.HGTConv and mod.DenseHGTConv is C.DenseHGTConv
        torch.manual_seed(0)
        gnn = ref_model.GNN(conv_name='hgt', in_dim=33, n_hid=64, n_heads=4, n_layers=2, dropout=0.2, num_types=3,
                            num_relations=5)                            # model.py unchanged, model.py:65-67
        assert all(type(gc.base_conv) is C.HGTConv for gc in gnn.gcs)
        assert {k: tuple(v.shape) for k, v in gnn.state_dict().items()} == names
        gnn.load_state_dict(pristine.state_dict())                     # reference checkpoints load (train_ogbn_mag.py:198)
        # the positional construction of GeneralConv (conv.py:308) and the dense variant
        gd = mod.GeneralConv('dense_hgt', 64, 64, 3, 5, 4, 0.2, True, True)
        assert type(gd.base_conv) is C.DenseHGTConv
        # whole-module pickle round trip (train_paper_field.py:279 / :286)
        buf = io.BytesIO()
        torch.save(gnn, buf)
        buf.seek(0)
        back = torch.load(buf, weights_only=False)
        assert all(type(gc.base_conv) is C.HGTConv for gc in back.gcs)
        for (k, a), (k2, b) in zip(gnn.state_dict().items(), back.state_dict().items()):
            assert k == k2 and torch.equal(a, b)
        for gc in back.gcs:
            assert gc.base_conv._packed is None and gc.base_conv.precision == C.DEFAULT_PRECISION == "f16x3"


@needs_ref
def test_modules_pickled_by_the_reference_class_load_into_ours():
    """A checkpoint written by the REFERENCE (its HGTConv has none of our runtime attributes) must unpickle into a working
    pyhgt_amd.HGTConv once pyHGT.conv.HGTConv resolves to it."""
    ref_model = load_reference_model()
    torch.manual_seed(1)
    theirs = ref_model.GNN(conv_name='hgt', in_dim=20, n_hid=32, n_heads=2, n_layers=2, dropout=0.2, num_types=2,
                           num_relations=3)
    blob = pickle.dumps(theirs)                                         # refers to pyHGT.conv.HGTConv by name
    assert b"pyhgt_amd" not in blob
    with _Installed():
        ours = pickle.loads(blob)
    for gc in ours.gcs:
        layer = gc.base_conv
        assert type(layer) is C.HGTConv
        for k in RUNTIME_KEYS:
            assert k in layer.__dict__, k
        assert layer.precision == C.DEFAULT_PRECISION and layer.keep_att is False and layer.d_k == 16
        layer._pack_parameters()                                        # every attribute forward() reads exists
        assert layer._packed["w_qkv"].shape == (2, 3 * layer._packed["lay"].d_pad, 32)
    for (k, a), (k2, b) in zip(theirs.state_dict().items(), ours.state_dict().items()):
        assert k == k2 and torch.equal(a, b)


def test_setstate_fills_runtime_defaults_without_the_reference():
    """Same property as above, emulated for the GPU box (no reference tree): a state dict stripped of every runtime
    attribute -- what the reference class would have pickled."""
    layer = C.HGTConv(32, 32, 2, 3, 4, use_RTE=True)
    state = {k: v for k, v in layer.__dict__.items() if k not in RUNTIME_KEYS and k not in ("d_k", "sqrt_dk")}
    clone = C.HGTConv.__new__(C.HGTConv)
    clone.__setstate__(state)
    for k in RUNTIME_KEYS:
        assert k in clone.__dict__
    assert clone.d_k == 8 and clone.precision == C.DEFAULT_PRECISION
    assert repr(clone) == repr(layer)
    dense = C.DenseHGTConv(32, 32, 2, 3, 4)
    state = {k: v for k, v in dense.__dict__.items() if k not in RUNTIME_KEYS}
    dclone = C.DenseHGTConv.__new__(C.DenseHGTConv)
    dclone.__setstate__(state)
    assert dclone._UPDATE_MODE == 1 and dclone._packed is None


def test_packed_parameter_cache_invalidation():
    """The packed arrays are keyed on (data_ptr, _version); `.data` writes do not bump the version, so
    invalidate() / load_state_dict / _apply / training mode must drop the cache."""
    torch.manual_seed(0)
    layer = C.HGTConv(16, 16, 2, 2, 2, use_RTE=False).eval()
    p1 = layer._pack_parameters()
    assert layer._pack_parameters() is p1                               # cached
    with torch.no_grad():
        layer.skip.add_(1.0)                                            # in-place op: version bump
    p2 = layer._pack_parameters()
    assert p2 is not p1 and torch.equal(p2["skip"], layer.skip.detach())
    layer.skip.data.fill_(7.0)                                          # no version bump: stale until invalidate()
    assert layer._pack_parameters() is p2
    layer.invalidate()
    p3 = layer._pack_parameters()
    assert p3 is not p2 and float(p3["skip"][0]) == 7.0
    sd = {k: v.clone() for k, v in layer.state_dict().items()}
    sd["skip"] = torch.full_like(sd["skip"], -2.0)
    layer.load_state_dict(sd)
    assert float(layer._pack_parameters()["skip"][0]) == -2.0
    p4 = layer._pack_parameters()
    layer.double().float()                                              # _apply
    assert layer._pack_parameters() is not p4
    layer.train()
    p5 = layer._pack_parameters()
    layer.skip.data.fill_(3.0)
    assert float(layer._pack_parameters()["skip"][0]) == 3.0            # training mode re-packs every call
    assert layer._prepared_valid is False and p5 is not None


def test_workspaces_are_per_stream_and_staged_calls_need_their_own():
    class FakeDev(str):
        pass
    a = C._Workspace.get(torch.device("cpu"), 128, stream=1)
    b = C._Workspace.get(torch.device("cpu"), 128, stream=2)
    assert a.data_ptr() != b.data_ptr()
    assert C._Workspace.get(torch.device("cpu"), 64, stream=1).data_ptr() == a.data_ptr()      # reused
    assert C._Workspace.get(torch.device("cpu"), 4096, stream=1).numel() >= 4096              # grown
    C._Workspace.clear()
