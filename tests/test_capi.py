"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol the header
declares, the ctypes table matches the header, and the host-side mirror keeps the reference's
parameter names.  No compute calls (there is no GPU here)."""
import ctypes
import os
import re

import pytest
import torch

from pyhgt_amd import _lib
from oracle import hgt_oracle as O
from oracle.reference_loader import reference_available, load_reference_conv, load_reference_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "hgt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hgt_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "libhgt_hip.so lacks %s" % n
    assert set(names) == set(_lib.SIGNATURES), "ctypes table and header disagree"


def test_abi_version_and_error_strings():
    lib = _lib.load()
    text = open(os.path.join(ROOT, "include", "hgt_hip.h")).read()
    assert lib.hgt_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define HGT_ABI_VERSION (\d+)", text).group(1))
    assert lib.hgt_strerror(0) == b"ok"
    assert b"invalid" in lib.hgt_strerror(-1)


@pytest.mark.parametrize("d,H,exp", [(64, 4, (16, 16, 64, 1)), (256, 8, (32, 32, 256, 4)), (512, 8, (64, 64, 512, 8)),
                                     (400, 8, (50, 64, 512, 8)), (100, 2, (50, 64, 128, 2)), (32, 2, (16, 32, 64, 1))])
def test_head_padded_layout(d, H, exp):
    lay = _lib.layout_for(d, H)
    assert (lay.d_k, lay.dk_pad, lay.d_pad, lay.vec) == exp


def test_layout_errors_are_codes_not_crashes():
    lay = _lib.HgtLayout()
    lib = _lib.load()
    assert lib.hgt_layout_for(65, 4, ctypes.byref(lay)) == -1       # d % heads != 0
    assert lib.hgt_layout_for(64, 32, ctypes.byref(lay)) == -2      # fewer than 4 lanes per head
    assert lib.hgt_layout_for(64, 4, None) == -1
    with pytest.raises(RuntimeError):
        _lib.layout_for(64, 32)
    # head counts that do not divide 64 (legal in the reference, conv.py:21) run in the next power-of-two layout
    for d, H, heads, dkp, dpad in [(96, 3, 4, 32, 128), (80, 5, 8, 16, 128), (96, 6, 8, 16, 128), (192, 12, 16, 16, 256)]:
        assert lib.hgt_layout_for(d, H, ctypes.byref(lay)) == 0
        assert (lay.heads, lay.d_k, lay.dk_pad, lay.d_pad) == (heads, d // H, dkp, dpad)


def test_conv_args_struct_matches_header_field_order():
    text = open(os.path.join(ROOT, "include", "hgt_hip.h")).read()
    body = text[text.index("typedef struct hgt_conv_args"):text.index("} hgt_conv_args;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for name in decl.split(","):
            fields.append(re.findall(r"([A-Za-z_0-9]+)\s*$", name.strip())[0])
    assert fields == [f[0] for f in _lib.HgtConvArgs._fields_]


def test_module_keeps_reference_state_dict_names():
    from pyhgt_amd import HGTConv
    T, R, H, d = 3, 4, 4, 64
    layer = HGTConv(d, d, T, R, H, 0.2, True, True)
    sd = O.make_state_dict(d, d, T, R, H, True, True, seed=0)
    assert set(layer.state_dict().keys()) == set(sd.keys())
    for k, v in layer.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    assert sum(p.numel() for p in layer.parameters()) == 78035        # SURVEY appendix B probe of the reference
    layer.load_state_dict(sd)
    # RTE table is the reference's sinusoid (conv.py:289-294)
    fresh = HGTConv(d, d, T, R, H).emb.emb.weight
    assert torch.allclose(fresh, O.sinusoid_table(d), atol=1e-7)


@pytest.mark.skipif(not reference_available(), reason="/root/reference only exists in the build container")
def test_state_dict_names_equal_the_live_reference():
    from pyhgt_amd import HGTConv
    conv = load_reference_conv()
    for use_norm, use_rte in ((True, True), (False, False)):
        ref = conv.HGTConv(32, 32, 2, 3, 4, 0.2, use_norm, use_rte)
        ours = HGTConv(32, 32, 2, 3, 4, 0.2, use_norm, use_rte)
        rs, os_ = ref.state_dict(), ours.state_dict()
        assert list(rs.keys()) == list(os_.keys())
        assert all(rs[k].shape == os_[k].shape for k in rs)
        ours.load_state_dict(rs)
        assert repr(ours) == repr(ref)


def test_cpu_input_fails_loudly():
    from pyhgt_amd import HGTConv
    layer = HGTConv(16, 16, 2, 2, 2, use_RTE=False).eval()
    x = torch.randn(4, 16)
    nt = torch.zeros(4, dtype=torch.long)
    ei = torch.zeros(2, 3, dtype=torch.long)
    et = torch.zeros(3, dtype=torch.long)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        layer(x, nt, ei, et)


def test_general_conv_rejects_out_of_scope_layers():
    from pyhgt_amd import GeneralConv, DenseHGTConv
    GeneralConv('hgt', 16, 16, 2, 2, 2, 0.2)
    assert isinstance(GeneralConv('dense_hgt', 16, 16, 2, 2, 2, 0.2).base_conv, DenseHGTConv)
    with pytest.raises(NotImplementedError):
        GeneralConv('gcn', 16, 16, 2, 2, 2, 0.2)


def test_dense_module_keeps_reference_state_dict_names():
    from pyhgt_amd import DenseHGTConv
    T, R, H, d = 3, 4, 4, 64
    layer = DenseHGTConv(d, d, T, R, H, 0.2, True, True)
    sd = O.make_state_dict(d, d, T, R, H, True, True, seed=0, dense=True)
    assert set(layer.state_dict().keys()) == set(sd.keys())       # no `skip`; mid_linear / out_linear / out_norm
    for k, v in layer.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    layer.load_state_dict(sd)


@pytest.mark.skipif(not reference_available(), reason="/root/reference only exists in the build container")
def test_dense_state_dict_names_equal_the_live_reference():
    from pyhgt_amd import DenseHGTConv
    conv = load_reference_conv()
    ref = conv.DenseHGTConv(32, 32, 2, 3, 4, 0.2, True, True)
    ours = DenseHGTConv(32, 32, 2, 3, 4, 0.2, True, True)
    rs, os_ = ref.state_dict(), ours.state_dict()
    assert set(rs.keys()) == set(os_.keys())
    assert all(rs[k].shape == os_[k].shape for k in rs)
    ours.load_state_dict(rs)
    assert repr(ours) == repr(ref)


def test_gnn_wrapper_matches_reference_parameter_count():
    """ogbn-mag/README.md:30: 21,173,389 parameters for GNN(129, 512, 4, 9, 8, 4, norms on, RTE) + Classifier(512, 349)."""
    from pyhgt_amd import GNN
    gnn = GNN(129, 512, 4, 9, 8, 4, prev_norm=True, last_norm=True, use_RTE=True)
    assert sum(p.numel() for p in gnn.parameters()) + (512 * 349 + 349) == 21173389


@pytest.mark.skipif(not reference_available(), reason="/root/reference only exists in the build container")
def test_gnn_state_dict_names_equal_the_live_reference():
    from oracle.reference_loader import load_reference_model
    from pyhgt_amd import GNN
    ref = load_reference_model().GNN(32, 64, 3, 4, 4, 2, prev_norm=True, last_norm=False, use_RTE=True)
    ours = GNN(32, 64, 3, 4, 4, 2, prev_norm=True, last_norm=False, use_RTE=True)
    rs, os_ = ref.state_dict(), ours.state_dict()
    assert list(rs.keys()) == list(os_.keys())
    assert all(rs[k].shape == os_[k].shape for k in rs)
    ours.load_state_dict(rs)


@pytest.mark.skipif(not reference_available(), reason="/root/reference only exists in the build container")
def test_heads_keep_reference_state_dict_names():
    """Classifier / Matcher (model.py:3-49): same parameter names, shapes and repr as the live reference."""
    from pyhgt_amd import Classifier, Matcher
    model = load_reference_model()
    for ours, ref in ((Classifier(32, 7), model.Classifier(32, 7)), (Matcher(32), model.Matcher(32))):
        rs, os_ = ref.state_dict(), ours.state_dict()
        assert list(rs.keys()) == list(os_.keys())
        assert all(rs[k].shape == os_[k].shape for k in rs)
        ours.load_state_dict(rs)
        if isinstance(ours, Classifier):      # the reference's Matcher.__repr__ reads an attribute it never sets (model.py:47-49)
            assert repr(ours) == repr(ref)
