"""Import the reference's own pyHGT/conv.py + model.py VERBATIM (build container only).

TEST INFRASTRUCTURE.  /root/reference is read-only and absent on the GPU box, so
this is used (a) by oracle/gen_golden.py to produce tests/golden/*.npz and (b) by
the `not gpu` tests that validate oracle/hgt_oracle.py against the real thing when
the reference tree is present.  Nothing is copied: the reference files are imported
from where they lie; the only thing supplied is the torch_geometric stand-in in
oracle/pyg_shim (the reference's un-vendored third-party dependency).
"""
import importlib
import os
import sys

REFERENCE_ROOT = os.environ.get("HGT_REFERENCE_ROOT", "/root/reference")
_SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pyg_shim")


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "pyHGT", "conv.py"))


def load_reference_conv():
    """Returns the reference module `pyHGT.conv` (HGTConv, GeneralConv, ...)."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    for p in (_SHIM, REFERENCE_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    return importlib.import_module("pyHGT.conv")


def load_reference_model():
    """Returns the reference module `pyHGT.model` (GNN, Classifier, Matcher)."""
    load_reference_conv()
    return importlib.import_module("pyHGT.model")


def load_reference_data():
    """Returns the reference module `pyHGT.data` (Graph, sample_subgraph, to_torch).  Its plotting imports (seaborn,
    matplotlib) are absent from this image and irrelevant to the sampler: empty stand-in modules are registered for them."""
    import types
    load_reference_conv()
    for name in ("seaborn", "matplotlib", "matplotlib.pyplot", "matplotlib.cm"):
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)
    import numpy as np
    for alias, typ in (("int", int), ("float", float), ("str", str), ("bool", bool)):      # removed in numpy >= 1.24 (utils.py:59,69)
        if not hasattr(np, alias):
            setattr(np, alias, typ)
    if not hasattr(np, "asfarray"):
        np.asfarray = lambda a, dtype=float: np.asarray(a, dtype=dtype)
    return importlib.import_module("pyHGT.data")
