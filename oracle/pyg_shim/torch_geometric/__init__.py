"""Minimal stand-in for torch_geometric, TEST INFRASTRUCTURE ONLY.

PyTorch-Geometric is a third-party dependency of the reference (pinned
torch-geometric==1.3.2, torch-scatter==1.3.2 in /root/reference/requirements.txt:4-9)
that is neither vendored under /root/reference nor installable offline.  The
reference's hot path uses exactly three of its symbols
(/root/reference/pyHGT/conv.py:5-8): MessagePassing, utils.softmax and
nn.inits.glorot.  This package restates their published semantics in pure
torch so that the reference's own conv.py / model.py can be imported and run
VERBATIM on CPU as the parity oracle.  Nothing in pyhgt_amd/ imports this.
"""
