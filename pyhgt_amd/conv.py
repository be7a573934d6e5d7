"""Host-side mirror of the reference's conv module for the HGT hot path.

`HGTConv` keeps the reference layer's constructor, parameter/state_dict names and forward
signature (/root/reference/pyHGT/conv.py:11-58), so it drops into the reference's
`GeneralConv` / `model.GNN` unchanged (`install_into(pyHGT.conv)`), but forward() enqueues the
hand-written HIP kernels of libhgt_hip.so (include/hgt_hip.h) instead of PyG message passing.

PyTorch is used for parameter storage, device memory and the current stream only.  There is no
CPU / eager fallback: a non-GPU input or a missing library raises.
"""
import ctypes as C
import math
import threading
import weakref
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib

__all__ = ["HGTConv", "GeneralConv", "RelTemporalEncoding", "GraphPlan", "install_into"]


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


class _Workspace:
    """Growing device scratch buffers, one per (device, stream): layers that run back to back on one stream share a
    buffer, layers enqueued on different streams (or from different threads on their own streams) never do -- the
    C ABI is re-entrant given distinct workspaces and never allocates (include/hgt_hip.h).
    Staged multi-GPU execution keeps Q/K/V in the workspace between calls and therefore passes its OWN buffer
    (`forward(..., workspace=...)`, pyhgt_amd.dist.PartitionedGraph), which nothing else can grow or overwrite."""
    _bufs = {}
    _lock = threading.Lock()

    @classmethod
    def get(cls, device, nbytes, stream=None):
        key = (str(device), int(stream if stream is not None else _stream()))
        with cls._lock:
            buf = cls._bufs.get(key)
            if buf is None or buf.numel() < nbytes:
                cls._bufs.pop(key, None)
                buf = None
                buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
                cls._bufs[key] = buf
            return buf

    @classmethod
    def clear(cls):
        with cls._lock:
            cls._bufs.clear()


_PLAN_SIZES = {}


def _plan_sizes(N, E, T, R):
    """(plan_bytes, tmp_bytes, max_items) of hgt_plan_sizes_for, memoised: sampled batches repeat a handful of shapes."""
    key = (N, E, T, R)
    v = _PLAN_SIZES.get(key)
    if v is None:
        sz = _lib.HgtPlanSizes()
        _lib.check(_lib.load().hgt_plan_sizes_for(N, E, T, R, C.byref(sz)), "hgt_plan_sizes_for")
        v = (int(sz.plan_bytes), int(sz.tmp_bytes), int(sz.max_items))
        if len(_PLAN_SIZES) > 4096:
            _PLAN_SIZES.clear()
        _PLAN_SIZES[key] = v
    return v


class _HeaderSlots:
    """A small ring of pinned host slots for the asynchronous read-back of plan headers (allocating pinned memory per plan
    costs ~20 us).  A plan whose slot is taken over before it looked at it simply never learns its hub count (safe)."""
    N = 64
    buf = None
    base = 0
    owners = [None] * 64
    nxt = 0

    @classmethod
    def acquire(cls, plan):
        if cls.buf is None:
            cls.buf = torch.zeros(cls.N, 4, dtype=torch.int32).pin_memory()
            cls.base = cls.buf.data_ptr()
        i = cls.nxt
        cls.nxt = (i + 1) % cls.N
        old = cls.owners[i]() if cls.owners[i] is not None else None
        if old is not None and old._hdr_slot == i:
            old._hdr_slot = None
        cls.owners[i] = weakref.ref(plan)
        return i

    @classmethod
    def release(cls, plan):
        if plan._hdr_slot is not None:
            cls.owners[plan._hdr_slot] = None
            plan._hdr_slot = None


class GraphPlan:
    """Device-resident plan of one typed (sub)graph: dst-tile/relation sorted int32 edges, segment
    table, wavefront work items, typed row lists.  Built once per graph by hgt_plan_build and
    reused by every layer (the reference re-derives all of this per layer per call through PyG
    gathers and the T*T*R mask loop, conv.py:57,71-84)."""

    _cache = OrderedDict()
    _cache_lock = threading.Lock()
    CACHE_SIZE = 4          # plans (and the graph tensors they were built from) kept alive; set to 0 to disable caching
    STRICT = False          # True: the first forward on a NEW plan waits for the plan build and raises IndexError for malformed ids
                            # before any layer kernel runs -- the reference's behaviour (index_select / nn.Embedding raise at once);
                            # costs one host synchronisation per new graph.  False: the flag arrives asynchronously and the error
                            # surfaces on the first forward after it has landed (raise_if_bad).  HGTConv(strict=...) overrides it.

    def __init__(self, node_type, edge_index, edge_type, edge_time, num_types, num_relations, n_q_rows=None, reverse=False):
        """reverse=True builds the plan of the TRANSPOSED graph (every edge j -> i taken as i -> j; used by the backward pass,
        pyhgt_amd/autograd.py) straight from the same edge_index tensor: the two rows are swapped through the row stride."""
        lib = _lib.load()
        if not node_type.is_cuda:
            raise RuntimeError("pyhgt_amd: graph tensors must live on the GPU (no CPU fallback)")
        for name, t in (("node_type", node_type), ("edge_index", edge_index), ("edge_type", edge_type)):
            if t.dtype != torch.int64:
                raise TypeError("pyhgt_amd: %s must be int64 like the reference's tensors, got %s" % (name, t.dtype))
        if edge_index.dim() != 2 or edge_index.size(0) != 2:
            raise ValueError("edge_index must be [2, E]")
        self.N = int(node_type.numel())
        self.E = int(edge_index.size(1))
        self.T, self.R = int(num_types), int(num_relations)
        self.NQ = self.N if n_q_rows is None else int(n_q_rows)
        if edge_type.numel() != self.E or (edge_time is not None and edge_time.numel() != self.E):
            raise ValueError("edge_type / edge_time must have E entries")
        node_type = node_type.contiguous()
        edge_type = edge_type.contiguous()
        if edge_time is not None:
            if edge_time.dtype != torch.int64:
                raise TypeError("pyhgt_amd: edge_time must be int64")
            edge_time = edge_time.contiguous()
        sz = _lib.HgtPlanSizes()
        _lib.check(lib.hgt_plan_sizes_for(self.N, self.E, self.T, self.R, C.byref(sz)), "hgt_plan_sizes_for")
        dev = node_type.device
        self.buf = torch.empty(int(sz.plan_bytes), dtype=torch.uint8, device=dev)
        tmp = torch.empty(int(sz.tmp_bytes), dtype=torch.uint8, device=dev)
        self.max_items = int(sz.max_items)
        # edge_index arrives as a (1,2)-strided view (data.py:254): hand the strides over, no copy
        sr, sc = (edge_index.stride(0), edge_index.stride(1)) if self.E > 0 else (0, 1)
        ei_ptr = _ptr(edge_index)
        if reverse and self.E > 0:
            ei_ptr, sr = ei_ptr + 8 * sr, -sr                  # row 0 := targets, row 1 := sources
        self.node_type = node_type
        self._graph = (node_type, edge_index, edge_type, edge_time)   # kept for transposed() / rte_plan()
        self._transposed = None
        self._rte_plan = None
        _lib.check(lib.hgt_plan_build(ei_ptr, sr, sc, _ptr(edge_type), _ptr(edge_time), _ptr(node_type),
                                      self.N, self.NQ, self.E, self.T, self.R, _ptr(self.buf), self.buf.numel(),
                                      _ptr(tmp), tmp.numel(), _stream()), "hgt_plan_build")
        # tmp is released by the caching allocator only after the stream ran past this point
        tmp.record_stream(torch.cuda.current_stream())
        self.device = dev
        self._start_header_readback()

    def _start_header_readback(self):
        # plan header (n_items, bad_index, n_hubs) copied to pinned host memory WITHOUT synchronising; `no_hubs` /
        # `raise_if_bad` read it once the copy has completed (from the second layer on, in practice)
        self._no_hubs = None
        self._no_unknown = None
        self._bad = None
        self._hdr_slot = _HeaderSlots.acquire(self)
        # (one hipMemcpyAsync through the C ABI: slicing + view + copy_ on the torch side cost ~8 us of host time per plan)
        _lib.check(_lib.load().hgt_plan_header_to_host(self.buf.data_ptr(), _HeaderSlots.base + 16 * self._hdr_slot, _stream()),
                   "hgt_plan_header_to_host")
        self._hdr_event = torch.cuda.Event()
        self._hdr_event.record()

    @classmethod
    def from_sorted(cls, node_type, edge_index, edge_type, edge_time, src32, dst32, time32, rel_ptr, type_off, num_types,
                    num_relations):
        """Plan of a graph that is already in the sampler's order: type-contiguous nodes
        (type_off int32[T+1]), edges grouped by relation (rel_ptr int32[R+1]) with non-decreasing targets inside a relation,
        ids as int32 device arrays -- hgt_plan_from_sorted: no radix sort, no int64 index traffic.  The int64 tensors of
        the reference's wire format (same edge order) are kept for the calls that still take them (forward() signature,
        transposed()/rte_plan() of the backward pass)."""
        lib = _lib.load()
        self = cls.__new__(cls)
        self.N, self.E = int(node_type.numel()), int(src32.numel())
        self.T, self.R = int(num_types), int(num_relations)
        self.NQ = self.N
        for t in (src32, dst32, rel_ptr, type_off) + ((time32,) if time32 is not None else ()):
            if t.dtype != torch.int32 or not t.is_cuda or not t.is_contiguous():
                raise TypeError("from_sorted takes contiguous int32 device arrays")
        if rel_ptr.numel() != self.R + 1 or type_off.numel() != self.T + 1:
            raise ValueError("rel_ptr must have R+1 and type_off T+1 entries")
        plan_bytes, tmp_bytes, self.max_items = _plan_sizes(self.N, self.E, self.T, self.R)
        dev = node_type.device
        self.buf = torch.empty(plan_bytes, dtype=torch.uint8, device=dev)
        tmp = torch.empty(tmp_bytes, dtype=torch.uint8, device=dev)
        self.node_type = node_type
        self._graph = (node_type, edge_index, edge_type, edge_time)
        self._transposed = None
        self._rte_plan = None
        _lib.check(lib.hgt_plan_from_sorted(_ptr(src32), _ptr(dst32), _ptr(time32), _ptr(rel_ptr), _ptr(type_off), self.N, self.NQ,
                                            self.E, self.T, self.R, _ptr(self.buf), self.buf.numel(), _ptr(tmp), tmp.numel(),
                                            _stream()), "hgt_plan_from_sorted")
        tmp.record_stream(torch.cuda.current_stream())
        self.device = dev
        self._start_header_readback()
        return self

    @property
    def no_hubs(self):
        """True once it is known (no synchronisation) that no target exceeds the hub in-degree threshold; False = unknown
        or hubs exist.  Lets hgt_conv_forward skip enqueueing the hub kernels (they would exit immediately)."""
        self._poll_header()
        return bool(self._no_hubs)

    @property
    def no_unknown_rows(self):
        """True once it is known (no synchronisation) that every target row has a valid node type (hgt_zero_rows has no work)."""
        self._poll_header()
        return bool(self._no_unknown)

    def _poll_header(self, wait=False):
        if self._no_hubs is None and self._hdr_slot is not None and (wait or self._hdr_event.query()):
            if wait:
                self._hdr_event.synchronize()
            row = _HeaderSlots.buf[self._hdr_slot]
            self._no_hubs = int(row[2]) == 0
            self._no_unknown = int(row[3]) == 0
            self._bad = int(row[1])
            _HeaderSlots.release(self)

    def raise_if_bad(self, wait=False, ignore_time=False):
        """The reference fails with an IndexError (index_select / nn.Embedding) when edge_index holds a node id outside
        [0, N) or edge_time a value outside [0, 240).  The plan build flags both on the device; the flag reaches the host
        asynchronously, so -- without a synchronisation -- the error surfaces on the first forward AFTER the header copy
        has landed (the second layer of a GNN, in practice); wait=True synchronises and checks now."""
        self._poll_header(wait)
        if self._bad is None and wait and self._hdr_slot is None:      # slot was recycled before we looked: read the device copy
            self._bad = int(self.buf[:16].view(torch.int32)[1].item())
        bad = (self._bad or 0) & (~2 if ignore_time else ~0)      # a layer with use_RTE=False never looks at edge_time (conv.py:91)
        if bad:
            what = []
            if bad & 1:
                what.append("edge_index contains node ids outside [0, num_nodes) (or targets beyond n_q_rows)")
            if bad & 2:
                what.append("edge_time contains values outside [0, %d)" % _lib.HGT_RTE_LEN)
            if bad & 4:
                what.append("from_sorted: edges are not grouped by relation with non-decreasing targets (or rel_ptr does not span [0, E])")
            raise IndexError("pyhgt_amd: " + "; ".join(what))

    @property
    def ptr(self):
        return self.buf.data_ptr()

    def row_lists(self):
        pr = _lib.HgtPlanRows()
        _lib.check(_lib.load().hgt_plan_row_lists(self.ptr, self.N, self.E, self.T, self.R, C.byref(pr)), "hgt_plan_row_lists")
        return pr

    def tile_items(self):
        """Host copy (synchronises; once per graph) of the plan's per-tile item table: the logits work items of destination tile
        t are [table[t], table[t + 1]).  Returns (int64 tensor [n_tiles + 1] on the CPU, targets per tile).  pyhgt_amd.dist uses
        it to launch the edge phase of one target block (hgt_conv_forward stage 5)."""
        off, nt = C.c_uint64(), C.c_int64()
        _lib.check(_lib.load().hgt_plan_tile_items_offset(self.N, self.E, self.T, self.R, C.byref(off), C.byref(nt)),
                   "hgt_plan_tile_items_offset")
        tile, item = C.c_int32(), C.c_int32()
        _lib.check(_lib.load().hgt_plan_constants(C.byref(tile), C.byref(item)), "hgt_plan_constants")
        tab = self.buf[int(off.value):int(off.value) + 4 * (int(nt.value) + 1)].view(torch.int32).cpu().to(torch.int64)
        return tab, int(tile.value)

    def check_indices(self):
        """Debug aid (synchronises): raise if an edge endpoint was outside [0, N) (the reference
        would have raised an IndexError inside index_select)."""
        self.raise_if_bad(wait=True)
        return int(self.buf[:8].view(torch.int32)[0].item())

    # -- derived plans of the backward pass (built on first use, kept with the plan) -----------------
    def transposed(self):
        """Plan of the reversed edges over the same nodes (every node is a target): dK / dV of the backward pass are
        aggregations over the OUT-edges of a node."""
        if self._transposed is None:
            nt, ei, et, _ = self._graph
            self._transposed = GraphPlan(nt, ei, et, None, self.T, self.R, reverse=True)
        return self._transposed

    def rte_plan(self, num_types, num_relations):
        """Plan whose targets are the rows of the temporal tables: edge e = (i -> row type(j) * 240 + dt_e), relation kept
        (relation id R = unclaimed for edges the forward did not claim).  Its aggregations are the table gradients.
        Returns (plan, number of table rows); node ids of the original graph are shifted by that number."""
        if self._rte_plan is None:
            nt, ei, et, tm = self._graph
            T, R, L = int(num_types), int(num_relations), _lib.HGT_RTE_LEN
            tab = T * L
            src, dst = ei[0], ei[1]
            tj, ti = nt[src], nt[dst]
            ok = (tj >= 0) & (tj < T) & (ti >= 0) & (ti < T) & (et >= 0) & (et < R)
            row = tj.clamp(0, T - 1) * L + tm.clamp(0, L - 1)
            ei_r = torch.stack([dst + tab, row], dim=0).contiguous()
            et_r = torch.where(ok, et, torch.full_like(et, R))
            nt_r = torch.cat([torch.arange(T, device=nt.device, dtype=nt.dtype).repeat_interleave(L), nt])
            self._rte_plan = (GraphPlan(nt_r, ei_r, et_r, None, T, R, n_q_rows=tab), tab)
        return self._rte_plan

    # -- cache: the reference passes the SAME tensors to every layer (model.py:78-79) ------------
    @staticmethod
    def _cache_key(node_type, edge_index, edge_type, edge_time, num_types, num_relations, n_q_rows):
        tensors = (node_type, edge_index, edge_type, edge_time)
        key = tuple((t.data_ptr(), t._version, tuple(t.shape), tuple(t.stride())) if t is not None else None for t in tensors)
        return key + (int(num_types), int(num_relations), n_q_rows, str(node_type.device))

    @classmethod
    def register(cls, plan, node_type, edge_index, edge_type, edge_time, num_types, num_relations):
        """Put a plan built elsewhere (from_sorted) into the cache under the tensors the model will be called with, so that
        the reference's unchanged call `gnn(node_feature, node_type, edge_time, edge_index, edge_type)` finds it."""
        tensors = (node_type, edge_index, edge_type, edge_time)
        with cls._cache_lock:
            for tm in ((edge_time, None) if edge_time is not None else (None,)):      # layers with and without use_RTE
                cls._cache[cls._cache_key(node_type, edge_index, edge_type, tm, num_types, num_relations, None)] = (plan, tensors)
            cls._evict(max(int(cls.CACHE_SIZE), 1))

    @classmethod
    def cached(cls, node_type, edge_index, edge_type, edge_time, num_types, num_relations, n_q_rows=None):
        tensors = (node_type, edge_index, edge_type, edge_time)
        key = cls._cache_key(node_type, edge_index, edge_type, edge_time, num_types, num_relations, n_q_rows)
        with cls._cache_lock:
            hit = cls._cache.get(key)
            if hit is not None:
                # a registered plan sits under two keys (with / without edge_time): refresh both, or the sibling of a hot plan
                # stays at the front of the eviction order
                for k in [k for k, v in cls._cache.items() if v[0] is hit[0]]:
                    cls._cache.move_to_end(k)
                return hit[0]
        plan = cls(node_type, edge_index, edge_type, edge_time, num_types, num_relations, n_q_rows)
        with cls._cache_lock:
            # keep the key tensors alive so their addresses cannot be recycled while the entry lives
            cls._cache[key] = (plan, tensors)
            cls._evict(max(int(cls.CACHE_SIZE), 0))
        return plan

    @classmethod
    def _evict(cls, keep):
        """Least-recently-used eviction counted in PLANS (a registered hand-off plan sits under two keys: with and without
        edge_time), so that CACHE_SIZE pre-registered batches really stay cached (caller holds _cache_lock)."""
        while len({id(v[0]) for v in cls._cache.values()}) > keep:
            victim = id(next(iter(cls._cache.values()))[0])
            for k in [k for k, v in cls._cache.items() if id(v[0]) == victim]:
                del cls._cache[k]

    @classmethod
    def clear_cache(cls):
        with cls._cache_lock:
            cls._cache.clear()


class RelTemporalEncoding(nn.Module):
    """Parameter container with the reference's names (conv.py:283-299): `emb` = fixed sinusoid
    table [max_len, n_hid] scaled by 1/sqrt(n_hid), `lin` = Linear(n_hid, n_hid).  HGTConv folds
    lin(emb[dt]) through W_k / W_v into per-(source type, dt) tables on the GPU
    (hgt_conv_forward step 3) instead of evaluating it per edge."""

    def __init__(self, n_hid, max_len=240, dropout=0.2):
        super().__init__()
        if max_len != _lib.HGT_RTE_LEN:
            raise ValueError("libhgt_hip is built for max_len = 240 (conv.py:287)")
        pos = torch.arange(0.0, max_len).unsqueeze(1)
        freq = torch.exp(torch.arange(0, n_hid, 2) * -(math.log(10000.0) / n_hid))
        table = nn.Embedding(max_len, n_hid)
        with torch.no_grad():
            table.weight[:, 0::2] = torch.sin(pos * freq) / math.sqrt(n_hid)
            table.weight[:, 1::2] = torch.cos(pos * freq) / math.sqrt(n_hid)
        self.emb = table
        self.lin = nn.Linear(n_hid, n_hid)


PRECISIONS = ("fp32", "bf16x3", "f16x3")
DEFAULT_PRECISION = "f16x3"      # round 6: the reference-accurate split is what a default-constructed layer runs (see HGTConv)


class HGTConv(nn.Module):
    """Heterogeneous Graph Transformer layer, forward on MI355X.

    Same constructor / parameters / forward as the reference (conv.py:11-58).  Extra keyword-only
    options: keep_att (export softmax weights into self.att like conv.py:108; off by default
    because nothing in the reference reads it), precision: "f16x3" (the default since round 6) evaluates every matrix-core product of
    the layer (typed Linears, relation transforms of the aggregation, fused a_linear) as three MFMA products of fp16 hi / lo parts
    with power-of-two row scales and fp32 accumulation -- as close to the fp64 result as the reference's own fp32 arithmetic
    (conv.py:96-104 is fp32 end to end): max |out - reference| <= 2e-6 over the tested configurations; "bf16x3" (the default of
    rounds 1-5, now the opt-in fast mode) runs the same three products on bf16 hi / mid parts (~16 mantissa bits per operand:
    <= 5e-5, bound 1e-4; ~5 % faster at c2); "fp32" uses the exact fp32 MFMA chain (<= 2e-6, 1.9x slower at c2).  Training (autograd) and the staged multi-GPU calls evaluate an "f16x3"
    layer with the "bf16x3" kernels.  strict=True (or GraphPlan.STRICT = True): node ids outside [0, N) / edge_time outside [0, 240)
    raise IndexError on the FIRST forward of a new graph, before any layer kernel runs, like the reference's index_select /
    nn.Embedding (one host synchronisation per new graph); by default the check is asynchronous and the same IndexError surfaces
    on the first forward after the plan's flag has reached the host.
    """

    def __init__(self, in_dim, out_dim, num_types, num_relations, n_heads, dropout=0.2, use_norm=True, use_RTE=True,
                 keep_att=False, precision=DEFAULT_PRECISION, strict=None, **kwargs):
        super().__init__()
        self.in_dim, self.out_dim = in_dim, out_dim
        self.num_types, self.num_relations = num_types, num_relations
        self.total_rel = num_types * num_relations * num_types
        self.n_heads = n_heads
        self.d_k = out_dim // n_heads
        self.sqrt_dk = math.sqrt(self.d_k)
        self.use_norm, self.use_RTE = use_norm, use_RTE
        self.keep_att = keep_att
        if precision not in PRECISIONS:
            raise ValueError("precision must be one of %s" % (PRECISIONS,))
        self.precision = precision
        self.kernel_flags = 0          # hgt_conv_args.flags (HGT_FLAG_*): explicit kernel selection for A/B runs and tests
        self.strict = strict           # None = GraphPlan.STRICT; True: malformed ids raise IndexError on the very first forward
        self.att = None

        self.k_linears = nn.ModuleList(nn.Linear(in_dim, out_dim) for _ in range(num_types))
        self.q_linears = nn.ModuleList(nn.Linear(in_dim, out_dim) for _ in range(num_types))
        self.v_linears = nn.ModuleList(nn.Linear(in_dim, out_dim) for _ in range(num_types))
        self.a_linears = nn.ModuleList(nn.Linear(out_dim, out_dim) for _ in range(num_types))
        self.norms = nn.ModuleList(nn.LayerNorm(out_dim) for _ in range(num_types)) if use_norm else nn.ModuleList()
        self.relation_pri = nn.Parameter(torch.ones(num_relations, n_heads))
        bound = math.sqrt(6.0 / (2 * self.d_k))                       # PyG glorot over the two trailing dims
        self.relation_att = nn.Parameter(torch.empty(num_relations, n_heads, self.d_k, self.d_k).uniform_(-bound, bound))
        self.relation_msg = nn.Parameter(torch.empty(num_relations, n_heads, self.d_k, self.d_k).uniform_(-bound, bound))
        self._init_update_parameters(num_types, out_dim)
        self.drop = nn.Dropout(dropout)
        if use_RTE:
            self.emb = RelTemporalEncoding(in_dim)
        self._init_runtime_state()

    _UPDATE_MODE = 0     # hgt_conv_args.update_mode
    _warned_eval_grad = False

    # -- state that is not part of the reference module: caches of packed parameters / device-side weight images ------
    EXTRA_KERNEL_FLAGS = 0      # OR-ed into every layer's kernel_flags (tests: tests/conftest.py sets it from HGT_TEST_KERNEL_FLAGS)
    _RUNTIME_DEFAULTS = dict(keep_att=False, precision=DEFAULT_PRECISION, kernel_flags=0, strict=None, att=None, _packed=None, _packed_key=None,
                             _prepared=None, _prepared_tag=None, _prepared_valid=False, _plist=None)

    def _init_runtime_state(self):
        for k, v in self._RUNTIME_DEFAULTS.items():
            if k not in self.__dict__:
                self.__dict__[k] = v

    def __getstate__(self):
        st = dict(self.__dict__)
        for k in ("_packed", "_packed_key", "_prepared", "_prepared_tag", "_plist"):     # derived device buffers: rebuilt on demand
            st[k] = None
        st["_prepared_valid"] = False
        st["att"] = None
        return st

    def __setstate__(self, state):
        """Whole-module pickles (torch.save(model), OAG/train_paper_field.py:279) written by the REFERENCE class carry none of
        this implementation's runtime attributes: fill in the defaults so that such a module loads and runs
        (pyHGT.conv.HGTConv must resolve to this class at load time: install_into)."""
        super().__setstate__(state)
        self._init_runtime_state()
        if "d_k" not in self.__dict__:
            self.d_k = self.out_dim // self.n_heads
        if "sqrt_dk" not in self.__dict__:
            self.sqrt_dk = math.sqrt(self.d_k)

    def invalidate(self):
        """Drop the packed-parameter cache and the device-side weight images.  They are keyed on the parameters' version
        counters, which optimizer steps, load_state_dict and in-place torch ops all change (.to()/.float() pass through
        _apply, which calls this) -- but writes through `.data` (`p.data.copy_()`, `p.data[...] = x`; EMA / manual
        initialisation code) do NOT bump `_version`, and neither does replacing a Parameter object of a sub-module
        (`layer.k_linears[0].weight = nn.Parameter(...)`): call invalidate() after such an update (training mode re-packs
        on every forward anyway).  Detected without a call: optimizer steps / in-place torch ops (version counters),
        `p.data = tensor` (storage address), re-assigning a Parameter or sub-module attribute of the layer itself."""
        self._packed = self._packed_key = None
        self._prepared_valid = False
        self.__dict__["_plist"] = None

    def _load_from_state_dict(self, *args, **kwargs):
        self.invalidate()
        return super()._load_from_state_dict(*args, **kwargs)

    def __setattr__(self, name, value):
        # `layer.skip = nn.Parameter(...)` (or any re-assigned sub-module) replaces objects the cached parameter list points at
        if isinstance(value, (nn.Parameter, nn.Module)) and "_plist" in self.__dict__:
            self.__dict__["_plist"] = None
            self.__dict__["_packed_key"] = None
        super().__setattr__(name, value)

    def _apply(self, fn, *args, **kwargs):
        self.invalidate()
        return super()._apply(fn, *args, **kwargs)

    def _init_update_parameters(self, num_types, out_dim):
        self.skip = nn.Parameter(torch.ones(num_types))                 # conv.py:47

    def _pack_update_parameters(self, grad=False):
        return dict(skip=(self.skip if grad else self.skip.detach()).float().contiguous())

    def _set_update_args(self, a, pk):
        a.update_mode = self._UPDATE_MODE
        a.skip = _ptr(pk["skip"])

    # ------------------------------------------------------------------------------------------
    def _pack_parameters(self, grad=False):
        """Stack the per-type Linear / LayerNorm parameters into the contiguous, head-padded arrays
        hgt_conv_forward takes (pure data movement; cached until a parameter changes).  grad=True: built with autograd
        recording (never cached), so that gradients of the packed arrays flow back to the reference-named parameters."""
        # cache key: the SUM of the parameters' version counters and storage addresses over a parameter list that is itself cached
        # -- a few microseconds per forward (the round-2 key, a 47-tuple of (data_ptr, _version), cost ~50 us of host time per layer and
        # would have bounded the latency regime).  Optimizer steps, in-place torch ops and load_state_dict bump a version;
        # .to() / .float() / load_state_dict also pass through _apply / _load_from_state_dict (invalidate()).
        plist = self.__dict__.get("_plist")
        if plist is None:
            plist = self.__dict__["_plist"] = list(self.parameters())
        key = 0
        for p in plist:      # (+ the storage address: `p.data = new_tensor` -- common in weight-loading code -- keeps the version)
            key += p._version + p.data_ptr()
        if not grad and self._packed is not None and self._packed_key == key and not self.training:
            return self._packed
        lay = _lib.layout_for(self.out_dim, self.n_heads)
        H, dk, dkp, dp = self.n_heads, lay.d_k, lay.dk_pad, lay.d_pad
        HL = lay.heads            # heads of the layout: H rounded up to a power of two; the extra heads are all-zero
        T, din, dout = self.num_types, self.in_dim, self.out_dim

        def pad_rows(w):          # [dout, *] -> [dp, *] (each head's dk rows followed by dkp-dk zero rows, then HL-H zero heads)
            if dkp == dk and HL == H:
                return w
            tail = w.shape[1:]
            w = w.reshape(H, dk, *tail)
            if dkp != dk:
                w = torch.cat([w, w.new_zeros(H, dkp - dk, *tail)], dim=1)
            if HL != H:
                w = torch.cat([w, w.new_zeros(HL - H, dkp, *tail)], dim=0)
            return w.reshape(dp, *tail)

        det = (lambda t: t) if grad else (lambda t: t.detach())
        with torch.set_grad_enabled(bool(grad)):
            w_qkv = torch.stack([torch.cat([pad_rows(self.q_linears[t].weight), pad_rows(self.k_linears[t].weight),
                                            pad_rows(self.v_linears[t].weight)], 0) for t in range(T)]).float().contiguous()
            b_qkv = torch.stack([torch.cat([pad_rows(self.q_linears[t].bias), pad_rows(self.k_linears[t].bias),
                                            pad_rows(self.v_linears[t].bias)], 0) for t in range(T)]).float().contiguous()
            w_a = torch.stack([pad_rows(self.a_linears[t].weight.t()).t() for t in range(T)]).float().contiguous()
            b_a = torch.stack([self.a_linears[t].bias for t in range(T)]).float().contiguous()
            ln_w = ln_b = None
            if self.use_norm:
                ln_w = torch.stack([self.norms[t].weight for t in range(T)]).float().contiguous()
                ln_b = torch.stack([self.norms[t].bias for t in range(T)]).float().contiguous()
            packed = dict(lay=lay, w_qkv=w_qkv, b_qkv=b_qkv, w_a=w_a, b_a=b_a, ln_w=ln_w, ln_b=ln_b,
                          ratt=det(self.relation_att).float().contiguous(),
                          rmsg=det(self.relation_msg).float().contiguous(),
                          rpri=det(self.relation_pri).float().contiguous())
            packed.update(self._pack_update_parameters(grad))
            if self.use_RTE:
                packed.update(rte_emb=det(self.emb.emb.weight).float().contiguous(),
                              rte_w=det(self.emb.lin.weight).float().contiguous(),
                              rte_b=det(self.emb.lin.bias).float().contiguous())
        assert w_qkv.shape == (T, 3 * dp, din) and w_a.shape == (T, dout, dp)
        if grad:
            return packed
        self._packed, self._packed_key = packed, key
        self._prepared_valid = False      # the device-side weight images (hgt_conv_args.prepared) are stale now
        return packed

    def _prepared_buffer(self, device, n_slices=1, prec=None):
        """Per-layer device buffer for the weight-only preprocessing hgt_conv_forward keeps across calls (packed relation
        matrices, split-bf16 weight tiles, temporal tables): valid until a parameter, the precision or the device changes."""
        key = (str(device), prec or self.precision, n_slices)
        if getattr(self, "_prepared", None) is None or self._prepared_tag != key:
            n = C.c_uint64()
            _lib.check(_lib.load().hgt_conv_prepared_bytes(self.in_dim, self.out_dim, self.num_types, self.num_relations * n_slices,
                                                          self.n_heads, int(self.use_RTE), C.byref(n)), "hgt_conv_prepared_bytes")
            self._prepared = torch.empty(max(int(n.value), 256), dtype=torch.uint8, device=device)
            self._prepared_tag = key
            self._prepared_valid = False
        return self._prepared

    # ------------------------------------------------------------------------------------------
    def forward(self, node_inp, node_type, edge_index, edge_type, edge_time=None, plan=None, n_q_rows=None,
                phase_events=None, stage=0, proj=None, workspace=None, slices=None, block=None, out=None, proj_c24=None):
        """node_inp f32[N,in_dim], node_type i64[N], edge_index i64[2,E] (row 0 = source, row 1 =
        target; any strides), edge_type i64[E], edge_time i64[E] (needed iff use_RTE).
        Returns f32[N,out_dim] (or [n_q_rows,out_dim] when only the first n_q_rows nodes are targets).

        stage / proj: staged execution for pyhgt_amd.dist (hgt_conv_args.stage): 1 = projections of the own rows,
        2 = K|V of the rows in proj = (rows int32[n], offsets int32[T+1]) (one call per received halo chunk),
        3 = edge phase + update (returns the output); stages 1 and 2 return None.
        slices = (index, count) with stage 4 (count alone matters for stages 1/2 of the same forward): the plan numbers
        relations `source bucket * num_relations + relation` (count buckets; pyhgt_amd.dist builds it), stage 4 runs the edge
        phase of bucket `index` with the softmax state carried in the workspace, and the last bucket's call returns the output.
        block = (q_begin, q_end, item_begin, item_end) with stage 5: edge phase + fused node update of the TARGET BLOCK
        [q_begin, q_end) (q_begin a multiple of the plan tile; its logits work items from GraphPlan.tile_items()); `out` must be
        the [n_q_rows, out_dim] tensor all blocks of the forward write into (returned).  pyhgt_amd.dist orders the halo rows by
        the first block that needs them, so block b runs as soon as halo chunks 0..b are projected -- no state between blocks.
        proj_c24 = (wire uint8 tensor, first local row) with stage 2: the rows of `proj` are projected straight off the 24-bit
        wire buffer of the exchange (hgt_gather_rows_c24 format) instead of from node_inp.
        workspace: caller-owned uint8 device buffer (>= workspace_bytes(N, E)) instead of the per-(device, stream) scratch;
        staged callers MUST pass one (Q/K/V live in it between the stages)."""
        lib = _lib.load()
        if not node_inp.is_cuda:
            raise RuntimeError("pyhgt_amd.HGTConv runs only on a ROCm GPU tensor; there is no CPU fallback "
                               "(the CPU oracle lives under oracle/ and is test infrastructure)")
        needs_grad = torch.is_grad_enabled() and (node_inp.requires_grad or any(p.requires_grad for p in self.parameters()))
        if needs_grad and (stage != 0 or n_q_rows is not None):
            raise RuntimeError("pyhgt_amd: the backward pass covers HGTConv / DenseHGTConv on a single GPU (SURVEY.md section 8f-2); "
                               "staged multi-GPU forwards run under torch.no_grad() only")
        if node_inp.dtype != torch.float32:
            raise TypeError("node_inp must be float32 (the reference layer is fp32-only, conv.py:68-69)")
        if self.in_dim != self.out_dim:
            raise RuntimeError("HGTConv needs in_dim == out_dim for the skip connection (conv.py:131)")
        if self.use_RTE and edge_time is None:
            raise ValueError("use_RTE=True needs edge_time (conv.py:91-92)")
        N = node_inp.size(0)
        if node_inp.size(1) != self.in_dim or node_type.numel() != N:
            raise ValueError("node_inp must be [N, in_dim] and node_type [N]")
        x = node_inp.detach().contiguous()
        if plan is None:
            plan = GraphPlan.cached(node_type, edge_index, edge_type, edge_time if self.use_RTE else None,
                                    self.num_types, self.num_relations, n_q_rows)
        n_slices = 1 if slices is None else int(slices[1])
        if n_slices < 1 or (stage == 4 and not (0 <= int(slices[0]) < n_slices)):
            raise ValueError("slices must be (index, count) with 0 <= index < count")
        # staged (multi-GPU) calls of an "f16x3" layer run the "bf16x3" kernels (include/hgt_hip.h: precision 2 is whole-layer only)
        prec = "bf16x3" if (self.precision == "f16x3" and stage != 0) else self.precision
        if n_slices > 1 and (stage == 0 or stage == 3 or prec != "bf16x3" or self._UPDATE_MODE != 0):
            raise ValueError("sliced edge phases run as stages 1 / 2 / 4 of an HGTConv with a split precision")
        R_plan = self.num_relations * n_slices
        if plan.N != N or plan.T != self.num_types or plan.R != R_plan:
            raise ValueError("plan was built for a different graph / schema")
        strict = GraphPlan.STRICT if self.strict is None else self.strict
        plan.raise_if_bad(wait=bool(strict) and plan._bad is None, ignore_time=not self.use_RTE)
        NQ, E = plan.NQ, plan.E
        if needs_grad and not self.training and not HGTConv._warned_eval_grad:
            HGTConv._warned_eval_grad = True
            import warnings
            warnings.warn("pyhgt_amd.HGTConv: eval-mode forward with autograd enabled takes the differentiable path (unfused kernels, "
                          "intermediates kept: about 1.4x the inference time and several times its memory).  Wrap inference in "
                          "torch.no_grad() to get the fused inference kernels.", stacklevel=2)
        if needs_grad:
            # training / differentiable path (pyhgt_amd/autograd.py): same kernels, intermediates kept, hand-written backward;
            # dropout on the a_linear output in training mode only (conv.py:125)
            from .autograd import hgt_conv_train
            out = hgt_conv_train(self, plan, node_inp.float(), self._pack_parameters(grad=True),
                                 float(self.drop.p) if self.training else 0.0)
            return out                  # (self.att was set by the Function when keep_att is on)
        pk = self._pack_parameters()
        if pk["w_qkv"].device != x.device:
            raise RuntimeError("module parameters and node_inp are on different devices")
        nbytes = C.c_uint64()
        # (the item-parallel aggregation's scratch only where this call can take that kernel: whole-layer calls of a split precision)
        # (a caller-owned workspace may come without it: hgt_conv_forward then simply rules that kernel out)
        item_scratch = int(workspace is None and stage == 0 and prec != "fp32" and not (self.kernel_flags & _lib.HGT_FLAG_NO_ITEM_AGGREGATE))
        options = item_scratch | (2 if (self.kernel_flags & _lib.HGT_FLAG_DETERMINISTIC_HUBS) else 0)
        _lib.check(lib.hgt_conv_workspace_bytes_ex(N, E, self.in_dim, self.out_dim, self.num_types, R_plan,
                                                   self.n_heads, int(self.use_RTE), options, C.byref(nbytes)), "hgt_conv_workspace_bytes_ex")
        if workspace is not None:
            if workspace.dtype != torch.uint8 or workspace.device != x.device or workspace.numel() < nbytes.value:
                raise ValueError("workspace must be a uint8 tensor of >= %d bytes on %s" % (nbytes.value, x.device))
            ws = workspace
        else:
            if stage != 0:
                raise ValueError("staged execution keeps Q/K/V in the workspace between calls: pass workspace=")
            ws = _Workspace.get(x.device, nbytes.value)
        final = stage in (0, 3) or (stage == 4 and int(slices[0]) == n_slices - 1)
        if stage == 5:
            if block is None or out is None or out.shape != (NQ, self.out_dim) or out.dtype != torch.float32 or not out.is_contiguous():
                raise ValueError("stage 5 takes block=(q_begin, q_end, item_begin, item_end) and out=f32[n_q_rows, out_dim]")
            if self.keep_att:
                raise ValueError("keep_att is not available in the target-blocked multi-GPU schedule")
        else:
            out = torch.empty(NQ, self.out_dim, dtype=torch.float32, device=x.device) if final else None
        att = torch.empty(E, self.n_heads, dtype=torch.float32, device=x.device) if (self.keep_att and final) else None
        ntype = node_type.contiguous()

        a = _lib.HgtConvArgs()
        a.n_nodes, a.n_edges = N, E
        a.in_dim, a.out_dim, a.n_types, a.n_relations, a.n_heads = (self.in_dim, self.out_dim, self.num_types, R_plan, self.n_heads)
        a.use_norm, a.use_rte = int(self.use_norm), int(self.use_RTE)
        a.precision = {"fp32": 0, "bf16x3": 1, "f16x3": 2}[prec]
        a.want_att = int(self.keep_att)
        a.n_q_rows = NQ
        a.x, a.node_type, a.plan = _ptr(x), _ptr(ntype), plan.ptr
        a.w_qkv, a.b_qkv, a.w_a, a.b_a = _ptr(pk["w_qkv"]), _ptr(pk["b_qkv"]), _ptr(pk["w_a"]), _ptr(pk["b_a"])
        if n_slices > 1:      # one copy of the relation parameters per source bucket (relation id = bucket * R + relation)
            rep = pk.get("rel_rep")
            if rep is None or rep[0] != n_slices:
                rep = pk["rel_rep"] = (n_slices, pk["ratt"].repeat(n_slices, 1, 1, 1), pk["rmsg"].repeat(n_slices, 1, 1, 1),
                                       pk["rpri"].repeat(n_slices, 1))
            a.relation_att, a.relation_msg, a.relation_pri = _ptr(rep[1]), _ptr(rep[2]), _ptr(rep[3])
        else:
            a.relation_att, a.relation_msg, a.relation_pri = _ptr(pk["ratt"]), _ptr(pk["rmsg"]), _ptr(pk["rpri"])
        a.ln_w, a.ln_b = _ptr(pk["ln_w"]), _ptr(pk["ln_b"])
        self._set_update_args(a, pk)
        a.rte_emb, a.rte_w, a.rte_b = _ptr(pk.get("rte_emb")), _ptr(pk.get("rte_w")), _ptr(pk.get("rte_b"))
        a.workspace, a.workspace_bytes = _ptr(ws), ws.numel()
        a.out, a.att_out = _ptr(out) if (final or stage == 5) else _ptr(ws), _ptr(att)   # stages 1/2 write no output (non-NULL placeholder)
        a.want_att = int(self.keep_att and final)
        a.stage = int(stage)
        if stage == 4:
            a.slice_index, a.slice_count = int(slices[0]), n_slices
        if stage == 5:
            a.q_begin, a.q_end, a.item_begin, a.item_end = (int(v) for v in block)
        a.plan_no_hubs = int(plan.no_hubs) | (2 if plan.no_unknown_rows else 0)
        a.flags = int(self.kernel_flags) | int(HGTConv.EXTRA_KERNEL_FLAGS)
        prep = self._prepared_buffer(x.device, n_slices, prec)          # after _pack_parameters: a re-pack has invalidated it
        a.prepared, a.prepared_bytes, a.prepared_valid = _ptr(prep), prep.numel(), int(self._prepared_valid)
        if stage == 2:
            rows, off = proj
            if rows.dtype != torch.int32 or off.dtype != torch.int32 or off.numel() != self.num_types + 1:
                raise TypeError("proj must be (int32 rows, int32 offsets[T+1])")
            a.proj_rows, a.proj_off, a.proj_n = _ptr(rows), _ptr(off), int(rows.numel())
            if proj_c24 is not None:
                wire, row0 = proj_c24
                if wire.dtype != torch.uint8 or not wire.is_contiguous() or wire.device != x.device:
                    raise TypeError("proj_c24 must be (contiguous uint8 device tensor of 24-bit rows, first local row)")
                a.proj_c24, a.proj_c24_row0 = _ptr(wire), int(row0)
        if phase_events is not None:      # ctypes array of HGT_N_PHASE_EVENTS hipEvent_t (bench.py instrumentation)
            a.phase_events = C.cast(phase_events, C.c_void_p)
        _lib.check(lib.hgt_conv_forward(C.byref(a), _stream()), "hgt_conv_forward")
        if final:
            self.att = att
            self._prepared_valid = True                 # every image was written by this forward (or an earlier one)
        return out

    def workspace_bytes(self, n_nodes, n_edges, n_slices=1, staged=True):
        """Size of a caller-owned workspace (staged callers: pyhgt_amd.dist).  staged=False adds the scratch of the item-parallel
        aggregation that only whole-layer calls on small graphs use."""
        n = C.c_uint64()
        _lib.check(_lib.load().hgt_conv_workspace_bytes_ex(int(n_nodes), int(n_edges), self.in_dim, self.out_dim, self.num_types,
                                                           self.num_relations * int(n_slices), self.n_heads, int(self.use_RTE),
                                                           int(not staged) | (2 if (self.kernel_flags & _lib.HGT_FLAG_DETERMINISTIC_HUBS) else 0),
                                                           C.byref(n)), "hgt_conv_workspace_bytes_ex")
        return int(n.value)

    def __repr__(self):
        return '{}(in_dim={}, out_dim={}, num_types={}, num_types={})'.format(
            self.__class__.__name__, self.in_dim, self.out_dim, self.num_types, self.num_relations)


class DenseHGTConv(HGTConv):
    """The reference's DenseHGTConv (conv.py:143-280): message() is HGTConv's (conv.py:197-248), update() differs
    (conv.py:250-274):

        y1  = LayerNorm_t(a_linear_t(agg) + x)          no gelu on the aggregate, plain residual, no `skip` gate
        out = out_norm(out_linear(gelu(mid_linear(y1))) + y1)            one dense layer shared by all types

    Same parameter names as the reference (no `skip`; mid_linear, out_linear, out_norm added)."""

    _UPDATE_MODE = 1

    def _init_update_parameters(self, num_types, out_dim):
        self.mid_linear = nn.Linear(out_dim, out_dim * 2)               # conv.py:189
        self.out_linear = nn.Linear(out_dim * 2, out_dim)               # conv.py:190
        self.out_norm = nn.LayerNorm(out_dim)                           # conv.py:191

    def _pack_update_parameters(self, grad=False):
        f = (lambda t: t.float().contiguous()) if grad else (lambda t: t.detach().float().contiguous())
        return dict(mid_w=f(self.mid_linear.weight), mid_b=f(self.mid_linear.bias), out_w=f(self.out_linear.weight),
                    out_b=f(self.out_linear.bias), out_ln_w=f(self.out_norm.weight), out_ln_b=f(self.out_norm.bias))

    def _set_update_args(self, a, pk):
        a.update_mode = self._UPDATE_MODE
        a.skip = None
        a.mid_w, a.mid_b, a.out_w, a.out_b = _ptr(pk["mid_w"]), _ptr(pk["mid_b"]), _ptr(pk["out_w"]), _ptr(pk["out_b"])
        a.out_ln_w, a.out_ln_b = _ptr(pk["out_ln_w"]), _ptr(pk["out_ln_b"])


class GeneralConv(nn.Module):
    """The reference's layer dispatcher (conv.py:303-323) for the in-scope convolutions: 'hgt' and 'dense_hgt'
    ('gcn' / 'gat' are PyG's own layers and out of scope)."""

    def __init__(self, conv_name, in_hid, out_hid, num_types, num_relations, n_heads, dropout, use_norm=True, use_RTE=True):
        super().__init__()
        self.conv_name = conv_name
        if conv_name == 'hgt':
            self.base_conv = HGTConv(in_hid, out_hid, num_types, num_relations, n_heads, dropout, use_norm, use_RTE)
        elif conv_name == 'dense_hgt':
            self.base_conv = DenseHGTConv(in_hid, out_hid, num_types, num_relations, n_heads, dropout, use_norm, use_RTE)
        else:
            raise NotImplementedError("pyhgt_amd implements conv_name 'hgt' and 'dense_hgt'; %r is outside the accelerated path"
                                      % conv_name)

    def forward(self, meta_xs, node_type, edge_index, edge_type, edge_time):
        return self.base_conv(meta_xs, node_type, edge_index, edge_type, edge_time)


def install_into(conv_module):
    """Plug this implementation into the reference: `import pyHGT.conv as c; install_into(c)` makes
    the reference's GeneralConv / model.GNN construct pyhgt_amd.HGTConv for conv_name='hgt'
    (GeneralConv looks the class up in its module globals at construction time, conv.py:308)."""
    conv_module.HGTConv = HGTConv
    conv_module.DenseHGTConv = DenseHGTConv
    return conv_module
