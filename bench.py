#!/usr/bin/env python
"""HGTConv forward benchmark (BASELINE.json metric: edges/s + achieved HBM GB/s).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one HGTConv.forward (eval mode, fp32, plan cached -- the reference reuses one sampled
graph for all layers and `repeat` optimisation steps, model.py:78-79 / train_paper_field.py:240)
over a synthetic typed graph that is already resident in HBM.  After the timed region the output
of the benchmarked configuration itself is checked against the fp64 CPU oracle on ~2000 sampled
target rows (`parity_max_abs_err`; the run fails above 1e-4), and secondary figures are taken:
the other precision, the plan-included rate, the median next to the mean.
  N = 1 : BASELINE.json configs[1]: T=4 R=8, 1M nodes / 10M edges, d=256, H=8, 4-argument form
          (use_RTE=False), LayerNorm on.
  N > 1 : weak scaling -- every rank owns 1M target nodes and their 10M in-edges; sources are
          uniform over all N*1M nodes; halo source rows (exact fp32) come over one RCCL all-to-all per
          step (pyhgt_amd/dist.py); value = total edges of all ranks / max-over-ranks time.
Rank 0 prints ONE JSON line (contract in the task statement) carrying `roofline` (dominant kernel,
timed with HIP events on the launch stream during the timed steps) and `cpu_baseline` (the CPU
port of the reference algorithm, oracle/hgt_oracle.py, on a bounded sample).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
PHASES = ["project_qkv", "edge_logits", "edge_softmax", "edge_aggregate", "a_linear", "node_update"]


def algorithmic_bytes(N, E, d, use_rte):
    """SURVEY.md section 8(d) minimal-traffic model, split per kernel:
    whole layer = E*(8d+24) + N*(28d+8) (+8E with RTE)."""
    per = {
        "project_qkv": N * (16 * d + 8),             # read x, write Q,K,V, node_type
        "edge_logits": E * (4 * d + 12 + (4 if use_rte else 0)) + N * 4 * d,   # K row + ids per edge, Q row per target
        "edge_softmax": 0,
        "edge_aggregate": E * (4 * d + 12 + (4 if use_rte else 0)),             # V row + ids per edge
        "a_linear": 0,
        "node_update": N * 8 * d,                    # read x (skip), write out
    }
    per["layer"] = sum(per.values())
    return per


class HipEvents:
    """hipEvent_t through ctypes on libamdhip64 (events are recorded on the launch stream by
    hgt_conv_forward itself, include/hgt_hip.h `phase_events`)."""

    def __init__(self):
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        self.hip.hipEventSynchronize.argtypes = [C.c_void_p]

    def make_set(self, n):
        arr = (C.c_void_p * n)()
        for i in range(n):
            ev = C.c_void_p()
            assert self.hip.hipEventCreate(C.byref(ev)) == 0
            arr[i] = ev
        return arr

    def elapsed_ms(self, a, b):
        ms = C.c_float()
        self.hip.hipEventSynchronize(b)
        rc = self.hip.hipEventElapsedTime(C.byref(ms), a, b)
        assert rc == 0, "hipEventElapsedTime rc=%d" % rc
        return float(ms.value)


def host_mem_gb():
    """(total, available) host RAM in GB from /proc/meminfo, or (None, None)."""
    try:
        info = {l.split(":")[0]: int(l.split()[1]) for l in open("/proc/meminfo") if ":" in l}
        return round(info["MemTotal"] / 2 ** 20, 1), round(info.get("MemAvailable", 0) / 2 ** 20, 1)
    except (OSError, KeyError, ValueError, IndexError):
        return None, None


def cpu_baseline_measure(d, H, T, R, threads, N=100_000, E=1_000_000, max_s=10.0):
    """One thread setting of the CPU leg (child process): the reference-cost CPU port (oracle.forward_meta_relation_port:
    per-meta-relation masks, per-EDGE projections, like conv.py:64-111) on the SURVEY.md section 8(d) fallback sample of the
    c2 recipe, E = 1M / N = 100k (the verbatim reference cannot travel to the GPU box; c2 itself needs 44 GB RSS and 75 s per
    forward)."""
    from oracle import hgt_oracle as O
    from pyhgt_amd.synth import synthetic_typed_graph
    torch.set_num_threads(threads)
    sd = O.make_state_dict(d, d, T, R, H, True, False, seed=0)
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=0)
    with torch.no_grad():
        O.forward_meta_relation_port(sd, T, R, H, x[:2000], nt[:2000], ei[:, :0], et[:0], None, use_RTE=False)  # warm
        t0 = time.time()
        reps = 0
        while reps < 1 or (time.time() - t0 < max_s and reps < 4):
            O.forward_meta_relation_port(sd, T, R, H, x, nt, ei, et, None, use_RTE=False)
            reps += 1
        dt = (time.time() - t0) / reps
    return {"threads": threads, "seconds_per_forward": dt, "forwards": reps, "edges": E, "nodes": N}


def cpu_baseline(d, H, T, R, limit_s=75, full=False):
    """The CPU leg, each thread setting in a child process with a hard time limit so that it can never stall the bench line:
    32 threads, 64 threads (the cap the round-3 review asked for), and os.cpu_count() threads on hosts with at most 64 of them.  The
    fastest setting is the reported value.  The default sample is E = 1M / N = 100k: the task statement bounds the CPU leg to
    10 - 30 s of work so that the default run finishes in minutes; `--cpu-baseline-full` runs the port once at the full c2 size
    (needs ~48 GB of RAM and minutes of CPU time) and reports it as `full_c2`."""
    import subprocess
    cores = os.cpu_count() or 1
    ram_total, ram_avail = host_mem_gb()
    cpu_model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    runs, notes = {}, []
    # os.cpu_count() threads only on hosts where that is a sane setting: the port is a chain of eager torch ops, and on the GPU
    # box (256 hardware threads) the over-subscribed 1M-edge scatter never finished a forward inside the 75 s limit
    settings = {min(32, cores), min(64, cores)} | ({cores} if cores <= 64 else set())
    for threads in sorted(settings):
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--cpu-threads", str(threads), "--dim", str(d),
               "--heads", str(H), "--types", str(T), "--relations", str(R)]
        try:
            res = subprocess.run(cmd, capture_output=True, text=True, timeout=limit_s)
            got = None
            for line in reversed(res.stdout.strip().splitlines()):
                if line.startswith("{"):
                    got = json.loads(line)
                    break
            if got is None:
                notes.append("%d threads: failed (%s)" % (threads, res.stderr[-120:].replace("\n", " ")))
            else:
                runs[threads] = got
                notes.append("%d threads: %d forwards of %.2f s" % (threads, got["forwards"], got["seconds_per_forward"]))
        except subprocess.TimeoutExpired:
            notes.append("%d threads: no forward finished within %d s" % (threads, limit_s))
    if not runs:
        return {"value": None, "unit": "edges/s", "cores": None, "kind": "port", "sample": "; ".join(notes), "host_cores": cores}
    best = min(runs, key=lambda t: runs[t]["seconds_per_forward"])
    E = runs[best]["edges"]
    full_c2 = None
    if full:      # the port once at the FULL c2 size, fastest thread setting (~44 GB RSS for the verbatim reference)
        if ram_avail is not None and ram_avail < 64:
            full_c2 = {"skipped": "only %.0f GB of host RAM available" % ram_avail}
        else:
            cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--cpu-threads", str(best), "--dim", str(d),
                   "--heads", str(H), "--types", str(T), "--relations", str(R), "--cpu-full-size"]
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=420)
                got = [json.loads(l) for l in r.stdout.strip().splitlines() if l.startswith("{")]
                full_c2 = ({"edges_per_s": got[-1]["edges"] / got[-1]["seconds_per_forward"], "seconds_per_forward": got[-1]["seconds_per_forward"],
                            "threads": best, "nodes": got[-1]["nodes"], "edges": got[-1]["edges"]} if got else
                           {"failed": r.stderr[-200:].replace("\n", " ")})
            except subprocess.TimeoutExpired:
                full_c2 = {"failed": "no forward finished within 420 s"}
    return {"value": E / runs[best]["seconds_per_forward"], "unit": "edges/s", "cores": best, "kind": "port",
            "host_ram_gb": ram_total, "host_ram_available_gb": ram_avail, "full_c2": full_c2,
            "why_a_sample": "`value` is the bounded sample the task statement asks for (10 - 30 s of CPU work); `full_c2` is ONE forward of the "
                            "port at the full c2 size, run in the same line on hosts with >= 96 GB of RAM available",
            "sample": "c2 recipe at E=1M / N=100k (SURVEY 8d fallback; T%d R%d d=%d H=%d, use_RTE=False), "
                      "oracle.forward_meta_relation_port; %s; fastest: %d threads" % (T, R, d, H, "; ".join(notes), best),
            "by_threads": {str(t): r["edges"] / r["seconds_per_forward"] for t, r in runs.items()},
            "cpu_model": cpu_model, "host_cores": cores}


def parity_check(layer_sd, out, x, node_type, edge_index, edge_type, edge_time, T, R, H, use_rte, n_q_rows=None):
    """After the timed region: compare the rows of `out` for ~2000 sampled targets (type-boundary tiles, first / last ragged
    tile, max in-degree rows, random rows) with the fp64 CPU oracle run on the sub-graph induced by ALL their in-edges
    (pyhgt_amd.synth.induced_in_neighbourhood: exact for those rows).  The oracle is the checker, never the thing measured."""
    from oracle import hgt_oracle as O
    from pyhgt_amd.synth import pick_check_targets, induced_in_neighbourhood
    nq = int(out.size(0)) if n_q_rows is None else int(n_q_rows)
    tg = pick_check_targets(node_type[:nq], edge_index[1])
    xs, nts, eis, ets, tms, pos = induced_in_neighbourhood(x, node_type, edge_index, edge_type, edge_time, tg)
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    ref = O.forward_closed_form(layer_sd, T, R, H, xs, nts, eis, ets, tms, use_norm=True, use_RTE=use_rte, dtype=torch.float64)
    err = (out[tg].cpu().double() - ref[pos]).abs().max().item()
    return {"max_abs_err": err, "rows": int(tg.numel()), "edges": int(eis.size(1)), "sub_nodes": int(xs.size(0)),
            "max_in_degree": int(torch.bincount(eis[1]).max()) if eis.numel() else 0}


def kernel_sources_sha16():
    """sha256 over the kernel sources + the ABI header: profiles/*_pmc_summary.json records it (tools/pmc_summary.py) so that
    counter-derived figures quoted in the bench line can be flagged when they were taken from a different build."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "pyhgt_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "pyhgt_amd", "csrc", "*.h")) +
                       [os.path.join(ROOT, "include", "hgt_hip.h")]):
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def wall_us(fn, iters, warm, repeats=3):
    """Host wall time per call, the FASTEST of `repeats` timed loops: these are 50 - 500 us measurements on a shared host, and a
    single loop is now and then inflated 2 - 3x by a host-side stall (seen once in a round-5 line: 200 us where every other run of the
    same build says 67)."""
    for _ in range(warm):
        fn()
    best = None
    for _ in range(repeats):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / iters * 1e6
        best = t if best is None else min(best, t)
    return best


def graph_replay_us(fn, iters=200, repeats=3):
    """One call of `fn` captured in a hipGraph (torch.cuda.CUDAGraph on the capture stream the library launches on) and replayed: host
    wall time per replay, fastest of `repeats` loops -- what a sampled-batch layer costs without the per-launch host work.  Returns
    (us, None) or (None, reason) when the call cannot be captured."""
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(g):
            fn()
        torch.cuda.synchronize()
        for _ in range(10):
            g.replay()
        best = None
        for _ in range(repeats):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                g.replay()
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / iters * 1e6
            best = t if best is None else min(best, t)
        del g
        return best, None
    except Exception as e:      # (a capture that fails must not take the rest of the line with it)
        try:
            torch.cuda.synchronize()
        except Exception:
            pass
        return None, ("%s: %s" % (type(e).__name__, e))[:300]


def small_regime(dev):
    """BASELINE.json configs[0], [2], [4] in the judged line (latency regime: the reference's real workloads are sampled
    sub-graphs of a few thousand nodes).  Wall-clock per call incl. launch overhead, plan cached; every entry carries its parity:
      c1       configs[0] at full size = tests/golden/c1_full.npz (inputs, parameters AND the reference's own output)
      c3       configs[2] surrogate: sampler-shaped ogbn-mag batch (T=4, R=9, d=256, H=8, RTE), one layer; fp64 oracle
      c5       configs[4] surrogate: sampler-shaped OAG batch, GNN in 1169 -> 400, 33 relations, 2 layers; rows of the verbatim
               reference GNN's output (tests/golden/gnn_oag2.npz)
      mag4     the published 4-layer n_hid=512 ogbn-mag model on the c3-sized batch (tests/golden/gnn_mag4.npz)"""
    import numpy as np
    from oracle import hgt_oracle as O
    from oracle.gen_golden_gnn import GNN_CASES, build_batch
    from pyhgt_amd import HGTConv, GNN, GraphPlan
    from pyhgt_amd.sampled import synthetic_sampled_batch, to_torch_layout, to_device_graph
    res = {"timing": "host wall time per call, fastest of 3 timed loops (wall_us; rounds 1-4 reported the mean of one loop)"}
    graph_cases = []      # (entry, divisor, fn): measured at the very end, see graph_replay_us
    gold = os.path.join(ROOT, "tests", "golden")
    # ---- c1
    z = np.load(os.path.join(gold, "c1_full.npz"))
    N, E, d, H, T, R, use_norm, use_rte = [int(v) for v in z["meta"]][:8]
    sd = {k[len("param::"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param::")}
    x = torch.from_numpy(z["node_feature"]).to(dev)
    nt = torch.from_numpy(z["node_type"]).long().to(dev)
    ei = torch.from_numpy(z["edge_index"]).long().to(dev)
    et = torch.from_numpy(z["edge_type"]).long().to(dev)
    tm = torch.from_numpy(z["edge_time"]).long().to(dev)
    ref = torch.from_numpy(z["out"])
    res["c1"] = {"workload": "BASELINE.json configs[0]: T=%d R=%d N=%d E=%d d=%d H=%d use_RTE=%s (tests/golden/c1_full.npz: the reference's "
                             "own output is the checker)" % (T, R, N, E, d, H, bool(use_rte))}
    for prec in ("bf16x3", "f16x3", "fp32"):
        layer = HGTConv(d, d, T, R, H, 0.2, bool(use_norm), bool(use_rte), precision=prec).eval()
        layer.load_state_dict(sd)
        layer = layer.to(dev)
        plan = GraphPlan(nt, ei, et, tm if use_rte else None, T, R)
        with torch.no_grad():
            out = layer(x, nt, ei, et, tm, plan=plan)
            us = wall_us(lambda: layer(x, nt, ei, et, tm, plan=plan), 200, 20)
        alg = algorithmic_bytes(N, E, d, bool(use_rte))["layer"]
        res["c1"][prec] = {"us_per_layer": us, "edges_per_s": E / (us * 1e-6), "parity_max_abs_err": float((out.cpu() - ref).abs().max()),
                           "layer_frac": round(alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
    # ---- c3
    batch = synthetic_sampled_batch("mag", n_seed=128, width=128, depth=6, feat_dim=256, mean_degree=4.0, seed=3)
    xc, ntc, tmc, eic, etc_, _, edge_dict = to_torch_layout(*batch)
    T, R, d, H = 4, len(edge_dict), 256, 8
    N, E = int(ntc.numel()), int(etc_.numel())
    sd = O.make_state_dict(d, d, T, R, H, True, True, seed=77)
    ref = O.forward_closed_form(sd, T, R, H, xc, ntc, eic, etc_, tmc, use_norm=True, use_RTE=True, dtype=torch.float64)
    x, nt, tm, ei, et = [t.to(dev) for t in (xc, ntc, tmc, eic, etc_)]
    dg = to_device_graph(*batch, device=dev)
    src32, dst32, time32 = dg[3][0].int().contiguous(), dg[3][1].int().contiguous(), dg[2].int().contiguous()
    rel_ptr = torch.searchsorted(dg[4], torch.arange(R + 1, device=dev)).int()
    type_off = torch.searchsorted(dg[1], torch.arange(T + 1, device=dev)).int()
    res["c3"] = {"workload": "BASELINE.json configs[2] surrogate: sampler-shaped ogbn-mag batch (sample_depth 6, sample_width 128), "
                             "T=%d R=%d N=%d E=%d d=%d H=%d use_RTE=True, one layer" % (T, R, N, E, d, H),
                 "plan_build_us": wall_us(lambda: GraphPlan(nt, ei, et, tm, T, R), 50, 5),
                 "plan_from_sorted_us": wall_us(lambda: GraphPlan.from_sorted(dg[1], dg[3], dg[4], dg[2], src32, dst32, time32, rel_ptr,
                                                                              type_off, T, R), 50, 5)}
    alg = algorithmic_bytes(N, E, d, True)["layer"]
    for prec in ("bf16x3", "f16x3", "fp32"):
        layer = HGTConv(d, d, T, R, H, 0.2, True, True, precision=prec).eval()
        layer.load_state_dict(sd)
        layer = layer.to(dev)
        plan = GraphPlan(nt, ei, et, tm, T, R)
        with torch.no_grad():
            out = layer(x, nt, ei, et, tm, plan=plan)
            us = wall_us(lambda: layer(x, nt, ei, et, tm, plan=plan), 200, 20)
        res["c3"][prec] = {"us_per_layer": us, "edges_per_s": E / (us * 1e-6),
                           "parity_max_abs_err": float((out.cpu().double() - ref).abs().max()),
                           "layer_frac": round(alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
        if prec == "f16x3":      # (the default precision)
            # every sampled batch of the reference's training loop is a NEW graph (round-3 advisor note): one layer with the plan
            # of the sampler-ordered hand-off built inside the timed call
            def new_graph_layer():
                p2 = GraphPlan.from_sorted(dg[1], dg[3], dg[4], dg[2], src32, dst32, time32, rel_ptr, type_off, T, R)
                return layer(x, nt, ei, et, tm, plan=p2)
            with torch.no_grad():
                res["c3"][prec]["us_per_layer_incl_plan_from_sorted"] = wall_us(new_graph_layer, 100, 10)
            graph_cases.append((res["c3"][prec], 1, (lambda layer=layer, x=x, nt=nt, ei=ei, et=et, tm=tm, plan=plan: layer(x, nt, ei, et, tm, plan=plan))))
    # ---- the scripts' DEFAULT batch (train_ogbn_mag.py:44-46: sample_width 520): N ~ 8.4k, E ~ 150k, one layer
    batch = synthetic_sampled_batch("mag", n_seed=128, width=520, depth=6, feat_dim=256, mean_degree=4.0, seed=5)
    xc, ntc, tmc, eic, etc_, _, edge_dict = to_torch_layout(*batch)
    N, E = int(ntc.numel()), int(etc_.numel())
    ref = O.forward_closed_form(sd, T, R, H, xc, ntc, eic, etc_, tmc, use_norm=True, use_RTE=True, dtype=torch.float64)
    x, nt, tm, ei, et = [t.to(dev) for t in (xc, ntc, tmc, eic, etc_)]
    layer = HGTConv(d, d, T, R, H, 0.2, True, True, precision="f16x3").eval()
    layer.load_state_dict(sd)
    layer = layer.to(dev)
    plan = GraphPlan(nt, ei, et, tm, T, R)
    with torch.no_grad():
        out = layer(x, nt, ei, et, tm, plan=plan)
        us = wall_us(lambda: layer(x, nt, ei, et, tm, plan=plan), 200, 20)
    res["c3_width520"] = {"workload": "ogbn-mag script default batch (sample_depth 6, sample_width 520; train_ogbn_mag.py:44-46), surrogate: "
                                      "T=%d R=%d N=%d E=%d d=%d H=%d use_RTE=True, one layer" % (T, R, N, E, d, H),
                          "plan_build_us": wall_us(lambda: GraphPlan(nt, ei, et, tm, T, R), 30, 5),
                          "f16x3": {"us_per_layer": us, "edges_per_s": E / (us * 1e-6),
                                     "parity_max_abs_err": float((out.cpu().double() - ref).abs().max()),
                                     "layer_frac": round(algorithmic_bytes(N, E, d, True)["layer"] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}}
    # ---- c5 and the published 4-layer model: whole GNN forwards against rows of the verbatim reference GNN
    for key, name in (("c5", "gnn_oag2"), ("mag4", "gnn_mag4")):
        c = GNN_CASES[name]
        z = np.load(os.path.join(gold, name + ".npz"))
        _, (xc, ntc, tmc, eic, etc_, _, _) = build_batch(c)
        sd = O.make_gnn_state_dict(c["in_dim"], c["n_hid"], c["T"], c["R"], c["H"], c["n_layers"], c["prev_norm"], c["last_norm"],
                                   c["use_RTE"], seed=c["seed"])
        rows = torch.from_numpy(z["rows"]).long()
        want = torch.from_numpy(z["layers"][-1])
        args = [t.to(dev) for t in (xc, ntc, tmc, eic, etc_)]
        res[key] = {"workload": "%s: sampler-shaped %s batch, N=%d E=%d, GNN in_dim %d -> n_hid %d, T=%d R=%d H=%d, %d layers "
                                "(tests/golden/%s.npz: rows of the verbatim reference GNN's output are the checker)" % (
                                    "BASELINE.json configs[4] surrogate" if key == "c5" else "published ogbn-mag model on a configs[2]-sized batch",
                                    c["schema"], ntc.numel(), etc_.numel(), c["in_dim"], c["n_hid"], c["T"], c["R"], c["H"], c["n_layers"], name)}
        for prec in ("bf16x3", "f16x3", "fp32"):
            gnn = GNN(c["in_dim"], c["n_hid"], c["T"], c["R"], c["H"], c["n_layers"], 0.2, "hgt", c["prev_norm"], c["last_norm"],
                      c["use_RTE"]).eval()
            gnn.load_state_dict(sd)
            gnn = gnn.to(dev)
            for gc in gnn.gcs:
                gc.base_conv.precision = prec
            with torch.no_grad():
                out = gnn(*args)
                us = wall_us(lambda: gnn(*args), 100, 10)
            res[key][prec] = {"us_per_forward": us, "us_per_layer": us / c["n_layers"], "edges_per_s_per_layer": etc_.numel() * c["n_layers"] / (us * 1e-6),
                              "parity_max_abs_err": float((out[rows.to(dev)].cpu() - want).abs().max())}
            if prec == "f16x3":      # a new graph per forward (plan cache emptied inside the timed call: radix plan build included)
                def new_graph_forward():
                    GraphPlan.clear_cache()
                    return gnn(*args)
                with torch.no_grad():
                    res[key][prec]["us_per_forward_new_graph"] = wall_us(new_graph_forward, 50, 5)
                if key == "mag4":
                    graph_cases.append((res[key][prec], c["n_layers"], (lambda gnn=gnn, args=args: gnn(*args))))
    # hipGraph replay next to the eager numbers (round-4 review): the same calls, captured once, replayed
    for entry, n_layers, fn in graph_cases:
        us, why = graph_replay_us(fn)
        entry["us_per_layer_graph_replay"] = None if us is None else us / n_layers
        if why:
            entry["graph_replay_note"] = why
    GraphPlan.clear_cache()
    return res


def replicas_c5(args, world, rank, dev, backend_name):
    """SURVEY.md section 8e, last row: BASELINE.json configs[4] does not shard -- a sampled batch is a few thousand nodes -- so N
    GPUs run N independent replicas (the reference prepares n_batch independent sub-graphs per epoch,
    OAG/train_paper_field.py:145-153).  A step = one 2-layer GNN forward (in 1169 -> 400, 33 relations, 8 heads) on the rank's
    next batch out of a pool of pre-built sampler-shaped batches (device-side hand-off, plans registered); no collective in
    the data path; value = layer-edges of all ranks / max-over-ranks time."""
    from pyhgt_amd import GNN, GraphPlan
    from pyhgt_amd.sampled import synthetic_sampled_batch, to_device_graph
    in_dim, n_hid, T, R, H, L, pool = 1169, 400, 5, 33, 8, 2, 4
    GraphPlan.CACHE_SIZE = pool
    torch.manual_seed(0)
    gnn = GNN(in_dim, n_hid, T, R, H, L, 0.2, "hgt", False, False, True).eval().to(dev)
    for gc in gnn.gcs:
        gc.base_conv.precision = args.precision
    batches = []
    for b in range(pool):
        raw = synthetic_sampled_batch("oag", n_seed=256, width=128, depth=6, feat_dim=in_dim, mean_degree=1.2, seed=1000 * rank + b)
        dg = to_device_graph(*raw, device=dev)
        batches.append((dg[0], dg[1], dg[2], dg[3], dg[4]))
    edges = [int(b[4].numel()) for b in batches]
    if world > 1:
        import torch.distributed as dist
        barrier = dist.barrier
    else:
        barrier = lambda: None
    with torch.no_grad():
        for i in range(args.warmup):
            out = gnn(*batches[i % pool])
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = gnn(*batches[i % pool])
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    my_edges = float(sum(edges[i % pool] for i in range(args.steps)) * L)
    tot = torch.tensor([elapsed, my_edges], device=dev if backend_name in ("nccl", "none") else "cpu", dtype=torch.float64)
    if world > 1:
        import torch.distributed as dist
        tmax = tot[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        esum = tot[1:].clone()
        dist.all_reduce(esum, op=dist.ReduceOp.SUM)
        elapsed, all_edges = float(tmax.item()), float(esum.item())
    else:
        all_edges = my_edges
    if rank == 0:
        # parity of the replica's model path against the verbatim reference GNN (same shape, golden rows): after the timed region
        parity = None
        if not args.no_parity:
            import numpy as np
            from oracle import hgt_oracle as O
            from oracle.gen_golden_gnn import GNN_CASES, build_batch
            c = GNN_CASES["gnn_oag2"]
            z = np.load(os.path.join(ROOT, "tests", "golden", "gnn_oag2.npz"))
            _, (xc, ntc, tmc, eic, etc_, _, _) = build_batch(c)
            g2 = GNN(in_dim, n_hid, T, R, H, L, 0.2, "hgt", False, False, True).eval()
            g2.load_state_dict(O.make_gnn_state_dict(in_dim, n_hid, T, R, H, L, False, False, True, seed=c["seed"]))
            g2 = g2.to(dev)
            for gc in g2.gcs:
                gc.base_conv.precision = args.precision
            with torch.no_grad():
                o2 = g2(*[t.to(dev) for t in (xc, ntc, tmc, eic, etc_)])
            rows = torch.from_numpy(z["rows"]).long().to(dev)
            parity = float((o2[rows].cpu() - torch.from_numpy(z["layers"][-1])).abs().max())
        line = {"metric": "GNN forward edges/sec (independent sampled batches)", "value": all_edges / elapsed, "unit": "edges/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": {"bf16x3": "f32 (typed linears and relation transforms as 3-term split-bf16 MFMA, fp32 accumulate)",
                          "f16x3": "f32 (typed linears and relation transforms as 3-term fp16 hi/lo MFMA, fp32 accumulate)"}.get(args.precision, "f32"),
                "data": "synthetic",
                "config": {"workload": "BASELINE.json configs[4] surrogate, replicas only: sampler-shaped OAG batches (T=5, R=33, ~4.1k nodes / "
                                       "~41k edges), 2-layer GNN in_dim 1169 -> n_hid 400, 8 heads, use_RTE=True, device-side hand-off, plans cached; "
                                       "one independent replica per GPU, no collective in the data path",
                           "batches_per_s": world * args.steps / elapsed, "edges_counted": "edges x layers",
                           "parallelism": "replicas x%d" % world, "backend": backend_name + (" (RCCL, barrier / timing only)" if backend_name == "nccl" else ""),
                           "precision": args.precision},
                "parity_max_abs_err": parity, "roofline": None, "cpu_baseline": None}
        print(json.dumps(line))
        if parity is not None and not (parity <= 1e-4):
            sys.stderr.write("PARITY FAILURE (> 1e-4 against the reference GNN golden)\n")
            sys.exit(3)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def configs3_share(dev, world, rank, Nl, El, T, R, use_rte, locality, skew, keep_global_types=False):
    """BASELINE.json configs[3] recipe: ONE global graph of world x (Nl nodes, El edges), generated identically on every rank (device
    generator, same seed), cut by the in-edge-balanced partitioner of pyhgt_amd.dist; returns this rank's share.
    locality p: a fraction p of the sources is drawn from the target's own natural partition (the 'schema/locality' variant of
    SURVEY 8e), the rest uniformly from ALL nodes; 0 = the uniform worst case (1 / world of the edges stay local)."""
    from pyhgt_amd.dist import partition
    gg = torch.Generator(device=dev).manual_seed(1234)
    Ng, Eg = Nl * world, El * world
    node_type_g = torch.randint(0, T, (world, Nl), generator=gg, device=dev).sort(dim=1).values.reshape(-1)
    dst_g = torch.randint(0, Ng, (Eg,), generator=gg, device=dev)
    if skew > 0.0:
        u = torch.rand(Eg, generator=gg, device=dev)
        dst_g = (dst_g // Nl) * Nl + (Nl * u ** (1.0 / (1.0 - skew))).long().clamp(0, Nl - 1)
    src_g = torch.randint(0, Ng, (Eg,), generator=gg, device=dev)
    if locality > 0.0:
        near = torch.rand(Eg, generator=gg, device=dev) < locality
        src_g = torch.where(near, (dst_g // Nl) * Nl + src_g % Nl, src_g)
        del near
    et_g = torch.randint(0, R, (Eg,), generator=gg, device=dev)
    tm_g = torch.randint(0, 240, (Eg,), generator=gg, device=dev) if use_rte else None
    share = partition(node_type_g, torch.stack([src_g, dst_g]), et_g, tm_g, world, rank)
    share["edge_ids"] = None
    del src_g, dst_g, et_g, tm_g
    torch.cuda.empty_cache()
    return share, (node_type_g if keep_global_types else None)


XGMI_LINK_GBS = 76.8      # per direction and peer link (7 peers per GPU: 538 GB/s egress), /opt/skills/guides/MI355X_MICROARCH.md / task statement


def stage_times(timeline_sets):
    """Mean milliseconds per stage label from the (label, event) marks PartitionedGraph.forward left on the compute stream."""
    acc, n = {}, 0
    for marks in timeline_sets:
        if len(marks) < 2:
            continue
        n += 1
        for (_, e0), (lab, e1) in zip(marks[:-1], marks[1:]):
            acc[lab] = acc.get(lab, 0.0) + e0.elapsed_time(e1)
    return {k: round(v / max(n, 1), 4) for k, v in acc.items()}


def emulate_rank(dev, W, Nl, El, d, H, T, R, locality, blocks, compress, steps, precision, block_shape="equal"):
    """ONE GPU plays rank 0 of a W-rank partition of the configs[3] recipe (pyhgt_amd.dist.HaloPlan(emulate=...)): the real receive
    side (ids, first-use chunks, types), a mirrored send side, and the all-to-all replaced by a device copy of the same size.  What is
    measured is everything a rank's GPU does in a step -- packing, own Q|K|V, halo K|V off the wire buffer, the target blocks --
    i.e. the compute side of the multi-GPU model in DESIGN.md section 6; the link time is bytes / (7 links x 76.8 GB/s x 0.7)."""
    from pyhgt_amd import HGTConv
    from pyhgt_amd.dist import HaloPlan, PartitionedGraph, target_blocks
    share, nt_g = configs3_share(dev, W, 0, Nl, El, T, R, False, locality, 0.0, keep_global_types=True)
    n_own = int(share["node_type_own"].numel())
    bounds = target_blocks(share["dst_local"], n_own, blocks, shape=block_shape)
    eblock = torch.searchsorted(torch.tensor(bounds[1:], device=dev), share["dst_local"], right=True).clamp(max=blocks - 1)
    hp = HaloPlan(share["node_type_own"], share["src_global"], share["node_offsets"], 0, W, n_chunks=blocks, edge_block=eblock,
                  emulate={"node_type_global": nt_g})
    del nt_g, eblock
    pg = PartitionedGraph(None, None, share["dst_local"], share["edge_type"], None, T, R, Nl, 0, W, node_offsets=share["node_offsets"],
                          halo=hp, compress=compress, mode="blocked", n_chunks=blocks, block_shape=block_shape)
    torch.manual_seed(0)
    layer = HGTConv(d, d, T, R, H, 0.2, True, False, precision=precision).eval().to(dev)
    pg.x_local = torch.empty(pg.n_local, d, dtype=torch.float32, device=dev)
    pg.x_local[:n_own].normal_(generator=torch.Generator(device=dev).manual_seed(7))
    x_own = pg.x_local[:n_own]
    sets, step_ms = [], []
    with torch.no_grad():
        for _ in range(3):
            pg.forward(layer, x_own)
        torch.cuda.synchronize()
        for _ in range(max(steps, 3)):
            pg.timeline = []
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = pg.forward(layer, x_own)
            e1.record()
            torch.cuda.synchronize()
            sets.append(pg.timeline)
            step_ms.append(e0.elapsed_time(e1))
    # the MEDIAN step and its own stage marks (single steps are occasionally inflated by milliseconds of host-side stalls in the pack
    # phase -- seen as idle time in rocprofv3 traces -- which a mean would charge to the GPU)
    order = sorted(range(len(step_ms)), key=lambda i: step_ms[i])
    mid = order[len(order) // 2]
    ms = step_ms[mid]
    sets = [sets[mid]]
    assert torch.isfinite(out).all()
    pg.timeline = None
    E_own = int(share["dst_local"].numel())
    halo_bytes = int(hp.n_halo) * d * (3 if compress else 4)
    link_ms = halo_bytes / (7 * XGMI_LINK_GBS * 1e9 * 0.7) * 1e3
    chunk_rows = [hp.recv_chunk_off[c + 1] - hp.recv_chunk_off[c] for c in range(blocks)]
    # The emulated all-to-all is a device copy ON THE COMPUTE STREAM (inside the `pack` stage); in a real step RCCL moves the rows on its own
    # stream.  Its time, measured alone on buffers of the same total size, is reported next to the step so that both readings are visible.
    copy_ms = 0.0
    if halo_bytes >= 16:
        cb_src = torch.empty(halo_bytes // 4, dtype=torch.int32, device=dev)
        cb_dst = torch.empty_like(cb_src)
        for _ in range(2):
            torch.bitwise_or(cb_src, 0, out=cb_dst)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(5):
            torch.bitwise_or(cb_src, 0, out=cb_dst)
        c1.record()
        torch.cuda.synchronize()
        copy_ms = c0.elapsed_time(c1) / 5
        del cb_src, cb_dst
    res = {"world": W, "locality": locality, "blocks": blocks, "block_shape": block_shape, "own_nodes": n_own, "own_edges": E_own, "halo_rows": int(hp.n_halo),
           "halo_rows_per_chunk": chunk_rows, "halo_format": "c24" if compress else "fp32", "halo_bytes": halo_bytes,
           "gpu_ms_per_step": round(ms, 4), "emulated_copy_ms": round(copy_ms, 4), "gpu_ms_per_step_minus_emulated_copy": round(ms - copy_ms, 4),
           "gpu_ms_per_step_min_max": [round(min(step_ms), 4), round(max(step_ms), 4)],
           "gpu_ms_per_step_all": [round(v, 3) for v in step_ms],
           "stage_ms": stage_times(sets),
           "link_ms_at_70pct_of_7x76.8GBs": round(link_ms, 3),
           "note": "GPU side of one rank's step measured on ONE GPU (exchange = device copies of the same size); a real step is "
                   "max(this, link time + the last block's tail), see DESIGN.md section 6"}
    del pg, hp, layer, out
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nodes-per-gpu", type=int, default=1_000_000)
    ap.add_argument("--edges-per-gpu", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=256)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--types", type=int, default=4)
    ap.add_argument("--relations", type=int, default=8)
    ap.add_argument("--rte", action="store_true", help="5-argument form with temporal encoding")
    ap.add_argument("--dst-skew", type=float, default=0.0, help="secondary variant: Zipf exponent a in (0,1) of the target in-degree distribution (hubs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the sampled-target oracle check after the timed region")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary measurements (other precision / other halo format)")
    ap.add_argument("--mode", default="blocked", choices=["blocked", "bucketed", "pipelined"],
                    help="multi-GPU schedule (pyhgt_amd/dist.py): target-blocked (first-use halo chunks, stage 5; default), source-bucketed "
                         "(stage 4) or the edge phase after the last chunk (stages 1/2/3)")
    ap.add_argument("--no-buckets", action="store_true", help=argparse.SUPPRESS)   # round-2/3 spelling of --mode pipelined
    ap.add_argument("--blocks", type=int, default=8, help="multi-GPU: target blocks = halo chunks of the blocked schedule")
    ap.add_argument("--block-shape", default="equal", choices=["equal", "geometric"], help="multi-GPU: in-edge shares of the target blocks "
                    "(pyhgt_amd.dist.target_blocks)")
    ap.add_argument("--locality", type=float, default=0.0, help="multi-GPU: fraction p of the edges whose source is drawn from the target's "
                    "own partition (SURVEY 8e 'schema/locality' variant); the rest is uniform over ALL nodes.  0 = the uniform worst case")
    ap.add_argument("--halo-fp32", action="store_true", help="multi-GPU: ship exact fp32 halo rows in the JUDGED run (default: the 24-bit "
                    "transport format: 16 significant bits, one rounding of <= 2^-16 relative per remote source element, the size of the "
                    "split-bf16 product's own dropped mid*mid term; the parity check always compares against the exact fp32 inputs)")
    ap.add_argument("--halo-c24", action="store_true", help=argparse.SUPPRESS)    # the default now; kept for old command lines
    ap.add_argument("--emulate-world", type=int, default=0, help="N = 1 only: ONE GPU plays rank 0 of a W-rank partition of the configs[3] "
                    "recipe (exchange replaced by device copies of the same size): the per-rank GPU work of a multi-GPU step, measured")
    ap.add_argument("--precision", default=None, choices=["bf16x3", "f16x3", "fp32"],
                    help="typed linears / relation transforms.  Default (round 6, N = 1) f16x3 = 3-term fp16 hi/lo MFMA with power-of-two "
                         "row scales, fp32 accumulation: the layer's default, the reference's own fp32 accuracy (<= 2e-6 from the fp64 "
                         "oracle).  bf16x3 = 3-term split-bf16 MFMA (the split SURVEY.md 7.2 prescribes, <= 3.5e-5, north-star bound 1e-4; "
                         "the judged mode of rounds 1-5, ~5 %% faster): measured in EVERY default run and reported at the top level of the "
                         "line as `fast_mode` (and under `secondary`, with exact fp32).  N > 1 defaults to bf16x3: the staged calls run it")
    ap.add_argument("--kernel-flags", type=int, default=0, help="hgt_conv_args.flags (HGT_FLAG_*), A/B runs")
    ap.add_argument("--workload", default="c2", choices=["c2", "c5"],
                    help="c2 (default): BASELINE.json configs[1] at N=1, the configs[3] recipe (dst partition + RCCL halo all-to-all) at N>1.  "
                         "c5: configs[4] surrogate in REPLICAS mode -- every GPU runs the 2-layer GNN forward on its own independent sampled "
                         "batches (OAG/train_paper_field.py:145-153 prepares n_batch independent sub-graphs), no collective in the data path")
    ap.add_argument("--small-only", action="store_true", help="tool mode: only the latency-regime measurements of the line (c1, c3, script-default "
                    "batch, c5, published 4-layer model), printed as one JSON object")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="also run the CPU port ONCE at the full c2 size (about a minute; needs >= 64 GB "
                    "of free RAM); the default on hosts with >= 96 GB of RAM available")
    ap.add_argument("--cpu-baseline-sample-only", action="store_true", help="never run the full-size CPU forward")
    ap.add_argument("--cpu-full-size", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-threads", type=int, default=32, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_only:
        if args.cpu_full_size:
            print(json.dumps(cpu_baseline_measure(args.dim, args.heads, args.types, args.relations, args.cpu_threads,
                                                  N=args.nodes_per_gpu, E=args.edges_per_gpu, max_s=0.0)))
        else:
            print(json.dumps(cpu_baseline_measure(args.dim, args.heads, args.types, args.relations, args.cpu_threads)))
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.precision is None:      # (see --precision)
        # round 6: the reference-accurate split is the layer's default and the judged mode at N = 1; the staged multi-GPU calls run the
        # bf16 split whatever the layer says (include/hgt_hip.h: precision 2 is whole-layer only), so N > 1 is labelled with what runs
        args.precision = "f16x3" if world == 1 else "bf16x3"
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    # HGT_BENCH_DEVICE / HGT_BENCH_BACKEND: development aids to walk the multi-GPU code path on a one-GPU box (all ranks on one
    # device, gloo with host-staged exchange); the judged runs use one GPU per rank over RCCL
    if "HGT_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["HGT_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("HGT_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from pyhgt_amd import HGTConv, GraphPlan
    from pyhgt_amd import _lib

    backend_name = "none" if world == 1 else os.environ.get("HGT_BENCH_BACKEND", "nccl")
    if args.workload == "c5":
        replicas_c5(args, world, rank, dev, backend_name)
        return

    d, H, T, R = args.dim, args.heads, args.types, args.relations
    Nl, El = args.nodes_per_gpu, args.edges_per_gpu
    use_rte = bool(args.rte)
    if world == 1 and args.small_only:
        print(json.dumps(small_regime(dev)))
        return
    if world == 1 and args.emulate_world > 1:      # tool mode: the GPU side of one rank's multi-GPU step, on one GPU
        print(json.dumps(emulate_rank(dev, args.emulate_world, Nl, El, d, H, T, R, args.locality, args.blocks, not args.halo_fp32,
                                      args.steps, args.precision if args.precision != "fp32" else "bf16x3", args.block_shape)))
        return

    # ---------------- synthetic inputs, generated on the device (SURVEY.md section 8d recipe) -------------
    share = None
    if world == 1:
        g = torch.Generator(device=dev).manual_seed(1234 + rank)
        node_type_own = torch.randint(0, T, (Nl,), generator=g, device=dev).sort().values
        x_own = torch.randn(Nl, d, generator=g, device=dev)
        src_global = torch.randint(0, Nl, (El,), generator=g, device=dev)
        dst_local = torch.randint(0, Nl, (El,), generator=g, device=dev)
        if args.dst_skew > 0.0:   # hub targets (SURVEY.md section 8d secondary variant)
            # Zipf-like: P(rank k) ~ k^-a with a = dst_skew in (0,1); rank 0 gets ~E*(1-a)/N^(1-a) edges
            u = torch.rand(El, generator=g, device=dev)
            dst_local = (Nl * u ** (1.0 / (1.0 - args.dst_skew))).long().clamp(0, Nl - 1)
        edge_type = torch.randint(0, R, (El,), generator=g, device=dev)
        edge_time = torch.randint(0, 240, (El,), generator=g, device=dev) if use_rte else None
    else:
        share, _ = configs3_share(dev, world, rank, Nl, El, T, R, use_rte, args.locality, args.dst_skew)
        # every rank generated the global graph itself (same seed, same device generator): make sure they all cut it the same way
        # before anything is exchanged -- a mismatch would otherwise show up as a hang or a fault deep inside the first step
        import torch.distributed as dist
        sig = torch.tensor([float(sum(share["node_offsets"])), float(share["dst_local"].numel() if rank == 0 else 0)], device=dev,
                           dtype=torch.float64)
        lo_, hi_ = sig[:1].clone(), sig[:1].clone()
        dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
        if float(lo_.item()) != float(hi_.item()):
            raise SystemExit("bench.py: the ranks derived different partitions of the global graph (device RNG not reproducible across GPUs?)")
        node_type_own, src_global, dst_local = share["node_type_own"], share["src_global"], share["dst_local"]
        edge_type, edge_time = share["edge_type"], share["edge_time"]
        Nl_own, El_own = int(node_type_own.numel()), int(dst_local.numel())
        x_own = torch.randn(Nl_own, d, generator=torch.Generator(device=dev).manual_seed(4321 + rank), device=dev)

    def make_layer(precision):
        torch.manual_seed(0)
        layer = HGTConv(d, d, T, R, H, 0.2, True, use_rte, precision=precision).eval()
        with torch.no_grad():
            layer.relation_pri.uniform_(0.5, 1.5)
            layer.skip.normal_()
        layer.kernel_flags = args.kernel_flags
        return layer.to(dev)

    layer = make_layer(args.precision)
    layer_sd = {k: v.detach().cpu() for k, v in layer.state_dict().items()}

    ev = HipEvents()
    pg = None
    if world == 1:
        edge_index = torch.stack([src_global, dst_local], dim=1).t()       # (1,2)-strided view like data.py:254
        plan_times = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            plan = GraphPlan(node_type_own, edge_index, edge_type, edge_time, T, R)
            torch.cuda.synchronize()
            plan_times.append((time.perf_counter() - t0) * 1e3)
        plan_ms = min(plan_times)

        def make_step(lay, compress=None):
            return lambda events=None: lay(x_own, node_type_own, edge_index, edge_type, edge_time, plan=plan, phase_events=events)
        barrier = lambda: None
    else:
        from pyhgt_amd.dist import PartitionedGraph
        import torch.distributed as dist
        mode = "pipelined" if args.no_buckets else args.mode
        pg = PartitionedGraph(node_type_own, src_global, dst_local, edge_type, edge_time, T, R, Nl, rank, world,
                              node_offsets=share["node_offsets"], compress=not args.halo_fp32, mode=mode,
                              n_chunks=args.blocks if mode == "blocked" else None, block_shape=args.block_shape)
        plan_ms = None
        Nl, El = Nl_own, El_own
        # own features live at the front of the [own ; halo] buffer, so a step does not copy them (pyhgt_amd/dist.py)
        pg.x_local = torch.empty(pg.n_local, d, dtype=torch.float32, device=dev)
        pg.x_local[:Nl].copy_(x_own)
        x_own = pg.x_local[:Nl]

        def make_step(lay, compress=None):
            def step(events=None):
                if compress is not None:
                    pg.compress = compress
                return pg.forward(lay, x_own, phase_events=events)
            return step
        barrier = dist.barrier

    def timed(step, steps, warmup):
        """W untimed steps, then EXACTLY K steps between barrier + synchronize pairs; max over ranks."""
        with torch.no_grad():
            for _ in range(warmup):
                out = step()
            # one GPU: hgt_conv_forward records HIP events at its phase boundaries (one call = one layer).  Several GPUs: a step is
            # many staged calls, so the phase events of single calls say nothing about the step (round-3 review: fractions > 1);
            # PartitionedGraph.forward marks its own stage boundaries on the compute stream instead (stage_times)
            event_sets = [ev.make_set(_lib.HGT_N_PHASE_EVENTS) for _ in range(steps)] if world == 1 else None
            timelines = []
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                if world == 1:
                    out = step(event_sets[i])
                else:
                    pg.timeline = []
                    out = step(None)
                    timelines.append(pg.timeline)
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist
            pg.timeline = None
            tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
            return out, elapsed, stage_times(timelines), elapsed / steps * 1e3
        phase_ms = {p: 0.0 for p in PHASES}
        per_step = []
        for es in event_sets:
            for i, p in enumerate(PHASES):
                phase_ms[p] += ev.elapsed_ms(es[i], es[i + 1])
            per_step.append(ev.elapsed_ms(es[0], es[len(PHASES)]))
        phase_ms = {p: v / steps for p, v in phase_ms.items()}
        per_step.sort()
        median = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2])
        return out, elapsed, phase_ms, median

    out, elapsed, phase_ms, median_ms = timed(make_step(layer), args.steps, args.warmup)
    assert torch.isfinite(out).all()

    ms_per_step = elapsed / args.steps * 1e3
    total_edges = El * world
    if world > 1:      # the partitioner balances in-edges to one plan tile of targets: count what the ranks really hold
        import torch.distributed as dist
        te = torch.tensor([El], device=dev, dtype=torch.int64)
        dist.all_reduce(te)
        total_edges = int(te.item())
    value = total_edges / (elapsed / args.steps)

    # ---------------- parity of the benchmarked configuration itself (after the timed region) ----------------
    def check(o, sd):
        if args.no_parity:
            return None
        if world == 1:
            return parity_check(sd, o, x_own, node_type_own, edge_index, edge_type, edge_time, T, R, H, use_rte)
        # a rank's local graph [own ; halo]: exact for its own targets only if the exchanged halo rows are right
        return parity_check(sd, o, pg.x_local, pg.node_type_local, pg.edge_index, pg.edge_type, pg.edge_time, T, R, H, use_rte,
                            n_q_rows=Nl)
    if world > 1 and not args.no_parity:
        # The checker wants the TRUE inputs: exact fp32 halo rows (compress=False), whatever format the timed steps shipped -- with
        # the rows of the 24-bit wire format fed to the oracle as well, the format's own rounding (<= 2^-16 relative per remote
        # source element) would cancel out of parity_max_abs_err and the 1e-4 gate could not see it (round-4 advisor finding).
        # The blocked schedule projects the halo rows straight off the wire buffer and never expands them, so one plain exchange
        # (a collective: every rank) fills them in after the timed region.
        with torch.no_grad():
            for c in range(pg.halo.n_chunks):
                pg.halo.exchange_chunk(c, x_own, pg.x_local, compress=False)
        torch.cuda.synchronize()
    parity = check(out, layer_sd) if rank == 0 else None
    del out

    n_local_nodes = Nl if world == 1 else pg.n_local
    alg = algorithmic_bytes(Nl, El, d, use_rte)
    pmc = {}
    if world > 1:
        # Several GPUs: ONE figure for the whole step of a rank -- the algorithmic bytes of the rank's own layer (SURVEY 8d on its
        # own nodes / edges: halo projections, packing and the exchange are overhead, not credited) over the step time -- and the
        # mean time of every stage on the compute stream.  No per-kernel fractions: no timed interval brackets a single kernel here.
        stage_ms = phase_ms
        ach = alg["layer"] / (ms_per_step * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "whole step of a rank (pack, own Q|K|V, halo K|V, target blocks; waits on the exchange included)",
                    "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                    "algorithmic_bytes_per_launch": alg["layer"], "avg_launch_ms": round(ms_per_step, 4),
                    "layer_achieved_GBs": round(ach, 1), "layer_frac": round(ach / HBM_PEAK_GBS, 4),
                    "stage_ms": stage_ms, "stage_note": "mean ms between marks on rank 0's compute stream: pack = gather kernels + "
                    "queueing the all-to-alls, wait = stalls on a chunk that has not arrived, halo_kv / edge_blocks summed over blocks"}
    else:
        fused_update = (phase_ms["a_linear"] + phase_ms["node_update"]) < 0.05 * phase_ms["edge_aggregate"]
        if fused_update:   # hgt_edge_aggregate_update: the node update runs as the epilogue of the aggregation kernel
            alg["edge_aggregate"] += alg["node_update"]
            alg["node_update"] = 0
        dom = max(("edge_logits", "edge_aggregate", "project_qkv"), key=lambda p: phase_ms[p])
        ach = alg[dom] / (phase_ms[dom] * 1e-3) / 1e9
        # counter-derived figures come from the committed rocprofv3 passes of this same command (profiles/pmc_summary.json names
        # the commit they were taken at); they are not re-measured inside the run
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_summary.json")))
        if cands:      # the newest round's passes
            try:
                pmc = json.load(open(cands[-1]))
            except Exception:
                pmc = {}
        traffic = (pmc.get("traffic_bytes") or {}).get(dom) if (not use_rte and args.dst_skew == 0.0) else None
        roofline = {"bound": "hbm", "kernel": dom + ("+node_update (fused)" if fused_update and dom == "edge_aggregate" else ""),
                    "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "algorithmic_bytes_per_launch": alg[dom], "avg_launch_ms": round(phase_ms[dom], 4),
                    "layer_achieved_GBs": round(alg["layer"] / (ms_per_step * 1e-3) / 1e9, 1),
                    "layer_frac": round(alg["layer"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "phase_ms": {p: round(v, 4) for p, v in phase_ms.items()},
                    "per_kernel_frac": {p: round(alg[p] / (phase_ms[p] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                        for p in ("project_qkv", "edge_logits", "edge_aggregate") if phase_ms[p] > 0},
                    "mfma_busy_pct": pmc.get("mfma_busy_pct"), "pmc_commit": pmc.get("commit"),
                    # counter figures are quoted from the committed passes: flagged when the kernel sources changed since
                    "pmc_stale": bool(pmc) and pmc.get("kernel_sources_sha16") != kernel_sources_sha16()}

    # ---------------- secondary measurements (never `value`) ----------------
    secondary = {}
    if not args.no_secondary:
        if world == 1:
            for other in [p for p in ("bf16x3", "f16x3", "fp32") if p != args.precision]:
                lay2 = make_layer(other)
                sd2 = {k: v.detach().cpu() for k, v in lay2.state_dict().items()}
                out2, el2, ph2, med2 = timed(make_step(lay2), args.steps, 2)
                ms2 = el2 / args.steps * 1e3
                par2 = check(out2, sd2)
                secondary["precision_" + other] = {
                    "ms_per_step": ms2, "median_ms_per_step": med2, "edges_per_s": El / (ms2 * 1e-3),
                    "layer_frac": round(alg["layer"] / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "phase_ms": {p: round(v, 4) for p, v in ph2.items()},
                    "parity_max_abs_err": None if par2 is None else par2["max_abs_err"]}
                del out2, lay2
        else:
            def refill():      # (the parity checker reads the EXACT fp32 halo rows: see above)
                if not args.no_parity:
                    with torch.no_grad():
                        for c in range(pg.halo.n_chunks):
                            pg.halo.exchange_chunk(c, x_own, pg.x_local, compress=False)
                    torch.cuda.synchronize()
            judged_compress = pg.compress
            out2, el2, st2, _ = timed(make_step(layer, compress=not judged_compress), args.steps, 2)
            ms2 = el2 / args.steps * 1e3
            refill()
            par2 = check(out2, layer_sd) if rank == 0 else None
            secondary["halo_" + ("fp32" if judged_compress else "c24")] = {
                "ms_per_step": ms2, "edges_per_s": total_edges / (ms2 * 1e-3), "stage_ms": st2,
                "parity_max_abs_err": None if par2 is None else par2["max_abs_err"],
                "note": "exact fp32 halo rows on the links (4/3 of the bytes)" if judged_compress else
                        "24-bit transport format (sign, 8 exponent, 15 mantissa bits)"}
            pg.compress = judged_compress
            del out2
            if pg.mode != "pipelined":      # the schedule without overlap of exchange and edge phase, same graph / chunks
                judged_mode, pg.mode = pg.mode, "pipelined"
                out3, el3, st3, _ = timed(make_step(layer), args.steps, 2)
                ms3 = el3 / args.steps * 1e3
                refill()
                par3 = check(out3, layer_sd) if rank == 0 else None
                secondary["edge_phase_after_last_chunk"] = {
                    "ms_per_step": ms3, "edges_per_s": total_edges / (ms3 * 1e-3),
                    "parity_max_abs_err": None if par3 is None else par3["max_abs_err"]}
                pg.mode = judged_mode
                del out3

    # ---------------- the other BASELINE.json configurations + shape variants of configs[1], each with its own parity ------------
    if world == 1 and not args.no_secondary and not (args.rte or args.dst_skew > 0.0):
        del x_own, src_global, dst_local, edge_type, edge_index, plan
        GraphPlan.clear_cache()
        torch.cuda.empty_cache()

        def large_variant(Nv, Ev, dv, Hv, rte, skew, seed):
            gv = torch.Generator(device=dev).manual_seed(seed)
            ntv = torch.randint(0, T, (Nv,), generator=gv, device=dev).sort().values
            xv = torch.randn(Nv, dv, generator=gv, device=dev)
            srcv = torch.randint(0, Nv, (Ev,), generator=gv, device=dev)
            dstv = torch.randint(0, Nv, (Ev,), generator=gv, device=dev)
            if skew > 0.0:
                u = torch.rand(Ev, generator=gv, device=dev)
                dstv = (Nv * u ** (1.0 / (1.0 - skew))).long().clamp(0, Nv - 1)
            etv = torch.randint(0, R, (Ev,), generator=gv, device=dev)
            tmv = torch.randint(0, 240, (Ev,), generator=gv, device=dev) if rte else None
            eiv = torch.stack([srcv, dstv], dim=1).t()
            planv = GraphPlan(ntv, eiv, etv, tmv, T, R)
            algv = algorithmic_bytes(Nv, Ev, dv, rte)["layer"]

            def one(prec):
                torch.manual_seed(0)
                lay = HGTConv(dv, dv, T, R, Hv, 0.2, True, rte, precision=prec).eval()
                with torch.no_grad():
                    lay.relation_pri.uniform_(0.5, 1.5)
                    lay.skip.normal_()
                lay = lay.to(dev)
                sdv = {k: v.detach().cpu() for k, v in lay.state_dict().items()}
                outv, elv, phv, medv = timed(lambda events=None: lay(xv, ntv, eiv, etv, tmv, plan=planv, phase_events=events), 10, 2)
                msv = elv / 10 * 1e3
                par = None if args.no_parity else parity_check(sdv, outv, xv, ntv, eiv, etv, tmv, T, R, Hv, rte)
                return {"precision": prec, "ms_per_step": msv, "median_ms_per_step": medv, "edges_per_s": Ev / (msv * 1e-3),
                        "layer_frac": round(algv / (msv * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "phase_ms": {p: round(v, 4) for p, v in phv.items()},
                        "parity_max_abs_err": None if par is None else par["max_abs_err"]}
            res = {"workload": "T=%d R=%d N=%d E=%d d=%d H=%d use_RTE=%s%s" % (T, R, Nv, Ev, dv, Hv, rte, ", Zipf(%.1f) targets" % skew if skew else "")}
            res.update(one(args.precision))
            other = {"f16x3": "bf16x3", "bf16x3": "f16x3"}.get(args.precision)      # both split modes for every variant
            if other:
                res[other] = one(other)
            return res
        secondary["c2_rte"] = large_variant(Nl, El, d, H, True, 0.0, 4321)
        torch.cuda.empty_cache()
        secondary["c2_zipf0.8"] = large_variant(Nl, El, d, H, False, 0.8, 4322)
        torch.cuda.empty_cache()
        if d == 256 and H == 8:     # the reference's published ogbn-mag width (n_hid 512 = 8 heads x 64) at half the node count
            secondary["d512_h8"] = large_variant(Nl // 2, El // 2, 512, 8, False, 0.0, 4323)
            torch.cuda.empty_cache()
            # ... and the OAG width (n_hid 400 = 8 heads x 50, padded to 64 columns per head: OAG/train_paper_field.py:31-32) on a large graph
            secondary["d400_h8"] = large_variant(Nl // 2, El // 2, 400, 8, False, 0.0, 4324)
            torch.cuda.empty_cache()
        secondary["latency_regime"] = small_regime(dev)
        if d == 256 and H == 8 and Nl == 1_000_000 and El == 10_000_000:
            # configs[3] on ONE GPU: rank 0 of an 8-rank partition, exchange emulated by device copies (what a rank's GPU does per step;
            # the model of DESIGN.md section 6 combines it with the link time).  Uniform sources = worst case; 0.75 = a partition with locality
            torch.cuda.empty_cache()
            secondary["rank_of_8_uniform"] = emulate_rank(dev, 8, Nl, El, d, H, T, R, 0.0, args.blocks, True, 5, "bf16x3")
            secondary["rank_of_8_locality0.75"] = emulate_rank(dev, 8, Nl, El, d, H, T, R, 0.75, args.blocks, True, 5, "bf16x3")

    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            # the one full-size forward of the port (about a minute on the GPU box's host) runs in every default line whose host has
            # the RAM for it (>= 96 GB available): the full-c2 CPU figure is then observed by whoever runs the line
            ram_ok = (host_mem_gb()[1] or 0) >= 96
            cpu = cpu_baseline(d, H, T, R, full=(args.cpu_baseline_full or ram_ok) and not args.cpu_baseline_sample_only)
        prec_note = {"fp32": "f32", "bf16x3": "f32 (typed linears and relation transforms as 3-term split-bf16 MFMA, fp32 accumulate)",
                     "f16x3": "f32 (typed linears and relation transforms as 3-term fp16 hi/lo MFMA with power-of-two row scales, "
                              "fp32 accumulate)"}
        line = {
            "metric": "HGTConv forward edges/sec", "value": value, "unit": "edges/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "median_ms_per_step": median_ms,
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            # said plainly (round-4 review): the builder's development loop has ONE GPU -- no step of the dst-partitioned path has been
            # timed on more than one device by the builder; `secondary.rank_of_8_*` is one GPU playing rank 0 of 8 (GPU side only)
            "multi_gpu_measured_by_builder": False if world == 1 else None,
            "dtype": prec_note[args.precision],
            "data": "synthetic",
            "config": {"workload": "%s: synthetic %d-type/%d-relation graph, %d nodes / %d edges per GPU, "
                                   "d=%d, n_heads=%d, use_RTE=%s, use_norm=True, plan cached%s" % (
                                       "BASELINE.json configs[1]" if world == 1 else
                                       "BASELINE.json configs[3] recipe (one global graph of configs[1] per GPU, cut by the in-edge-balanced "
                                       "partitioner; sources %s)" % ("uniform over all ranks" if args.locality <= 0 else
                                                                      "%.2f from the target's partition, the rest uniform" % args.locality),
                                       T, R, Nl, El, d, H, use_rte, (", Zipf(%.2f) targets" % args.dst_skew) if args.dst_skew > 0 else ""),
                       "nodes_per_gpu": Nl, "edges_per_gpu": El, "local_nodes_incl_halo": int(n_local_nodes),
                       "halo_exchange_bytes_per_gpu_per_step": 0 if world == 1 else int(pg.halo.n_halo) * d * (3 if pg.compress else 4),
                       "halo_format": None if world == 1 else ("24-bit (sign, 8 exp, 15 mantissa; fp32 arithmetic)" if pg.compress else "fp32"),
                       "halo_chunks": 0 if world == 1 else int(pg.halo.n_chunks),
                       "edge_phase": None if world == 1 else {
                           "blocked": "target-blocked: %d blocks, halo chunks by first use, every block = the single-GPU kernel pair on a tile "
                                      "range (hgt_conv_forward stages 1/2/5)" % pg.halo.n_chunks,
                           "bucketed": "source-bucketed, softmax state carried between buckets (stages 1/2/4)",
                           "pipelined": "after the last halo chunk (stages 1/2/3)"}[pg.layer_mode(layer)],
                       "locality": None if world == 1 else args.locality,
                       "node_offsets": None if world == 1 else share["node_offsets"],
                       "parallelism": "single" if world == 1 else "dst-partition x%d + RCCL all-to-all halo" % world,
                       "backend": backend_name + (" (RCCL)" if backend_name == "nccl" else ""),
                       "plan_build_ms": plan_ms, "precision": args.precision, "kernel_flags": args.kernel_flags},
            "parity_max_abs_err": None if parity is None else parity["max_abs_err"],
            "parity": parity,
            "plan_included": None if plan_ms is None else {
                "ms_per_step": ms_per_step + plan_ms, "edges_per_s": El / ((ms_per_step + plan_ms) * 1e-3),
                "layer_frac": round(alg["layer"] / ((ms_per_step + plan_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "roofline": roofline, "cpu_baseline": cpu, "secondary": secondary,
        }
        f16 = secondary.get("precision_f16x3")
        if f16:      # the same layer, same graph, same run in the reference's own accuracy (see --precision)
            line["fp32_accurate"] = {"precision": "f16x3", "ms_per_step": f16["ms_per_step"], "edges_per_s": f16["edges_per_s"],
                                     "layer_frac": f16["layer_frac"], "parity_max_abs_err": f16["parity_max_abs_err"]}
        elif args.precision == "f16x3":      # (round 6: the headline IS the reference-accurate mode)
            line["fp32_accurate"] = {"precision": "f16x3", "ms_per_step": ms_per_step, "edges_per_s": value,
                                     "layer_frac": roofline.get("layer_frac") if roofline else None,
                                     "parity_max_abs_err": None if parity is None else parity["max_abs_err"]}
        fast = secondary.get("precision_bf16x3")
        if fast:     # the opt-in fast mode (rounds 1-5's judged mode) on the same graph in the same run
            line["fast_mode"] = {"precision": "bf16x3", "ms_per_step": fast["ms_per_step"], "edges_per_s": fast["edges_per_s"],
                                 "layer_frac": fast["layer_frac"], "parity_max_abs_err": fast["parity_max_abs_err"]}
        # scalar copies of figures that live in nested objects (the driver's `parsed` view flattens those away: round-5 review)
        if plan_ms is not None:
            line["plan_included_ms"] = ms_per_step + plan_ms
            line["plan_included_frac"] = round(alg["layer"] / ((ms_per_step + plan_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        if isinstance(cpu, dict):
            full = cpu.get("full_c2")
            line["cpu_full_c2_edges_per_s"] = full.get("edges_per_s") if isinstance(full, dict) else None
            line["cpu_sample_edges_per_s"] = cpu.get("value")
        lat = secondary.get("latency_regime")
        if isinstance(lat, dict):
            try:
                line["c3_us_per_layer"] = lat["c3"][args.precision]["us_per_layer"]
                line["c5_us_per_forward"] = lat["c5"][args.precision]["us_per_forward"]
                line["mag4_us_per_layer"] = lat["mag4"][args.precision]["us_per_layer"]
            except KeyError:
                pass
        print(json.dumps(line))
        def parities(prefix, node):
            if isinstance(node, dict):
                for kk, vv in node.items():
                    if kk == "parity_max_abs_err":
                        yield prefix, vv
                    else:
                        yield from parities(prefix + "." + kk if prefix else kk, vv)
        bad = [k for k, v in [("headline", line["parity_max_abs_err"])] + list(parities("", secondary))
               if v is not None and not (v <= 1e-4)]
        if bad:
            sys.stderr.write("PARITY FAILURE (> 1e-4 against the fp64 oracle): %s\n" % bad)
            if world > 1:
                import torch.distributed as dist
                dist.destroy_process_group()
            sys.exit(3)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
