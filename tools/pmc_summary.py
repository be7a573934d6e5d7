#!/usr/bin/env python
"""Condense the rocprofv3 CSVs of tools/profile_pmc.sh into small files (run on the GPU box; gpurun_out is capped at 64 MiB):

    python tools/pmc_summary.py gpurun_out/prof_<tag> <tag> [commit]

writes <dir>/<tag>_kernel_stats.csv (top kernels of the --stats pass), <dir>/<tag>_pmc_per_kernel.csv (every counter averaged
per dispatch of the hot kernels) and <dir>/<tag>_pmc_summary.json (traffic bytes per launch with the gfx950 FETCH_SIZE x2
correction of MI355X_MICROARCH.md, MFMA-busy / VALU / LDS-conflict / occupancy figures), then deletes the raw rocprofv3 output.
Copy the three files into profiles/ to have them judged."""
import csv
import glob
import json
import os
import re
import shutil
import sys

HOT = [("project_qkv", r"k_typed_linear_xs<0|k_typed_linear_xsILi0E|k_typed_linear_pc<0,\s*false(,\s*(false|true))?>|k_typed_linear_pcILi0ELb0E"),
       ("edge_logits", r"k_edge_logits"),
       ("edge_aggregate", r"k_edge_aggregate"),
       ("plan_sort", r"radix|onesweep")]
N_CU, N_SIMD, N_XCD = 256, 1024, 8     # GRBM_GUI_ACTIVE is summed over the 8 XCDs: busy cycles of the chip = value / 8


def short(name):
    return re.sub(r"\(anonymous namespace\)::", "", name)[:90]


def load_counters(d):
    """-> {kernel regex label: {counter: [values per dispatch]}}"""
    out = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row.get("Kernel_Name", "")
                for label, rx in HOT:
                    if re.search(rx, k):
                        key = label
                        if label == "edge_aggregate":
                            key = "edge_aggregate" if "update" in k or "aggregate" in k else label
                        out.setdefault(key, {}).setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
                        out[key][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
                        out[key].setdefault("_meta", {})
                        for m in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size"):
                            if m in row:
                                out[key]["_meta"][m] = row[m]
                        out[key]["_meta"]["kernel"] = short(k)
                        break
    return out


def main():
    d, tag = sys.argv[1], sys.argv[2]
    commit = sys.argv[3] if len(sys.argv) > 3 else None
    # ---- kernel stats
    stats = glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True)
    dur = {}
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        with open(os.path.join(d, tag + "_kernel_stats.csv"), "w") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows[:24]:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
        for r in rows:
            for label, rx in HOT:
                if re.search(rx, r["Name"]) and label not in dur:
                    dur[label] = float(r["AverageNs"]) * 1e-6
    sha = None
    try:      # the same hash bench.py computes at run time: lets the bench line flag counter figures of another build (pmc_stale)
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        sha = bench.kernel_sources_sha16()
    except Exception:
        pass
    summary = {"tag": tag, "commit": commit, "kernel_sources_sha16": sha, "avg_kernel_ms": dur, "traffic_bytes": {}, "mfma_busy_pct": {}, "valu_busy_pct": {},
               "lds_bank_conflict_pct": {}, "waves_per_simd_avg": {}, "kernel_meta": {}, "counters_avg_per_dispatch": {},
               "_note": "traffic = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (KB units, gfx950 FETCH_SIZE x2 correction, MI355X_MICROARCH.md HBM "
                        "section; WRITE_SIZE uncalibrated).  cyc = GRBM_GUI_ACTIVE / 8 XCDs.  mfma_busy_pct = SQ_VALU_MFMA_BUSY_CYCLES / (cyc * 1024 SIMDs) "
                        "(cycles summed over the SIMDs); valu_busy_pct = 4 * SQ_ACTIVE_INST_VALU (quad-cycles) / (cyc * 1024); "
                        "waves_per_simd_avg = 4 * SQ_WAVE_CYCLES / (cyc * 1024); wait_pct = SQ_WAIT_ANY / SQ_WAVE_CYCLES (share of the resident wave cycles parked in s_waitcnt / barriers).  GRBM_GUI_ACTIVE comes from the sqA pass; "
                        "counters of other passes are normalised with it (same command, same clocks to within DVFS noise)."}
    merged = {}
    for sub in sorted(os.listdir(d)):
        p = os.path.join(d, sub)
        if os.path.isdir(p) and sub != "stats":
            for k, v in load_counters(p).items():
                tgt = merged.setdefault(k, {})
                for cn, vals in v.items():
                    if cn == "_meta":
                        tgt.setdefault("_meta", {}).update(vals)
                    else:
                        xs = list(vals.values())
                        tgt[cn] = sum(xs) / max(1, len(xs))
    with open(os.path.join(d, tag + "_pmc_per_kernel.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "avg_per_dispatch"])
        for k, v in merged.items():
            for cn, val in sorted(v.items()):
                if cn != "_meta":
                    w.writerow([k, cn, "%.6g" % val])
    for k, v in merged.items():
        summary["kernel_meta"][k] = v.get("_meta", {})
        summary["counters_avg_per_dispatch"][k] = {cn: val for cn, val in v.items() if cn != "_meta"}
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            summary["traffic_bytes"][k] = int(2 * v["FETCH_SIZE"] * 1024 + v["WRITE_SIZE"] * 1024)
        gui = v.get("GRBM_GUI_ACTIVE")
        if gui:
            gui = gui / N_XCD
        if gui and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            # SQ_VALU_MFMA_BUSY_CYCLES: cycles summed over the SIMDs (MI355X_MICROARCH.md: = 32 x N_mfma for 32x32x16 bf16)
            summary["mfma_busy_pct"][k] = round(100.0 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * N_SIMD), 2)
        if "SQ_ACTIVE_INST_VALU" in v and gui:
            # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs
            summary["valu_busy_pct"][k] = round(100.0 * 4.0 * v["SQ_ACTIVE_INST_VALU"] / (gui * N_SIMD), 2)
        if "SQ_LDS_BANK_CONFLICT" in v and v.get("SQ_LDS_IDX_ACTIVE"):
            summary["lds_bank_conflict_pct"][k] = round(100.0 * v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"], 2)
        if "SQ_WAIT_ANY" in v and v.get("SQ_WAVE_CYCLES"):
            summary.setdefault("wait_pct", {})[k] = round(100.0 * v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 1)
        if "SQ_WAVE_CYCLES" in v and gui:
            # SQ_WAVE_CYCLES counts quad-cycles of resident waves: average resident waves per SIMD = 4 * WAVE_CYCLES / (GUI * SIMDs)
            summary["waves_per_simd_avg"][k] = round(4.0 * v["SQ_WAVE_CYCLES"] / (gui * N_SIMD), 2)
    json.dump(summary, open(os.path.join(d, tag + "_pmc_summary.json"), "w"), indent=1)
    # ---- drop the bulky raw output
    for sub in os.listdir(d):
        p = os.path.join(d, sub)
        if os.path.isdir(p):
            shutil.rmtree(p, ignore_errors=True)
    print(json.dumps({k: summary[k] for k in ("avg_kernel_ms", "traffic_bytes", "mfma_busy_pct", "lds_bank_conflict_pct", "waves_per_simd_avg")}))


if __name__ == "__main__":
    main()
