"""Per-kernel table from a rocprofv3 kernel trace and an SQ counter pass of
tools/profile_variants.py (development aid).

    python tools/summarise_variants.py <trace_dir> <sq_dir> <out.json>

For every kernel (largest grid of each name): launches, mean duration, registers, waves,
VALU instructions per wave and the share of the dispatch's SIMD issue slots they fill
(SQ_ACTIVE_INST_VALU counts quad-cycles, one per wave64 VALU instruction;
SQ_BUSY_CYCLES is summed over the 32 shader engines' SQs -- MI355X_MICROARCH.md)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

SIMDS = 256 * 4


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name


def trace(d):
    by = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if "pb::" not in r["Kernel_Name"]:
                    continue
                grid = int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"])
                by[short(r["Kernel_Name"])].append((grid, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    out = {}
    for k, v in by.items():
        gmax = max(g for g, _ in v)
        sel = [t for g, t in v if g == gmax]
        out[k] = {"launches_at_largest_grid": len(sel), "grid": gmax, "mean_ms": sum(sel) / len(sel) / 1e6,
                  "min_ms": min(sel) / 1e6, "launches_total": len(v)}
    return out


def counters(d):
    acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
    meta = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if "pb::" not in r["Kernel_Name"]:
                    continue
                k = short(r["Kernel_Name"])
                disp = int(r["Dispatch_Id"])
                acc[k][disp][r["Counter_Name"]] += float(r["Counter_Value"])
                meta[(k, disp)] = (int(r["Grid_Size"]), int(r["VGPR_Count"]), int(r.get("Accum_VGPR_Count", 0) or 0),
                                   int(r["SGPR_Count"]), int(r["LDS_Block_Size"]))
    out = {}
    for k, disps in acc.items():
        gmax = max(meta[(k, d_)][0] for d_ in disps)
        keep = [d_ for d_ in disps if meta[(k, d_)][0] == gmax]
        m = meta[(k, keep[0])]
        mean = defaultdict(float)
        for d_ in keep:
            for c, v in disps[d_].items():
                mean[c] += v / len(keep)
        e = {"vgpr": m[1], "agpr": m[2], "sgpr": m[3], "lds_bytes_per_block": m[4], "waves": mean.get("SQ_WAVES")}
        if mean.get("SQ_WAVES"):
            e["valu_insts_per_wave"] = mean["SQ_INSTS_VALU"] / mean["SQ_WAVES"]
        if mean.get("SQ_BUSY_CYCLES"):
            # busy cycles of one SQ ~ dispatch duration in cycles; VALU quad-cycles * 4 / (SIMDs * cycles)
            cycles = mean["SQ_BUSY_CYCLES"] / 32.0
            e["simd_valu_utilisation"] = mean["SQ_ACTIVE_INST_VALU"] * 4.0 / (SIMDS * cycles)
        out[k] = e
    return out


def main():
    t = trace(sys.argv[1])
    c = counters(sys.argv[2]) if os.path.isdir(sys.argv[2]) else {}
    rows = {}
    for k in sorted(t, key=lambda k: -t[k]["mean_ms"] * t[k]["launches_total"]):
        rows[k] = dict(t[k], **c.get(k, {}))
    json.dump({"command": "python3 tools/profile_variants.py under rocprofv3 --kernel-trace / --pmc SQ_* (two runs)",
               "kernels": rows}, open(sys.argv[3], "w"), indent=1)
    for k, e in rows.items():
        print("%-110s n=%3d grid=%9d %9.3f ms  vgpr=%s valu/wave=%s util=%s" % (
            k[:110], e["launches_at_largest_grid"], e["grid"], e["mean_ms"], e.get("vgpr"),
            "%.0f" % e["valu_insts_per_wave"] if "valu_insts_per_wave" in e else "-",
            "%.3f" % e["simd_valu_utilisation"] if "simd_valu_utilisation" in e else "-"))


main()
