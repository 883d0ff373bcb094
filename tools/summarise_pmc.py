"""Summarise rocprofv3 counter-collection CSVs of the bench command into the small JSON files
kept under profiles/ (development aid).

    python tools/summarise_pmc.py traffic <fetch_dir> <write_dir> <out.json> [--update-traffic]
    python tools/summarise_pmc.py sq <sq_dir> <out.json>        (any SQ_* pass)

Only dispatches of the dominant kernel (name contains KERNEL, default 'fista_pair') with the
largest grid are kept (the bench also launches it on fewer problems)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

KERNEL = os.environ.get("PB_PMC_KERNEL", "fista_pair")


def rows(d):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if KERNEL in r["Kernel_Name"]:
                    yield r


def per_dispatch(d):
    """{counter: [value per dispatch]} for the largest grid, plus kernel name/grid/duration."""
    by = defaultdict(lambda: defaultdict(float))
    meta = {}
    for r in rows(d):
        key = int(r["Dispatch_Id"])
        by[key][r["Counter_Name"]] += float(r["Counter_Value"])
        meta[key] = (int(r["Grid_Size"]), r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                     int(r["VGPR_Count"]), int(r["SGPR_Count"]), int(r["LDS_Block_Size"]))
    if not meta:
        raise SystemExit("no dispatch of a kernel matching %r under %s" % (KERNEL, d))
    gmax = max(m[0] for m in meta.values())
    keep = sorted(k for k, m in meta.items() if m[0] == gmax)
    out = defaultdict(list)
    for k in keep:
        for c, v in by[k].items():
            out[c].append(v)
    m = meta[keep[0]]
    info = {"kernel": m[1][:120], "grid_size": m[0], "dispatches": len(keep),
            "mean_dispatch_ms": sum(meta[k][2] for k in keep) / len(keep) / 1e6,
            "vgpr": m[3], "sgpr": m[4], "lds_bytes_per_block": m[5]}
    return out, info


def mean(x):
    return sum(x) / len(x)


def main():
    mode = sys.argv[1]
    if mode == "traffic":
        fetch, info = per_dispatch(sys.argv[2])
        write, _ = per_dispatch(sys.argv[3])
        voxels, scans, iters = (int(os.environ.get(k, d)) for k, d in
                                (("PB_VOXELS", "98304"), ("PB_SCANS", "300"), ("PB_ITERS", "500")))
        f_kb, w_kb = mean(fetch["FETCH_SIZE"]), mean(write["WRITE_SIZE"])
        # the bench starts cold (PB_FLAG_COLD_START): y float32 is the only input; PB_WARM=1: + w float64
        known_in = voxels * scans * (4 + (8 if os.environ.get("PB_WARM") == "1" else 0))
        known_out = voxels * scans * 8
        hbm = 2.0 * f_kb * 1024 + w_kb * 1024
        out = {"command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) --output-format csv "
                          "-- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0",
               **info, "voxels": voxels, "scans": scans, "iters": iters,
               "raw": {"FETCH_SIZE_KB_per_dispatch": fetch["FETCH_SIZE"], "WRITE_SIZE_KB_per_dispatch": write["WRITE_SIZE"]},
               "known_input_bytes": known_in, "known_output_bytes": known_out,
               "fetch_correction": "gfx950 FETCH_SIZE reads 1/2 of the fetched bytes (MI355X_MICROARCH.md, HBM): "
                                   "2*FETCH = %.1f MB vs %.1f MB of known input" % (2 * f_kb * 1024 / 1e6, known_in / 1e6),
               "hbm_bytes_per_launch": hbm,
               "algorithmic_bytes_per_launch": 12.0 * scans * voxels * iters}
        json.dump(out, open(sys.argv[4], "w"), indent=1)
        if "--update-traffic" in sys.argv:
            json.dump({"voxels": voxels, "scans": scans, "iters": iters, "hbm_bytes_per_launch": hbm,
                       "source": sys.argv[4]}, open(os.path.join(os.path.dirname(sys.argv[4]), "traffic.json"), "w"),
                      indent=1)
        print(json.dumps(out)[:400])
    else:
        c, info = per_dispatch(sys.argv[2])
        m = {k: mean(v) for k, v in c.items()}
        iters = int(os.environ.get("PB_ITERS", "500"))
        waves = m.get("SQ_WAVES", 0.0)
        d = {}
        if waves:
            d["valu_insts_per_wave_per_iteration"] = m["SQ_INSTS_VALU"] / waves / iters
            vpw = float(os.environ.get("PB_VOXELS_PER_WAVE", "8"))        # pair form 8, matrix-pipe form 16
            d["voxels_per_wave"] = vpw
            d["valu_simd_cycles_per_voxel_iteration"] = m["SQ_INSTS_VALU"] / waves / iters * 4.0 / vpw   # 4 cycles per instruction
            d["active_inst_valu_over_wave_cycles"] = m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"]
            secs = info["mean_dispatch_ms"] * 1e-3
            d["sustained_clock_GHz_from_SQ_BUSY_CYCLES_over_32_SE"] = m["SQ_BUSY_CYCLES"] / 32.0 / secs / 1e9
            simds = 1024.0
            busy_per_simd = m["SQ_ACTIVE_INST_VALU"] * 4.0 / simds          # quad-cycles -> cycles
            d["simd_valu_utilisation"] = busy_per_simd / (m["SQ_BUSY_CYCLES"] / 32.0)
            d["note_units"] = ("SQ_WAVE_CYCLES and SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md); one "
                               "wave64 VALU instruction = 1 quad-cycle = 4 cycles")
        counters = " ".join(sorted(m))
        out = {"command": "rocprofv3 --pmc %s --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0" % counters,
               **info, "mean_per_dispatch": m, "derived": d}
        json.dump(out, open(sys.argv[3], "w"), indent=1)
        print(json.dumps(out)[:600])


if __name__ == "__main__":
    main()
