"""Per-kernel means of the counters of one rocprofv3 --pmc pass: python tools/r3_pmc_table.py <dir>"""
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list); meta = {}
seen = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        k = r["Kernel_Name"][:150]
        key = (k, r["Dispatch_Id"], r["Counter_Name"])
        seen[key] = seen.get(key, 0.0) + float(r["Counter_Value"])
        meta[k] = (r["Grid_Size"], r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"])
        dur[(k, r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for (k, d, c), v in seen.items():
    acc[k][c].append(v)
for k in acc:
    ds = [v for (kk, d), v in dur.items() if kk == k]
    print(k)
    print("   grid/vgpr/sgpr/lds", meta[k], "dispatches", len(ds), "mean ms %.3f" % (sum(ds) / len(ds) / 1e6))
    for c, v in sorted(acc[k].items()):
        print("   %-28s %.4e" % (c, sum(v) / len(v)))
    a = {c: sum(v) / len(v) for c, v in acc[k].items()}
    if "SQ_WAVES" in a and "SQ_INSTS_VALU" in a:
        print("   VALU instructions / wave       %.1f" % (a["SQ_INSTS_VALU"] / a["SQ_WAVES"]))
    if "SQ_WAVES" in a and "SQ_INSTS_SALU" in a:
        print("   SALU instructions / wave       %.1f" % (a["SQ_INSTS_SALU"] / a["SQ_WAVES"]))
    if "SQ_WAVES" in a and "SQ_INSTS_LDS" in a:
        print("   LDS instructions / wave        %.1f" % (a["SQ_INSTS_LDS"] / a["SQ_WAVES"]))
    if "SQ_WAVE_CYCLES" in a and "SQ_ACTIVE_INST_VALU" in a:
        print("   ACTIVE_INST_VALU / WAVE_CYCLES %.3f" % (a["SQ_ACTIVE_INST_VALU"] / a["SQ_WAVE_CYCLES"]))
    if "SQ_WAVE_CYCLES" in a and "SQ_WAIT_INST_ANY" in a:
        print("   WAIT_INST_ANY / WAVE_CYCLES    %.3f" % (a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"]))
    if "SQ_WAVE_CYCLES" in a and "SQ_WAIT_ANY" in a:
        print("   WAIT_ANY / WAVE_CYCLES         %.3f" % (a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"]))
