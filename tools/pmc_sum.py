"""Sum rocprofv3 --pmc counters per kernel name over every dispatch.  usage: pmc_sum.py <dir> [<dir> ...]  (directories given to `rocprofv3 -d`)
Prints one line per kernel: dispatches and the counters' sums, plus the ratios this build's notes use."""
import csv, glob, os, sys
tot = {}
for d in sys.argv[1:]:
    for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"].split("(")[0]
            t = tot.setdefault(k, {})
            t.setdefault("_disp", set()).add((p, r["Dispatch_Id"]))
            t[r["Counter_Name"]] = t.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k, t in sorted(tot.items()):
    if not k.startswith("ka_"): continue
    n = len(t.pop("_disp"))
    print("%s  dispatches %d" % (k, n))
    for c in sorted(t): print("    %-28s %.4g" % (c, t[c]))
    wc = t.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_FLAT", "SQ_ACTIVE_INST_MISC"):
            if c in t: print("    %-28s / SQ_WAVE_CYCLES = %.3f" % (c, t[c] / wc))
    if "SQ_THREAD_CYCLES_VALU" in t and "SQ_ACTIVE_INST_VALU" in t and t["SQ_ACTIVE_INST_VALU"]:
        print("    active lanes per VALU cycle (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU / 4) = %.1f of 64" % (t["SQ_THREAD_CYCLES_VALU"] / t["SQ_ACTIVE_INST_VALU"] / 4.0))
    if "SQ_INSTS_VALU" in t:
        for c in ("SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH", "SQ_INSTS_FLAT"):
            if c in t: print("    %-28s / SQ_INSTS_VALU = %.3f" % (c, t[c] / t["SQ_INSTS_VALU"]))
