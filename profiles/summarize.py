"""Fold the rocprofv3 outputs of profiles/collect.sh into small tracked summaries.

usage: python profiles/summarize.py <tag> <dir with kt/ pmc_fetch/ pmc_write/ pmc_sq/ bench.json>
writes profiles/<tag>_kernel_stats.csv, <tag>_last_step_launches.csv, <tag>_pmc_sq.csv,
<tag>_bench.json and profiles/r01_pmc_traffic.json (the file bench.py reads for roofline.traffic).
"""
import csv, glob, json, os, sys

tag, d = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))


def find(sub, suffix):
    hits = glob.glob(os.path.join(d, sub, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def rows(path):
    with open(path) as f:
        return list(csv.DictReader(f))


LAUNCHES_FOR_TRACE = 13
_b = os.path.join(d, "bench.json")
if os.path.exists(_b):
    for _l in open(_b).read().splitlines():
        if _l.startswith("{"):
            LAUNCHES_FOR_TRACE = int(json.loads(_l)["roofline"]["launches_per_step"])
# ---- kernel trace: stats + the launches of the last timed step ----
st = find("kt", "kernel_stats.csv")
if st:
    with open(st) as f, open(os.path.join(here, tag + "_kernel_stats.csv"), "w") as g:
        g.write(f.read())
tr = find("kt", "kernel_trace.csv")
if tr:
    r = [x for x in rows(tr) if x["Kernel_Name"].startswith("ka_task_kernel")]
    r.sort(key=lambda x: int(x["Start_Timestamp"]))
    last = r[-LAUNCHES_FOR_TRACE:]
    with open(os.path.join(here, tag + "_last_step_launches.csv"), "w") as g:
        g.write("kernel,grid,workgroup,lds_bytes,vgprs,sgprs,duration_us\n")
        for x in last:
            g.write("%s,%s,%s,%s,%s,%s,%.1f\n" % (x["Kernel_Name"].split("(")[0], x["Grid_Size_X"], x["Workgroup_Size_X"], x.get("LDS_Block_Size", ""),
                                            x.get("VGPR_Count", ""), x.get("SGPR_Count", ""),
                                            (int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3))

# ---- PMC: HBM traffic per step ----
def pmc_sum(sub, counter):
    p = find(sub, "counter_collection.csv")
    if not p:
        return None, 0
    tot, n = 0.0, 0
    for x in rows(p):
        if x["Kernel_Name"].startswith("ka_task_kernel") and x["Counter_Name"] == counter:
            tot += float(x["Counter_Value"]); n += 1
    return tot, n

fetch, nf = pmc_sum("pmc_fetch", "FETCH_SIZE")
write, nw = pmc_sum("pmc_write", "WRITE_SIZE")
LAUNCHES = 13
b0 = os.path.join(d, "bench.json")
if os.path.exists(b0):
    for l in open(b0).read().splitlines():
        if l.startswith("{"):
            LAUNCHES = int(json.loads(l)["roofline"]["launches_per_step"])
if fetch is not None and write is not None and nf and nw:
    steps_f, steps_w = nf / LAUNCHES, nw / LAUNCHES
    fkb, wkb = fetch / steps_f, write / steps_w
    corrected = (2.0 * fkb + wkb) * 1024.0
    out = {"workload": "bench.py default (1024 protein x ~400, dssim seed 1)", "tag": tag, "launches_per_step": LAUNCHES,
           "FETCH_SIZE_KB_per_step": fkb, "WRITE_SIZE_KB_per_step": wkb,
           "hbm_bytes_per_step_corrected": corrected, "hbm_bytes_per_launch_corrected": corrected / LAUNCHES,
           "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); WRITE_SIZE uncalibrated"}
    json.dump(out, open(os.path.join(here, "r01_pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(out))

# ---- PMC: SQ issue counters, summed over the task kernels ----
p = find("pmc_sq", "counter_collection.csv")
if p:
    acc = {}
    for x in rows(p):
        if x["Kernel_Name"].startswith("ka_task_kernel"):
            k = (x["Kernel_Name"].split("(")[0], x["Counter_Name"])
            acc[k] = acc.get(k, 0.0) + float(x["Counter_Value"])
    with open(os.path.join(here, tag + "_pmc_sq.csv"), "w") as g:
        g.write("kernel,counter,sum_over_launches\n")
        for (k, c), v in sorted(acc.items()):
            g.write("%s,%s,%.0f\n" % (k, c, v))

b = os.path.join(d, "bench.json")
if os.path.exists(b):
    line = [l for l in open(b).read().splitlines() if l.startswith("{")]
    if line:
        open(os.path.join(here, tag + "_bench.json"), "w").write(line[-1] + "\n")
        print(line[-1][:300])
