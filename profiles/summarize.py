"""Fold the rocprofv3 outputs of profiles/collect.sh into small tracked summaries.

usage: python profiles/summarize.py <tag> <workload> <dir with kt/ pmc_fetch/ pmc_write/ [pmc_sq/] bench.json>
writes profiles/<tag>_<workload>_kernel_stats.csv, _last_step_launches.csv, [_pmc_sq.csv], _bench.json and updates
profiles/<tag>_pmc_traffic.json (the file bench.py reads for roofline.traffic; one entry per workload).
"""
import csv, glob, json, os, sys

tag, wl, d = sys.argv[1], sys.argv[2], sys.argv[3]
here = os.path.dirname(os.path.abspath(__file__))
KEY = {"headline": "headline", "c2": "c2_1024x400", "c3": "c3_dna_4096x2000"}.get(wl, wl)
pre = "%s_%s" % (tag, wl)


def find(sub, suffix):
    hits = glob.glob(os.path.join(d, sub, "**", "*" + suffix), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None      # (a re-collected directory keeps the older runs' files)


def rows(path):
    with open(path) as f:
        return list(csv.DictReader(f))


bench = None
_b = os.path.join(d, "bench.json")
if os.path.exists(_b):
    for _l in open(_b).read().splitlines():
        if _l.startswith("{"):
            bench = json.loads(_l)
LAUNCHES = int(bench["roofline"]["launches_per_step"]) if bench else 1
# ---- kernel trace: stats + the launches of the last timed step ----
st = find("kt", "kernel_stats.csv")
if st:
    with open(st) as f, open(os.path.join(here, pre + "_kernel_stats.csv"), "w") as g:
        g.write(f.read())
tr = find("kt", "kernel_trace.csv")
if tr:
    r = [x for x in rows(tr) if x["Kernel_Name"].startswith("ka_task_kernel")]
    r.sort(key=lambda x: int(x["Start_Timestamp"]))
    last = r[-LAUNCHES:]
    tot = 0.0
    with open(os.path.join(here, pre + "_last_step_launches.csv"), "w") as g:
        g.write("kernel,grid,workgroup,lds_bytes,vgprs,sgprs,start_us,duration_us\n")
        for x in last:
            us = (int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3
            tot += us
            g.write("%s,%s,%s,%s,%s,%s,%.1f,%.1f\n" % (x["Kernel_Name"].split("(")[0], x["Grid_Size_X"], x["Workgroup_Size_X"], x.get("LDS_Block_Size", ""),
                                            x.get("VGPR_Count", ""), x.get("SGPR_Count", ""), (int(x["Start_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e3, us))
    # mean over every traced step (13 = 3 warm-up + 10 timed)
    nsteps = len(r) // LAUNCHES if LAUNCHES else 0
    if nsteps:
        allus = sum((int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3 for x in r)
        # Round 5: the chained launch runs on its own stream beside the queued one (it starts with it and waits for its operands),
        # so the durations of a step's launches overlap and their sum is more than the step takes.  What bench.py times with HIP
        # events -- and what the roofline divides by -- is the time the step's launches COVER: the union of their intervals.
        cover = 0.0
        for s in range(nsteps):
            iv = sorted((int(x["Start_Timestamp"]), int(x["End_Timestamp"])) for x in r[s * LAUNCHES:(s + 1) * LAUNCHES])
            lo, hi = iv[0]
            for a, b in iv[1:]:
                if a > hi:
                    cover += hi - lo; lo, hi = a, b
                else:
                    hi = max(hi, b)
            cover += hi - lo
        line = ("kernel-trace: %d task-kernel launches = %d steps x %d; time covered by a step's launches %.3f ms (bench.py's HIP events: kernel_ms_per_step %.3f); "
                "sum of the launch durations per step %.3f ms (last step %.3f ms) -- the chained launch overlaps the queued one" % (
                    len(r), nsteps, LAUNCHES, cover / nsteps / 1e6, bench["roofline"]["kernel_ms_per_step"] if bench else -1, allus / nsteps / 1e3, tot / 1e3))
        print(line)
        open(os.path.join(here, pre + "_kernel_trace_summary.txt"), "w").write(line + "\n")


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
if fetch is not None and write is not None and nf and nw:
    steps_f, steps_w = nf / LAUNCHES, nw / LAUNCHES
    fkb, wkb = fetch / steps_f, write / steps_w
    corrected = (2.0 * fkb + wkb) * 1024.0
    path = os.path.join(here, tag + "_pmc_traffic.json")
    allw = json.load(open(path)) if os.path.exists(path) else {}
    allw[KEY] = {"workload": bench["config"]["workload"] if bench else wl, "tag": pre, "launches_per_step": LAUNCHES,
                 "FETCH_SIZE_KB_per_step": fkb, "WRITE_SIZE_KB_per_step": wkb, "hbm_bytes_per_step_corrected": corrected,
                 "algorithmic_bytes_per_step": bench["roofline"]["algorithmic_bytes_per_step"] if bench else None,
                 "kernel_ms_per_step": bench["roofline"]["kernel_ms_per_step"] if bench else None,     # bench.py matches its live kernel time against this (5 %)
                 "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE), summed over the task-kernel launches of a step; FETCH_SIZE doubled per "
                         "MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); WRITE_SIZE uncalibrated"}
    json.dump(allw, open(path, "w"), indent=1)
    print(json.dumps(allw[KEY]))

# ---- PMC: SQ issue counters, summed over the task kernels ----
p = find("pmc_sq", "counter_collection.csv")
if p:
    acc = {}
    for x in rows(p):
        if x["Kernel_Name"].startswith("ka_task_kernel"):
            k = (x["Kernel_Name"].split("(")[0], x["Counter_Name"])
            acc[k] = acc.get(k, 0.0) + float(x["Counter_Value"])
    with open(os.path.join(here, pre + "_pmc_sq.csv"), "w") as g:
        g.write("kernel,counter,sum_over_launches\n")
        for (k, c), v in sorted(acc.items()):
            g.write("%s,%s,%.0f\n" % (k, c, v))

if bench:
    open(os.path.join(here, pre + "_bench.json"), "w").write(json.dumps(bench) + "\n")
    print(json.dumps(bench)[:300])
