"""Build-time check (no GPU): disassemble a task-kernel object and report, for every run of strip steps (consecutive ds_write_b96 --
the out-ring write of ka_wstrip's step), instructions and scratch accesses per step, plus the kernel's register / spill metadata.
The STEADY-state octets must have NO scratch access: exit 1 when fewer than --steady (default 6: two-row and one-row steps, first and
later strips, 20 / 23 / 5 residues as the unit has them) octets are scratch-free, or when any octet carries more than --max-scratch
(default 20) scratch accesses per step.  The head / tail / edge forms of the step do spill (8-19 accesses per step in unit 0, DESIGN.md
section 4f / 6d): reported, and bounded by --max-scratch so that they cannot grow unnoticed.  Run by __graft_entry__.build().
usage: check_hot_loops.py [kalign_amd/csrc/build/ka_kernels_u0.o] [--steady N] [--max-scratch M] [--json OUT: the octets as data -- bench.py reads
the instructions per step of its VALU roofline from it (kalign_amd/hot_loops.json, written by __graft_entry__.build())]"""
import os, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin/"
args = [a for a in sys.argv[1:] if not a.startswith("--")]
def opt(name, dflt):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else dflt
args = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and not sys.argv[i - 1].startswith("--")]
STEADY, MAXS = opt("--steady", 6), opt("--max-scratch", 20)
obj = args[0] if args else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kalign_amd/csrc/build/ka_kernels_u0.o")
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, "x.fat"), os.path.join(d, "x.co")
    subprocess.check_call([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj])
    subprocess.check_call([LLVM + "clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co, "--unbundle"])
    notes = subprocess.check_output([LLVM + "llvm-readelf", "--notes", co]).decode()
    for ln in notes.split("\n"):
        if any(k in ln for k in (".name:", "private_segment_fixed_size", "sgpr_spill_count", "vgpr_spill_count", ".vgpr_count")) and ".kd" not in ln:
            print(ln.strip())
    dis = subprocess.check_output([LLVM + "llvm-objdump", "-d", co]).decode().split("\n")
idx = [i for i, l in enumerate(dis) if "ds_write_b96" in l]
def count(a, b, pat=None):
    n = 0
    for l in dis[a + 1:b + 1]:
        t = l.strip()
        if not t or t.startswith(";") or t.endswith(":"): continue
        if pat is None or pat in t: n += 1
    return n
runs, cur = [], []
for a, b in zip(idx, idx[1:]):
    if b - a < 260: cur.append((count(a, b), count(a, b, "scratch_"), count(a, b, "v_pk_mul_f32"), count(a, b, "v_")))
    else:
        if len(cur) >= 6: runs.append(cur)
        cur = []
if len(cur) >= 6: runs.append(cur)
bad, worst = 0, 0
octets = []
for r in runs:
    n = sorted(x[0] for x in r)[len(r) // 2]
    sc = max(x[1] for x in r[1:-1]) if len(r) > 2 else max(x[1] for x in r)
    pk = sorted(x[2] for x in r)[len(r) // 2]
    valu = sorted(x[3] for x in r)[len(r) // 2]
    print("octet: %d steps, median %d instructions per step (%d vector, %d v_pk_mul_f32), scratch accesses per step (inner steps) <= %d" % (len(r) + 1, n, valu, pk, sc))
    octets.append({"steps": len(r) + 1, "instructions_per_step": n, "vector_instructions_per_step": valu, "v_pk_mul_per_step": pk, "scratch_per_step": sc})
    bad += sc > 0
    worst = max(worst, sc)
if "--json" in sys.argv:
    import json
    with open(sys.argv[sys.argv.index("--json") + 1], "w") as f:
        json.dump({"object": os.path.basename(obj), "octets": octets}, f)
clean = len(runs) - bad
print("octets with scratch traffic: %d of %d (the edge forms); scratch-free: %d (needed: %d); worst %d accesses per step (allowed: %d)" % (bad, len(runs), clean, STEADY, worst, MAXS))
sys.exit(0 if (clean >= STEADY and worst <= MAXS) else 1)
