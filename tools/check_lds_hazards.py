"""Build-time check (no GPU): does any instruction touch the destination of an LDS read that may still be in flight?

The strip steps read their next operands with `ds_read_b128` in inline asm and wait for them with a hand-placed `s_waitcnt
lgkmcnt(0)` one step later (ka_wstrip.h, ka_pass.h, ka_subtree.h): between the two the compiler believes the registers are already
written.  Where it then SPILLS one of them (scratch_store) or moves it, the stale content is saved and the data that lands later
is lost -- round 4's intermittent wrong prefix of a strip's last row.  The compiler's own (tracked) loads never show this: it
puts a wait in front of every use.  So: a forward may-analysis over every function of the disassembled objects --

  state   VGPR -> the least number of lgkm-counted operations issued after the read that targets it, over all paths
  ds_read / other lgkm operation: every age + 1, the read's destination registers enter at age 0
  s_waitcnt lgkmcnt(N): registers of age >= N are retired (LDS returns a wave's operations in order)
  a call (s_swappc_b64): everything retired -- the callee opens with s_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0) (checked)
  any other appearance of a tracked register in an instruction: reported

usage: check_lds_hazards.py [object ...]   (default: every kalign_amd/csrc/build/ka_kernels_u*.o); exit status 1 when anything is reported"""
import glob, os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
objs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "kalign_amd/csrc/build/ka_kernels_u*.o")))

RE_LINE = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
RE_V = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
RE_LGKM = re.compile(r"lgkmcnt\((\d+)\)")


def vregs(text):
    out = set()
    for m in RE_V.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def disassemble(obj):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "x.fat"), os.path.join(d, "x.co")
        subprocess.check_call([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj])
        subprocess.check_call([LLVM + "clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co, "--unbundle"])
        return subprocess.check_output([LLVM + "llvm-objdump", "-d", co]).decode().split("\n")


def functions(dis):
    name, cur = None, []
    for ln in dis:
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m:
            if name and cur:
                yield name, cur
            name, cur = m.group(1), []
            continue
        m = RE_LINE.match(ln)
        if m and name:
            cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
    if name and cur:
        yield name, cur


def is_lgkm(op):
    return op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_memtime") or \
        op.startswith("s_memrealtime") or op.startswith("flat_") or op.startswith("s_sendmsg") or op.startswith("s_dcache") or \
        op.startswith("s_scratch_load") or op.startswith("s_atc")


def long_target(ins, i):
    reg = ins[i][2].strip()
    m = re.match(r"s\[(\d+):(\d+)\]", reg)
    if not m: return None
    lo, hi = "s" + m.group(1), "s" + m.group(2)
    add_lo = add_hi = None
    for j in range(i - 1, max(i - 6, -1), -1):
        a, op, args = ins[j]
        parts = [x.strip() for x in args.split(",")]
        if op == "s_add_u32" and parts[0] == lo and parts[1] == lo: add_lo = int(parts[2], 0)
        elif op == "s_addc_u32" and parts[0] == hi and parts[1] == hi: add_hi = int(parts[2], 0)
        elif op == "s_getpc_b64" and args.strip() == reg:
            if add_lo is None or add_hi is None: return None
            off = (add_hi << 32) | (add_lo & 0xffffffff)
            if off >= 1 << 63: off -= 1 << 64
            return a + 4 + off
    return None


def analyse(name, ins):
    unresolved = []
    addr2i = {a: i for i, (a, _, _) in enumerate(ins)}
    n = len(ins)
    succ = [[] for _ in range(n)]
    for i, (a, op, args) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            try:
                off = int(args.split()[0])
            except ValueError:
                off = None
            if off is not None:
                if off >= 32768: off -= 65536
                t = addr2i.get(a + 4 + 4 * off)
                if t is not None: succ[i].append(t)
            if op != "s_branch" and i + 1 < n: succ[i].append(i + 1)
        elif op == "s_setpc_b64":
            # a return (s[30:31]), or a LONG BRANCH of a kernel past the 16-bit offsets:
            #   s_getpc_b64 s[x:y]; s_add_u32 sx, sx, lo; s_addc_u32 sy, sy, hi; s_setpc_b64 s[x:y]
            t = long_target(ins, i)
            if t is not None:
                if t in addr2i: succ[i].append(addr2i[t])
                else: unresolved.append((a, args))
            elif args.strip() != "s[30:31]":
                unresolved.append((a, args))
        elif op == "s_endpgm":
            pass
        elif i + 1 < n:
            succ[i].append(i + 1)
    state = [None] * n                                                # state BEFORE instruction i
    pred = {}                                                          # (j, reg) -> i: the edge the register's entry came in by (for the witness path)
    state[0] = {}
    work = [0]
    reports = {}
    longest, nlong = [0], [0]
    while work:
        i = work.pop()
        st = dict(state[i])
        a, op, args = ins[i]
        if op == "s_swappc_b64":
            st = {}                                                     # (a call: every function starts with s_waitcnt 0 -- checked below)
        elif op == "s_waitcnt":
            m = RE_LGKM.search(args)
            if m:
                k = int(m.group(1))
                for r, (g, at) in st.items():
                    # (statistics over straight-line stretches only: the index distance means nothing across a branch)
                    if g >= k and 0 < i - at < 400 and i - at > longest[0]: longest[0] = i - at
                    if g >= k and 24 < i - at < 400: nlong[0] += 1
                st = {r: ga for r, ga in st.items() if ga[0] < k}
        else:
            touched = vregs(args)
            hit = touched & st.keys()
            if hit:
                if i not in reports:
                    # witness: back along the edges the first register came in by, jumps only
                    r0, j, hops = sorted(hit)[0], i, []
                    while (j, r0) in pred and len(hops) < 40:
                        pj = pred[(j, r0)]
                        if pj + 1 != j: hops.append("%x->%x" % (ins[pj][0], ins[j][0]))
                        j = pj
                        if ins[j][1].startswith("ds_read") and r0 in vregs(ins[j][2].split(",")[0]): break
                    reports[i] = (a, op, args + "    [read at %x; jumps %s]" % (ins[j][0], " ".join(reversed(hops)) or "none"), sorted(hit))
                for r in hit: st.pop(r, None)                           # (reported once)
            if is_lgkm(op):
                st = {r: (g + 1, at) for r, (g, at) in st.items()}
                if op.startswith("ds_read") or op.startswith("ds_bpermute") or op.startswith("ds_permute") or op.startswith("ds_swizzle") or \
                        (op.startswith("ds_") and "_rtn" in op):
                    dst = args.split(",")[0]
                    for r in vregs(dst): st[r] = (0, i)
        for j in succ[i]:
            old = state[j]
            if old is None:
                state[j] = st; work.append(j)
                for r in st: pred[(j, r)] = i
            else:
                new = dict(old); changed = False
                for r, ga in st.items():
                    if r not in new or ga[0] < new[r][0]:
                        new[r] = ga; changed = True; pred[(j, r)] = i
                if changed:
                    state[j] = new; work.append(j)
    rep = [reports[k] for k in sorted(reports)]
    rep += [(a, 's_setpc_b64', args + '   (target not resolved: the analysis is incomplete)', []) for a, args in unresolved]
    return rep, longest[0], nlong[0]


bad = 0
for obj in objs:
    dis = disassemble(obj)
    nf = nr = 0
    for name, ins in functions(dis):
        nf += 1
        reads = sum(1 for _, op, _ in ins if op.startswith("ds_read"))
        rep, longest, nlong = analyse(name, ins)
        # a device function (anything that returns with s_setpc_b64) must open with the wait a call relies on
        if any(op == "s_setpc_b64" and ar.strip() == "s[30:31]" for _, op, ar in ins) and not (ins[0][1] == "s_waitcnt" and "lgkmcnt(0)" in ins[0][2]):
            rep.append((ins[0][0], ins[0][1], ins[0][2] + "   (function entry without s_waitcnt lgkmcnt(0))", []))
        print("%s: %s: %d instructions, %d LDS reads (longest straight-line flight: %d instructions; registers in flight for > 24: %d), %d suspicious uses"
              % (os.path.basename(obj), name[:70], len(ins), reads, longest, nlong, len(rep)))
        for a, op, args, regs in (rep if os.environ.get("KA_HZ_ALL") else rep[:12]):
            print("    %08x  %s %s    <- in flight: %s" % (a, op, args, ", ".join("v%d" % r for r in regs)))
        nr += len(rep)
    bad += nr
print("suspicious uses in total: %d" % bad)
sys.exit(1 if bad else 0)
