"""Wall time of `kalign --ensemble=8 --realign 1` (BASELINE config 5's member loop) through the drop-in binary on one GPU: members
one after the other (KALIGN_AMD_ENSEMBLE_SLOTS=1) against side by side on contexts sharing the GPU (the default), and the
reference binary beside it.  The consensus stage behind the members (POAR tables over all pairs of rows, host, the reference's own
code in both binaries) grows with N^2 L and takes minutes at 2048 sequences -- the default here is 512.
python tools/dropin_ensemble_time.py [NSEQ 512] [LEN 300] [MEMBERS 8]"""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 512
length = int(sys.argv[2]) if len(sys.argv) > 2 else 300
members = int(sys.argv[3]) if len(sys.argv) > 3 else 8
seqs = bench.workload_letters(nseq, length, False, 5)
tmp = tempfile.mkdtemp()
inp = os.path.join(tmp, "in.fa")
with open(inp, "w") as fh:
    for i, s in enumerate(seqs):
        fh.write(">s%d\n%s\n" % (i, s if isinstance(s, str) else s.decode()))
flags = ["--ensemble=%d" % members, "--realign", "1"]
outs = {}
for name, binary, env in (("drop-in, members one after the other", "dropin/kalign", {"KALIGN_AMD_ENSEMBLE_SLOTS": "1"}),
                          ("drop-in, members side by side (default)", "dropin/kalign", {}),
                          ("drop-in, --ensemble off (one alignment: process start + I/O + one member)", "dropin/kalign", None),
                          ("reference, 16 threads", "kalign_ref", {})):
    out = os.path.join(tmp, "out_%d.fa" % len(outs))
    f = ["--realign", "1"] if env is None else flags
    for rep in range(2):
        t0 = time.perf_counter()
        r = subprocess.run([os.path.join(root, "oracle", "_ref", binary), "-i", inp, "-o", out, "-n", "16"] + f,
                           env=dict(os.environ, OMP_NUM_THREADS="16", KALIGN_AMD_GLUE_REPORT="1", **(env or {})), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        dt = time.perf_counter() - t0
        if binary == "kalign_ref":
            break
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    rep = [ln for ln in r.stderr.decode().splitlines() if ln.startswith("kalign_amd_glue:")]
    outs[name] = open(out, "rb").read()
    print("%-80s %8.0f ms   %s" % (name, dt * 1e3, rep[-1][:200] if rep else ""), flush=True)
vals = [v for k, v in outs.items() if "off" not in k]
print("aligned FASTA identical across the three ensemble runs:", all(v == vals[0] for v in vals))
