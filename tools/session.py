"""One GPU-box session = one preset: the experiment drivers of every round folded into one file (rounds 3 and 4 kept a dozen
five-line shell scripts each).  Run on the box from the repo root:  python tools/session.py PRESET[,PRESET...] [LOG]  -- e.g.
`gpurun -- 'python tools/session.py saturation,pairs r05_shapes'`; the log lands in gpurun_out/LOG.log.

A preset is a list of steps: ("title", {env}, "command").  `variants.py` takes NSEQ LEN DNA 'K=V,K=V;...' (';' separates the
variants, an empty one = the defaults); SESSION_LIBS=reg,alt runs every step once per library build (tools/build_alt.sh NAME ...
makes kalign_amd/libkalign_amd_NAME.so; `reg` is the regular one)."""
import os, shutil, subprocess, sys

V = "python tools/variants.py"
PRESETS = {
    # the headline tree, C2, a C3-shaped tree under a variant string (SESSION_VARIANTS, default: the defaults alone)
    "headline": [("4096 x 400 aa", {}, V + " 4096 400 0 '{v}'"), ("1024 x 400 aa", {}, V + " 1024 400 0 '{v}'"),
                 ("1024 x 2000 nt", {"VAR_STEPS": "4"}, V + " 1024 2000 1 '{v}'")],
    "c3": [("4096 x 2000 nt", {"VAR_STEPS": "3"}, V + " 4096 2000 1 '{v}'")],
    "big": [("16384 x 500 aa", {"VAR_STEPS": "3"}, V + " 16384 500 0 '{v}'"), ("2048 x 1000 aa", {"VAR_STEPS": "3"}, V + " 2048 1000 0 '{v}'"),
            ("512 x 3000 nt", {"VAR_STEPS": "3"}, V + " 512 3000 1 '{v}'")],
    # 16 trees in flight as one forest (the saturated regime), four C3 trees
    "saturation": [("16 x (4096 x 400 aa)", {"VAR_COPIES": "16", "VAR_STEPS": "3"}, V + " 4096 400 0 '{v}'"),
                   ("4 x (4096 x 2000 nt)", {"VAR_COPIES": "4", "VAR_STEPS": "3"}, V + " 4096 2000 1 '{v}'")],
    # the N x 5 seq-seq batch of anchor consistency; default mode trees
    "pairs": [("pairs 4096 x 400", {"VAR_PAIRS": "5"}, V + " 4096 400 0 '{v}'"), ("pairs 16384 x 500", {"VAR_PAIRS": "5"}, V + " 16384 500 0 '{v}'")],
    "default_mode": [("4096 x 400 aa, 5 anchors", {"VAR_ANCHORS": "5", "VAR_STEPS": "4"}, V + " 4096 400 0 '{v}'")],
    # round 5's launch-shape and prefix-reuse questions as variant strings
    "shapes": [("16 trees", {"VAR_COPIES": "16", "VAR_STEPS": "3"}, V + " 4096 400 0 ';KA_QW=2;KA_QW=1;KA_LW=2;KA_LW=1;KA_QW=2,KA_LW=2'"),
               ("one tree", {}, V + " 4096 400 0 ';KA_QW=2;KA_QW=1;KA_LW=2;KA_LW=1'"),
               ("pairs", {"VAR_PAIRS": "5"}, V + " 16384 500 0 ';KA_REUSE=0;KA_PW=4;KA_PW=4,KA_REUSE=0;KA_PW=1'")],
    # per-level task times and the critical path; phase times alone and under load; refinement
    "levels": [("levels 4096 x 400", {}, "python tools/levels_real.py 0 4096 400 0 0 9,15,17"), ("levels 1024 x 400", {}, "python tools/levels_real.py 0 1024 400")],
    "phases": [("phases alone / loaded", {"KA_LAUNCH_EV": "1"}, "python tools/phases_loaded.py 16")],
    "refine": [("refinement 1024 x 400", {}, "python tools/refine_time.py 1024 400")],
    # correctness nets worth running with any kernel change
    "parity": [("parity", {}, "python -m pytest tests/test_gpu_parity.py tests/test_gpu_consistency.py tests/test_gpu_handover.py -x -q")],
    "stress": [("stress", {}, "python -m pytest tests/test_gpu_stress.py tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q")],
    "dropin": [("drop-in", {}, "python -m pytest tests/test_gpu_dropin.py tests/test_gpu_multi.py -x -q")],
}


def main():
    root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.chdir(root)
    os.makedirs("gpurun_out", exist_ok=True)
    names = sys.argv[1].split(",")
    log = os.path.join("gpurun_out", (sys.argv[2] if len(sys.argv) > 2 else "session_" + names[0]) + ".log")
    variants = os.environ.get("SESSION_VARIANTS", "")
    libs = os.environ.get("SESSION_LIBS", "reg").split(",")
    reg = os.path.join("kalign_amd", "libkalign_amd.so")
    shutil.copy(reg, "/tmp/reg.so")
    with open(log, "w") as out:
        for lib in libs:
            if lib != "reg":
                alt = os.path.join("kalign_amd", "libkalign_amd_%s.so" % lib)
                if not os.path.exists(alt):
                    continue
                shutil.copy(alt, reg)
            if len(libs) > 1:
                out.write("==== library: %s\n" % lib)
            for name in names:
                for title, env, cmd in PRESETS[name]:
                    out.write("== %s: %s\n" % (name, title))
                    out.flush()
                    r = subprocess.run("timeout 900 " + cmd.format(v=variants), shell=True, env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
                    text = "\n".join(l for l in r.stdout.decode().splitlines() if "amdgpu.ids" not in l)
                    out.write(text[-20000:] + "\n")
                    out.flush()
            shutil.copy("/tmp/reg.so", reg)
    print(open(log).read())


if __name__ == "__main__":
    main()
