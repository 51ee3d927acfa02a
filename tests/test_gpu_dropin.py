"""The drop-in boundary, for real: Kalign's own library and CLI with the MI355X dispatcher underneath.

oracle/_ref/dropin/libkalign.so.3 is the reference's lib/src compiled where it lies with six seams replaced by
oracle/dropin/kalign_amd_glue.c (= the text of INTEGRATION.md) and linked against kalign_amd/libkalign_amd.so;
oracle/_ref/dropin/kalign is the reference's CLI (src/run_kalign.c) on that library.  Built by `make -C oracle dropin`
in the build container (__graft_entry__.build), shipped prebuilt to the GPU box.  Every result must be byte-identical
to the unmodified reference (oracle/_ref/libkalign_ref.so, oracle/_ref/kalign_ref) on the same input."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
DATA = os.path.join(ROOT, "tests", "golden", "data")

KALIGN_TYPE_DNA_INTERNAL = 1          # lib/include/kalign/kalign.h:18-27
KALIGN_TYPE_PROTEIN = 3
KALIGN_TYPE_UNDEFINED = 8


def _need(name):
    p = os.path.join(REFDIR, name)
    assert os.path.exists(p), "%s missing: run `make -C oracle dropin cli` in the build container" % p
    return p


def _cli(binary, infile, outfile, *flags, counters=None, env_extra=None):
    """counters: a dict that receives the seam counters the drop-in prints at exit (KALIGN_AMD_GLUE_REPORT=1): how often
    every seam ran on the device and how often it handed the call to the reference's own function"""
    env = dict(os.environ, OMP_NUM_THREADS="8", KALIGN_AMD_GLUE_REPORT="1")
    env.update(env_extra or {})
    r = subprocess.run([_need(binary), "-i", infile, "-o", outfile, "-n", "8"] + list(flags), env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    if counters is not None:
        lines = [ln for ln in r.stderr.decode().splitlines() if ln.startswith("kalign_amd_glue:")]
        assert lines, "the drop-in did not report its seam counters:\n" + r.stderr.decode()[-1500:]
        counters.update({k: int(v) for k, v in (kv.split("=") for kv in lines[-1].split()[1:])})
    with open(outfile, "rb") as fh:
        return fh.read()


def _device_ran(c, flags):
    """the seams a run with these CLI flags must have taken ON THE DEVICE -- a byte-identical FASTA alone would also come out
    of a run that silently fell back to the reference's own functions"""
    for k in ("refine_ref", "finalise_ref", "cons_ref", "alndist_ref", "alntree_ref", "inline_ref"):
        assert c[k] == 0, (k, c)
    precise = "--precise" in flags
    realign = int(flags[flags.index("--realign") + 1]) if "--realign" in flags else (1 if precise else 0)
    # --precise: ensemble of 3 + one realignment (src/run_kalign.c:375-383)
    ens = [f for f in flags if f.startswith("--ensemble=")]
    members = int(ens[0].split("=")[1]) if ens else (3 if precise else 1)
    assert c["tree"] + c["inline"] >= members * (1 + realign), c
    assert c["finalise"] >= members * (1 + realign), c
    assert c["kmeans"] + c["kmeans_noisy"] >= members, c
    if members > 1:
        # the members ran ahead of the reference's loop, side by side on contexts sharing the GPU (kalign_ensemble in the glue)
        assert c["ensemble_multi"] == 1 and c["member_ahead"] == members and c["member_missed"] == 0, c
    if members > 1 and not realign:
        assert c["kmeans_noisy"] >= 1, c                      # kalign_run_seeded: members after the first build their trees on noisy distances
    if realign:
        assert c["alndist"] >= members * realign and c["alntree"] == c["alndist"], c
    if "--fast" not in flags and members == 1:
        assert c["cons"] >= 1, c                               # default mode: the N x K batch and the position maps
    if "--refine" in flags:
        assert c["refine"] >= 1, c


def _write_fasta(path, seqs):
    with open(path, "w") as fh:
        for i, s in enumerate(seqs):
            fh.write(">seq%d\n%s\n" % (i, s))


CLI_CASES = [
    ("BB11001.tfa", []), ("BB11001.tfa", ["--fast"]), ("BB11001.tfa", ["--precise"]), ("BB12006.tfa", ["--fast", "--realign", "2"]),
    ("BB11001.tfa", ["--ensemble=3"]),
    ("BB30014.tfa", []), ("BB30014.tfa", ["--fast"]),
    ("BB12006.tfa", []), ("BB12006.tfa", ["--realign", "1"]),
    ("BB30014.tfa", ["--precise"]),
    # refinement (aln_refine.c) on the device: ka_tree_refine behind refine_alignment
    ("BB11001.tfa", ["--refine", "all"]), ("BB30014.tfa", ["--refine", "all"]), ("BB30014.tfa", ["--refine", "confident"]),
    ("BB12006.tfa", ["--fast", "--refine", "confident"]), ("BB12006.tfa", ["--refine", "all", "--realign", "1"]),
    ("BB30014.tfa", ["--refine", "all", "--adaptive-budget"]), ("BB12006.tfa", ["--refine", "confident", "--adaptive-budget"]),
    # more than five anchors (round 4): the second set of consistency kernels
    ("BB11001.tfa", ["--consistency", "8"]), ("BB30014.tfa", ["--consistency", "10", "--refine", "all"]), ("BB12006.tfa", ["--consistency", "6", "--realign", "1"]),
    # round 5: any K up to 32 on the device (the entries of a DP row are walked, not held in registers)
    ("BB30014.tfa", ["--consistency", "16"]), ("BB30014.tfa", ["--consistency", "32", "--refine", "confident"]),
]


@pytest.mark.parametrize("name,flags", CLI_CASES, ids=["%s%s" % (n.split(".")[0], "_".join([""] + f).replace("--", "").replace("__", "_")) for n, f in CLI_CASES])
def test_cli_output_is_byte_identical(tmp_path, name, flags):
    """`kalign -i in -o out [flags]`: the aligned FASTA written by the drop-in equals the reference's."""
    inp = os.path.join(DATA, name)
    c = {}
    got = _cli("dropin/kalign", inp, str(tmp_path / "dropin.fa"), *flags, counters=c)
    want = _cli("kalign_ref", inp, str(tmp_path / "ref.fa"), *flags)
    assert len(want) > 100 and got == want
    _device_ran(c, flags)


def test_cli_with_more_anchors_than_the_kernels_carry_takes_the_reference_seams(tmp_path):
    """`--consistency 160` (the device kernels walk up to KA_CONS_MAX_ANCHORS = 128 entries per DP row): the library declines the table
    (cons_ref), the dispatcher hands the trees to the reference's own create_msa_tree (tree_ref) -- same bytes, no error."""
    from kalign_amd import synth
    inp = str(tmp_path / "in.fa")
    _write_fasta(inp, synth.dssim(176, 90, dna=False, seed=3))   # (the reference caps the anchors at the number of sequences)
    c = {}
    got = _cli("dropin/kalign", inp, str(tmp_path / "dropin.fa"), "--consistency", "160", counters=c)
    want = _cli("kalign_ref", inp, str(tmp_path / "ref.fa"), "--consistency", "160")
    assert got == want
    assert c["cons_ref"] >= 1 and c["tree_ref"] >= 1 and c["cons"] == 0, c


def test_cli_with_forty_and_a_hundred_anchors_stays_on_the_device(tmp_path):
    """round 6: `--consistency K` up to 128 on the device (a row's entries are collected and walked in the task's scratch, whatever K)"""
    from kalign_amd import synth
    inp = str(tmp_path / "in.fa")
    _write_fasta(inp, synth.dssim(120, 120, dna=False, seed=4))
    for k in ("40", "100"):
        c = {}
        got = _cli("dropin/kalign", inp, str(tmp_path / ("dropin%s.fa" % k)), "--consistency", k, counters=c)
        want = _cli("kalign_ref", inp, str(tmp_path / ("ref%s.fa" % k)), "--consistency", k)
        assert got == want
        assert c["cons"] >= 1 and c["cons_ref"] == 0 and c["tree_ref"] == 0, c


@pytest.mark.parametrize("dna,n,length,flags", [(False, 32, 200, []), (False, 32, 200, ["--fast"]),
                                                (True, 16, 300, ["--type", "dna"]), (False, 200, 150, []),
                                                (False, 40, 150, ["--consistency", "8"]), (True, 24, 300, ["--type", "dna", "--consistency", "10", "--refine", "all"])])
def test_cli_on_dssim_sets(tmp_path, dna, n, length, flags):
    from kalign_amd import synth
    inp = str(tmp_path / "in.fa")
    _write_fasta(inp, synth.dssim(n, length, dna=dna, seed=1))
    c = {}
    got = _cli("dropin/kalign", inp, str(tmp_path / "dropin.fa"), *flags, counters=c)
    want = _cli("kalign_ref", inp, str(tmp_path / "ref.fa"), *flags)
    assert got == want
    _device_ran(c, flags)


def _lib_kalign(libname, seqs, type_, n_threads=4):
    """kalign() of lib/include/kalign/kalign.h:36-40 through ctypes"""
    L = C.CDLL(_need(libname))
    L.kalign.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                         C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(C.c_int)]
    n = len(seqs)
    arr = (C.c_char_p * n)(*[s.encode() for s in seqs])
    lens = (C.c_int * n)(*[len(s) for s in seqs])
    out = C.POINTER(C.c_char_p)()
    alen = C.c_int(0)
    rc = L.kalign(arr, lens, n, n_threads, type_, -1.0, -1.0, -1.0, C.byref(out), C.byref(alen))
    assert rc == 0
    return [out[i].decode() for i in range(n)], alen.value


def _distinct_lengths(seqs):
    """kalign_arr_to_msa leaves the sequence names uninitialised (msa_op.c:482) and msa_sort_len_name breaks length
    ties by name (msa_sort.c:62-80): with equal lengths the REFERENCE's own result changes from call to call.  Sets
    whose lengths are all different have one defined answer."""
    seen, out = set(), []
    for s in seqs:
        if len(s) not in seen:
            seen.add(len(s))
            out.append(s)
    return out


def test_library_kalign_entry_point():
    """the 4-sequence DNA case of the reference's tests/kalign_lib_test.c:33-46 and DSSim protein sets through
    kalign(): same rows from the drop-in and from the reference; residues preserved"""
    from kalign_amd import synth
    dna = ["ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTTGCATGCATGCATGCATGCATGCATGCA"] * 3 + \
          ["ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT"]
    prot_a = _distinct_lengths(synth.dssim(64, 200, seed=1))
    prot_b = _distinct_lengths(synth.dssim(96, 120, seed=3))
    assert len(prot_a) >= 16 and len(prot_b) >= 12
    for seqs, type_ in ((dna, KALIGN_TYPE_DNA_INTERNAL), (prot_a, KALIGN_TYPE_PROTEIN), (prot_b, KALIGN_TYPE_UNDEFINED),
                        (prot_a, KALIGN_TYPE_PROTEIN)):
        got, glen = _lib_kalign("dropin/libkalign.so.3", seqs, type_)
        want, wlen = _lib_kalign("libkalign_ref.so", seqs, type_)
        assert glen == wlen and got == want
        assert [r.replace("-", "") for r in got] == list(seqs)
        assert all(len(r) == glen for r in got)


def _lib_run_file(libname, infile, outfile, refine, n_threads=4):
    """kalign_read_input + kalign_run + kalign_write_msa of lib/include/kalign/kalign.h:36-50"""
    L = C.CDLL(_need(libname))
    L.kalign_read_input.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.c_int]
    L.kalign_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int]
    L.kalign_write_msa.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    L.kalign_free_msa.argtypes = [C.c_void_p]
    msa = C.c_void_p()
    assert L.kalign_read_input(infile.encode(), C.byref(msa), 1) == 0
    assert L.kalign_run(msa, n_threads, KALIGN_TYPE_UNDEFINED, -1.0, -1.0, -1.0, refine, 0) == 0
    assert L.kalign_write_msa(msa, outfile.encode(), b"fasta") == 0
    L.kalign_free_msa(msa)
    with open(outfile, "rb") as fh:
        return fh.read(), L


@pytest.mark.parametrize("refine", [1, 2, 3], ids=["all", "confident", "inline"])
def test_library_kalign_run_with_refinement(tmp_path, refine):
    """kalign_run(..., refine, 0): KALIGN_REFINE_ALL / _CONFIDENT go through refine_alignment, KALIGN_REFINE_INLINE
    (not reachable from the CLI) through create_msa_tree_inline_refine -- all three on the device in the drop-in"""
    inp = os.path.join(DATA, "BB30014.tfa")
    L0 = C.CDLL(_need("dropin/libkalign.so.3"))
    before = [L0.kalign_amd_glue_count(k) for k in range(6)]
    got, L = _lib_run_file("dropin/libkalign.so.3", inp, str(tmp_path / "dropin.fa"), refine)
    want, _ = _lib_run_file("libkalign_ref.so", inp, str(tmp_path / "ref.fa"), refine)
    assert len(want) > 100 and got == want
    # which seams ran on the device: (tree, inline tree, refine, refine by the reference, finalise, finalise by the reference)
    delta = [L.kalign_amd_glue_count(k) - b for k, b in enumerate(before)]
    assert delta == ([0, 1, 0, 0, 1, 0] if refine == 3 else [1, 0, 1, 0, 1, 0])


# ---- several GPUs under the drop-in (kalign_amd_glue.c: glue_multi_context; ka_multi_* in the library) ----
# KALIGN_AMD_GLUE_WORLD=n: n ranks of the sharded path as threads of the kalign process, all on the box's one GPU over the
# library's in-process transport -- on a node with several GPUs the same code runs one rank per device over RCCL
# (KALIGN_AMD_DEVICES).  create_msa_tree and anchor_consistency_build must have gone through the sharded path (tree_multi /
# cons_multi counters), the single-device forms of those seams must NOT have run, and the FASTA must be the reference's.
MULTI_CASES = [("BB11001.tfa", []), ("BB11001.tfa", ["--fast"]), ("BB30014.tfa", []), ("BB30014.tfa", ["--fast"]),
               ("BB12006.tfa", []), ("BB12006.tfa", ["--fast", "--realign", "2"]), ("BB11001.tfa", ["--ensemble=3"]),
               ("BB30014.tfa", ["--refine", "all"]), ("BB30014.tfa", ["--refine", "confident", "--realign", "1"]),
               ("BB11001.tfa", ["--ensemble=8"]), ("BB11001.tfa", ["--precise"]), ("BB30014.tfa", ["--ensemble=5", "--fast"])]


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("name,flags", MULTI_CASES, ids=["%s%s" % (n.split(".")[0], "_".join([""] + f).replace("--", "").replace("__", "_")) for n, f in MULTI_CASES])
def test_cli_with_several_ranks_under_the_dropin(tmp_path, name, flags, world):
    inp = os.path.join(DATA, name)
    c = {}
    got = _cli("dropin/kalign", inp, str(tmp_path / "dropin.fa"), *flags, counters=c, env_extra={"KALIGN_AMD_GLUE_WORLD": str(world)})
    want = _cli("kalign_ref", inp, str(tmp_path / "ref.fa"), *flags)
    assert len(want) > 100 and got == want
    realign = int(flags[flags.index("--realign") + 1]) if "--realign" in flags else 0
    ens = [f for f in flags if f.startswith("--ensemble=")]
    members = int(ens[0].split("=")[1]) if ens else 1
    precise = "--precise" in flags
    if precise:                                               # ensemble of 3 + one realignment (src/run_kalign.c:375-383)
        members, realign = 3, 1
    if members > 1:
        # round 5: the members of an ensemble run side by side, member k on device (here: context) k mod G, each through the
        # single-device seams on its own thread (kalign_ensemble in the glue); the reference's loop then takes them over
        assert c["ensemble_multi"] == 1 and c["member_ahead"] == members and c["member_missed"] == 0, c
        assert c["tree"] >= members * (1 + realign), c
    else:
        assert c["tree_multi"] >= members * (1 + realign) and c["tree"] == 0, c
    if "--fast" not in flags and members == 1:
        assert c["cons_multi"] >= 1 and c["cons"] == 0 and c["cons_ref"] == 0, c
    # round 5: after a sharded run rank 0's context takes the alignment over (ka_multi_adopt) -- the stages behind the
    # dispatcher stay on the device, exactly as the single-device cases assert
    for k in ("refine_ref", "finalise_ref", "alndist_ref", "alntree_ref"):
        assert c[k] == 0, (k, c)
    assert c["finalise"] >= members * (1 + realign), c
    if realign:
        assert c["alndist"] >= members * realign and c["alntree"] == c["alndist"], c
    if "--refine" in flags:
        assert c["refine"] >= 1, c


_SEEDED_SCRIPT = r"""
import ctypes as C, sys
lib, inp, out, refine, anchors = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
L = C.CDLL(lib)
L.kalign_read_input.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.c_int]
L.kalign_run_seeded.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_uint64, C.c_float,
                                C.c_float, C.c_float, C.c_float, C.c_int, C.c_float]
L.kalign_write_msa.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
msa = C.c_void_p()
assert L.kalign_read_input(inp.encode(), C.byref(msa), 1) == 0
assert L.kalign_run_seeded(msa, 4, 8, -1.0, -1.0, -1.0, refine, 0, 0, 0.0, 0.0, -1.0, -1.0, anchors, 2.0) == 0
assert L.kalign_write_msa(msa, out.encode(), b"fasta") == 0
if hasattr(L, "kalign_amd_glue_count"):
    print("COUNTS", " ".join(str(L.kalign_amd_glue_count(k)) for k in range(19)))
"""


@pytest.mark.parametrize("world", [1, 2, 3])
def test_inline_refinement_with_consistency_under_several_ranks(tmp_path, world):
    """kalign_run_seeded(refine = KALIGN_REFINE_INLINE, 5 anchors) -- anchor_consistency_build runs on the
    ranks of the node, create_msa_tree_inline_refine on the single-GPU context, which never saw that table: it must build its
    own instead of trusting a table that lives on other contexts (the output silently lost its bonus before round 5)."""
    import sys
    inp = os.path.join(DATA, "BB30014.tfa")
    script = tmp_path / "seeded.py"
    script.write_text(_SEEDED_SCRIPT)
    outs = {}
    for name, lib, env in (("dropin", "dropin/libkalign.so.3", {"KALIGN_AMD_GLUE_WORLD": str(world)}), ("ref", "libkalign_ref.so", {})):
        out = str(tmp_path / (name + ".fa"))
        r = subprocess.run([sys.executable, str(script), _need(lib), inp, out, "3", "5"], env=dict(os.environ, OMP_NUM_THREADS="4", **env),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        outs[name] = (open(out, "rb").read(), r.stdout.decode())
    assert len(outs["ref"][0]) > 100 and outs["dropin"][0] == outs["ref"][0]
    counts = [int(x) for x in outs["dropin"][1].split("COUNTS")[1].split()]
    # (enum order of kalign_amd_glue.c: 1 inline, 6 cons, 7 cons_ref, 14 inline_ref, 16 cons_multi)
    assert counts[1] == 1 and counts[14] == 0 and counts[7] == 0, counts
    assert (counts[16] if world > 1 else counts[6]) == 1, counts


@pytest.mark.parametrize("flags", [[], ["--fast"]], ids=["default", "fast"])
def test_cli_with_several_ranks_on_a_dssim_set(tmp_path, flags):
    from kalign_amd import synth
    inp = str(tmp_path / "in.fa")
    _write_fasta(inp, synth.dssim(200, 150, seed=1))
    c = {}
    got = _cli("dropin/kalign", inp, str(tmp_path / "dropin.fa"), *flags, counters=c, env_extra={"KALIGN_AMD_GLUE_WORLD": "3"})
    want = _cli("kalign_ref", inp, str(tmp_path / "ref.fa"), *flags)
    assert got == want
    assert c["tree_multi"] >= 1 and c["tree"] == 0, c
