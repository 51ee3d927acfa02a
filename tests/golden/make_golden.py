"""Regenerates tests/golden/*.npz from the REAL reference (oracle/_ref).

Runs only in the build container (needs `make -C oracle ref`, i.e.
/root/reference).  The committed .npz files hold data only: inputs (sequences),
what the reference's own pipeline produced up to the dispatcher seam (encoded
sequences, guide-tree task list, seq_distances, scoring parameters) and what its
dispatcher / DP produced (per-task coded paths, top-level meetup, hashes of the
merged profiles and f/b rows, final gap arrays, aligned rows).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kalign_amd import synth            # noqa: E402
from oracle import oracledrv, refdrv    # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def tree_case(name, seqs, **kw):
    job = refdrv.RefJob(seqs, **kw)
    dump_task = 0
    recs, paths, gaps, dump = job.run_tree_traced(dump_task=dump_task)
    rows = job.finalise()
    # cross-check: the real create_msa_tree must give the same gap arrays as the traced replay
    job2 = refdrv.RefJob(seqs, **kw)
    gaps2, _ = job2.run_tree()
    assert all(np.array_equal(a, b) for a, b in zip(gaps, gaps2)), name
    used = recs[-1].path_off + recs[-1].plen + 2
    d = dict(
        seqs=np.array(seqs), kw=np.array(repr(sorted(kw.items()))),
        lens=job.lens, ranks=job.ranks, codes=np.concatenate(job.codes), tree_codes=np.concatenate(job.tree_codes),
        seq_distances=job.seq_distances if job.seq_distances is not None else np.zeros(0, np.float32),
        tasks=job.tasks, subm=job.subm,
        scal=np.array([job.gpo, job.gpe, job.tgpe, job.dist_scale, job.vsm_amax, job.use_seq_weights], np.float32),
        biotype=np.int32(job.biotype),
        paths=paths[:used], gaps=np.concatenate(gaps), rows=np.array(rows),
        dump_task=np.int32(dump_task), dump=dump[:64 * (recs[dump_task].plen + 2)],
    )
    for k, v in oracledrv.recs_to_dict(recs).items():
        d["rec_" + k] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(name, "n=%d" % job.n, "alnlen=%d" % len(rows[0]))


def cons_case(name, seqs, n_anchors=5, weight=2.0, **kw):
    """Default mode: anchor_consistency_build + bonus matrices in every DP (aln_wrap.c:207-214)."""
    job = refdrv.RefJob(seqs, **kw)
    job.build_consistency(n_anchors, weight)
    ids, maps = job.consistency()
    bh = np.zeros(job.ntasks, np.uint64)
    recs, paths, gaps, _ = job.run_tree_traced(bonus_hash=bh)
    rows = job.finalise()
    job2 = refdrv.RefJob(seqs, **kw)
    job2.build_consistency(n_anchors, weight)
    gaps2, _ = job2.run_tree()
    assert all(np.array_equal(a, b) for a, b in zip(gaps, gaps2)), name
    used = recs[-1].path_off + recs[-1].plen + 2
    d = dict(
        seqs=np.array(seqs), kw=np.array(repr(sorted(kw.items()))),
        lens=job.lens, ranks=job.ranks, codes=np.concatenate(job.codes), tree_codes=np.concatenate(job.tree_codes),
        seq_distances=job.seq_distances, tasks=job.tasks, subm=job.subm,
        scal=np.array([job.gpo, job.gpe, job.tgpe, job.dist_scale, job.vsm_amax, job.use_seq_weights], np.float32),
        biotype=np.int32(job.biotype),
        paths=paths[:used], gaps=np.concatenate(gaps), rows=np.array(rows),
        n_anchors=np.int32(n_anchors), weight=np.float32(weight), anchor_ids=ids,
        maps=np.concatenate([m for row in maps for m in row]), bonus_hash=bh,
    )
    for k, v in oracledrv.recs_to_dict(recs).items():
        d["rec_" + k] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(name, "n=%d" % job.n, "alnlen=%d" % len(rows[0]), "anchors", ids)


def cons_cases():
    data = os.path.join(HERE, "data")
    cons_case("cons_BB11001", synth.read_fasta(os.path.join(data, "BB11001.tfa"))[1])
    cons_case("cons_BB30014", synth.read_fasta(os.path.join(data, "BB30014.tfa"))[1])
    cons_case("cons_prot32x200", synth.family(32, 200, seed=1))
    cons_case("cons_dna16x300", synth.family(16, 300, dna=True, seed=1), type_=0)
    cons_case("cons_prot48_k3", synth.family(48, 130, seed=21), n_anchors=3, weight=1.5)
    cons_case("cons_ragged", ["ACDEFGHIKL", "AC", "ACDEFGHIKLMNPQRSTVWYACDEFGHIKLMNPQRSTVWY", "MKV", "ACDKL", "WYACDEFG"])


def pairwise_case(name, seqs, type_):
    job = refdrv.RefJob(seqs, type_=type_)
    n = job.n
    ia, ib = np.meshgrid(np.arange(n), np.arange(min(n, 4)), indexing="ij")
    keep = ia.ravel() != ib.ravel()
    ia, ib = ia.ravel()[keep].astype(np.int32), ib.ravel()[keep].astype(np.int32)
    paths, _ = refdrv.pairwise_batch(job.codes, ia, ib, job.subm, float(job.gpo), float(job.gpe), float(job.tgpe))
    np.savez_compressed(os.path.join(HERE, name + ".npz"),
                        lens=job.lens, codes=np.concatenate(job.codes), ia=ia, ib=ib, subm=job.subm,
                        scal=np.array([job.gpo, job.gpe, job.tgpe], np.float32),
                        paths=np.concatenate(paths), plen=np.array([p[0] for p in paths], np.int32))
    print(name, "pairs=%d" % len(ia))


def bpm_case(name, seed):
    """Distance estimation: bpm_block through calc_distance on sequences over the 13-letter reduced alphabet
    (what kalign_run converts to before tree building, aln_wrap.c:155-160) -- related and unrelated pairs,
    lengths from 1 to beyond the 1024-position cap of the pattern."""
    rng = np.random.RandomState(seed)
    codes = []
    for L in (1, 2, 63, 64, 65, 127, 128, 129, 300, 301, 400, 640, 1023, 1024, 1025, 1500):
        base = rng.randint(0, 13, L).astype(np.uint8)
        codes.append(base)
        mut = base.copy()
        idx = rng.rand(L) < 0.2
        mut[idx] = rng.randint(0, 13, int(idx.sum()))
        cut = rng.randint(0, max(1, L // 10) + 1)
        codes.append(np.concatenate([mut[cut:], rng.randint(0, 13, rng.randint(0, 20)).astype(np.uint8)]) if L > 4 else mut)
    codes = [c for c in codes if len(c) > 0]
    n = len(codes)
    ia, ib = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    ia, ib = ia.ravel().astype(np.int32), ib.ravel().astype(np.int32)
    dist = refdrv.bpm_batch(codes, ia, ib)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), lens=np.array([len(c) for c in codes], np.int32),
                        codes=np.concatenate(codes), ia=ia, ib=ib, dist=dist)
    print(name, "pairs=%d" % len(ia), "max dist", int(dist.max()))


def param_tables():
    out = {}
    for biotype, types in ((0, (3, 4, 5, 6, 8)), (1, (0, 1, 2, 8))):
        for t in types:
            subm, scal = refdrv.param_table(biotype, t)
            out["subm_%d_%d" % (biotype, t)] = subm
            out["scal_%d_%d" % (biotype, t)] = scal
    np.savez_compressed(os.path.join(HERE, "param_tables.npz"), **out)
    print("param_tables", sorted(out))


def main():
    data = os.path.join(HERE, "data")
    for f in ("BB11001", "BB12006", "BB30014"):
        tree_case("tree_" + f, synth.read_fasta(os.path.join(data, f + ".tfa"))[1])
    tree_case("tree_prot32x200", synth.family(32, 200, seed=1))
    tree_case("tree_dna16x300", synth.family(16, 300, dna=True, seed=1), type_=0)
    tree_case("tree_rna16x300", synth.family(16, 300, dna=True, seed=2))
    tree_case("tree_prot24_scaled", synth.family(24, 120, seed=5), dist_scale=0.5, use_seq_weights=1.0)
    tree_case("tree_prot64_gon", synth.family(64, 150, seed=7), type_=4)
    tree_case("tree_ragged", ["ACDEFGHIKL", "A", "ACDEFGHIKLMNPQRSTVWYACDEFGHIKLMNPQRSTVWY", "MKV", "ACDKL", "WYACDEFG"])
    # the 4-sequence DNA case of the reference's own library test (tests/kalign_lib_test.c:33-46 shape)
    tree_case("tree_dna4", ["GAGGTCCATCAAGTTGCGAGCGGGGCGTTTCTG", "GAGGTCATCAAGTTGCAGCGAGGGGCGTTTCTGA",
                            "GAGGTCCATCAAGTTGCGAGCGGGGCGTTCTG", "GAGGTCCATCAGTTGCGAGCGGGGCGTTTCTGAAAA"])
    pairwise_case("pairs_prot12x90", synth.family(12, 90, seed=11), -1)
    pairwise_case("pairs_dna8x200", synth.family(8, 200, dna=True, seed=12), 0)
    param_tables()


def guide_case(name, seqs, n_threads=1, tree_seed=0, tree_noise=0.0):
    """build_tree_kmeans (bisectingKmeans.c:177-271) -- or, with a seed, build_tree_kmeans_noisy (:76-175): the
    sequences in the alphabet the tree builder saw, the task list it made (sorted TASK_ORDER_TREE), msa->seq_distances
    and, for the noisy variant, the multipliers the reference's generator drew."""
    job = refdrv.RefJob(seqs, n_threads=n_threads, tree_seed=tree_seed, tree_noise=tree_noise)
    assert job.tree_codes is not None
    extra = {}
    if tree_seed:
        extra["dm_scale"] = refdrv.noise_multipliers(tree_seed, tree_noise, job.n * min(32, job.n))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), lens=job.lens, tree_codes=np.concatenate(job.tree_codes),
                        tasks=job.tasks, seq_distances=job.seq_distances, biotype=np.int32(job.biotype), **extra)
    print(name, "n=%d" % job.n, "tasks=%d" % job.ntasks)


def refine_case(name, seqs, mode, n_anchors=0, weight=2.0, **kw):
    """refine_alignment (aln_refine.c:36-346) after the first alignment: mode 1 = KALIGN_REFINE_ALL, 2 = _CONFIDENT;
    mode 3 = KALIGN_REFINE_INLINE: the tree aligned from scratch by create_msa_tree_inline_refine.
    Everything but `paths` / `path_off` comes from the real reference; the per-task coded paths (the reference frees them)
    are the oracle's, stored only after its gap arrays, confidences and lengths have been found identical to the
    reference's."""
    job = refdrv.RefJob(seqs, **kw)
    if n_anchors:
        job.build_consistency(n_anchors, weight)
    gaps1, _ = job.run_tree()
    gaps, cb, ca, plen = job.refine(mode)
    rows = job.finalise()
    scal = np.array([job.gpo, job.gpe, job.tgpe, job.dist_scale, job.vsm_amax, job.use_seq_weights], np.float32)
    recs, paths, og = oracledrv.msa_tree_refine(job.codes, job.tasks, job.subm, scal, job.seq_distances, mode=mode, conf_in=cb,
                                               n_anchors=n_anchors, weight=weight)
    assert all(np.array_equal(a, b) for a, b in zip(gaps, og)), name
    assert np.array_equal(np.array([r.confidence for r in recs], np.float32), ca), name
    assert all(recs[t].plen == plen[job.tasks[t][2]] for t in range(len(recs))), name
    used = recs[-1].path_off + recs[-1].plen + 2
    d = dict(
        seqs=np.array(seqs), kw=np.array(repr(sorted(kw.items()))), mode=np.int32(mode),
        lens=job.lens, ranks=job.ranks, codes=np.concatenate(job.codes), tree_codes=np.concatenate(job.tree_codes),
        seq_distances=job.seq_distances if job.seq_distances is not None else np.zeros(0, np.float32),
        tasks=job.tasks, subm=job.subm, scal=scal, biotype=np.int32(job.biotype),
        n_anchors=np.int32(n_anchors), weight=np.float32(weight),
        gaps_first=np.concatenate(gaps1), conf_before=cb, conf_after=ca, plen_after=plen,
        gaps=np.concatenate(gaps), rows=np.array(rows),
        paths=paths[:used], path_off=np.array([r.path_off for r in recs], np.int32),
        n_differ=np.int32(sum(not np.array_equal(a, b) for a, b in zip(gaps1, gaps))),
    )
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(name, "n=%d" % job.n, "alnlen=%d" % len(rows[0]), "sequences whose gaps changed:", int(d["n_differ"]))


def refine_cases():
    data = os.path.join(HERE, "data")
    refine_case("refine_prot32x200_all", synth.dssim(32, 200, seed=1), 1)
    refine_case("refine_prot32x200_conf", synth.dssim(32, 200, seed=1), 2)
    refine_case("refine_dna16x300_all", synth.dssim(16, 300, dna=True, seed=1), 1, type_=0)
    refine_case("refine_cons_prot48_all", synth.dssim(48, 150, seed=5), 1, n_anchors=5)
    refine_case("refine_cons_prot24_conf", synth.dssim(24, 120, seed=7), 2, n_anchors=3)
    refine_case("refine_BB11001_all", synth.read_fasta(os.path.join(data, "BB11001.tfa"))[1], 1, n_anchors=5)
    refine_case("refine_BB30014_conf", synth.read_fasta(os.path.join(data, "BB30014.tfa"))[1], 2)
    refine_case("refine_prot24_scaled_all", synth.dssim(24, 150, seed=11), 1, dist_scale=0.5, use_seq_weights=1.0)
    refine_case("refine_ragged_all", [s[:40 + 13 * i] for i, s in enumerate(synth.dssim(20, 400, seed=13))], 1)
    inline_cases()
    adaptive_cases()


def adaptive_cases():
    """mode + 256 = --adaptive-budget (aln_refine.c:255-282): the number of trials of an edge (1..8) follows from the share
    of very uncertain meetups of its baseline trial"""
    data = os.path.join(HERE, "data")
    refine_case("refine_prot32x200_all_adaptive", synth.dssim(32, 200, seed=1), 1 + 256)
    refine_case("refine_cons_prot24_conf_adaptive", synth.dssim(24, 120, seed=7), 2 + 256, n_anchors=3)
    refine_case("refine_BB30014_all_adaptive", synth.read_fasta(os.path.join(data, "BB30014.tfa"))[1], 1 + 256)
    refine_case("refine_dna16x300_all_adaptive", synth.dssim(16, 300, dna=True, seed=1), 1 + 256, type_=0)
    refine_case("refine_ragged_all_adaptive", [s[:40 + 13 * i] for i, s in enumerate(synth.dssim(20, 400, seed=13))], 1 + 256)


def inline_cases():
    """mode 3 = KALIGN_REFINE_INLINE: create_msa_tree_inline_refine (aln_run.c:448-475) with three trials per edge"""
    data = os.path.join(HERE, "data")
    refine_case("refine_prot32x200_inline", synth.dssim(32, 200, seed=1), 3)
    refine_case("refine_dna16x300_inline", synth.dssim(16, 300, dna=True, seed=1), 3, type_=0)
    refine_case("refine_cons_prot24_inline", synth.dssim(24, 120, seed=7), 3, n_anchors=3)
    refine_case("refine_BB30014_inline", synth.read_fasta(os.path.join(data, "BB30014.tfa"))[1], 3)
    refine_case("refine_prot24_scaled_inline", synth.dssim(24, 150, seed=11), 3, dist_scale=0.5, use_seq_weights=1.0)


def realign_case(name, seqs, n_anchors=0, weight=2.0, **kw):
    """One iteration of kalign_run_realign (aln_wrap.c:361-527): first alignment on the BPM/k-means tree, identity
    distances from that alignment, UPGMA tree from them, second alignment on that tree."""
    job = refdrv.RefJob(seqs, **kw)
    if n_anchors:
        job.build_consistency(n_anchors, weight)
    tasks1, sd1 = job.tasks.copy(), job.seq_distances.copy()
    job.run_tree()
    rows_sorted, dm, _, _ = job.realign_tree()
    tasks2, sd2 = job.tasks.copy(), job.seq_distances.copy()
    job.run_tree()
    final_rows = job.finalise()
    np.savez_compressed(os.path.join(HERE, name + ".npz"),
                        seqs=np.array(seqs), lens=job.lens, ranks=job.ranks, codes=np.concatenate(job.codes),
                        subm=job.subm, biotype=np.int32(job.biotype), n_anchors=np.int32(n_anchors), weight=np.float32(weight),
                        scal=np.array([job.gpo, job.gpe, job.tgpe, job.dist_scale, job.vsm_amax, job.use_seq_weights], np.float32),
                        tasks1=tasks1, seq_distances1=sd1, rows_sorted=np.array(rows_sorted), dm=dm,
                        tasks2=tasks2, seq_distances2=sd2, final_rows=np.array(final_rows))
    print(name, "n=%d" % job.n, "alnlen %d -> %d" % (len(rows_sorted[0]), len(final_rows[0])))


def realign_cases():
    realign_case("realign_prot40", synth.family(40, 80, seed=3))
    fam = synth.family(30, 60, seed=4)
    realign_case("realign_dups", fam + fam[:7] + fam[3:5])                        # identical rows: distance ties in UPGMA
    realign_case("realign_dna24_cons", synth.family(24, 150, dna=True, seed=5), n_anchors=5)
    realign_case("realign_prot150", synth.family(150, 70, seed=6))


def guide_cases():
    rng = np.random.RandomState(5)
    guide_case("guide_prot300", synth.family(300, 150, seed=21))                  # k-means levels + UPGMA leaves
    guide_case("guide_dna200", synth.family(200, 120, dna=True, seed=22))
    guide_case("guide_prot64", synth.family(64, 90, seed=23), n_threads=4)        # one split; thread count must not matter
    guide_case("guide_prot49", synth.family(49, 80, seed=24))                     # UPGMA only
    guide_case("guide_prot20", synth.family(20, 60, seed=25))                     # fewer sequences than anchors
    same = ["".join("ACDEFGHIKLMNPQRSTVWY"[k] for k in rng.randint(0, 20, size=70)) for _ in range(130)]
    guide_case("guide_samelen130", same)                                          # all lengths tie in the anchor sort
    fam = synth.family(90, 200, seed=26)
    guide_case("guide_ragged", [s[:10 + (7 * i) % 190] for i, s in enumerate(fam)])
    guide_case("guide_prot1100", synth.family(1100, 60, seed=27), n_threads=8)    # deeper k-means recursion
    guide_case("guide_two", synth.family(2, 50, seed=28))
    # the trees of ensemble members (kalign_ensemble -> kalign_run_seeded with tree_seed / tree_noise)
    guide_case("guide_noisy_prot300", synth.family(300, 150, seed=21), tree_seed=42, tree_noise=0.2)
    guide_case("guide_noisy_dna200", synth.family(200, 120, dna=True, seed=22), tree_seed=7, tree_noise=0.5)
    guide_case("guide_noisy_prot40", synth.family(40, 70, seed=29), tree_seed=99, tree_noise=1.5)   # clamps at 0.1


if __name__ == "__main__":
    if not refdrv.available():
        sys.exit("oracle/_ref/libkalign_ref.so missing: run `make -C oracle ref` (needs /root/reference)")
    if len(sys.argv) > 1 and sys.argv[1] == "cons":        # only the consistency cases
        cons_cases()
    elif len(sys.argv) > 1 and sys.argv[1] == "guide":
        guide_cases()
    elif len(sys.argv) > 1 and sys.argv[1] == "realign":
        realign_cases()
    elif len(sys.argv) > 1 and sys.argv[1] == "refine":
        refine_cases()
    elif len(sys.argv) > 1 and sys.argv[1] == "inline":
        inline_cases()
    elif len(sys.argv) > 1 and sys.argv[1] == "adaptive":
        adaptive_cases()
    elif len(sys.argv) > 1 and sys.argv[1] == "bpm":
        bpm_case("bpm_mixed", 31)
        guide_cases()
        realign_cases()
        refine_cases()
    else:
        main()
        cons_cases()
        bpm_case("bpm_mixed", 31)
        guide_cases()
        realign_cases()
        refine_cases()
