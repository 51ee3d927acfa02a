"""ctypes driver for oracle/libkalign_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this; the product (kalign_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "libkalign_oracle.so")


class TaskRec(C.Structure):
    """ko_task_rec == refh_task_rec == ka_task_rec"""
    _fields_ = [
        ("a", C.c_int), ("b", C.c_int), ("c", C.c_int),
        ("len_a", C.c_int), ("len_b", C.c_int),
        ("nsip_a", C.c_int), ("nsip_b", C.c_int),
        ("plen", C.c_int), ("kind", C.c_int), ("swapped", C.c_int),
        ("meet", C.c_int), ("transition", C.c_int), ("path_off", C.c_int),
        ("gap_scale", C.c_float), ("subm_off", C.c_float),
        ("score", C.c_float), ("confidence", C.c_float),
        ("prof_hash", C.c_uint64), ("fhash", C.c_uint64), ("bhash", C.c_uint64),
    ]


REC_FIELDS = [f[0] for f in TaskRec._fields_]


def recs_to_dict(recs):
    """array-of-struct -> dict of numpy arrays (what the golden files store)."""
    out = {}
    for name, ctype in TaskRec._fields_:
        dt = {C.c_int: np.int32, C.c_float: np.float32, C.c_uint64: np.uint64}[ctype]
        out[name] = np.array([getattr(r, name) for r in recs], dtype=dt)
    return out


def build():
    if (not os.path.exists(ORACLE_SO)
            or os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(_HERE, "kalign_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(ORACLE_SO)
        vp = C.c_void_p
        L.ko_msa_tree.argtypes = [C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, vp,
                                  C.POINTER(TaskRec), vp, C.c_longlong, vp, C.c_int, vp]
        L.ko_msa_tree_cons.argtypes = [C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_float,
                                       C.POINTER(TaskRec), vp, C.c_longlong, vp, C.c_int, vp, vp, vp, vp]
        L.ko_msa_tree_refine.argtypes = [C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_float, C.c_int, vp,
                                         C.POINTER(TaskRec), vp, C.c_longlong, vp]
        L.ko_convert_raw_path.argtypes = [vp, C.c_int, C.c_int, vp]
        L.ko_pairwise_batch.argtypes = [vp, vp, vp, vp, vp, C.c_int, vp, C.c_float, C.c_float, C.c_float,
                                        vp, vp, vp]
        L.ko_dp_single.argtypes = [C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, vp,
                                   C.c_float, C.c_float, C.c_float, C.c_float, C.c_int,
                                   vp, C.c_int, vp, vp, vp,
                                   C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.ko_make_profile.argtypes = [vp, C.c_int, vp, C.c_float, C.c_float, C.c_float, C.c_float, vp]
        L.ko_set_gap_penalties.argtypes = [vp, C.c_int, C.c_int]
        L.ko_code_path.argtypes = [vp, C.c_int, C.c_int, vp]
        L.ko_mirror_path.argtypes = [vp, C.c_int, C.c_int, vp]
        L.ko_update_profile.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, vp,
                                        C.c_float, C.c_float, C.c_float, C.c_float]
        L.ko_bpm_batch.argtypes = [vp, vp, vp, vp, vp, C.c_int, vp]
        L.ko_aln_pairwise_dist.argtypes = [vp, C.c_int, C.c_longlong, C.c_int, C.c_ubyte, vp]
        L.ko_tree_from_pairwise.argtypes = [vp, C.c_int, vp, vp]
        L.ko_fnv1a.argtypes = [vp, C.c_uint64]
        L.ko_fnv1a.restype = C.c_uint64
        L.ko_set_prefix_reuse.argtypes = [C.c_int]
        L.ko_set_prefix_reuse.restype = None
        L.ko_prefix_reuse_cells.argtypes = [vp, vp]
        L.ko_prefix_reuse_cells.restype = None
        L.ko_set_carried_votes.argtypes = [C.c_int]
        L.ko_set_carried_votes.restype = None
        L.ko_carried_votes_cells.argtypes = [vp, vp]
        L.ko_carried_votes_cells.restype = None
        _lib = L
    return _lib


def set_prefix_reuse(on):
    """Hirschberg prefix reuse in the oracle's recursion (the device's rule; results must not change)."""
    lib().ko_set_prefix_reuse(1 if on else 0)


def set_carried_votes(on):
    """anchor votes carried up the tree (the device's KA_CARRY=1 rule; results must not change)"""
    lib().ko_set_carried_votes(1 if on else 0)


def carried_votes_cells():
    """(cells of the vote tables merged, cells that needed a count over an operand's members) since set_carried_votes"""
    cells, counted = C.c_longlong(0), C.c_longlong(0)
    lib().ko_carried_votes_cells(C.byref(cells), C.byref(counted))
    return cells.value, counted.value


def prefix_reuse_cells():
    """(DP cells of the passes that ran, of the passes taken over) since set_prefix_reuse"""
    run, reused = C.c_longlong(0), C.c_longlong(0)
    lib().ko_prefix_reuse_cells(C.byref(run), C.byref(reused))
    return run.value, reused.value


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def flatten(codes):
    lens = np.array([len(c) for c in codes], np.int32)
    off = np.zeros(len(codes), np.int32)
    off[1:] = np.cumsum(lens)[:-1]
    flat = np.ascontiguousarray(np.concatenate(codes), np.uint8)
    return flat, off, lens


def msa_tree(codes, tasks, subm, scal, seq_distances=None, dump_task=-1):
    """Oracle dispatcher.  Returns (recs, paths, gaps per seq, dumped profile or None)."""
    flat, off, lens = flatten(codes)
    tasks = np.ascontiguousarray(tasks, np.int32)
    nt = len(tasks)
    recs = (TaskRec * nt)()
    cap = (int(lens.max()) * 2 + 4) * nt + int(lens.sum()) * 2 * 40
    paths = np.zeros(cap, np.int32)
    gaps = np.zeros(int(lens.sum()) + len(codes), np.int32)
    dump = np.zeros(64 * (2 * int(lens.sum()) + 4), np.float32) if dump_task >= 0 else None
    sd = None if seq_distances is None else np.ascontiguousarray(seq_distances, np.float32)
    rc = lib().ko_msa_tree(len(codes), _ptr(flat), _ptr(off), _ptr(lens), _ptr(sd), nt, _ptr(tasks),
                           _ptr(np.ascontiguousarray(subm, np.float32).reshape(-1)),
                           _ptr(np.ascontiguousarray(scal, np.float32)),
                           recs, _ptr(paths), cap, _ptr(gaps), dump_task, _ptr(dump))
    if rc:
        raise RuntimeError("ko_msa_tree rc=%d" % rc)
    g, o = [], 0
    for n in lens:
        g.append(gaps[o:o + int(n) + 1].copy())
        o += int(n) + 1
    return recs, paths, g, dump


def msa_tree_cons(codes, tasks, subm, scal, seq_distances, n_anchors=5, weight=2.0):
    """Oracle dispatcher in default mode (anchor consistency).  Returns
    (recs, paths, gaps per seq, anchor_ids, maps[i][k], bonus_hash[n_tasks])."""
    flat, off, lens = flatten(codes)
    tasks = np.ascontiguousarray(tasks, np.int32)
    nt = len(tasks)
    recs = (TaskRec * nt)()
    cap = (int(lens.max()) * 2 + 4) * nt + int(lens.sum()) * 2 * 40
    paths = np.zeros(cap, np.int32)
    gaps = np.zeros(int(lens.sum()) + len(codes), np.int32)
    K = min(n_anchors, len(codes))
    ids = np.full(max(K, 1), -1, np.int32)
    mflat = np.zeros(max(1, int(lens.sum()) * max(K, 1)), np.int32)
    bh = np.zeros(nt, np.uint64)
    sd = None if seq_distances is None else np.ascontiguousarray(seq_distances, np.float32)
    rc = lib().ko_msa_tree_cons(len(codes), _ptr(flat), _ptr(off), _ptr(lens), _ptr(sd), nt, _ptr(tasks),
                                _ptr(np.ascontiguousarray(subm, np.float32).reshape(-1)),
                                _ptr(np.ascontiguousarray(scal, np.float32)), n_anchors, weight,
                                recs, _ptr(paths), cap, _ptr(gaps), -1, None, _ptr(ids), _ptr(mflat), _ptr(bh))
    if rc:
        raise RuntimeError("ko_msa_tree_cons rc=%d" % rc)
    g, o = [], 0
    for n in lens:
        g.append(gaps[o:o + int(n) + 1].copy())
        o += int(n) + 1
    maps, o = [], 0
    for i in range(len(codes)):
        row = []
        for k in range(K):
            row.append(mflat[o:o + int(lens[i])].copy())
            o += int(lens[i])
        maps.append(row)
    return recs, paths, g, ids[:K], maps, bh


def msa_tree_refine(codes, tasks, subm, scal, seq_distances, mode=1, conf_in=None, n_anchors=0, weight=2.0):
    """refine_alignment as the oracle restates it (ko_msa_tree_refine): the second pass over every edge.
    Returns (recs, paths, gaps per seq)."""
    flat, off, lens = flatten(codes)
    tasks = np.ascontiguousarray(tasks, np.int32)
    nt = len(tasks)
    recs = (TaskRec * nt)()
    cap = (int(lens.max()) * 2 + 4) * nt + int(lens.sum()) * 2 * 40
    paths = np.zeros(cap, np.int32)
    gaps = np.zeros(int(lens.sum()) + len(codes), np.int32)
    sd = None if seq_distances is None else np.ascontiguousarray(seq_distances, np.float32)
    cf = None if conf_in is None else np.ascontiguousarray(conf_in, np.float32)
    rc = lib().ko_msa_tree_refine(len(codes), _ptr(flat), _ptr(off), _ptr(lens), _ptr(sd), nt, _ptr(tasks),
                                  _ptr(np.ascontiguousarray(subm, np.float32).reshape(-1)),
                                  _ptr(np.ascontiguousarray(scal, np.float32)), int(n_anchors), float(weight), int(mode), _ptr(cf),
                                  recs, _ptr(paths), cap, _ptr(gaps))
    if rc:
        raise RuntimeError("ko_msa_tree_refine rc=%d" % rc)
    g, o = [], 0
    for n in lens:
        g.append(gaps[o:o + int(n) + 1].copy())
        o += int(n) + 1
    return recs, paths, g


def pairwise_batch(codes, ia, ib, subm, gpo, gpe, tgpe):
    flat, off, lens = flatten(codes)
    ia = np.ascontiguousarray(ia, np.int32)
    ib = np.ascontiguousarray(ib, np.int32)
    sizes = lens[ia].astype(np.int64) + lens[ib] + 3
    poff = np.zeros(len(ia), np.int64)
    poff[1:] = np.cumsum(sizes)[:-1]
    paths = np.zeros(int(sizes.sum()), np.int32)
    scores = np.zeros(len(ia), np.float32)
    lib().ko_pairwise_batch(_ptr(flat), _ptr(off), _ptr(lens), _ptr(ia), _ptr(ib), len(ia),
                            _ptr(np.ascontiguousarray(subm, np.float32).reshape(-1)), gpo, gpe, tgpe,
                            _ptr(paths), _ptr(poff), _ptr(scores))
    return [paths[poff[k]:poff[k] + paths[poff[k]] + 2].copy() for k in range(len(ia))], scores


def dp_single(kind, len_a, len_b, subm, gpo, gpe, tgpe, soff=0.0, sip=1,
              seq1=None, seq2=None, prof1=None, prof2=None, bonus=None):
    raw = np.zeros(len_a + len_b + 2, np.int32)
    f = np.zeros(3 * (max(len_a, len_b) + 2), np.float32)
    b = np.zeros(3 * (max(len_a, len_b) + 2), np.float32)
    meet, tr = C.c_int(0), C.c_int(0)
    score, conf = C.c_float(0), C.c_float(0)
    lib().ko_dp_single(kind, _ptr(seq1), _ptr(seq2), _ptr(prof1), _ptr(prof2), len_a, len_b,
                       _ptr(np.ascontiguousarray(subm, np.float32).reshape(-1)),
                       gpo, gpe, tgpe, soff, sip, _ptr(bonus), len_b if bonus is not None else 0,
                       _ptr(raw), _ptr(f), _ptr(b),
                       C.byref(meet), C.byref(tr), C.byref(score), C.byref(conf))
    return dict(raw=raw[:len_a + 2], f=f[:3 * (len_b + 1)].reshape(-1, 3), b=b[:3 * (len_b + 1)].reshape(-1, 3),
                meet=meet.value, transition=tr.value, score=score.value, confidence=conf.value)


def rows_from_gaps(seqs_sorted, gaps):
    """finalise_alignment (msa_op.c:546-598): gaps[] -> '-' padded strings."""
    rows = []
    for s, g in zip(seqs_sorted, gaps):
        out = []
        for j, ch in enumerate(s):
            out.append("-" * int(g[j]))
            out.append(ch)
        out.append("-" * int(g[len(s)]))
        rows.append("".join(out))
    return rows


def bpm_batch(codes, ia, ib):
    """calc_distance / bpm_block restatement for a list of pairs (codes < 13)."""
    flat, off, lens = flatten(codes)
    ia = np.ascontiguousarray(ia, np.int32)
    ib = np.ascontiguousarray(ib, np.int32)
    out = np.zeros(len(ia), np.int32)
    lib().ko_bpm_batch(_ptr(flat), _ptr(off), _ptr(lens), _ptr(ia), _ptr(ib), len(ia), _ptr(out))
    return out


def aln_guide_tree(rows, gap=b"-"):
    """compute_aln_pairwise_dist + build_tree_from_pairwise restated: (tasks, seq_distances, dm) from equal-length rows."""
    n, alnlen = len(rows), len(rows[0])
    flat = np.frombuffer(b"".join(r if isinstance(r, bytes) else r.encode() for r in rows), np.uint8).copy()
    dm = np.zeros((n, n), np.float32)
    lib().ko_aln_pairwise_dist(_ptr(flat), n, alnlen, alnlen, gap[0], _ptr(dm))
    work = dm.copy()
    tasks = np.zeros((n - 1, 3), np.int32)
    sd = np.zeros(n, np.float32)
    if lib().ko_tree_from_pairwise(_ptr(work), n, _ptr(tasks), _ptr(sd)):
        raise RuntimeError("ko_tree_from_pairwise failed")
    return tasks, sd, dm
