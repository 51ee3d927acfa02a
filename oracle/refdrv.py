"""ctypes driver for oracle/_ref/libkalign_ref.so -- TEST INFRASTRUCTURE ONLY.

The .so is the real Kalign library (compiled from /root/reference by
`make -C oracle ref`) plus oracle/ref_harness.c.  Only tests/, the golden
generator, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product (kalign_amd/) never does.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(_HERE, "_ref", "libkalign_ref.so")


class TaskRec(C.Structure):
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


def available():
    return os.path.exists(REF_SO)


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(REF_SO)
        L.refh_prepare.restype = C.c_void_p
        L.refh_prepare_noisy.restype = C.c_void_p
        L.refh_prepare_noisy.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int, C.c_int,
                                         C.c_float, C.c_float, C.c_float, C.c_int,
                                         C.c_float, C.c_float, C.c_float, C.c_uint64, C.c_float]
        L.refh_noise_multipliers.argtypes = [C.c_uint64, C.c_float, C.c_int, C.c_void_p]
        L.refh_prepare.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int, C.c_int,
                                   C.c_float, C.c_float, C.c_float, C.c_int,
                                   C.c_float, C.c_float, C.c_float]
        for name in ("refh_numseq", "refh_biotype", "refh_ntasks", "refh_alnlen_from_gaps"):
            getattr(L, name).restype = C.c_int
            getattr(L, name).argtypes = [C.c_void_p]
        L.refh_free.argtypes = [C.c_void_p]
        L.refh_free.restype = None
        L.refh_get_seq_info.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.refh_get_seq_codes.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.refh_get_seq_distances.argtypes = [C.c_void_p, C.c_void_p]
        L.refh_get_tree_codes.argtypes = [C.c_void_p, C.c_void_p]
        L.refh_tree_seconds.argtypes = [C.c_void_p]
        L.refh_realign_tree.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_void_p,
                                        C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.refh_tree_seconds.restype = C.c_double
        L.refh_get_seq_distances.restype = C.c_int
        L.refh_get_params.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.refh_param_table.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.refh_get_tasks.argtypes = [C.c_void_p, C.c_void_p]
        L.refh_run_tree.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        L.refh_finalise.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_int)]
        L.refh_run_tree_traced.argtypes = [C.c_void_p, C.POINTER(TaskRec), C.c_void_p, C.c_void_p,
                                           C.c_int, C.c_void_p]
        L.refh_pairwise_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_float,
                                          C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        L.refh_set_tasks.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.refh_prepare_encoded.restype = C.c_void_p
        L.refh_prepare_encoded.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                           C.c_float, C.c_float, C.c_float, C.c_int,
                                           C.c_float, C.c_float, C.c_float]
        L.refh_build_consistency.argtypes = [C.c_void_p, C.c_int, C.c_float]
        L.refh_refine.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.refh_get_consistency.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.refh_set_bonus_hash_out.argtypes = [C.c_void_p]
        L.refh_set_bonus_hash_out.restype = None
        L.refh_bpm_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.refh_kalign.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int,
                                  C.c_float, C.c_float, C.c_float, C.POINTER(C.c_char_p), C.POINTER(C.c_int)]
        _lib = L
    return _lib


def noise_multipliers(seed, sigma, n):
    """what build_tree_kmeans_noisy multiplies the anchor distances with (the reference's own generator)"""
    out = np.zeros(n, np.float32)
    if lib().refh_noise_multipliers(seed, sigma, n, _ptr(out)):
        raise RuntimeError("rng")
    return out


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class RefJob:
    """One alignment job prepared by the real reference up to the dispatcher seam."""

    def __init__(self, seqs, type_=-1, gpo=-1.0, gpe=-1.0, tgpe=-1.0, n_threads=1,
                 dist_scale=0.0, vsm_amax=-1.0, use_seq_weights=-1.0, tree_seed=0, tree_noise=0.0):
        L = lib()
        bs = [s.encode() if isinstance(s, str) else s for s in seqs]
        n = len(bs)
        arr = (C.c_char_p * n)(*bs)
        lens = (C.c_int * n)(*[len(b) for b in bs])
        # the reference treats type 8 (UNDEFINED) / anything unknown as "auto"
        if tree_seed and tree_noise > 0.0:          # the guide tree of an ensemble member (build_tree_kmeans_noisy)
            self.h = L.refh_prepare_noisy(arr, lens, n, 8 if type_ < 0 else type_, gpo, gpe, tgpe, n_threads,
                                          dist_scale, vsm_amax, use_seq_weights, tree_seed, tree_noise)
        else:
            self.h = L.refh_prepare(arr, lens, n, 8 if type_ < 0 else type_, gpo, gpe, tgpe, n_threads,
                                    dist_scale, vsm_amax, use_seq_weights)
        if not self.h:
            raise RuntimeError("reference refused the input")
        self.n = L.refh_numseq(self.h)
        self.biotype = L.refh_biotype(self.h)
        self.lens = np.zeros(self.n, np.int32)
        self.ranks = np.zeros(self.n, np.int32)
        L.refh_get_seq_info(self.h, _ptr(self.lens), _ptr(self.ranks))
        self.codes = []
        for i in range(self.n):
            buf = np.zeros(int(self.lens[i]), np.uint8)
            L.refh_get_seq_codes(self.h, i, _ptr(buf))
            self.codes.append(buf)
        flat = np.zeros(max(int(self.lens.sum()), 1), np.uint8)
        self.tree_codes = None                       # the alphabet build_tree_kmeans saw (reduced protein / nucleotides)
        if L.refh_get_tree_codes(self.h, _ptr(flat)) == 0:
            o = np.concatenate([[0], np.cumsum(self.lens)])
            self.tree_codes = [flat[o[i]:o[i + 1]].copy() for i in range(self.n)]
        self.tree_seconds = float(L.refh_tree_seconds(self.h))      # wall time of build_tree_kmeans
        self.seq_distances = np.zeros(self.n, np.float32)
        if not L.refh_get_seq_distances(self.h, _ptr(self.seq_distances)):
            self.seq_distances = None
        self.subm = np.zeros(23 * 23, np.float32)
        scal = np.zeros(6, np.float32)
        L.refh_get_params(self.h, _ptr(self.subm), _ptr(scal))
        self.gpo, self.gpe, self.tgpe, self.dist_scale, self.vsm_amax, self.use_seq_weights = \
            [np.float32(x) for x in scal]
        self.ntasks = L.refh_ntasks(self.h)
        self.tasks = np.zeros((self.ntasks, 3), np.int32)
        L.refh_get_tasks(self.h, _ptr(self.tasks))

    def close(self):
        if self.h:
            lib().refh_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _gaps_buf(self):
        return np.zeros(int(self.lens.sum()) + self.n, np.int32)

    def split_gaps(self, flat):
        out, o = [], 0
        for i in range(self.n):
            out.append(flat[o:o + int(self.lens[i]) + 1].copy())
            o += int(self.lens[i]) + 1
        return out

    def run_tree(self):
        """The reference's own create_msa_tree.  Returns (gaps per sorted seq, seconds)."""
        g = self._gaps_buf()
        secs = C.c_double(0.0)
        if lib().refh_run_tree(self.h, _ptr(g), C.byref(secs)):
            raise RuntimeError("create_msa_tree failed")
        return self.split_gaps(g), secs.value

    def build_consistency(self, n_anchors=5, weight=2.0):
        """anchor_consistency_build (aln_wrap.c:207-214): from now on the tree runs in default mode."""
        if lib().refh_build_consistency(self.h, n_anchors, weight):
            raise RuntimeError("anchor_consistency_build failed")

    def consistency(self):
        """(anchor_ids[K], maps[i][k] = int array of len_i) or None when no table is attached."""
        K = lib().refh_get_consistency(self.h, None, None)
        if K == 0:
            return None
        ids = np.zeros(K, np.int32)
        flat = np.zeros(int(self.lens.sum()) * K, np.int32)
        lib().refh_get_consistency(self.h, _ptr(ids), _ptr(flat))
        maps, o = [], 0
        for i in range(self.n):
            row = []
            for k in range(K):
                row.append(flat[o:o + int(self.lens[i])].copy())
                o += int(self.lens[i])
            maps.append(row)
        return ids, maps

    def refine(self, mode=1):
        """refine_alignment (aln_refine.c:36-88) after run_tree(): mode 1 = KALIGN_REFINE_ALL, 2 = _CONFIDENT;
        3 = KALIGN_REFINE_INLINE (create_msa_tree_inline_refine, aln_run.c:448-475, from scratch).
        Returns (gaps per sorted sequence, task.confidence before, after, plen of every node)."""
        g = self._gaps_buf()
        cb = np.zeros(self.ntasks, np.float32)
        ca = np.zeros(self.ntasks, np.float32)
        plen = np.zeros(2 * self.n - 1, np.int32)
        if lib().refh_refine(self.h, int(mode), _ptr(g), _ptr(cb), _ptr(ca), _ptr(plen)):
            raise RuntimeError("refine_alignment failed")
        return self.split_gaps(g), cb, ca, plen

    def run_tree_traced(self, dump_task=-1, bonus_hash=None):
        """bonus_hash: optional uint64[ntasks] array receiving the FNV hash of each task's bonus matrix."""
        lib().refh_set_bonus_hash_out(_ptr(bonus_hash) if bonus_hash is not None else None)
        recs = (TaskRec * self.ntasks)()
        total = 0
        # upper bound on the path storage
        total = int(self.lens.sum()) * 2 * max(1, int(np.ceil(np.log2(self.n + 1)))) + 8 * self.n + 64
        total = max(total, (int(self.lens.max()) * 2 + 4) * self.ntasks)
        paths = np.zeros(total, np.int32)
        g = self._gaps_buf()
        dump = np.zeros(64 * (2 * int(self.lens.sum()) + 4), np.float32) if dump_task >= 0 else None
        rc = lib().refh_run_tree_traced(self.h, recs, _ptr(paths), _ptr(g), dump_task,
                                        _ptr(dump) if dump is not None else None)
        if rc:
            raise RuntimeError("traced replay failed")
        return recs, paths, self.split_gaps(g), dump

    def realign_tree(self, want_dm=True):
        """One iteration of kalign_run_realign's loop up to the new tree (aln_wrap.c:449-495), after run_tree():
        returns (finalised rows in sorted order, identity distances N x N, seconds in compute_aln_pairwise_dist,
        seconds in build_tree_from_pairwise) and refreshes self.tasks / self.seq_distances; run_tree() then
        aligns on the new tree."""
        L = lib()
        cap = int(self.lens.sum()) + 1
        bufs = [C.create_string_buffer(cap) for _ in range(self.n)]
        rows = (C.c_char_p * self.n)(*[C.cast(b, C.c_char_p) for b in bufs])
        alnlen = C.c_int(0)
        dm = np.zeros((self.n, self.n), np.float32) if want_dm else None
        sd, st = C.c_double(0), C.c_double(0)
        if L.refh_realign_tree(self.h, rows, C.byref(alnlen), _ptr(dm), C.byref(sd), C.byref(st)):
            raise RuntimeError("realign step failed")
        self.ntasks = L.refh_ntasks(self.h)
        self.tasks = np.zeros((self.ntasks, 3), np.int32)
        L.refh_get_tasks(self.h, _ptr(self.tasks))
        L.refh_get_seq_distances(self.h, _ptr(self.seq_distances))
        return [b.value.decode() for b in bufs], dm, sd.value, st.value

    def finalise(self):
        alnlen = lib().refh_alnlen_from_gaps(self.h)
        bufs = [C.create_string_buffer(alnlen + 1) for _ in range(self.n)]
        rows = (C.c_char_p * self.n)(*[C.cast(b, C.c_char_p) for b in bufs])
        n = C.c_int(0)
        if lib().refh_finalise(self.h, rows, C.byref(n)):
            raise RuntimeError("finalise failed")
        return [b.value.decode() for b in bufs]


class EncodedJob(RefJob):
    """The reference dispatcher on caller-supplied encoded sequences + task list (no tree
    building): what bench.py's cpu_baseline leg times."""

    def __init__(self, codes, tasks, seq_distances, biotype, type_=-1, gpo=-1.0, gpe=-1.0, tgpe=-1.0,
                 n_threads=1, dist_scale=0.0, vsm_amax=-1.0, use_seq_weights=-1.0):
        L = lib()
        self.lens = np.array([len(c) for c in codes], np.int32)
        off = np.zeros(len(codes), np.int32)
        off[1:] = np.cumsum(self.lens)[:-1]
        flat = np.ascontiguousarray(np.concatenate(codes), np.uint8)
        self.n = len(codes)
        self.h = L.refh_prepare_encoded(_ptr(flat), _ptr(off), _ptr(self.lens), self.n, biotype,
                                        8 if type_ < 0 else type_, gpo, gpe, tgpe, n_threads,
                                        dist_scale, vsm_amax, use_seq_weights)
        if not self.h:
            raise RuntimeError("reference refused the input")
        tasks = np.ascontiguousarray(tasks, np.int32)
        sd = None if seq_distances is None else np.ascontiguousarray(seq_distances, np.float32)
        if L.refh_set_tasks(self.h, _ptr(tasks), len(tasks), _ptr(sd) if sd is not None else None):
            raise RuntimeError("bad task list")
        self.codes = codes
        self.tasks = tasks
        self.ntasks = len(tasks)
        self.seq_distances = sd
        self.subm = np.zeros(23 * 23, np.float32)
        scal = np.zeros(6, np.float32)
        L.refh_get_params(self.h, _ptr(self.subm), _ptr(scal))
        self.scal = scal


def kalign(seqs, type_=-1, gpo=-1.0, gpe=-1.0, tgpe=-1.0, n_threads=1):
    """The reference's one-call library API (lib/include/kalign/kalign.h:46)."""
    L = lib()
    bs = [s.encode() for s in seqs]
    n = len(bs)
    arr = (C.c_char_p * n)(*bs)
    lens = (C.c_int * n)(*[len(b) for b in bs])
    cap = sum(len(b) for b in bs) + 16
    bufs = [C.create_string_buffer(cap) for _ in range(n)]
    rows = (C.c_char_p * n)(*[C.cast(b, C.c_char_p) for b in bufs])
    alnlen = C.c_int(0)
    if L.refh_kalign(arr, lens, n, n_threads, 8 if type_ < 0 else type_, gpo, gpe, tgpe, rows, C.byref(alnlen)):
        raise RuntimeError("kalign() failed")
    return [b.value.decode() for b in bufs]


def param_table(biotype, type_):
    subm = np.zeros(23 * 23, np.float32)
    scal = np.zeros(6, np.float32)
    if lib().refh_param_table(biotype, type_, _ptr(subm), _ptr(scal)):
        raise RuntimeError("aln_param_init failed")
    return subm.reshape(23, 23), scal


def pairwise_batch(codes, ia, ib, subm, gpo, gpe, tgpe, n_threads=1, want_paths=True):
    """N independent seq-seq alignments through the reference's aln_runner."""
    lens = np.array([len(c) for c in codes], np.int32)
    off = np.zeros(len(codes), np.int32)
    off[1:] = np.cumsum(lens)[:-1]
    flat = np.concatenate(codes).astype(np.uint8)
    ia = np.ascontiguousarray(ia, np.int32)
    ib = np.ascontiguousarray(ib, np.int32)
    sizes = (lens[ia].astype(np.int64) + lens[ib] + 3)
    poff = np.zeros(len(ia), np.int64)
    poff[1:] = np.cumsum(sizes)[:-1]
    paths = np.zeros(int(sizes.sum()), np.int32) if want_paths else None
    secs = C.c_double(0.0)
    sub = np.ascontiguousarray(subm, np.float32).reshape(-1)
    rc = lib().refh_pairwise_batch(_ptr(flat), _ptr(off), _ptr(lens), _ptr(ia), _ptr(ib), len(ia),
                                   _ptr(sub), gpo, gpe, tgpe, n_threads,
                                   _ptr(paths) if want_paths else None, _ptr(poff), C.byref(secs))
    if rc:
        raise RuntimeError("pairwise batch failed")
    out = None
    if want_paths:
        out = [paths[poff[k]:poff[k] + paths[poff[k]] + 2].copy() for k in range(len(ia))]
    return out, secs.value


def bpm_batch(codes, ia, ib):
    """calc_distance -> bpm_block of the reference for a list of pairs (codes < 13)."""
    lens = np.array([len(c) for c in codes], np.int32)
    off = np.zeros(len(codes), np.int32)
    off[1:] = np.cumsum(lens)[:-1]
    flat = np.ascontiguousarray(np.concatenate(codes), np.uint8)
    ia = np.ascontiguousarray(ia, np.int32)
    ib = np.ascontiguousarray(ib, np.int32)
    out = np.zeros(len(ia), np.int32)
    lib().refh_bpm_batch(_ptr(flat), _ptr(off), _ptr(lens), _ptr(ia), _ptr(ib), len(ia), _ptr(out))
    return out
