"""A CPU stand-in for kalign_amd.Context's partial-run interface, built from the oracle's per-task
primitives (tests only).  It lets the world_size-2 gloo test exercise kalign_amd.dist.sharded_tree --
the subtree plan, the profile hand-over between ranks and the final gather -- without a GPU."""
import ctypes as C

import numpy as np

from oracle import oracledrv


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleExecutor:
    def __init__(self, codes, tasks, subm, scal, gap_scale, subm_off):
        """gap_scale / subm_off: per task, as do_align derives them (aln_run.c:226-237); taken from a
        whole-tree oracle run by the caller."""
        self.L = oracledrv.lib()
        self.codes = [np.ascontiguousarray(c, np.uint8) for c in codes]
        self.tasks = np.asarray(tasks)
        self.subm = np.ascontiguousarray(subm, np.float32).reshape(-1)
        self.scal = np.asarray(scal, np.float32)
        self.gap_scale, self.subm_off = gap_scale, subm_off
        n = len(codes)
        self.n = n
        self.plen = {i: len(c) for i, c in enumerate(codes)}
        self.nsip = {i: 1 for i in range(n)}
        for a, b, c in self.tasks:
            self.nsip[int(c)] = self.nsip[int(a)] + self.nsip[int(b)]
        self.prof = {}
        self.done = {}

    def _leaf(self, i, gpo, gpe, tgpe, soff):
        out = np.zeros(64 * (self.plen[i] + 2), np.float32)
        self.L.ko_make_profile(_p(self.codes[i]), self.plen[i], _p(self.subm), C.c_float(gpo), C.c_float(gpe), C.c_float(tgpe), C.c_float(soff), _p(out))
        return out

    def tree_run_tasks(self, ids):
        for t in sorted(int(x) for x in ids):
            a, b, c = (int(v) for v in self.tasks[t])
            gs, soff = np.float32(self.gap_scale[t]), np.float32(self.subm_off[t])
            gpo, gpe, tgpe = self.scal[0], self.scal[1], self.scal[2]
            if gs < 1.0 or soff > 0.0:
                gpo, gpe, tgpe = np.float32(gpo * gs), np.float32(gpe * gs), np.float32(tgpe * gs)
            else:
                soff = np.float32(0.0)
            la, lb = self.plen[a], self.plen[b]
            na, nb = self.nsip[a], self.nsip[b]
            pa = self._leaf(a, gpo, gpe, tgpe, soff) if na == 1 else self.prof[a]
            pb = self._leaf(b, gpo, gpe, tgpe, soff) if nb == 1 else self.prof[b]
            if na > 1:
                self.L.ko_set_gap_penalties(_p(pa), la, nb)
            if nb > 1:
                self.L.ko_set_gap_penalties(_p(pb), lb, na)
            swapped = 0
            if na == 1 and nb == 1:
                kind = 0
                if la < lb:
                    kw = dict(seq1=self.codes[a], seq2=self.codes[b])
                else:
                    swapped = 1
                    kw = dict(seq1=self.codes[b], seq2=self.codes[a])
                sip = 1
            elif na == 1:
                kind, swapped, sip = 1, 1, nb
                kw = dict(seq2=self.codes[a], prof1=pb)
            elif nb == 1:
                kind, sip = 1, na
                kw = dict(seq2=self.codes[b], prof1=pa)
            else:
                kind, sip = 2, 1
                if la < lb:
                    kw = dict(prof1=pa, prof2=pb)
                else:
                    swapped = 1
                    kw = dict(prof1=pb, prof2=pa)
            dla, dlb = (lb, la) if swapped else (la, lb)
            r = oracledrv.dp_single(kind, dla, dlb, self.subm, float(gpo), float(gpe), float(tgpe), float(soff), sip, **kw)
            raw = np.zeros(la + lb + 4, np.int32)
            raw[:dla + 2] = r["raw"]
            if swapped:
                raw2 = np.zeros(la + lb + 4, np.int32)
                self.L.ko_mirror_path(_p(raw), la, lb, _p(raw2))
                raw = raw2
            coded = np.zeros(la + lb + 3, np.int32)
            self.L.ko_code_path(_p(raw), la, lb, _p(coded))
            plen = int(coded[0])
            merged = np.zeros(64 * (plen + 2), np.float32)
            if t != len(self.tasks) - 1:
                self.L.ko_update_profile(_p(pa), _p(pb), _p(merged), _p(coded), na, nb, _p(self.subm),
                                         C.c_float(self.scal[0]), C.c_float(self.scal[1]), C.c_float(self.scal[2]), C.c_float(self.scal[5]))
            self.prof[c] = merged
            self.plen[c] = plen
            rec = oracledrv.TaskRec()
            rec.a, rec.b, rec.c = a, b, c
            rec.len_a, rec.len_b, rec.nsip_a, rec.nsip_b = la, lb, na, nb
            rec.plen, rec.kind, rec.swapped = plen, kind, swapped
            rec.meet, rec.transition, rec.score, rec.confidence = r["meet"], r["transition"], r["score"], r["confidence"]
            rec.gap_scale, rec.subm_off = float(gs), float(soff)
            self.done[t] = (rec, coded[:plen + 2].copy())

    def tree_get_node(self, node):
        return self.prof[node]

    def tree_set_node(self, node, prof):
        self.prof[node] = np.array(prof, np.float32)
        self.plen[node] = len(prof) // 64 - 2

    def tree_download_tasks(self, ids):
        recs, chunks, off = [], [], 0
        for t in ids:
            rec, path = self.done[int(t)]
            rec.path_off = off
            recs.append(rec)
            chunks.append(path)
            off += len(path)
        return recs, (np.concatenate(chunks) if chunks else np.zeros(0, np.int32))
