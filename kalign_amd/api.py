"""ctypes binding of include/kalign_amd.h (host mirror of the reference's dispatcher seam).

Function names follow the reference: msa_tree() stands where create_msa_tree()
(lib/src/aln_run.c:43) is called, pairwise_batch() where anchor_consistency_build loops over
pairwise_align_map() (lib/src/anchor_consistency.c:246-267).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

FLAG_DEBUG_ROWS = 1
FLAG_TIMING = 2
FLAG_DEVICE_GAPS = 4
FLAG_KEEP_CONSISTENCY = 8
FLAG_EXACT_CONFIDENCE = 16
FLAG_LEAF_PROFILES = 32


class KalignAmdError(RuntimeError):
    pass


class TaskRec(C.Structure):
    """ka_task_rec (include/kalign_amd.h)"""
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


# ka_dist_fn of include/kalign_amd.h
DIST_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int))

EXPORTS = ["ka_tree_profile_dev", "ka_tree_reserve_profile_dev", "ka_tree_build_consistency_part",
           "ka_tree_consistency_part_range", "ka_tree_consistency_maps_dev", "ka_debug_set_hooks", "ka_debug_reload_env", "ka_debug_tp_launches", "ka_ctx_fallback_runs", "ka_ctx_helped_tasks", "ka_ctx_create", "ka_ctx_destroy", "ka_ctx_set_stream", "ka_ctx_set_shared", "ka_last_error", "ka_abi_version",
           "ka_msa_tree", "ka_tree_upload", "ka_tree_run", "ka_tree_refine", "ka_tree_sync", "ka_tree_paths_size",
           "ka_tree_download", "ka_tree_get_profile", "ka_tree_get_timing", "ka_debug_trace", "ka_tree_cells", "ka_tree_kernel_ms", "ka_tree_launch_ms",
           "ka_pairwise_batch", "ka_pairwise_kernel_ms", "ka_tree_build_consistency", "ka_tree_get_consistency",
           "ka_tree_run_tasks", "ka_tree_reset", "ka_tree_node_len", "ka_tree_set_profile", "ka_tree_download_tasks", "ka_weave_gaps",
           "ka_tree_node_cols_size", "ka_tree_get_node_cols", "ka_tree_set_node_cols", "ka_bpm_batch",
           "ka_tree_aligned_rows", "ka_guide_tree", "ka_guide_tree_from",
           "ka_aln_guide_tree", "ka_run_encoded", "ka_run_encoded_refine", "ka_tree_plan_tasks", "ka_tree_run_planned",
           "ka_dist_unique_id", "ka_dist_create", "ka_dist_destroy", "ka_dist_plan_subtrees", "ka_dist_plan", "ka_dist_get_plan",
           "ka_dist_consistency", "ka_dist_tree_run", "ka_dist_paths_size", "ka_dist_download", "ka_dist_last_ms", "ka_dist_retries",
           "ka_dist_loopback_new", "ka_dist_loopback_free", "ka_dist_create_loopback",
           "ka_guide_last_bisect_ms", "ka_device_count", "ka_multi_create", "ka_multi_destroy", "ka_multi_world", "ka_multi_runs", "ka_multi_last_error",
           "ka_multi_consistency", "ka_multi_tree_run", "ka_multi_paths_size", "ka_multi_download", "ka_multi_ctx", "ka_multi_adopt",
           "ka_tree_adopt_alignment"]


def lib_path():
    return os.path.join(_HERE, "libkalign_amd.so")


_lib = None


def load_library():
    """Loads the HIP library.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise KalignAmdError("%s not built: run `make -C kalign_amd/csrc` or __graft_entry__.build()" % p)
    L = C.CDLL(p)
    vp = C.c_void_p
    L.ka_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.ka_ctx_destroy.argtypes = [vp]
    L.ka_ctx_destroy.restype = None
    L.ka_ctx_set_stream.argtypes = [vp, vp]
    L.ka_ctx_set_shared.argtypes = [vp, C.c_int]
    L.ka_last_error.restype = C.c_char_p
    L.ka_debug_set_hooks.argtypes = [vp, C.c_int]
    L.ka_debug_reload_env.argtypes = [vp]
    L.ka_tree_profile_dev.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_int)]
    L.ka_tree_reserve_profile_dev.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.ka_tree_build_consistency_part.argtypes = [vp, C.c_int, C.c_float, C.c_int, C.c_int]
    L.ka_tree_consistency_part_range.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    L.ka_tree_consistency_maps_dev.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_longlong)]
    L.ka_ctx_fallback_runs.argtypes = [vp]
    L.ka_ctx_helped_tasks.argtypes = [vp]
    L.ka_ctx_helped_tasks.restype = C.c_longlong
    L.ka_debug_tp_launches.argtypes = []
    L.ka_debug_tp_launches.restype = C.c_longlong
    L.ka_abi_version.restype = C.c_int
    L.ka_msa_tree.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int,
                              C.POINTER(TaskRec), vp, C.c_longlong, vp]
    L.ka_tree_upload.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int]
    L.ka_tree_run.argtypes = [vp]
    L.ka_tree_refine.argtypes = [vp, C.c_int, vp]
    L.ka_tree_sync.argtypes = [vp]
    L.ka_tree_paths_size.argtypes = [vp]
    L.ka_tree_paths_size.restype = C.c_longlong
    L.ka_tree_download.argtypes = [vp, C.POINTER(TaskRec), vp, C.c_longlong, vp]
    L.ka_tree_get_profile.argtypes = [vp, C.c_int, vp, C.c_longlong]
    L.ka_tree_get_timing.argtypes = [vp, vp]
    L.ka_debug_trace.argtypes = [vp, vp]
    L.ka_pairwise_kernel_ms.argtypes = [vp]
    L.ka_pairwise_kernel_ms.restype = C.c_float
    L.ka_tree_cells.argtypes = [vp]
    L.ka_tree_cells.restype = C.c_double
    L.ka_tree_kernel_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.ka_tree_launch_ms.argtypes = [vp, vp, C.c_int]
    L.ka_tree_run_tasks.argtypes = [vp, vp, C.c_int]
    L.ka_tree_plan_tasks.argtypes = [vp, vp, C.c_int]
    L.ka_tree_run_planned.argtypes = [vp]
    L.ka_dist_unique_id.argtypes = [vp]
    L.ka_dist_create.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
    L.ka_dist_destroy.argtypes = [vp]
    L.ka_dist_destroy.restype = None
    L.ka_dist_plan_subtrees.argtypes = [C.c_int, vp, C.c_int, vp, C.c_int, vp, vp, C.POINTER(C.c_int)]
    L.ka_dist_plan.argtypes = [vp]
    L.ka_dist_get_plan.argtypes = [vp, vp, vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ka_dist_consistency.argtypes = [vp, C.c_int, C.c_float]
    L.ka_dist_tree_run.argtypes = [vp]
    L.ka_dist_paths_size.argtypes = [vp]
    L.ka_dist_paths_size.restype = C.c_longlong
    L.ka_dist_download.argtypes = [vp, vp, vp, C.c_longlong, C.POINTER(C.c_longlong)]
    L.ka_dist_last_ms.argtypes = [vp]
    L.ka_dist_last_ms.restype = C.c_double
    L.ka_dist_retries.argtypes = [vp]
    L.ka_device_count.argtypes = []
    L.ka_guide_last_bisect_ms.argtypes = [vp]
    L.ka_guide_last_bisect_ms.restype = C.c_double
    L.ka_multi_create.argtypes = [C.c_int, vp, C.c_int, C.POINTER(vp)]
    L.ka_multi_destroy.argtypes = [vp]
    L.ka_multi_destroy.restype = None
    L.ka_multi_world.argtypes = [vp]
    L.ka_multi_runs.argtypes = [vp]
    L.ka_multi_runs.restype = C.c_longlong
    L.ka_multi_last_error.argtypes = []
    L.ka_multi_last_error.restype = C.c_char_p
    L.ka_multi_consistency.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_float, vp, vp]
    L.ka_multi_tree_run.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_float]
    L.ka_multi_paths_size.argtypes = [vp]
    L.ka_multi_paths_size.restype = C.c_longlong
    L.ka_multi_download.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp, C.c_longlong, vp]
    L.ka_multi_ctx.argtypes = [vp, C.c_int]
    L.ka_multi_ctx.restype = vp
    L.ka_multi_adopt.argtypes = [vp, vp, vp]
    L.ka_tree_adopt_alignment.argtypes = [vp, vp, vp]
    L.ka_dist_loopback_new.argtypes = [C.c_int]
    L.ka_dist_loopback_new.restype = vp
    L.ka_dist_loopback_free.argtypes = [vp]
    L.ka_dist_loopback_free.restype = None
    L.ka_dist_create_loopback.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
    L.ka_tree_reset.argtypes = [vp]
    L.ka_tree_node_len.argtypes = [vp, C.c_int]
    L.ka_tree_set_profile.argtypes = [vp, C.c_int, vp, C.c_int]
    L.ka_tree_node_cols_size.argtypes = [vp, C.c_int]
    L.ka_tree_node_cols_size.restype = C.c_longlong
    L.ka_tree_get_node_cols.argtypes = [vp, C.c_int, vp]
    L.ka_tree_set_node_cols.argtypes = [vp, C.c_int, vp]
    L.ka_tree_download_tasks.argtypes = [vp, vp, C.c_int, C.POINTER(TaskRec), vp, C.c_longlong, C.POINTER(C.c_longlong)]
    L.ka_weave_gaps.argtypes = [C.c_int, vp, C.c_int, C.POINTER(TaskRec), vp, vp]
    L.ka_bpm_batch.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp, C.c_int, vp]
    L.ka_tree_build_consistency.argtypes = [vp, C.c_int, C.c_float]
    L.ka_tree_aligned_rows.argtypes = [vp, vp, C.c_ubyte, vp, C.c_longlong, vp]
    L.ka_run_encoded.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_float, C.c_int, vp, C.c_int,
                                 C.c_ubyte, vp, C.c_longlong, vp]
    L.ka_run_encoded_refine.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_float, C.c_int, vp, C.c_int, C.c_int,
                                        C.c_ubyte, vp, C.c_longlong, vp]
    L.ka_aln_guide_tree.argtypes = [vp, C.c_int, vp, C.c_longlong, C.c_int, C.c_ubyte, vp, vp, vp]
    L.ka_guide_tree.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp]
    L.ka_guide_tree_from.argtypes = [C.c_int, vp, DIST_FN, vp, C.c_int, vp, vp, vp]
    L.ka_tree_get_consistency.argtypes = [vp, vp, vp]
    L.ka_pairwise_batch.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp, C.c_int, vp,
                                    C.c_float, C.c_float, C.c_float, vp, vp, vp]
    _lib = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _flatten(codes):
    lens = np.array([len(c) for c in codes], np.int32)
    off = np.zeros(len(codes), np.int32)
    off[1:] = np.cumsum(lens)[:-1]
    flat = np.ascontiguousarray(np.concatenate(codes), np.uint8)
    return flat, off, lens


class Context:
    """One GPU context (ka_ctx)."""

    def __init__(self, device=0, stream=None, shared=False):
        """shared=True: other streams / processes use the GPU at the same time (ka_ctx_set_shared)."""
        self.L = load_library()
        h = C.c_void_p()
        if self.L.ka_ctx_create(device, C.byref(h)):
            raise KalignAmdError(self.L.ka_last_error().decode())
        self.h = h
        if stream is not None:
            self.L.ka_ctx_set_stream(self.h, C.c_void_p(stream))
        if shared:
            self.L.ka_ctx_set_shared(self.h, 1)
        self._job = None

    @classmethod
    def borrowed(cls, handle, job=None):
        """A view of a context somebody else owns (Multi.ctx): close() does not destroy it."""
        self = cls.__new__(cls)
        self.L = load_library()
        self.h = C.c_void_p(handle) if not isinstance(handle, C.c_void_p) else handle
        self._job = job
        self._borrowed = True
        return self

    def close(self):
        if getattr(self, "h", None):
            if not getattr(self, "_borrowed", False):
                self.L.ka_ctx_destroy(self.h)
            self.h = None

    def adopt_alignment(self, recs, gaps):
        """ka_tree_adopt_alignment: this context (same uploaded job) takes over an alignment made elsewhere."""
        arr = (TaskRec * len(recs))(*recs)
        flat = np.ascontiguousarray(np.concatenate([np.asarray(g, np.int32) for g in gaps]), np.int32)
        self._chk(self.L.ka_tree_adopt_alignment(self.h, arr, _ptr(flat)))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def debug_set_hooks(self, hooks):
        """tests only: ka_debug_set_hooks (KA_DEBUG_SMALL_ARENAS = 1, KA_DEBUG_STARVE_ROOT_JOIN = 2, KA_DEBUG_STARVE_REFINE_MEMBER = 4, KA_DEBUG_CHAIN_FIRST = 8, KA_DEBUG_POISON_ARENAS = 16)"""
        self._chk(self.L.ka_debug_set_hooks(self.h, int(hooks)))

    def reload_env(self):
        """tools / tests: the KA_* environment switches are read once at context creation; read them again"""
        self._chk(self.L.ka_debug_reload_env(self.h))

    def tp_launches(self):
        """launches of the throughput kernel (KA_TP=1) since the library was loaded (ka_debug_tp_launches)"""
        return int(self.L.ka_debug_tp_launches())

    def fallback_runs(self):
        return int(self.L.ka_ctx_fallback_runs(self.h))

    def helped_tasks(self):
        """queue tasks of the last run that workgroups of the chained launch took over (ka_ctx_helped_tasks)"""
        return int(self.L.ka_ctx_helped_tasks(self.h))

    def _chk(self, rc):
        if rc:
            raise KalignAmdError("rc=%d: %s" % (rc, self.L.ka_last_error().decode()))

    # ---- staged dispatcher -------------------------------------------------------------
    def tree_upload(self, codes, tasks, subm, scal, seq_distances=None, flags=0):
        flat, off, lens = _flatten(codes)
        tasks = np.ascontiguousarray(tasks, np.int32)
        sd = None if seq_distances is None else np.ascontiguousarray(seq_distances, np.float32)
        sub = np.ascontiguousarray(subm, np.float32).reshape(-1)
        sc = np.ascontiguousarray(scal, np.float32)
        self._chk(self.L.ka_tree_upload(self.h, len(codes), _ptr(flat), _ptr(off), _ptr(lens), _ptr(sd),
                                        len(tasks), _ptr(tasks), _ptr(sub), _ptr(sc), flags))
        self._job = dict(lens=lens, ntasks=len(tasks), n=len(codes))

    def tree_run(self):
        self._chk(self.L.ka_tree_run(self.h))

    def tree_refine(self, mode, conf_in=None):
        """refine_alignment (aln_refine.c:34-85) over the uploaded tree: mode 1 = all edges, 2 = edges whose
        first-pass confidence (conf_in; None: computed on the device) is at or below the median, 3 = inline
        refinement (create_msa_tree_inline_refine), 4 = the plain first pass, depth first (exact confidences)."""
        cf = None if conf_in is None else np.ascontiguousarray(conf_in, np.float32)
        if cf is not None and len(cf) != self._job["ntasks"]:
            raise KalignAmdError("tree_refine: one confidence per task")
        self._chk(self.L.ka_tree_refine(self.h, int(mode), None if cf is None else _ptr(cf)))

    def tree_sync(self):
        self._chk(self.L.ka_tree_sync(self.h))

    def tree_kernel_ms(self):
        ms, n = C.c_float(0), C.c_int(0)
        self._chk(self.L.ka_tree_kernel_ms(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def tree_plan_tasks(self, task_ids):
        """the listed tasks as ONE planned run (queued / chained launches); None = the whole tree again"""
        if task_ids is None:
            self._chk(self.L.ka_tree_plan_tasks(self.h, None, 0))
        else:
            ids = np.ascontiguousarray(task_ids, np.int32)
            self._chk(self.L.ka_tree_plan_tasks(self.h, _ptr(ids), len(ids)))

    def tree_run_planned(self):
        self._chk(self.L.ka_tree_run_planned(self.h))

    def tree_launch_ms(self):
        """per-launch kernel times of the last run (KA_LAUNCH_EV=1 when the context was created / reload_env)"""
        out = np.zeros(96, np.float32)
        n = self.L.ka_tree_launch_ms(self.h, _ptr(out), 96)
        if n < 0:
            raise RuntimeError(self.L.ka_last_error().decode())
        return out[:n].tolist()

    def tree_cells(self):
        return self.L.ka_tree_cells(self.h)

    def tree_download(self, want_gaps=True):
        j = self._job
        self.tree_sync()
        cap = self.L.ka_tree_paths_size(self.h)
        recs = (TaskRec * j["ntasks"])()
        paths = np.zeros(max(int(cap), 1), np.int32)
        gaps = np.zeros(int(j["lens"].sum()) + j["n"], np.int32) if want_gaps else None
        self._chk(self.L.ka_tree_download(self.h, recs, _ptr(paths), cap, _ptr(gaps)))
        g = None
        if want_gaps:
            g, o = [], 0
            for n in j["lens"]:
                g.append(gaps[o:o + int(n) + 1].copy())
                o += int(n) + 1
        return recs, paths, g

    def tree_aligned_rows(self, letters, gap=b"-"):
        """finalise_alignment on the device: `letters` holds, per sequence, what to print for each residue (bytes,
        str or uint8 array); returns one bytes object per sequence, as long as the alignment of its tree."""
        j = self._job

        def as_u8(x):
            if isinstance(x, np.ndarray):
                return x.astype(np.uint8)
            return np.frombuffer(x.encode() if isinstance(x, str) else bytes(x), np.uint8)

        flat = np.ascontiguousarray(np.concatenate([as_u8(x) for x in letters]))
        if len(flat) != int(j["lens"].sum()) or any(len(x) != n for x, n in zip(letters, j["lens"])):
            raise KalignAmdError("letters do not match the uploaded sequences")
        alen = np.zeros(j["n"], np.int32)
        self._chk(self.L.ka_tree_aligned_rows(self.h, _ptr(flat), gap[0], None, 0, _ptr(alen)))   # size query
        stride = int(alen.max()) + 1
        rows = np.zeros((j["n"], stride), np.uint8)
        self._chk(self.L.ka_tree_aligned_rows(self.h, _ptr(flat), gap[0], _ptr(rows), stride, _ptr(alen)))
        assert all(rows[i, alen[i]] == 0 for i in range(j["n"]))
        return [rows[i, :alen[i]].tobytes() for i in range(j["n"])]

    def tree_timing(self):
        n = self._job["ntasks"]
        out = np.zeros(8 * n + 48 + 512, np.int64)
        self._chk(self.L.ka_tree_get_timing(self.h, _ptr(out)))
        self.root_levels = out[8 * n:8 * n + 48].reshape(16, 3)      # per recursion level of the root task: n, pass, meetup
        self.prof = out[8 * n + 48:].reshape(8, 8, 8)                 # KA_PROF builds: [level][wave][slot]
        return out[:8 * n].reshape(n, 8)

    def pairwise_kernel_ms(self):
        return float(self.L.ka_pairwise_kernel_ms(self.h))

    def debug_trace(self):
        out = np.zeros(64, np.int32)
        self._chk(self.L.ka_debug_trace(self.h, _ptr(out)))
        return out

    def tree_profile(self, node, max_cols):
        out = np.zeros(64 * (max_cols + 2), np.float32)
        self._chk(self.L.ka_tree_get_profile(self.h, node, _ptr(out), out.size))
        return out

    # ---- one-shot -----------------------------------------------------------------------
    # ---- partial runs (single-tree multi-GPU sharding, kalign_amd/dist.py:sharded_tree) ----
    def tree_run_tasks(self, task_ids):
        ids = np.ascontiguousarray(task_ids, np.int32)
        self._chk(self.L.ka_tree_run_tasks(self.h, _ptr(ids), len(ids)))

    def tree_reset(self):
        self._chk(self.L.ka_tree_reset(self.h))

    def tree_node_len(self, node):
        n = self.L.ka_tree_node_len(self.h, int(node))
        if n < 0:
            raise KalignAmdError("ka_tree_node_len failed")
        return n

    def tree_get_node(self, node):
        """State of an internal node as one float32 array: the merged profile [(plen+2)*64] followed, when the job
        has a consistency table, by the residue->column table of its members (ints viewed as float32)."""
        self.tree_sync()
        prof = self.tree_profile(node, self.tree_node_len(node))
        if self.L.ka_tree_get_consistency(self.h, None, None) > 0:
            n = int(self.L.ka_tree_node_cols_size(self.h, int(node)))
            cols = np.zeros(n, np.int32)
            self._chk(self.L.ka_tree_get_node_cols(self.h, int(node), _ptr(cols)))
            return np.concatenate([prof, cols.view(np.float32)])
        return prof

    def tree_set_node(self, node, state):
        state = np.ascontiguousarray(state, np.float32).reshape(-1)
        ncols = 0
        if self.L.ka_tree_get_consistency(self.h, None, None) > 0:
            ncols = int(self.L.ka_tree_node_cols_size(self.h, int(node)))
        prof = np.ascontiguousarray(state[:len(state) - ncols])
        self._chk(self.L.ka_tree_set_profile(self.h, int(node), _ptr(prof), len(prof) // 64 - 2))
        if ncols:
            cols = np.ascontiguousarray(state[len(state) - ncols:]).view(np.int32)
            self._chk(self.L.ka_tree_set_node_cols(self.h, int(node), _ptr(cols)))

    # ---- device-to-device hand-over (sharded trees; kalign_amd/dist.py wraps the pointers as torch tensors) ----
    def tree_profile_dev(self, node):
        """(device pointer, plen) of a node's merged profile: (plen+2)*64 float32 in this context's HBM"""
        ptr, plen = C.c_void_p(), C.c_int(0)
        self._chk(self.L.ka_tree_profile_dev(self.h, int(node), C.byref(ptr), C.byref(plen)))
        return ptr.value, plen.value

    def tree_reserve_profile_dev(self, node, plen):
        """room for an incoming profile of `plen` columns; returns the device pointer to fill before the parent runs"""
        ptr = C.c_void_p()
        self._chk(self.L.ka_tree_reserve_profile_dev(self.h, int(node), int(plen), C.byref(ptr)))
        return ptr.value

    def tree_node_cols(self, node):
        """residue -> column table of a node's members (None without a consistency table)"""
        if self.L.ka_tree_get_consistency(self.h, None, None) <= 0:
            return None
        n = int(self.L.ka_tree_node_cols_size(self.h, int(node)))
        cols = np.zeros(n, np.int32)
        self._chk(self.L.ka_tree_get_node_cols(self.h, int(node), _ptr(cols)))
        return cols

    def tree_set_node_cols(self, node, cols):
        cols = np.ascontiguousarray(cols, np.int32)
        self._chk(self.L.ka_tree_set_node_cols(self.h, int(node), _ptr(cols)))

    def tree_build_consistency_part(self, n_anchors, weight, part, nparts):
        """this rank's share of the N x K batch of a sharded consistency build (ka_tree_build_consistency_part)"""
        self._chk(self.L.ka_tree_build_consistency_part(self.h, int(n_anchors), float(weight), int(part), int(nparts)))

    def tree_consistency_part_range(self, part, nparts):
        lo, hi = C.c_longlong(0), C.c_longlong(0)
        self._chk(self.L.ka_tree_consistency_part_range(self.h, int(part), int(nparts), C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def tree_consistency_maps_dev(self):
        """(device pointer, number of int32) of the position-map table in HBM"""
        ptr, n = C.c_void_p(), C.c_longlong(0)
        self._chk(self.L.ka_tree_consistency_maps_dev(self.h, C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

    # (the executor interface of kalign_amd.dist.sharded_consistency)
    def cons_build_part(self, n_anchors, weight, part, nparts):
        self.tree_build_consistency_part(n_anchors, weight, part, nparts)

    def cons_part_range(self, part, nparts):
        return self.tree_consistency_part_range(part, nparts)

    def cons_table(self):
        import torch
        from . import dist as kd
        if self.L.ka_tree_get_consistency(self.h, None, None) <= 0:
            return None
        ptr, n = self.tree_consistency_maps_dev()
        return kd.dev_tensor(ptr, n, torch.int32)

    def tree_download_tasks(self, task_ids):
        """(recs, paths) of the listed tasks; recs[i].path_off indexes `paths`."""
        ids = np.ascontiguousarray(task_ids, np.int32)
        recs = (TaskRec * max(len(ids), 1))()
        cap = int(self._job["lens"].sum()) * 2 * max(len(ids), 1) + 8 * len(ids) + 16
        cap = min(cap, (int(self._job["lens"].sum()) + 2) * (len(ids) + 1) + 16)
        paths = np.zeros(cap, np.int32)
        used = C.c_longlong(0)
        self._chk(self.L.ka_tree_download_tasks(self.h, _ptr(ids), len(ids), recs, _ptr(paths), cap, C.byref(used)))
        return list(recs)[:len(ids)], paths[:used.value]

    def tree_build_consistency(self, n_anchors=5, weight=2.0):
        """anchor_consistency_build: call between tree_upload and tree_run (the reference's default mode)."""
        self._chk(self.L.ka_tree_build_consistency(self.h, int(n_anchors), float(weight)))

    def tree_consistency(self):
        """(anchor_ids[K], maps[i][k]) or None when the job has no consistency table."""
        K = self.L.ka_tree_get_consistency(self.h, None, None)
        if K <= 0:
            return None
        lens = self._job["lens"]
        ids = np.full(max(len(lens), K), -1, np.int32)        # a forest job: K anchors per alignment
        flat = np.zeros(int(lens.sum()) * K, np.int32)
        self.L.ka_tree_get_consistency(self.h, _ptr(ids), _ptr(flat))
        ids = ids[ids >= 0]
        maps, o = [], 0
        for n in lens:
            row = []
            for _ in range(K):
                row.append(flat[o:o + int(n)].copy())
                o += int(n)
            maps.append(row)
        return ids, maps

    def msa_tree(self, codes, tasks, subm, scal, seq_distances=None, flags=0, n_anchors=0, weight=2.0):
        # like ka_msa_tree: the gap arrays are wanted, so let the device keep every residue's column
        self.tree_upload(codes, tasks, subm, scal, seq_distances, flags | FLAG_DEVICE_GAPS)
        if n_anchors > 0:
            self.tree_build_consistency(n_anchors, weight)
        self.tree_run()
        return self.tree_download()

    def pairwise_batch(self, codes, ia, ib, subm, gpo, gpe, tgpe):
        flat, off, lens = _flatten(codes)
        ia = np.ascontiguousarray(ia, np.int32)
        ib = np.ascontiguousarray(ib, np.int32)
        sizes = lens[ia].astype(np.int64) + lens[ib] + 3
        poff = np.zeros(len(ia), np.int64)
        poff[1:] = np.cumsum(sizes)[:-1]
        paths = np.zeros(int(sizes.sum()), np.int32)
        scores = np.zeros(len(ia), np.float32)
        sub = np.ascontiguousarray(subm, np.float32).reshape(-1)
        self._chk(self.L.ka_pairwise_batch(self.h, _ptr(flat), _ptr(off), _ptr(lens), len(codes),
                                           _ptr(ia), _ptr(ib), len(ia), _ptr(sub),
                                           float(gpo), float(gpe), float(tgpe), _ptr(paths), _ptr(poff), _ptr(scores)))
        return [paths[poff[k]:poff[k] + paths[poff[k]] + 2].copy() for k in range(len(ia))], scores


def _bpm_batch(self, codes, ia, ib):
    """calc_distance / bpm_block for a list of pairs (codes < 13): int32 distances."""
    flat, off, lens = _flatten(codes)
    ia = np.ascontiguousarray(ia, np.int32)
    ib = np.ascontiguousarray(ib, np.int32)
    out = np.zeros(len(ia), np.int32)
    self._chk(self.L.ka_bpm_batch(self.h, _ptr(flat), _ptr(off), _ptr(lens), len(codes), _ptr(ia), _ptr(ib), len(ia), _ptr(out)))
    return out


Context.bpm_batch = _bpm_batch


def _dm_scale(noise, n):
    if noise is None:
        return None
    noise = np.ascontiguousarray(noise, np.float32).reshape(-1)
    if len(noise) != n * min(32, n):
        raise KalignAmdError("dm_scale needs numseq * min(32, numseq) multipliers")
    return noise


def _guide_tree(self, codes, n_threads=1, dm_scale=None):
    """build_tree_kmeans with both distance batches on the device: (tasks[n-1, 3], seq_distances[n]).
    `codes` in the alphabet the reference builds its tree in (reduced protein alphabet / nucleotides);
    dm_scale: the multipliers of build_tree_kmeans_noisy, or None."""
    flat, off, lens = _flatten(codes)
    tasks = np.zeros((len(codes) - 1, 3), np.int32)
    sd = np.zeros(len(codes), np.float32)
    sc = _dm_scale(dm_scale, len(codes))
    self._chk(self.L.ka_guide_tree(self.h, len(codes), _ptr(flat), _ptr(off), _ptr(lens), int(n_threads), _ptr(sc), _ptr(tasks), _ptr(sd)))
    return tasks, sd


Context.guide_tree = _guide_tree


def _aln_guide_tree(self, rows=None, n=None, gap=b"-", want_dm=False):
    """The guide tree of a realignment pass (compute_aln_pairwise_dist + build_tree_from_pairwise on the device):
    (tasks, seq_distances[, dm]).  rows: equal-length byte strings, or None for the rows the last
    tree_aligned_rows left in HBM (then n = number of sequences of that job)."""
    if rows is None:
        n = self._job["n"] if n is None else n
        flat, stride, alnlen = None, 0, 0
    else:
        n = len(rows)
        alnlen = len(rows[0])
        if any(len(r) != alnlen for r in rows):
            raise KalignAmdError("rows of one alignment have one length")
        flat = np.frombuffer(b"".join(bytes(r) for r in rows), np.uint8)
        stride = alnlen
    tasks = np.zeros((n - 1, 3), np.int32)
    sd = np.zeros(n, np.float32)
    dm = np.zeros((n, n), np.float32) if want_dm else None
    self._chk(self.L.ka_aln_guide_tree(self.h, n, _ptr(flat), stride, alnlen, gap[0], _ptr(tasks), _ptr(sd), _ptr(dm)))
    return (tasks, sd, dm) if want_dm else (tasks, sd)


Context.aln_guide_tree = _aln_guide_tree


def _run_encoded(self, tree_codes, codes, letters, subm, scal, n_anchors=0, weight=2.0, realign=0, dm_scale=None,
                 n_threads=1, gap=b"-", refine=0):
    """ka_run_encoded(_refine): guide tree, (consistency,) alignment, `realign` realignment iterations, (refinement,)
    rows -- one call.  refine: 0, 1, 2 (| 256 adaptive budget) or 3, as kalign_run_seeded's argument.
    Returns the aligned rows (bytes) in the order of the input sequences."""
    tflat, off, lens = _flatten(tree_codes)
    cflat, _, _ = _flatten(codes)
    lflat = np.ascontiguousarray(np.concatenate([np.frombuffer(x.encode() if isinstance(x, str) else bytes(x), np.uint8) for x in letters]))
    if len(lflat) != len(cflat) or len(tflat) != len(cflat):
        raise KalignAmdError("the three encodings of the sequences differ in length")
    n = len(codes)
    sub = np.ascontiguousarray(subm, np.float32).reshape(-1)
    sc = np.ascontiguousarray(scal, np.float32)
    dms = _dm_scale(dm_scale, n)
    alen = np.zeros(n, np.int32)
    args = (self.h, n, _ptr(tflat), _ptr(cflat), _ptr(lflat), _ptr(off), _ptr(lens), _ptr(sub), _ptr(sc),
            int(n_anchors), float(weight), int(realign), _ptr(dms), int(n_threads), gap[0])
    if refine:
        self._chk(self.L.ka_run_encoded_refine(*args[:14], int(refine), args[14], None, 0, _ptr(alen)))
    else:
        self._chk(self.L.ka_run_encoded(*args, None, 0, _ptr(alen)))        # the alignment stays in HBM: how long is it?
    self._job = dict(lens=lens, ntasks=n - 1, n=n)
    return self.tree_aligned_rows(letters, gap=gap)


Context.run_encoded = _run_encoded


def guide_tree_from(lens, dist, n_threads=1, dm_scale=None):
    """build_tree_kmeans with the caller's distance source (ka_guide_tree_from; host only, no GPU needed):
    dist(ia, ib) -> calc_distance of every pair, as an int array."""
    L = load_library()
    lens = np.ascontiguousarray(lens, np.int32)

    def cb(_user, n, ia, ib, out):
        try:
            d = dist(np.ctypeslib.as_array(ia, (n,)).copy(), np.ctypeslib.as_array(ib, (n,)).copy())
            np.ctypeslib.as_array(out, (n,))[:] = d
            return 0
        except Exception:                     # reported as "the distance source failed"
            return 1

    tasks = np.zeros((len(lens) - 1, 3), np.int32)
    sd = np.zeros(len(lens), np.float32)
    sc = _dm_scale(dm_scale, len(lens))
    if L.ka_guide_tree_from(len(lens), _ptr(lens), DIST_FN(cb), None, int(n_threads), _ptr(sc), _ptr(tasks), _ptr(sd)):
        raise KalignAmdError(L.ka_last_error().decode())
    return tasks, sd


def guide_last_bisect():
    """(milliseconds, ran on the device) of the 2-means bisection inside the last ka_guide_tree of this process"""
    L = load_library()
    on = C.c_int(0)
    ms = float(L.ka_guide_last_bisect_ms(C.byref(on)))
    return ms, bool(on.value)


def weave_gaps(lens, recs, paths):
    """Host-only make_seq/update_gaps over all tasks in tree order (ka_weave_gaps); needs the library but no GPU.
    recs: sequence of TaskRec in task order with path_off into `paths`.  Returns the gap array per sequence."""
    L = load_library()
    lens = np.ascontiguousarray(lens, np.int32)
    arr = (TaskRec * len(recs))(*recs)
    paths = np.ascontiguousarray(paths, np.int32)
    gaps = np.zeros(int(lens.sum()) + len(lens), np.int32)
    if L.ka_weave_gaps(len(lens), _ptr(lens), len(recs), arr, _ptr(paths), _ptr(gaps)):
        raise KalignAmdError(L.ka_last_error().decode())
    out, o = [], 0
    for n in lens:
        out.append(gaps[o:o + int(n) + 1].copy())
        o += int(n) + 1
    return out


def msa_tree(codes, tasks, subm, scal, seq_distances=None, flags=0, device=0):
    ctx = Context(device)
    try:
        return ctx.msa_tree(codes, tasks, subm, scal, seq_distances, flags)
    finally:
        ctx.close()


def pairwise_batch(codes, ia, ib, subm, gpo, gpe, tgpe, device=0):
    ctx = Context(device)
    try:
        return ctx.pairwise_batch(codes, ia, ib, subm, gpo, gpe, tgpe)
    finally:
        ctx.close()


def dist_plan_subtrees(lens, tasks, world):
    """ka_dist_plan_subtrees: (run_rank[n_tasks], top task ids) -- pure host logic, needs no GPU"""
    L = load_library()
    lens = np.ascontiguousarray(lens, np.int32)
    tasks = np.ascontiguousarray(tasks, np.int32)
    run_rank = np.zeros(len(tasks), np.int32)
    top = np.zeros(len(tasks), np.int32)
    n_top = C.c_int(0)
    if L.ka_dist_plan_subtrees(len(lens), _ptr(lens), len(tasks), _ptr(tasks), int(world), _ptr(run_rank), _ptr(top), C.byref(n_top)):
        raise RuntimeError(L.ka_last_error().decode())
    return run_rank, top[:n_top.value].tolist()


def dist_unique_id():
    """rank 0: the 128 bytes every rank hands to Dist (ncclGetUniqueId)"""
    L = load_library()
    buf = np.zeros(128, np.uint8)
    if L.ka_dist_unique_id(_ptr(buf)):
        raise RuntimeError(L.ka_last_error().decode())
    return buf


class Dist:
    """Thin caller of the C multi-GPU layer (ka_dist_*): ONE alignment over the GPUs of a node, RCCL driven from C.
    unique_id: the 128 bytes of dist_unique_id() from rank 0 (None with world == 1: no communicator)."""

    def __init__(self, ctx, rank, world, unique_id=None, loopback=None):
        self.ctx, self.L = ctx, ctx.L
        self.h = C.c_void_p()
        if loopback is not None:                      # tests: threads of one process (ka_dist_loopback_new)
            rc = self.L.ka_dist_create_loopback(ctx.h, int(rank), int(world), loopback, C.byref(self.h))
        else:
            uid = None if unique_id is None else np.ascontiguousarray(unique_id, np.uint8)
            rc = self.L.ka_dist_create(ctx.h, int(rank), int(world), _ptr(uid) if uid is not None else None, C.byref(self.h))
        if rc:
            raise RuntimeError(self.L.ka_last_error().decode())

    def _chk(self, rc):
        if rc:
            raise RuntimeError(self.L.ka_last_error().decode())

    def plan(self):
        self._chk(self.L.ka_dist_plan(self.h))

    def get_plan(self):
        n = self.ctx._job["ntasks"]
        run_rank, top = np.zeros(n, np.int32), np.zeros(n, np.int32)
        n_top, n_moves = C.c_int(0), C.c_int(0)
        self._chk(self.L.ka_dist_get_plan(self.h, _ptr(run_rank), _ptr(top), C.byref(n_top), C.byref(n_moves)))
        return run_rank, top[:n_top.value].tolist(), n_moves.value

    def consistency(self, n_anchors, weight):
        self._chk(self.L.ka_dist_consistency(self.h, int(n_anchors), float(weight)))

    def tree_run(self):
        self._chk(self.L.ka_dist_tree_run(self.h))

    def download(self):
        n = self.ctx._job["ntasks"]
        recs = (TaskRec * n)()
        cap = int(self.L.ka_dist_paths_size(self.h))
        paths = np.zeros(max(cap, 1), np.int32)
        used = C.c_longlong(0)
        self._chk(self.L.ka_dist_download(self.h, recs, _ptr(paths), cap, C.byref(used)))
        return list(recs), paths[:used.value]

    def last_ms(self):
        return float(self.L.ka_dist_last_ms(self.h))

    def retries(self):
        return int(self.L.ka_dist_retries(self.h))

    def close(self):
        if self.h:
            self.L.ka_dist_destroy(self.h)
            self.h = C.c_void_p()


class Multi:
    """ka_multi_*: the GPUs of one node under one caller -- one context and one rank of the sharded path per device, the ranks
    as threads inside the library.  loopback=True: every rank on device 0 over the in-process transport (one-GPU boxes)."""

    def __init__(self, world, loopback=False, devices=None):
        self.L = load_library()
        self.h = C.c_void_p()
        dev = None if devices is None else np.ascontiguousarray(devices, np.int32)
        if self.L.ka_multi_create(int(world), _ptr(dev) if dev is not None else None, 1 if loopback else 0, C.byref(self.h)):
            raise RuntimeError(self.L.ka_multi_last_error().decode())
        self._job = None

    def _chk(self, rc):
        if rc:
            raise RuntimeError(self.L.ka_multi_last_error().decode())

    def _args(self, codes, tasks, subm, scal, seq_distances):
        flat, off, lens = _flatten(codes)
        tasks = np.ascontiguousarray(tasks, np.int32)
        sd = np.ascontiguousarray(seq_distances, np.float32)
        sub = np.ascontiguousarray(subm, np.float32).reshape(-1)
        sc = np.ascontiguousarray(scal, np.float32)
        self._job = {"lens": lens, "ntasks": len(tasks), "keep": (flat, off, lens, tasks, sd, sub, sc)}
        return flat, off, lens, tasks, sd, sub, sc

    def consistency(self, codes, tasks, subm, scal, seq_distances, n_anchors, weight):
        flat, off, lens, tasks, sd, sub, sc = self._args(codes, tasks, subm, scal, seq_distances)
        ids = np.zeros(max(n_anchors, 1), np.int32)
        maps = np.zeros(int(lens.sum()) * max(n_anchors, 1) + 1, np.int32)
        k = self.L.ka_multi_consistency(self.h, len(lens), _ptr(flat), _ptr(off), _ptr(lens), _ptr(sd), len(tasks), _ptr(tasks),
                                        _ptr(sub), _ptr(sc), 0, int(n_anchors), float(weight), _ptr(ids), _ptr(maps))
        if k < 0:
            raise RuntimeError(self.L.ka_multi_last_error().decode())
        return ids[:k], maps[:int(lens.sum()) * k]

    def tree_run(self, codes, tasks, subm, scal, seq_distances, n_anchors=0, weight=0.0, keep_consistency=False):
        flat, off, lens, tasks, sd, sub, sc = self._args(codes, tasks, subm, scal, seq_distances)
        self._chk(self.L.ka_multi_tree_run(self.h, len(lens), _ptr(flat), _ptr(off), _ptr(lens), _ptr(sd), len(tasks), _ptr(tasks),
                                           _ptr(sub), _ptr(sc), FLAG_KEEP_CONSISTENCY if keep_consistency else 0, int(n_anchors), float(weight)))

    def download(self):
        lens, n = self._job["lens"], self._job["ntasks"]
        recs = (TaskRec * n)()
        cap = int(self.L.ka_multi_paths_size(self.h))
        paths = np.zeros(max(cap, 1), np.int32)
        gaps = np.zeros(int(lens.sum()) + len(lens), np.int32)
        self._chk(self.L.ka_multi_download(self.h, len(lens), _ptr(lens), n, recs, _ptr(paths), cap, _ptr(gaps)))
        out, g = [], 0
        for ln in lens:
            out.append(gaps[g:g + int(ln) + 1].copy())
            g += int(ln) + 1
        return list(recs), paths, out

    def runs(self):
        return int(self.L.ka_multi_runs(self.h))

    def ctx(self, rank=0):
        """rank's single-GPU context as a borrowed Context (ka_multi_ctx)"""
        h = self.L.ka_multi_ctx(self.h, int(rank))
        if not h:
            raise RuntimeError("ka_multi_ctx: bad rank")
        flat, off, lens, tasks, sd, sub, sc = self._job["keep"]
        return Context.borrowed(h, job=dict(lens=lens, ntasks=len(tasks), n=len(lens)))

    def adopt(self, recs, gaps):
        """ka_multi_adopt: rank 0's context takes the alignment of the last sharded run over"""
        arr = (TaskRec * len(recs))(*recs)
        flat = np.ascontiguousarray(np.concatenate([np.asarray(g, np.int32) for g in gaps]), np.int32)
        self._chk(self.L.ka_multi_adopt(self.h, arr, _ptr(flat)))

    def close(self):
        if self.h:
            self.L.ka_multi_destroy(self.h)
            self.h = C.c_void_p()
