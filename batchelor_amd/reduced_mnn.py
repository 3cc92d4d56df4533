"""reducedMNN() / the `.fast_mnn` cut, served by the MI355X engine.

Mirrors R/reducedMNN.R:61-95 and the argument handling of R/fastMNN.R:398-429 (names, merge.order, restrict), the
batch splitting of R/divideIntoBatches.R:36-84 and the re-ordering of R/utils_reorder.R.  Conventions follow R:
cells x dims matrices, 1-based batch ids / cell indices in the result, `None` for NULL / NA.

All arithmetic happens in HIP kernels behind include/batchelor_mi355x.h; there is no CPU path here.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Any, List, Optional, Sequence

import numpy as np

from . import _lib
from .merge_tree import encode_postorder, resolve_merge_order


class BmxParams(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_int32), ("k", ctypes.c_int32), ("prop_k", ctypes.c_double), ("ndist", ctypes.c_double),
                ("min_batch_skip", ctypes.c_double), ("auto_merge", ctypes.c_int32), ("var_adj", ctypes.c_int32),
                ("sigma", ctypes.c_double)]


ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64)


@dataclass
class MergeInfo:
    """metadata(output)$merge.info (R/fastMNN.R:551-560)."""
    left: List[List[Any]]
    right: List[List[Any]]
    pairs: List[Any]          # per merge: (left, right) 1-based output-row indices
    batch_size: np.ndarray
    skipped: np.ndarray
    lost_var: np.ndarray      # (B-1) x B, columns in input batch order


@dataclass
class MnnResult:
    corrected: np.ndarray     # N x d, rows in input order
    batch: np.ndarray         # batch id (1-based) or name per row
    merge_info: MergeInfo
    stats: Optional[list] = None   # per merge {nL, nR, U, P, nL_all, nR_all}: sizes behind the flop / byte counts


class MnnEngine:
    """A device-resident engine (bmx_engine_t): upload once, run many times, download when needed."""

    def __init__(self, device: int = 0):
        _lib.require_gpu()
        self._h = ctypes.c_void_p()
        _lib.check(_lib.lib().bmx_engine_create(int(device), ctypes.byref(self._h)))
        self._keep = None
        self._cb = None
        self.nbatches = 0
        self.nrows: List[int] = []
        self.d = 0

    def close(self):
        if self._h:
            _lib.lib().bmx_engine_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_shard(self, rank, world, allgather):
        """allgather(buf_ptr: int, bytes_per_rank: int) -> None; in-place all-gather of a device buffer."""
        def _tramp(_ctx, buf, nbytes):
            try:
                allgather(int(buf), int(nbytes))
                return 0
            except Exception as exc:  # never unwind into C
                self._cb_error = exc
                return 1
        self._cb_error = None
        self._cb = ALLGATHER_FN(_tramp) if world > 1 else ctypes.cast(None, ALLGATHER_FN)
        _lib.check(_lib.lib().bmx_engine_set_shard(self._h, int(rank), int(world), self._cb, None))

    def upload(self, batches: Sequence[np.ndarray], restrict=None):
        mats = [_lib.as_f(b) for b in batches]
        if len(mats) < 2:
            raise ValueError("at least two batches must be specified")  # R/fastMNN.R:345
        d = mats[0].shape[1]
        for m in mats:
            if m.ndim != 2 or m.shape[1] != d:
                raise ValueError("number of columns is not the same across batches")  # R/checkInputs.R:64-71
        B = len(mats)
        data = (ctypes.c_void_p * B)(*[m.ctypes.data for m in mats])
        nrows = np.asarray([m.shape[0] for m in mats], dtype=np.int32)
        rlist, rptr, rn = [], (ctypes.c_void_p * B)(), np.full(B, -1, dtype=np.int32)
        if restrict is not None:
            if len(restrict) != B:
                raise ValueError("'restrictions' must of length equal to the number of batches")
            for b, r in enumerate(restrict):
                if r is None:
                    rlist.append(None)
                    continue
                r = np.asarray(r)
                r = (np.flatnonzero(r) + 1) if r.dtype == bool else r
                r = np.ascontiguousarray(r, dtype=np.int32)
                if r.size == 0:
                    raise ValueError("no cells remaining in a batch after restriction")
                rlist.append(r)
                rptr[b] = r.ctypes.data
                rn[b] = r.size
        _lib.check(_lib.lib().bmx_engine_upload(self._h, B, d, data, _lib.i32p(nrows),
                                                rptr if restrict is not None else None, _lib.i32p(rn)))
        self._keep = (mats, rlist)
        self.nbatches, self.nrows, self.d = B, nrows.tolist(), d

    def run(self, k=20, prop_k=None, ndist=3.0, min_batch_skip=0.0, merge_tree=None, auto_merge=False, var_adj=False,
            sigma=0.1):
        p = BmxParams(ctypes.sizeof(BmxParams), int(k), float("nan") if prop_k is None else float(prop_k), float(ndist),
                      float("nan") if min_batch_skip is None else float(min_batch_skip), 1 if auto_merge else 0,
                      1 if var_adj else 0, float(sigma))
        code = encode_postorder(merge_tree if merge_tree is not None
                                else resolve_merge_order(self.nbatches))
        rc = _lib.lib().bmx_engine_run(self._h, ctypes.byref(p), _lib.i32p(code), int(code.size))
        if rc != 0 and getattr(self, "_cb_error", None) is not None:
            err, self._cb_error = self._cb_error, None
            raise err
        _lib.check(rc)

    def emulate(self, mode, rank=0, world=1):
        """Measurement hook (bmx_engine_emulate): mode 1 records a single-rank run's exchanges, mode 2 makes the engine
        rank `rank` of `world` with the other ranks' results replayed from the recording, mode 0 switches it off."""
        _lib.check(_lib.lib().bmx_engine_emulate(self._h, int(mode), int(rank), int(world)))

    def exchange_stats(self):
        calls, nbytes = ctypes.c_int64(0), ctypes.c_int64(0)
        _lib.check(_lib.lib().bmx_engine_exchange_stats(self._h, ctypes.byref(calls), ctypes.byref(nbytes)))
        return {"calls": calls.value, "bytes": nbytes.value}

    def set_watchdog(self, base_ms):
        """Deadline base of every wait of a run in milliseconds (<= 0: off); see bmx_engine_set_watchdog."""
        _lib.check(_lib.lib().bmx_engine_set_watchdog(self._h, ctypes.c_double(float(base_ms))))

    def _debug_stall(self, ms):
        """Testing hook: keep the engine's stream busy for `ms` milliseconds."""
        _lib.check(_lib.lib().bmx_engine_debug_stall(self._h, int(ms)))

    def set_profiling(self, on=True):
        _lib.check(_lib.lib().bmx_engine_set_profiling(self._h, 1 if on else 0))

    def profile(self):
        ms, n, fb = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_int64(0)
        _lib.check(_lib.lib().bmx_engine_profile(self._h, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fb)))
        return {"topk_ms": ms.value, "topk_launches": n.value, "exact_fallbacks": fb.value,
                "variant": int(_lib.lib().bmx_engine_knn_variant(self._h))}

    def set_snapshot(self, merge):
        """Diagnostics: keep the two matrices merge `merge` (0-based) searches; None switches it off."""
        _lib.check(_lib.lib().bmx_engine_set_snapshot(self._h, -1 if merge is None else int(merge)))

    def snapshot(self):
        """(left, right) of the snapshot merge, row-major [n x d], as handed to findMutualNN."""
        nl, nr = ctypes.c_int64(0), ctypes.c_int64(0)
        _lib.check(_lib.lib().bmx_engine_snapshot(self._h, None, None, ctypes.byref(nl), ctypes.byref(nr)))
        left = np.zeros((nl.value, self.d), dtype=np.float64)
        right = np.zeros((nr.value, self.d), dtype=np.float64)
        _lib.check(_lib.lib().bmx_engine_snapshot(self._h, _lib.f64p(left), _lib.f64p(right), None, None))
        return left, right

    def snapshot_var_adj(self):
        """What the snapshot merge's adjust_shift_variance was handed and returned (var_adj runs): dict of left, right,
        correction (row-major [n x d]), scaling [n_right] (before pmax(., 1)), restrict1, restrict2 (0-based)."""
        sz = (ctypes.c_int64 * 4)()
        f = _lib.lib().bmx_engine_snapshot_var_adj
        _lib.check(f(self._h, None, None, None, None, None, None, sz))
        nl, nr, n1, n2 = (int(x) for x in sz)
        out = {"left": np.zeros((nl, self.d)), "right": np.zeros((nr, self.d)), "correction": np.zeros((nr, self.d)),
               "scaling": np.zeros(nr), "restrict1": np.zeros(n1, dtype=np.int32), "restrict2": np.zeros(n2, dtype=np.int32)}
        _lib.check(f(self._h, _lib.f64p(out["left"]), _lib.f64p(out["right"]), _lib.f64p(out["correction"]),
                     _lib.f64p(out["scaling"]), _lib.i32p(out["restrict1"]), _lib.i32p(out["restrict2"]), None))
        return out

    def var_adj_tally(self):
        """Per merge (runs made with the testing hook "asv_modes" on): cells the tiled adjust_shift_variance re-ran in the
        reference's order of operations, flagged beyond the re-run, handled in all (-1: not recorded)."""
        out = []
        for m in range(self.nbatches - 1):
            a = np.zeros(3, dtype=np.int64)
            _lib.check(_lib.lib().bmx_engine_var_adj_tally(self._h, m, a.ctypes.data_as(_lib.c_i64p)))
            out.append(dict(zip(("rerun", "beyond", "tiled"), a.tolist())))
        return out

    def snapshot_var_adj_modes(self, n):
        """The way every right cell of the snapshot merge's tiled adjust_shift_variance went (0 / 1 / 2; 255: not recorded)."""
        a = np.zeros(int(n), dtype=np.uint8)
        _lib.check(_lib.lib().bmx_engine_snapshot_var_adj_modes(self._h, a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                                                              ctypes.c_int64(int(n))))
        return a

    def profile_detail(self):
        a = np.zeros(10, dtype=np.float64)
        _lib.check(_lib.lib().bmx_engine_profile_detail(self._h, _lib.f64p(a)))
        buf = ctypes.create_string_buffer(128)
        _lib.check(_lib.lib().bmx_engine_knn_kernel(self._h, buf, 128))
        return {"f16_ms": a[0], "f16_launches": int(a[1]), "bf16_ms": a[2], "bf16_launches": int(a[3]),
                "sample_ms": a[4], "sample_launches": int(a[5]), "streaming_ms": a[6],
                "exact_fallbacks": int(a[7]), "tier2_queries": int(a[8]), "optimistic_retries": int(a[9]),
                "kernel": buf.value.decode()}

    def profile_var_adj(self):
        a = np.zeros(3, dtype=np.float64)
        _lib.check(_lib.lib().bmx_engine_profile_var_adj(self._h, _lib.f64p(a)))
        return {"asv_ms": a[0], "asv_launches": int(a[1]), "asv_pairs": a[2]}

    def merge_stats(self):
        out = []
        for m in range(self.nbatches - 1):
            a = np.zeros(6, dtype=np.int64)
            _lib.check(_lib.lib().bmx_engine_merge_stats(self._h, m, a.ctypes.data_as(_lib.c_i64p)))
            out.append(dict(zip(("nL", "nR", "U", "P", "nL_all", "nR_all"), a.tolist())))
        return out

    def _pairs(self):
        """Every merge's pair list: size queries, the caller's arrays (R's integer vectors), one filling call."""
        L, nm = _lib.lib(), self.nbatches - 1
        pairs, cap = [], np.zeros(nm, dtype=np.int64)
        for m in range(nm):
            n = ctypes.c_int64(0)
            _lib.check(L.bmx_engine_pairs_into(self._h, m, None, None, ctypes.c_int64(0), ctypes.byref(n)))
            cap[m] = n.value
            pairs.append((np.empty(n.value, dtype=np.int32), np.empty(n.value, dtype=np.int32)))
        lp = (ctypes.c_void_p * nm)(*[p[0].ctypes.data for p in pairs])
        rp = (ctypes.c_void_p * nm)(*[p[1].ctypes.data for p in pairs])
        _lib.check(L.bmx_engine_pairs_all_into(self._h, nm, lp, rp, cap.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))))
        return pairs

    def download(self, with_pairs=True, c_order=True) -> MnnResult:
        """Results to the host.  The boundary writes `corrected` column-major (as R holds matrices); `c_order=False`
        returns it that way instead of converting it to numpy's row-major default."""
        B, d = self.nbatches, self.d
        N = int(sum(self.nrows))
        nm = B - 1
        corrected = np.empty((N, d), dtype=np.float64, order="F")  # (every element is written by the library)
        batch = np.empty(N, dtype=np.int32)
        ml = np.zeros((nm, B), dtype=np.int32)
        mr = np.zeros((nm, B), dtype=np.int32)
        bs = np.zeros(nm, dtype=np.float64)
        sk = np.zeros(nm, dtype=np.int32)
        lv = np.zeros((nm, B), dtype=np.float64, order="F")
        _lib.check(_lib.lib().bmx_engine_download(self._h, _lib.f64p(corrected), _lib.i32p(batch), _lib.i32p(ml),
                                                  _lib.i32p(mr), _lib.f64p(bs), _lib.i32p(sk), _lib.f64p(lv)))
        pairs = self._pairs() if with_pairs else []
        info = MergeInfo(left=[[int(x) for x in row if x] for row in ml], right=[[int(x) for x in row if x] for row in mr],
                         pairs=pairs, batch_size=bs, skipped=sk.astype(bool), lost_var=np.ascontiguousarray(lv))
        return MnnResult(corrected=np.ascontiguousarray(corrected) if c_order else corrected, batch=batch, merge_info=info,
                         stats=self.merge_stats())


def fast_mnn_one_shot(batches, restrict=None, k=20, prop_k=None, ndist=3.0, min_batch_skip=0.0, merge_tree=None,
                      auto_merge=False, var_adj=False, sigma=0.1, with_pairs=True, c_order=True) -> MnnResult:
    """bmx_fast_mnn(): the single call INTEGRATION.md's .Call shim makes -- host matrices in (R layout: cells x d,
    column-major), host results out; the library pulls the batches through its pinned staging ring while the first
    merges already run.  `merge_tree`: binary tree with 1-based integer leaves (None: 1..B progressive)."""
    _lib.require_gpu()
    mats = [_lib.as_f(b) for b in batches]
    if len(mats) < 2:
        raise ValueError("at least two batches must be specified")  # R/fastMNN.R:345
    d = mats[0].shape[1]
    for m in mats:
        if m.ndim != 2 or m.shape[1] != d:
            raise ValueError("number of columns is not the same across batches")  # R/checkInputs.R:64-71
    B = len(mats)
    data = (ctypes.c_void_p * B)(*[m.ctypes.data for m in mats])
    nrows = np.asarray([m.shape[0] for m in mats], dtype=np.int32)
    rlist, rptr, rn = [], (ctypes.c_void_p * B)(), np.full(B, -1, dtype=np.int32)
    if restrict is not None:
        if len(restrict) != B:
            raise ValueError("'restrictions' must of length equal to the number of batches")
        for b, r in enumerate(restrict):
            if r is None:
                continue
            r = np.asarray(r)
            r = (np.flatnonzero(r) + 1) if r.dtype == bool else r
            r = np.ascontiguousarray(r, dtype=np.int32)
            if r.size == 0:
                raise ValueError("no cells remaining in a batch after restriction")
            rlist.append(r)
            rptr[b] = r.ctypes.data
            rn[b] = r.size
    p = BmxParams(ctypes.sizeof(BmxParams), int(k), float("nan") if prop_k is None else float(prop_k), float(ndist),
                  float("nan") if min_batch_skip is None else float(min_batch_skip), 1 if auto_merge else 0,
                  1 if var_adj else 0, float(sigma))
    code = encode_postorder(merge_tree if merge_tree is not None else resolve_merge_order(B))
    N, nm = int(nrows.sum()), B - 1
    corrected = np.empty((N, d), dtype=np.float64, order="F")  # (every element is written by the library)
    batch = np.empty(N, dtype=np.int32)
    ml = np.zeros((nm, B), dtype=np.int32)
    mr = np.zeros((nm, B), dtype=np.int32)
    bs = np.zeros(nm, dtype=np.float64)
    sk = np.zeros(nm, dtype=np.int32)
    lv = np.zeros((nm, B), dtype=np.float64, order="F")
    eng = MnnEngine.__new__(MnnEngine)
    eng._h, eng._keep, eng._cb = ctypes.c_void_p(), (mats, rlist), None
    eng.nbatches, eng.nrows, eng.d = B, nrows.tolist(), d
    _lib.check(_lib.lib().bmx_fast_mnn(B, d, data, _lib.i32p(nrows), rptr if restrict is not None else None, _lib.i32p(rn),
                                       ctypes.byref(p), _lib.i32p(code), int(code.size), _lib.f64p(corrected),
                                       _lib.i32p(batch), _lib.i32p(ml), _lib.i32p(mr), _lib.f64p(bs), _lib.i32p(sk),
                                       _lib.f64p(lv), ctypes.byref(eng._h)))
    try:
        pairs = eng._pairs() if with_pairs else []
        stats = eng.merge_stats()
    finally:
        eng.close()
    info = MergeInfo(left=[[int(x) for x in row if x] for row in ml], right=[[int(x) for x in row if x] for row in mr],
                     pairs=pairs, batch_size=bs, skipped=sk.astype(bool), lost_var=np.ascontiguousarray(lv))
    return MnnResult(corrected=np.ascontiguousarray(corrected) if c_order else corrected, batch=batch, merge_info=info,
                     stats=stats)


def _fast_mnn(batches, k, prop_k, restrict, ndist, merge_order, auto_merge, min_batch_skip, names, device=0,
              var_adj=False, sigma=0.1):
    """.fast_mnn (R/fastMNN.R:398-429)."""
    if names is not None and len(set(names)) != len(names):
        raise ValueError("names of batches should be unique")  # R/fastMNN.R:422
    # the one call the .Call shim makes (bmx_fast_mnn: upload hidden behind the first merges), on the device asked for
    _lib.require_gpu()
    _lib.check(_lib.lib().bmx_set_device(int(device)))
    tree = None if auto_merge else resolve_merge_order(len(batches), merge_order, names)
    out = fast_mnn_one_shot(batches, restrict, k=k, prop_k=prop_k, ndist=ndist, min_batch_skip=min_batch_skip,
                            merge_tree=tree, auto_merge=auto_merge, var_adj=var_adj, sigma=sigma)
    if names is not None:  # R/fastMNN.R:419-427
        nm = np.asarray(list(names), dtype=object)
        out.batch = nm[out.batch - 1]
        out.merge_info.left = [[names[i - 1] for i in s] for s in out.merge_info.left]
        out.merge_info.right = [[names[i - 1] for i in s] for s in out.merge_info.right]
    return out


def divideIntoBatches(x, batch, restrict=None):
    """R/divideIntoBatches.R:36-84 with byrow=TRUE: levels are the sorted unique values of `batch`."""
    x = np.asarray(x, dtype=np.float64)
    batch = np.asarray(batch)
    if batch.shape[0] != x.shape[0]:
        raise ValueError("'length(batch)' and 'nrow(x)' are not the same")
    levels = sorted(set(batch.tolist()))
    mask = None
    if restrict is not None:
        r = np.asarray(restrict)
        mask = np.zeros(x.shape[0], dtype=bool)
        if r.dtype == bool:
            mask[:] = r
        else:
            mask[r.astype(np.int64) - 1] = True
    out, restricted = [], ([] if mask is not None else None)
    reorder = np.zeros(x.shape[0], dtype=np.int64)
    last = 0
    for lev in levels:
        keep = batch == lev
        cur = x[keep]
        if mask is not None:
            cr = np.flatnonzero(mask[keep]) + 1
            if cr.size == 0:
                raise ValueError("no cells remaining in a batch after restriction")
            restricted.append(cr.astype(np.int32))
        out.append(cur)
        reorder[keep] = last + np.arange(1, cur.shape[0] + 1)
        last += cur.shape[0]
    return {"batches": out, "levels": levels, "reorder": reorder, "restricted": restricted}


def _reindex_pairings(pairings, new_order):
    """R/utils_reorder.R:23-36."""
    new_order = np.asarray(new_order, dtype=np.int64)
    rev = np.zeros(new_order.size + 1, dtype=np.int64)
    rev[new_order] = np.arange(1, new_order.size + 1)
    return [(rev[l], rev[r]) for l, r in pairings]


def reducedMNN(*batches, batch=None, k=20, prop_k=None, restrict=None, ndist=3, merge_order=None, auto_merge=False,
               min_batch_skip=0.0, names=None, device=0, var_adj=False, sigma=0.1) -> MnnResult:
    """reducedMNN(..., batch=, k=, prop.k=, restrict=, ndist=, merge.order=, auto.merge=, min.batch.skip=)
    (R/reducedMNN.R:61-95).  `names` plays the role of the argument names of `...`."""
    if len(batches) == 1 and isinstance(batches[0], (list, tuple)):
        batches = tuple(batches[0])
    if len(batches) == 0:
        raise ValueError("at least two batches must be specified")
    d = np.asarray(batches[0]).shape[1]
    for b in batches:
        if np.asarray(b).ndim != 2 or np.asarray(b).shape[1] != d:
            raise ValueError("number of columns is not the same across batches")
    if restrict is not None and len(restrict) != len(batches):
        raise ValueError("'restrictions' must of length equal to the number of batches")
    if len(batches) == 1:
        if batch is None:
            raise ValueError("'batch' must be specified if '...' has only one object")  # R/checkInputs.R:128
        div = divideIntoBatches(batches[0], batch, None if restrict is None else restrict[0])
        out = _fast_mnn(div["batches"], k, prop_k, div["restricted"], ndist, merge_order, auto_merge, min_batch_skip,
                        [str(l) for l in div["levels"]], device, var_adj, sigma)
        reo = div["reorder"]
        out.corrected = out.corrected[reo - 1]
        out.batch = out.batch[reo - 1]
        out.merge_info.pairs = _reindex_pairings(out.merge_info.pairs, reo)
        return out
    return _fast_mnn([np.asarray(b, dtype=np.float64) for b in batches], k, prop_k, restrict, ndist, merge_order,
                     auto_merge, min_batch_skip, names, device, var_adj, sigma)
