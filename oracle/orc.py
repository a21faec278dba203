"""ctypes driver for the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product."""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FLAG_SCORE_ONLY = 0x01
FLAG_RIGHT = 0x02
FLAG_EXTZ_ONLY = 0x40


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".hpp"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.orc_index_load.restype = ctypes.c_void_p
        L.orc_index_load.argtypes = [ctypes.c_char_p]
        L.orc_index_create.restype = ctypes.c_void_p
        L.orc_index_create.argtypes = [ctypes.c_uint64] * 4 + [ctypes.c_void_p] * 9 + [ctypes.c_char_p]
        L.orc_index_free.argtypes = [ctypes.c_void_p]
        L.orc_index_set_lifts.argtypes = [ctypes.c_void_p, ctypes.c_uint64] + [ctypes.c_void_p] * 6
        L.orc_lift.restype = ctypes.c_uint64
        L.orc_lift.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
        L.orc_lift_cigar.restype = ctypes.c_uint64
        L.orc_lift_cigar.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64]
        L.orc_index_n.restype = ctypes.c_uint64
        L.orc_index_n.argtypes = [ctypes.c_void_p]
        L.orc_index_r.restype = ctypes.c_uint64
        L.orc_index_r.argtypes = [ctypes.c_void_p]
        L.orc_ms_query.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_void_p]
        L.orc_phi_lcp.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p]
        L.orc_seed_batch.restype = ctypes.c_void_p
        L.orc_seed_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64,
                                     ctypes.c_uint64, ctypes.c_int, ctypes.c_uint64, ctypes.c_int]
        L.orc_seed_n_mems.restype = ctypes.c_uint64
        L.orc_seed_n_mems.argtypes = [ctypes.c_void_p]
        L.orc_seed_n_occs.restype = ctypes.c_uint64
        L.orc_seed_n_occs.argtypes = [ctypes.c_void_p]
        L.orc_seed_get.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.orc_seed_free.argtypes = [ctypes.c_void_p]
        L.orc_ksw_align.restype = None
        L.orc_ksw_align.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.orc_align_pe.restype = ctypes.c_void_p
        L.orc_align_pe.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 4 + [ctypes.c_uint64] + [ctypes.c_void_p] * 6 + [ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p]
        L.orc_align_batch.restype = ctypes.c_void_p
        L.orc_align_batch.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_uint64] + [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int,
                                                                                                     ctypes.c_void_p, ctypes.c_void_p]
        L.orc_free.argtypes = [ctypes.c_void_p]
        L.orc_align_csv.restype = ctypes.c_void_p
        L.orc_align_csv.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_uint64] + [ctypes.c_void_p] * 3
        L.orc_index_set_no_lcp.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_report_mems_batch.restype = ctypes.c_void_p
        L.orc_report_mems_batch.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_uint64] + [ctypes.c_void_p] * 4
        L.orc_ms_lengths.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_extz.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int8,
                               ctypes.c_void_p, ctypes.c_int8, ctypes.c_int8, ctypes.c_int, ctypes.c_int,
                               ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        _LIB = L
    return _LIB


class OracleIndex:
    def __init__(self, path: str = None, fi=None):
        self._L = lib()
        if fi is not None:
            names = b"".join(s.encode() + b"\0" for s in fi.names)
            self._h = self._L.orc_index_create(fi.n, fi.r, fi.w, len(fi.seq_starts) - 1, fi.F.ctypes.data, fi.heads.ctypes.data,
                                               fi.starts.ctypes.data, fi.ssa.ctypes.data, fi.esa.ctypes.data, fi.thr.ctypes.data,
                                               fi.slcp.ctypes.data, fi.text.ctypes.data, fi.seq_starts.ctypes.data, names)
            lf = getattr(fi, "lifts", None)
            if self._h and lf is not None:
                self._L.orc_index_set_lifts(self._h, len(fi.seq_starts) - 1, lf.second.ctypes.data, lf.len.ctypes.data, lf.ins_off.ctypes.data,
                                            lf.ins.ctypes.data, lf.del_off.ctypes.data, lf.dele.ctypes.data)
        else:
            self._h = self._L.orc_index_load(path.encode())
        if not self._h:
            raise IOError("oracle: cannot load " + str(path))
        self.n = self._L.orc_index_n(self._h)
        self.r = self._L.orc_index_r(self._h)

    def set_no_lcp(self, on: bool = True):
        """the `-n` form: the occurrence walks use Phi / Phi_inv and a bounded LCE on the text instead of the sampled LCP"""
        self._L.orc_index_set_no_lcp(self._h, int(on))

    def close(self):
        if self._h:
            self._L.orc_index_free(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def lift(self, pos: int) -> int:
        return int(self._L.orc_lift(self._h, pos))

    def lift_cigar(self, cigar: np.ndarray, pos: int) -> np.ndarray:
        cigar = np.ascontiguousarray(cigar, dtype=np.uint32)
        cap = int((cigar >> 4).sum()) * 2 + 16
        out = np.zeros(cap, dtype=np.uint32)
        n = self._L.orc_lift_cigar(self._h, cigar.ctypes.data, len(cigar), pos, out.ctypes.data, cap)
        return out[:n].copy()

    def ms_lengths(self, pattern: bytes):
        """legacy `moni ms`: (pointers, lengths) of a pattern"""
        ptr = np.empty(len(pattern), dtype=np.uint64)
        ln = np.empty(len(pattern), dtype=np.uint64)
        self._L.orc_ms_lengths(self._h, pattern, len(pattern), ptr.ctypes.data, ln.ctypes.data)
        return ptr, ln

    def ms_query(self, pattern: bytes) -> np.ndarray:
        out = np.empty(len(pattern), dtype=np.uint64)
        self._L.orc_ms_query(self._h, pattern, len(pattern), out.ctypes.data)
        return out

    def phi_lcp(self, i: int, inverse: bool = False):
        out = np.empty(2, dtype=np.uint64)
        self._L.orc_phi_lcp(self._h, i, int(inverse), out.ctypes.data)
        return int(out[0]), int(out[1])

    def seed_batch(self, seqs: np.ndarray, offsets: np.ndarray, min_len: int = 25, filter_seeds: bool = True,
                   n_seeds_thr: int = 1000, threads: int = 1) -> Dict[str, np.ndarray]:
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n_reads = len(offsets) - 1
        r = self._L.orc_seed_batch(self._h, seqs.ctypes.data, offsets.ctypes.data, n_reads, min_len,
                                   int(filter_seeds), n_seeds_thr, threads)
        try:
            nm = self._L.orc_seed_n_mems(r)
            no = self._L.orc_seed_n_occs(r)
            names = ["read", "pos", "len", "idx", "mate", "rpos", "total_occ", "num_filtered", "occ_off", "occ_cnt"]
            out = {}
            for f, nme in enumerate(names):
                a = np.empty(nm, dtype=np.uint64)
                self._L.orc_seed_get(r, f, a.ctypes.data)
                out[nme] = a
            a = np.empty(no, dtype=np.uint64)
            self._L.orc_seed_get(r, 10, a.ctypes.data)
            out["occs"] = a
            a = np.empty(n_reads + 1, dtype=np.uint64)
            self._L.orc_seed_get(r, 11, a.ctypes.data)
            out["read_mem_off"] = a
            a = np.empty(4, dtype=np.uint64)
            self._L.orc_seed_get(r, 12, a.ctypes.data)
            out["counters"] = a      # lf_steps, jumps, phi_steps, text_cmp
            return out
        finally:
            self._L.orc_seed_free(r)


def make_names(n: int, prefix: str = "simulated"):
    names = [("%s.%d" % (prefix, i)).encode() for i in range(n)]
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in names])
    return np.frombuffer(b"".join(names), dtype=np.uint8).copy(), off


def align_batch(oidx: "OracleIndex", seqs: np.ndarray, offsets: np.ndarray, names=None, name_off=None, quals=None,
                with_header: bool = False, threads: int = 1):
    """SAM text (bytes) of the reference's single-end path, plus work counters."""
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(offsets) - 1
    if names is None:
        names, name_off = make_names(n)
    names = np.ascontiguousarray(names, dtype=np.uint8)
    name_off = np.ascontiguousarray(name_off, dtype=np.uint64)
    if quals is not None:
        quals = np.ascontiguousarray(quals, dtype=np.uint8)
    out_len = ctypes.c_uint64()
    cnt = np.zeros(8, dtype=np.uint64)
    p = lib().orc_align_batch(oidx._h, seqs.ctypes.data, offsets.ctypes.data, n, names.ctypes.data, name_off.ctypes.data,
                              quals.ctypes.data if quals is not None else None, int(with_header), threads,
                              ctypes.byref(out_len), cnt.ctypes.data)
    try:
        sam = ctypes.string_at(p, out_len.value)
    finally:
        lib().orc_free(p)
    keys = ["lf_steps", "jumps", "phi_steps", "text_cmp", "dp_cells", "dp_calls", "ref_bytes", "aligned"]
    return sam, {k: int(v) for k, v in zip(keys, cnt)}


def align_csv(oidx: "OracleIndex", seqs: np.ndarray, offsets: np.ndarray, names, name_off) -> bytes:
    """the `-c` MEM statistics of the batch (one CSV line per read, no header), one thread"""
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    names = np.ascontiguousarray(names, dtype=np.uint8)
    name_off = np.ascontiguousarray(name_off, dtype=np.uint64)
    out_len = ctypes.c_uint64()
    p = lib().orc_align_csv(oidx._h, seqs.ctypes.data, offsets.ctypes.data, len(offsets) - 1, names.ctypes.data, name_off.ctypes.data, ctypes.byref(out_len))
    try:
        return ctypes.string_at(p, out_len.value)
    finally:
        lib().orc_free(p)


def align_pe(oidx: "OracleIndex", seqs1, offs1, seqs2, offs2, names1, noff1, names2, noff2, quals1=None, quals2=None, b_size: int = 512,
             find_orphan: bool = False, report_mems: bool = False, filter_dir: bool = True, secondary_chains: bool = False, csv: bool = False):
    """SAM text (bytes) of the reference's paired-end path without orphan recovery (oracle/align_pe.hpp), one thread, st_align's
    batch order, plus {"aligned", "ins_count", "ins_mean", "ins_std_dev", "ins_complete"}.  csv: the pairs' `-c` lines instead of the SAM text."""
    c = lambda a, t: np.ascontiguousarray(a, dtype=t)
    seqs1, seqs2, names1, names2 = c(seqs1, np.uint8), c(seqs2, np.uint8), c(names1, np.uint8), c(names2, np.uint8)
    offs1, offs2, noff1, noff2 = c(offs1, np.uint64), c(offs2, np.uint64), c(noff1, np.uint64), c(noff2, np.uint64)
    if quals1 is not None:
        quals1, quals2 = c(quals1, np.uint8), c(quals2, np.uint8)
    n = len(offs1) - 1
    out_len = ctypes.c_uint64()
    st = np.zeros(7, dtype=np.float64)
    p = lib().orc_align_pe(oidx._h, seqs1.ctypes.data, offs1.ctypes.data, seqs2.ctypes.data, offs2.ctypes.data, n, names1.ctypes.data, noff1.ctypes.data,
                           names2.ctypes.data, noff2.ctypes.data, quals1.ctypes.data if quals1 is not None else None,
                           quals2.ctypes.data if quals2 is not None else None, b_size, int(find_orphan) | (2 if report_mems else 0) | (0 if filter_dir else 4) | (8 if secondary_chains else 0) | (16 if csv else 0), ctypes.byref(out_len), st.ctypes.data)
    try:
        sam = ctypes.string_at(p, out_len.value)
    finally:
        lib().orc_free(p)
    return sam, {"aligned": int(st[0]), "ins_count": int(st[1]), "ins_mean": float(st[2]), "ins_std_dev": float(st[3]), "ins_complete": bool(st[4]),
                 "orphan_pairs": int(st[5]), "orphan_recovered": int(st[6])}


DEFAULT_MAT = np.array([2, -4, -4, -4, 0, -4, 2, -4, -4, 0, -4, -4, 2, -4, 0, -4, -4, -4, 2, 0, 0, 0, 0, 0, 0],
                       dtype=np.int8)   # ksw_gen_simple_mat(5, mat, 2, -4), aligner_ksw2.hpp:3199-3211


def extz(query: np.ndarray, target: np.ndarray, flag: int, m: int = 5, mat: np.ndarray = DEFAULT_MAT, q: int = 4,
         e: int = 2, w: int = -1, zdrop: int = -1, end_bonus: int = 400):
    query = np.ascontiguousarray(query, dtype=np.uint8)
    target = np.ascontiguousarray(target, dtype=np.uint8)
    out = np.zeros(11, dtype=np.int32)
    cap = len(query) + len(target) + 2
    cig = np.zeros(cap, dtype=np.uint32)
    lib().orc_extz(len(query), query.ctypes.data, len(target), target.ctypes.data, m, mat.ctypes.data, q, e, w, zdrop,
                   end_bonus, flag, out.ctypes.data, cig.ctypes.data, cap)
    keys = ["max", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q", "score", "reach_end", "n_cigar", "zdropped"]
    res = {k: int(v) for k, v in zip(keys, out)}
    res["cigar"] = cig[: res["n_cigar"]].copy()
    return res


def ksw_align(query: np.ndarray, target: np.ndarray, mat: np.ndarray = None, gapo: int = 4, gape: int = 2):
    """klib's ksw_align(.., KSW_XSTART) as restated in align_pe.hpp (nt4 codes): dict score / te / qe / tb / qb."""
    query = np.ascontiguousarray(query, dtype=np.uint8)
    target = np.ascontiguousarray(target, dtype=np.uint8)
    mat = DEFAULT_MAT if mat is None else np.ascontiguousarray(mat, dtype=np.int8)
    out = np.zeros(5, dtype=np.int32)
    lib().orc_ksw_align(query.ctypes.data, len(query), target.ctypes.data, len(target), mat.ctypes.data, gapo, gape, out.ctypes.data)
    return dict(zip(["score", "te", "qe", "tb", "qb"], (int(x) for x in out)))


def report_mems_batch(oidx: "OracleIndex", seqs, offsets, names, name_off, quals=None) -> bytes:
    """aligner::align with report_mems (-m): the MEM records of the batch"""
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    names = np.ascontiguousarray(names, dtype=np.uint8)
    name_off = np.ascontiguousarray(name_off, dtype=np.uint64)
    if quals is not None:
        quals = np.ascontiguousarray(quals, dtype=np.uint8)
    out_len = ctypes.c_uint64()
    p = lib().orc_report_mems_batch(oidx._h, seqs.ctypes.data, offsets.ctypes.data, len(offsets) - 1, names.ctypes.data, name_off.ctypes.data,
                                    quals.ctypes.data if quals is not None else None, ctypes.byref(out_len))
    try:
        return ctypes.string_at(p, out_len.value)
    finally:
        lib().orc_free(p)


def legacy_mems(pointers: np.ndarray, lengths: np.ndarray):
    """src/mems.cpp:241-262: the (offset, length) pairs `moni mems` reports from ms pointers / lengths"""
    out = []
    for i in range(len(lengths)):
        if i == 0 or lengths[i] >= lengths[i - 1]:
            out.append((i, int(lengths[i])))
    return out
