"""ctypes binding of libmoni_hip.so (include/moni_hip.h).

Python is only the test/bench driver here; the product is the shared library.  There is no
Python or CPU implementation behind these calls: if the library is missing or no HIP device is
present the calls raise."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np

# streams beyond the runtime's hardware queues share one (the paired path keeps ~11 busy): a default for processes that set none, before HIP initialises
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.environ.get("MONI_HIP_LIB") or os.path.join(CSRC, "libmoni_hip.so")      # override: kernel-variant sweeps

EXPORTS = [
    "moni_version", "moni_index_create", "moni_index_load", "moni_index_destroy", "moni_index_n", "moni_index_r",
    "moni_index_device_bytes", "moni_index_text", "moni_ctx_create", "moni_ctx_destroy", "moni_reads_upload", "moni_reads_swap", "moni_ms_run",
    "moni_ms_query_batch", "moni_seed_run", "moni_seed_counts", "moni_seed_fetch", "moni_seed_batch", "moni_free",
    "moni_phi_lcp_batch", "moni_extz_batch", "moni_last_kernel_ms", "moni_last_counters",
    "moni_align_params_default", "moni_align_batch", "moni_align_csv_batch", "moni_align_run", "moni_align_stream", "moni_sam_header",
    "moni_ldx_info", "moni_ldx_rewrite", "moni_ldx_lift_batch", "moni_ldx_write",
    "moni_ms_file_info", "moni_ms_file_read", "moni_ms_file_write", "moni_index_load_reference", "moni_ms_lengths_batch", "moni_report_mems_batch",
    "moni_pe_params_default", "moni_pe_learn_batch", "moni_pe_align_batch", "moni_pe_align_stream", "moni_pe_align_run", "moni_pe_align_csv_batch", "moni_pe_report_mems_batch",
]


class FlatIndexC(C.Structure):
    _fields_ = [("n", C.c_uint64), ("r", C.c_uint64), ("w", C.c_uint64), ("n_seq", C.c_uint64),
                ("F", C.c_void_p), ("heads", C.c_void_p), ("starts", C.c_void_p), ("ssa", C.c_void_p),
                ("esa", C.c_void_p), ("thr", C.c_void_p), ("slcp", C.c_void_p), ("text", C.c_void_p),
                ("seq_starts", C.c_void_p), ("seq_names", C.c_char_p),
                ("lift_second", C.c_void_p), ("lift_len", C.c_void_p), ("lift_ins_off", C.c_void_p), ("lift_ins", C.c_void_p),
                ("lift_del_off", C.c_void_p), ("lift_del", C.c_void_p)]


class ReadBatchC(C.Structure):
    _fields_ = [("seq", C.c_void_p), ("offsets", C.c_void_p), ("n_reads", C.c_uint64)]


class SeedParamsC(C.Structure):
    _fields_ = [("min_len", C.c_uint32), ("filter_seeds", C.c_uint32), ("n_seeds_thr", C.c_uint32),
                ("report_mems", C.c_uint32)]


class AlignParamsC(C.Structure):
    _fields_ = [("min_len", C.c_uint32), ("ext_len", C.c_uint32), ("check_k", C.c_uint32), ("region_dist", C.c_uint32),
                ("filter_seeds", C.c_uint32), ("n_seeds_thr", C.c_uint32), ("filter_freq", C.c_uint32), ("left_mem_check", C.c_uint32),
                ("freq_thr", C.c_double),
                ("smatch", C.c_int8), ("smismatch", C.c_int8), ("gapo", C.c_int8), ("gapo2", C.c_int8), ("gape", C.c_int8), ("gape2", C.c_int8),
                ("end_bonus", C.c_int32), ("w", C.c_int32), ("zdrop", C.c_int32),
                ("max_dist_x", C.c_int64), ("max_dist_y", C.c_int64), ("max_iter", C.c_int64), ("max_pred", C.c_int64),
                ("min_chain_score", C.c_int64), ("min_chain_length", C.c_int64),
                ("host_threads", C.c_uint32), ("reserved", C.c_uint32)]


class AlignStatsC(C.Structure):
    _fields_ = [("reads", C.c_uint64), ("aligned", C.c_uint64), ("dp_tasks", C.c_uint64), ("dp_cells", C.c_uint64), ("dp_rounds", C.c_uint64),
                ("t_seed", C.c_double), ("t_chain", C.c_double), ("t_dp", C.c_double), ("t_host", C.c_double),
                ("t_dp_kernel", C.c_double), ("handed_back", C.c_uint64), ("dp_reused", C.c_uint64), ("dp_cells_reused", C.c_uint64),
                ("kernel_fallback", C.c_uint64), ("dp_ref_bytes", C.c_uint64),
                ("t_k_chain", C.c_double), ("t_k_dp", C.c_double), ("t_k_select", C.c_double), ("t_k_finish", C.c_double),
                ("handover_why", C.c_uint64 * 12), ("dp_cells_cut", C.c_uint64), ("dp_slots", C.c_uint64)]


class PeParamsC(C.Structure):
    _fields_ = [("filter_dir", C.c_uint32), ("find_orphan", C.c_uint32), ("dir_thr", C.c_double), ("ins_learning_n", C.c_uint64),
                ("ins_learning_score_gap_threshold", C.c_uint64), ("secondary_chains", C.c_uint32), ("reserved", C.c_uint32)]


class PeModelC(C.Structure):
    _fields_ = [("mean", C.c_double), ("std_dev", C.c_double), ("variance", C.c_double), ("sample_variance", C.c_double), ("m2", C.c_double),
                ("count", C.c_uint64), ("complete", C.c_uint32), ("reserved", C.c_uint32)]


class DpParamsC(C.Structure):
    _fields_ = [("m", C.c_int8), ("mat", C.c_int8 * 25), ("q", C.c_int8), ("e", C.c_int8),
                ("w", C.c_int32), ("zdrop", C.c_int32), ("end_bonus", C.c_int32)]


MEM_DTYPE = np.dtype([("pos", "<u8"), ("len", "<u4"), ("idx", "<u4"), ("rpos", "<u4"), ("mate", "<u4"),
                      ("total_occ", "<u4"), ("num_filtered", "<u4"), ("occ_off", "<u8"), ("occ_cnt", "<u4"),
                      ("read", "<u4")])
DP_TASK_DTYPE = np.dtype([("q_off", "<u8"), ("t_off", "<u8"), ("qlen", "<i4"), ("tlen", "<i4"), ("flag", "<i4"),
                          ("reserved", "<i4")])
DP_RESULT_DTYPE = np.dtype([("max", "<i4"), ("max_q", "<i4"), ("max_t", "<i4"), ("mqe", "<i4"), ("mqe_t", "<i4"),
                            ("mte", "<i4"), ("mte_q", "<i4"), ("score", "<i4"), ("reach_end", "<i4"),
                            ("zdropped", "<i4"), ("n_cigar", "<u4"), ("cigar_off", "<u4")])
assert MEM_DTYPE.itemsize == 48 and DP_TASK_DTYPE.itemsize == 32 and DP_RESULT_DTYPE.itemsize == 48

DEFAULT_MAT = [2, -4, -4, -4, 0, -4, 2, -4, -4, 0, -4, -4, 2, -4, 0, -4, -4, -4, 2, 0, 0, 0, 0, 0, 0]

_lib = None


def build_lib(force: bool = False) -> str:
    """hipcc cross-compiles for gfx950 without a GPU."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".hpp", ".inc", ".cpp"))]
    srcs.append(os.path.join(os.path.dirname(HERE), "include", "moni_hip.h"))
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        # pe_big.cpp: host-only translation unit (the paired state machine with large capacities, its own namespace)
        big_o = os.path.join(CSRC, "pe_big.o")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-c", os.path.join(CSRC, "pe_big.cpp"), "-o", big_o])
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               "-o", LIB_PATH, os.path.join(CSRC, "moni_hip.hip"), "-Wl," + big_o])      # (as a linker input: hipcc would read a bare .o as HIP source)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libmoni_hip.so is not built (run __graft_entry__.build()); there is no fallback path")
        L = C.CDLL(LIB_PATH)
        L.moni_version.restype = C.c_char_p
        L.moni_index_n.restype = C.c_uint64
        L.moni_index_r.restype = C.c_uint64
        L.moni_index_device_bytes.restype = C.c_uint64
        for name in ("moni_index_n", "moni_index_r", "moni_index_device_bytes", "moni_index_destroy",
                     "moni_ctx_destroy", "moni_free"):
            getattr(L, name).argtypes = [C.c_void_p]
        L.moni_index_text.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.moni_index_create.argtypes = [C.POINTER(FlatIndexC), C.c_int, C.POINTER(C.c_void_p)]
        L.moni_index_load.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
        L.moni_ctx_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.moni_reads_upload.argtypes = [C.c_void_p, C.POINTER(ReadBatchC)]
        L.moni_reads_swap.argtypes = [C.c_void_p, C.c_uint32]
        L.moni_ms_run.argtypes = [C.c_void_p]
        L.moni_ms_query_batch.argtypes = [C.c_void_p, C.POINTER(ReadBatchC), C.c_void_p]
        L.moni_seed_run.argtypes = [C.c_void_p, C.POINTER(SeedParamsC)]
        L.moni_seed_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.moni_seed_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.moni_phi_lcp_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]
        L.moni_extz_batch.argtypes = [C.c_void_p, C.POINTER(DpParamsC), C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                      C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.moni_align_params_default.argtypes = [C.POINTER(AlignParamsC)]
        L.moni_align_params_default.restype = None
        L.moni_align_batch.argtypes = [C.c_void_p, C.POINTER(ReadBatchC), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(AlignParamsC),
                                       C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(AlignStatsC)]
        L.moni_align_stream.argtypes = L.moni_align_batch.argtypes
        L.moni_align_csv_batch.argtypes = [C.c_void_p, C.POINTER(ReadBatchC), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(AlignParamsC),
                                           C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(AlignStatsC)]
        L.moni_pe_params_default.argtypes = [C.POINTER(PeParamsC)]
        L.moni_pe_params_default.restype = None
        L.moni_pe_learn_batch.argtypes = [C.c_void_p, C.POINTER(ReadBatchC), C.POINTER(AlignParamsC), C.POINTER(PeParamsC), C.POINTER(PeModelC)]
        L.moni_pe_align_batch.argtypes = [C.c_void_p, C.POINTER(ReadBatchC), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(AlignParamsC),
                                          C.POINTER(PeParamsC), C.POINTER(PeModelC), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(AlignStatsC)]
        L.moni_pe_align_stream.argtypes = L.moni_pe_align_batch.argtypes
        L.moni_pe_align_csv_batch.argtypes = L.moni_pe_align_batch.argtypes[:-1] + [C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(AlignStatsC)]
        L.moni_pe_align_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(AlignParamsC), C.POINTER(PeParamsC), C.POINTER(PeModelC),
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(AlignStatsC)]
        L.moni_pe_report_mems_batch.argtypes = [C.c_void_p, C.POINTER(ReadBatchC), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(AlignParamsC), C.POINTER(PeParamsC),
                                                C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.moni_align_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(AlignParamsC),
                                     C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(AlignStatsC)]
        L.moni_sam_header.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.moni_last_kernel_ms.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
        L.moni_last_counters.argtypes = [C.c_void_p, C.c_void_p]
        L.moni_ms_lengths_batch.argtypes = [C.c_void_p, C.POINTER(ReadBatchC), C.c_void_p, C.c_void_p]
        L.moni_report_mems_batch.argtypes = [C.c_void_p, C.POINTER(ReadBatchC), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(AlignParamsC),
                                             C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.moni_ldx_info.argtypes = [C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
        L.moni_ldx_rewrite.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.moni_ldx_write.argtypes = [C.POINTER(FlatIndexC), C.c_char_p, C.c_int]
        L.moni_ms_file_info.argtypes = [C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.moni_ms_file_read.argtypes = [C.c_char_p, C.c_uint64] + [C.c_void_p] * 7 + [C.c_char_p, C.c_uint64]
        L.moni_ms_file_write.argtypes = [C.POINTER(FlatIndexC), C.c_char_p]
        L.moni_index_load_reference.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
        L.moni_ldx_lift_batch.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p]
        _lib = L
    return _lib


HANDOVER_WHY = ("long_read", "anchors", "chains", "chains_to_score", "chain_anchors", "dp_size", "unused", "wildcard_or_dirs", "loop_depends_on_score",
                "extension_short_of_end", "capacity", "cigar_md_line")


def _stats_dict(st) -> dict:
    d = {k: getattr(st, k) for k, _ in AlignStatsC._fields_ if k != "handover_why"}
    d["handover_why"] = {HANDOVER_WHY[i]: int(st.handover_why[i]) for i in range(12) if st.handover_why[i]}
    return d


def _chk(rc: int, what: str):
    if rc != 0:
        raise RuntimeError("%s failed with code %d" % (what, rc))


def flat_struct(fi, without_text: bool = False, without_lcp: bool = False) -> FlatIndexC:
    """fi: moni_align_amd.index_build.FlatIndex (arrays are kept alive by the caller).  without_text: text = NULL (rebuilt from the BWT);
    without_lcp: slcp = NULL (the `-n` form: <prefix>.thrbv.full.ms)."""
    s = FlatIndexC()
    s.n, s.r, s.w, s.n_seq = fi.n, fi.r, fi.w, len(fi.seq_starts) - 1
    for k in ("F", "heads", "starts", "ssa", "esa", "thr", "seq_starts") + (() if without_text else ("text",)) + (() if without_lcp else ("slcp",)):
        a = getattr(fi, k)
        assert a.flags["C_CONTIGUOUS"]
        setattr(s, k, a.ctypes.data)
    s._names_keep = b"".join(x.encode() + b"\0" for x in fi.names)
    s.seq_names = s._names_keep
    lf = getattr(fi, "lifts", None)
    if lf is not None:           # VCF-built index: one lift per sequence (liftidx.hpp:131-143); absent = null lifts
        s.lift_second, s.lift_len = lf.second.ctypes.data, lf.len.ctypes.data
        s.lift_ins_off, s.lift_ins = lf.ins_off.ctypes.data, lf.ins.ctypes.data
        s.lift_del_off, s.lift_del = lf.del_off.ctypes.data, lf.dele.ctypes.data
    return s


class Index:
    def __init__(self, fi=None, path: Optional[str] = None, device: int = 0, reference=None, without_text: bool = False, without_lcp: bool = False):
        """fi: a FlatIndex; path: an .mfi file; reference: (<prefix>.thrbv.full.lcp.ms, <prefix>.ldx, text file) of the reference."""
        self._L = lib()
        self._h = C.c_void_p()
        self._keep = fi
        if reference is not None:
            ms, ldx, text = reference
            _chk(self._L.moni_index_load_reference(ms.encode(), ldx.encode(), text.encode() if text else None, device, C.byref(self._h)), "moni_index_load_reference")
        elif fi is not None:
            st = flat_struct(fi, without_text, without_lcp)
            _chk(self._L.moni_index_create(C.byref(st), device, C.byref(self._h)), "moni_index_create")
        else:
            _chk(self._L.moni_index_load(path.encode(), device, C.byref(self._h)), "moni_index_load")
        self.n = self._L.moni_index_n(self._h)
        self.r = self._L.moni_index_r(self._h)
        self.device_bytes = self._L.moni_index_device_bytes(self._h)

    def text(self) -> np.ndarray:
        """the text of the index (n - 1 bytes), as handed over or as rebuilt from the BWT"""
        out = np.empty(self.n - 1, dtype=np.uint8)
        _chk(self._L.moni_index_text(self._h, out.ctypes.data, out.size), "moni_index_text")
        return out

    def close(self):
        if self._h:
            self._L.moni_index_destroy(self._h)
            self._h = C.c_void_p()


class Ctx:
    def __init__(self, index: Index):
        self._L = lib()
        self.index = index
        self._h = C.c_void_p()
        _chk(self._L.moni_ctx_create(index._h, C.byref(self._h)), "moni_ctx_create")
        self.n_reads = 0
        self._parked = {}

    def close(self):
        if self._h:
            self._L.moni_ctx_destroy(self._h)
            self._h = C.c_void_p()

    @staticmethod
    def _batch(seq: np.ndarray, offsets: np.ndarray):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        b = ReadBatchC(seq.ctypes.data, offsets.ctypes.data, len(offsets) - 1)
        return b, (seq, offsets)

    def upload(self, seq: np.ndarray, offsets: np.ndarray):
        b, keep = self._batch(seq, offsets)
        _chk(self._L.moni_reads_upload(self._h, C.byref(b)), "moni_reads_upload")
        self.n_reads = len(offsets) - 1

    def swap(self, slot: int):
        """exchange the resident batch with the one parked in `slot` (moni_reads_swap)"""
        _chk(self._L.moni_reads_swap(self._h, slot), "moni_reads_swap")
        self.n_reads, self._parked[slot] = self._parked.get(slot, 0), self.n_reads

    def ms_run(self):
        _chk(self._L.moni_ms_run(self._h), "moni_ms_run")

    def ms_query_batch(self, seq: np.ndarray, offsets: np.ndarray) -> np.ndarray:
        b, keep = self._batch(seq, offsets)
        total = int(keep[1][-1] - keep[1][0])
        out = np.empty(2 * total, dtype=np.uint64)
        _chk(self._L.moni_ms_query_batch(self._h, C.byref(b), out.ctypes.data), "moni_ms_query_batch")
        self.n_reads = len(offsets) - 1
        return out

    def seed_run(self, min_len: int = 25, filter_seeds: bool = True, n_seeds_thr: int = 1000, report_mems: bool = False):
        p = SeedParamsC(min_len, int(filter_seeds), n_seeds_thr, int(report_mems))
        _chk(self._L.moni_seed_run(self._h, C.byref(p)), "moni_seed_run")

    def seed_fetch(self) -> Dict[str, np.ndarray]:
        nm, no = C.c_uint64(), C.c_uint64()
        _chk(self._L.moni_seed_counts(self._h, C.byref(nm), C.byref(no)), "moni_seed_counts")
        mems = np.empty(nm.value, dtype=MEM_DTYPE)
        occs = np.empty(no.value, dtype=np.uint64)
        rmo = np.empty(self.n_reads + 1, dtype=np.uint64)
        _chk(self._L.moni_seed_fetch(self._h, mems.ctypes.data, occs.ctypes.data, rmo.ctypes.data), "moni_seed_fetch")
        return {"mems": mems, "occs": occs, "read_mem_off": rmo}

    def phi_lcp_batch(self, pos: np.ndarray, inverse: bool = False):
        pos = np.ascontiguousarray(pos, dtype=np.uint64)
        a = np.empty(len(pos), dtype=np.uint64)
        b = np.empty(len(pos), dtype=np.uint64)
        _chk(self._L.moni_phi_lcp_batch(self._h, pos.ctypes.data, len(pos), int(inverse), a.ctypes.data, b.ctypes.data),
             "moni_phi_lcp_batch")
        return a, b

    def extz_batch(self, qseq: np.ndarray, tseq: np.ndarray, tasks: np.ndarray, cigar_cap: Optional[int] = None,
                   mat=DEFAULT_MAT, q: int = 4, e: int = 2, w: int = -1, zdrop: int = -1, end_bonus: int = 400):
        qseq = np.ascontiguousarray(qseq, dtype=np.uint8)
        tseq = np.ascontiguousarray(tseq, dtype=np.uint8)
        tasks = np.ascontiguousarray(tasks, dtype=DP_TASK_DTYPE)
        prm = DpParamsC()
        prm.m = 5
        for i, v in enumerate(mat):
            prm.mat[i] = v
        prm.q, prm.e, prm.w, prm.zdrop, prm.end_bonus = q, e, w, zdrop, end_bonus
        res = np.zeros(len(tasks), dtype=DP_RESULT_DTYPE)
        if cigar_cap is None:
            cigar_cap = int((tasks["qlen"].astype(np.int64) + tasks["tlen"].astype(np.int64)).sum()) + 16
        pool = np.zeros(cigar_cap, dtype=np.uint32)
        used = C.c_uint64()
        _chk(self._L.moni_extz_batch(self._h, C.byref(prm), qseq.ctypes.data, len(qseq), tseq.ctypes.data, len(tseq),
                                     tasks.ctypes.data, len(tasks), res.ctypes.data, pool.ctypes.data, cigar_cap,
                                     C.byref(used)), "moni_extz_batch")
        return res, pool[: used.value]

    def align_batch(self, seq: np.ndarray, offsets: np.ndarray, names: np.ndarray, name_off: np.ndarray, quals=None,
                    host_threads: Optional[int] = None, stream: bool = False, want_text: bool = True, **overrides):
        """SAM text (bytes) of the batch + stats dict; the whole single-end path on the GPU + host stages.
        stream=True: moni_align_stream (text in the context-owned pinned buffer, lines ordered on the GPU); want_text=False
        (with stream) returns the text's length instead of copying it into a Python bytes object."""
        b, keep = self._batch(seq, offsets)
        names = np.ascontiguousarray(names, dtype=np.uint8)
        name_off = np.ascontiguousarray(name_off, dtype=np.uint64)
        if quals is not None:
            quals = np.ascontiguousarray(quals, dtype=np.uint8)
        prm = AlignParamsC()
        self._L.moni_align_params_default(C.byref(prm))
        if host_threads is not None:
            prm.host_threads = host_threads
        for k, v in overrides.items():
            setattr(prm, k, v)
        out = C.c_void_p()
        ln = C.c_uint64()
        st = AlignStatsC()
        fn = self._L.moni_align_stream if stream else self._L.moni_align_batch
        _chk(fn(self._h, C.byref(b), names.ctypes.data, name_off.ctypes.data,
                quals.ctypes.data if quals is not None else None, C.byref(prm), C.byref(out), C.byref(ln),
                C.byref(st)), "moni_align_stream" if stream else "moni_align_batch")
        self.n_reads = len(offsets) - 1
        try:
            sam = C.string_at(out, ln.value) if (want_text or not stream) else int(ln.value)
        finally:
            if not stream:
                self._L.moni_free(out)
        return sam, _stats_dict(st)

    def align_csv_batch(self, seq, offsets, names, name_off, quals=None, host_threads: Optional[int] = None, **overrides):
        """(SAM text, CSV lines, stats) of moni_align_csv_batch (`-c`)"""
        b, keep = self._batch(seq, offsets)
        names = np.ascontiguousarray(names, dtype=np.uint8)
        name_off = np.ascontiguousarray(name_off, dtype=np.uint64)
        if quals is not None:
            quals = np.ascontiguousarray(quals, dtype=np.uint8)
        prm = AlignParamsC()
        self._L.moni_align_params_default(C.byref(prm))
        if host_threads is not None:
            prm.host_threads = host_threads
        for k, v in overrides.items():
            setattr(prm, k, v)
        sam, sl, csv, cl = C.c_void_p(), C.c_uint64(), C.c_void_p(), C.c_uint64()
        st = AlignStatsC()
        _chk(self._L.moni_align_csv_batch(self._h, C.byref(b), names.ctypes.data, name_off.ctypes.data, quals.ctypes.data if quals is not None else None, C.byref(prm),
                                          C.byref(sam), C.byref(sl), C.byref(csv), C.byref(cl), C.byref(st)), "moni_align_csv_batch")
        self.n_reads = len(offsets) - 1
        try:
            return C.string_at(sam, sl.value), C.string_at(csv, cl.value), _stats_dict(st)
        finally:
            self._L.moni_free(sam); self._L.moni_free(csv)

    def _pe_params(self, host_threads, overrides):
        prm = AlignParamsC()
        self._L.moni_align_params_default(C.byref(prm))
        if host_threads is not None:
            prm.host_threads = host_threads
        pe = PeParamsC()
        self._L.moni_pe_params_default(C.byref(pe))
        for k, v in overrides.items():
            setattr(pe if hasattr(pe, k) and not hasattr(prm, k) else prm, k, v)
        return prm, pe

    def pe_learn(self, seq: np.ndarray, offsets: np.ndarray, model: Optional["PeModelC"] = None, **overrides):
        """learn_fragment_model over one batch of interleaved pairs (reads 2p, 2p+1); returns the updated model."""
        b, keep = self._batch(seq, offsets)
        prm, pe = self._pe_params(None, overrides)
        model = model if model is not None else PeModelC()
        _chk(self._L.moni_pe_learn_batch(self._h, C.byref(b), C.byref(prm), C.byref(pe), C.byref(model)), "moni_pe_learn_batch")
        return model

    def pe_align(self, seq: np.ndarray, offsets: np.ndarray, names: np.ndarray, name_off: np.ndarray, quals, model: "PeModelC",
                 host_threads: Optional[int] = None, want_text: bool = True, stream: bool = False, **overrides):
        """SAM text (bytes) of the interleaved pairs + stats dict (moni_pe_align_batch; stream=True: moni_pe_align_stream, the text in the
        context-owned buffer); want_text=False returns the text's length instead of copying it into a Python bytes object (the library has
        still produced the text in host memory)."""
        b, keep = self._batch(seq, offsets)
        names = np.ascontiguousarray(names, dtype=np.uint8)
        name_off = np.ascontiguousarray(name_off, dtype=np.uint64)
        if quals is not None:
            quals = np.ascontiguousarray(quals, dtype=np.uint8)
        prm, pe = self._pe_params(host_threads, overrides)
        out = C.c_void_p()
        ln = C.c_uint64()
        st = AlignStatsC()
        fn = self._L.moni_pe_align_stream if stream else self._L.moni_pe_align_batch
        _chk(fn(self._h, C.byref(b), names.ctypes.data, name_off.ctypes.data, quals.ctypes.data if quals is not None else None,
                C.byref(prm), C.byref(pe), C.byref(model), C.byref(out), C.byref(ln), C.byref(st)), "moni_pe_align_stream" if stream else "moni_pe_align_batch")
        try:
            sam = C.string_at(out, ln.value) if want_text else int(ln.value)
        finally:
            if not stream:
                self._L.moni_free(out)
        return sam, _stats_dict(st)

    def pe_align_csv(self, seq, offsets, names, name_off, quals, model: "PeModelC", host_threads: Optional[int] = None, **overrides):
        """(SAM text, CSV lines, stats) of moni_pe_align_csv_batch (`-c` for pairs: one line per pair)"""
        b, keep = self._batch(seq, offsets)
        names = np.ascontiguousarray(names, dtype=np.uint8)
        name_off = np.ascontiguousarray(name_off, dtype=np.uint64)
        if quals is not None:
            quals = np.ascontiguousarray(quals, dtype=np.uint8)
        prm, pe = self._pe_params(host_threads, overrides)
        sam, sl, csv, cl, st = C.c_void_p(), C.c_uint64(), C.c_void_p(), C.c_uint64(), AlignStatsC()
        _chk(self._L.moni_pe_align_csv_batch(self._h, C.byref(b), names.ctypes.data, name_off.ctypes.data, quals.ctypes.data if quals is not None else None,
                                             C.byref(prm), C.byref(pe), C.byref(model), C.byref(sam), C.byref(sl), C.byref(csv), C.byref(cl), C.byref(st)), "moni_pe_align_csv_batch")
        try:
            return C.string_at(sam, sl.value), C.string_at(csv, cl.value), _stats_dict(st)
        finally:
            self._L.moni_free(sam)
            self._L.moni_free(csv)

    def pe_align_run(self, names: np.ndarray, name_off: np.ndarray, quals, model: "PeModelC", host_threads: Optional[int] = None, want_text: bool = True, **overrides):
        """moni_pe_align_run over the interleaved pairs made resident by upload(); the text is in the context's buffer (want_text=False: its length)"""
        names = np.ascontiguousarray(names, dtype=np.uint8)
        name_off = np.ascontiguousarray(name_off, dtype=np.uint64)
        if quals is not None:
            quals = np.ascontiguousarray(quals, dtype=np.uint8)
        prm, pe = self._pe_params(host_threads, overrides)
        out, ln, st = C.c_void_p(), C.c_uint64(), AlignStatsC()
        _chk(self._L.moni_pe_align_run(self._h, names.ctypes.data, name_off.ctypes.data, quals.ctypes.data if quals is not None else None, C.byref(prm), C.byref(pe),
                                       C.byref(model), C.byref(out), C.byref(ln), C.byref(st)), "moni_pe_align_run")
        return (C.string_at(out, ln.value) if want_text else int(ln.value)), _stats_dict(st)

    def pe_report_mems(self, seq, offsets, names, name_off, quals=None, **overrides) -> bytes:
        """-m for pairs (moni_pe_report_mems_batch): the MEM records of the interleaved pairs"""
        b, keep = self._batch(seq, offsets)
        names = np.ascontiguousarray(names, dtype=np.uint8)
        name_off = np.ascontiguousarray(name_off, dtype=np.uint64)
        if quals is not None:
            quals = np.ascontiguousarray(quals, dtype=np.uint8)
        prm, pe = self._pe_params(None, overrides)
        out, ln = C.c_void_p(), C.c_uint64()
        _chk(self._L.moni_pe_report_mems_batch(self._h, C.byref(b), names.ctypes.data, name_off.ctypes.data, quals.ctypes.data if quals is not None else None,
                                               C.byref(prm), C.byref(pe), C.byref(out), C.byref(ln)), "moni_pe_report_mems_batch")
        self.n_reads = len(offsets) - 1
        try:
            return C.string_at(out, ln.value)
        finally:
            self._L.moni_free(out)

    def align_run(self, names: np.ndarray, name_off: np.ndarray, quals=None, host_threads: Optional[int] = None,
                  want_text: bool = True, **overrides):
        """moni_align_run over the batch made resident by upload(); want_text=False skips the copy into a Python bytes
        object (the library has still produced the SAM text) and returns its length instead."""
        names = np.ascontiguousarray(names, dtype=np.uint8)
        name_off = np.ascontiguousarray(name_off, dtype=np.uint64)
        if quals is not None:
            quals = np.ascontiguousarray(quals, dtype=np.uint8)
        prm = AlignParamsC()
        self._L.moni_align_params_default(C.byref(prm))
        if host_threads is not None:
            prm.host_threads = host_threads
        for k, v in overrides.items():
            setattr(prm, k, v)
        out = C.c_void_p()
        ln = C.c_uint64()
        st = AlignStatsC()
        _chk(self._L.moni_align_run(self._h, names.ctypes.data, name_off.ctypes.data, quals.ctypes.data if quals is not None else None,
                                    C.byref(prm), C.byref(out), C.byref(ln), C.byref(st)), "moni_align_run")
        sam = C.string_at(out, ln.value) if want_text else int(ln.value)      # the buffer belongs to the context
        return sam, _stats_dict(st)

    def ms_lengths_batch(self, seq: np.ndarray, offsets: np.ndarray):
        """legacy `moni ms`: (pointers, lengths) of the forward strand of every read"""
        b, keep = self._batch(seq, offsets)
        total = int(keep[1][-1] - keep[1][0])
        ptr = np.empty(total, dtype=np.uint64)
        ln = np.empty(total, dtype=np.uint64)
        _chk(self._L.moni_ms_lengths_batch(self._h, C.byref(b), ptr.ctypes.data, ln.ctypes.data), "moni_ms_lengths_batch")
        self.n_reads = len(offsets) - 1
        return ptr, ln

    def report_mems_batch(self, seq, offsets, names, name_off, quals=None, **overrides) -> bytes:
        b, keep = self._batch(seq, offsets)
        names = np.ascontiguousarray(names, dtype=np.uint8)
        name_off = np.ascontiguousarray(name_off, dtype=np.uint64)
        if quals is not None:
            quals = np.ascontiguousarray(quals, dtype=np.uint8)
        prm = AlignParamsC()
        self._L.moni_align_params_default(C.byref(prm))
        for k, v in overrides.items():
            setattr(prm, k, v)
        out, ln = C.c_void_p(), C.c_uint64()
        _chk(self._L.moni_report_mems_batch(self._h, C.byref(b), names.ctypes.data, name_off.ctypes.data, quals.ctypes.data if quals is not None else None,
                                            C.byref(prm), C.byref(out), C.byref(ln)), "moni_report_mems_batch")
        self.n_reads = len(offsets) - 1
        try:
            return C.string_at(out, ln.value)
        finally:
            self._L.moni_free(out)

    def sam_header(self) -> bytes:
        out = C.c_void_p()
        ln = C.c_uint64()
        _chk(self._L.moni_sam_header(self.index._h, C.byref(out), C.byref(ln)), "moni_sam_header")
        try:
            return C.string_at(out, ln.value)
        finally:
            self._L.moni_free(out)

    def kernel_ms(self, which: int) -> float:
        v = C.c_float()
        _chk(self._L.moni_last_kernel_ms(self._h, which, C.byref(v)), "moni_last_kernel_ms")
        return float(v.value)

    def counters(self) -> np.ndarray:
        out = np.zeros(4, dtype=np.uint64)
        _chk(self._L.moni_last_counters(self._h, out.ctypes.data), "moni_last_counters")
        return out


def ldx_info(path: str):
    n_seq, u, w, has_w = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_int()
    _chk(lib().moni_ldx_info(path.encode(), C.byref(n_seq), C.byref(u), C.byref(w), C.byref(has_w)), "moni_ldx_info")
    return {"n_seq": n_seq.value, "u": u.value, "w": w.value, "has_w": bool(has_w.value)}


def ldx_rewrite(src: str, dst: str, with_w: bool):
    _chk(lib().moni_ldx_rewrite(src.encode(), dst.encode(), int(with_w)), "moni_ldx_rewrite")


def ldx_write(fi, path: str, with_w: bool = True):
    st = flat_struct(fi)
    _chk(lib().moni_ldx_write(C.byref(st), path.encode(), int(with_w)), "moni_ldx_write")


def ms_file_info(path: str):
    n, r = C.c_uint64(), C.c_uint64()
    _chk(lib().moni_ms_file_info(path.encode(), C.byref(n), C.byref(r)), "moni_ms_file_info")
    return n.value, r.value


def ms_file_read(path: str):
    """<prefix>.thrbv.full.lcp.ms -> dict of the flat arrays (n, r, F, heads, starts, ssa, esa, thr, slcp)."""
    n, r = ms_file_info(path)
    a = {"F": np.zeros(256, np.uint64), "heads": np.zeros(r, np.uint8), "starts": np.zeros(r + 1, np.uint64), "ssa": np.zeros(r, np.uint64),
         "esa": np.zeros(r, np.uint64), "thr": np.zeros(r, np.uint64), "slcp": np.zeros(r, np.uint64)}
    err = C.create_string_buffer(256)
    rc = lib().moni_ms_file_read(path.encode(), r, *[a[k].ctypes.data for k in ("F", "heads", "starts", "ssa", "esa", "thr", "slcp")], err, 256)
    if rc != 0:
        raise RuntimeError("moni_ms_file_read failed with code %d: %s" % (rc, err.value.decode()))
    a["n"], a["r"] = n, r
    return a


def ms_file_write(fi, path: str, without_lcp: bool = False):
    """moni_lcp::serialize (<prefix>.thrbv.full.lcp.ms) or, without_lcp, ms_pointers::serialize (<prefix>.thrbv.full.ms)"""
    st = flat_struct(fi, without_lcp=without_lcp)
    _chk(lib().moni_ms_file_write(C.byref(st), path.encode()), "moni_ms_file_write")


def ldx_lift_batch(path: str, pos: np.ndarray, device: int = 0) -> np.ndarray:
    pos = np.ascontiguousarray(pos, dtype=np.uint64)
    out = np.empty(len(pos), dtype=np.uint64)
    _chk(lib().moni_ldx_lift_batch(path.encode(), device, pos.ctypes.data, len(pos), out.ctypes.data), "moni_ldx_lift_batch")
    return out
