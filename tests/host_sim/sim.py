"""ctypes driver for tests/host_sim/libhost_sim.so (TEST HARNESS: the kernels' per-lane code compiled for the host)."""
import ctypes as C
import os
import subprocess

import numpy as np

from moni_align_amd import capi

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(HERE, "libhost_sim.so")
        srcs = [os.path.join(HERE, "host_sim.cpp"), os.path.join(ROOT, "oracle", "ksw2.hpp")] + \
               [os.path.join(capi.CSRC, f) for f in ("seed_core.h", "image.hpp", "layout.h", "align_host.hpp", "align_core.h", "pe_core.h", "pe_host.hpp", "pe_big.h", "pe_big.cpp", "../../oracle/align_pe.hpp", "sort_emul.h", "lift_core.h", "lift_build.hpp")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", so, os.path.join(HERE, "host_sim.cpp"),
                                   os.path.join(capi.CSRC, "pe_big.cpp")])
        L = C.CDLL(so)
        L.sim_create.restype = C.c_void_p
        L.sim_create.argtypes = [C.POINTER(capi.FlatIndexC)]
        L.sim_destroy.argtypes = [C.c_void_p]
        L.sim_seed_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(capi.SeedParamsC), C.c_uint32, C.c_uint32]
        L.sim_n_mems.restype = C.c_uint64
        L.sim_n_mems.argtypes = [C.c_void_p]
        L.sim_n_occs.restype = C.c_uint64
        L.sim_n_occs.argtypes = [C.c_void_p]
        L.sim_fetch.argtypes = [C.c_void_p] * 5
        L.sim_fetch_pointers.argtypes = [C.c_void_p] * 3
        L.sim_phi.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
        L.sim_align_batch.restype = C.c_void_p
        L.sim_align_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                      C.POINTER(C.c_uint64), C.c_void_p]
        L.sim_align_core_batch.restype = C.c_void_p
        L.sim_align_core_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.POINTER(C.c_uint64), C.c_void_p]
        L.sim_align_pe_batch.restype = C.c_void_p
        L.sim_align_pe_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double,
                                         C.c_double, C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]
        L.sim_align_pe_big_batch.restype = C.c_void_p
        L.sim_align_pe_big_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int,
                                             C.POINTER(C.c_uint64), C.c_void_p]
        L.sim_free.argtypes = [C.c_void_p]
        L.liftsim_create.restype = C.c_void_p
        L.liftsim_create.argtypes = [C.POINTER(capi.FlatIndexC)]
        L.liftsim_destroy.argtypes = [C.c_void_p]
        L.liftsim_n_runs.restype = C.c_uint64
        L.liftsim_n_runs.argtypes = [C.c_void_p, C.c_uint64]
        L.liftsim_lift.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.liftsim_cigar.restype = C.c_int64
        L.liftsim_cigar.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        _lib = L
    return _lib


class Sim:
    def __init__(self, fi, without_lcp=False):
        self.fi = fi
        st = capi.flat_struct(fi, without_lcp=without_lcp)
        self.h = lib().sim_create(C.byref(st))
        if not self.h:
            raise RuntimeError("host_sim: index rejected")

    def seed_run(self, seq, offsets, min_len=25, filter_seeds=True, n_seeds_thr=1000, report_mems=False, tmp_cap=16,
                 pool_rows=64):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        p = capi.SeedParamsC(min_len, int(filter_seeds), n_seeds_thr, int(report_mems))
        rc = lib().sim_seed_run(self.h, seq.ctypes.data, offsets.ctypes.data, len(offsets) - 1, C.byref(p), tmp_cap, pool_rows)
        if rc:
            raise RuntimeError("sim_seed_run rc=%d" % rc)
        mems = np.empty(lib().sim_n_mems(self.h), dtype=capi.MEM_DTYPE)
        occs = np.empty(lib().sim_n_occs(self.h), dtype=np.uint64)
        rmo = np.empty(len(offsets), dtype=np.uint64)
        cnt = np.empty(4, dtype=np.uint64)
        lib().sim_fetch(self.h, mems.ctypes.data, occs.ctypes.data, rmo.ctypes.data, cnt.ctypes.data)
        ptr = np.empty(2 * int(offsets[-1] - offsets[0]), dtype=np.uint64)
        lib().sim_fetch_pointers(self.h, offsets.ctypes.data, ptr.ctypes.data)
        return {"mems": mems, "occs": occs, "read_mem_off": rmo, "counters": cnt, "pointers": ptr}

    def phi(self, i, inverse=False):
        out = np.empty(2, dtype=np.uint64)
        lib().sim_phi(self.h, i, int(inverse), out.ctypes.data)
        return int(out[0]), int(out[1])


    def align_batch(self, seq, offsets, names, name_off, quals=None, threads=1):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        names = np.ascontiguousarray(names, dtype=np.uint8)
        name_off = np.ascontiguousarray(name_off, dtype=np.uint64)
        if quals is not None:
            quals = np.ascontiguousarray(quals, dtype=np.uint8)
        ln = C.c_uint64()
        st = np.zeros(5, dtype=np.uint64)
        p = lib().sim_align_batch(self.h, seq.ctypes.data, offsets.ctypes.data, len(offsets) - 1, names.ctypes.data, name_off.ctypes.data,
                                  quals.ctypes.data if quals is not None else None, threads, C.byref(ln), st.ctypes.data)
        if not p:
            raise RuntimeError("sim_align_batch failed")
        try:
            return C.string_at(p, ln.value), st
        finally:
            lib().sim_free(p)

    def align_core_batch(self, seq, offsets, names, name_off, quals=None):
        """SAM via the device-side per-read logic (align_core.h) replayed on the host; stats = reads, aligned, dp tasks, overflowed, rounds.  The harness also checks, per read, that the host stage's one-pass emitter (emit_record) spells the same record as finish_record + sam_write, with MD/NM computed and handed in; it fails on any difference."""
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        names = np.ascontiguousarray(names, dtype=np.uint8)
        name_off = np.ascontiguousarray(name_off, dtype=np.uint64)
        if quals is not None:
            quals = np.ascontiguousarray(quals, dtype=np.uint8)
        ln = C.c_uint64()
        st = np.zeros(5, dtype=np.uint64)
        p = lib().sim_align_core_batch(self.h, seq.ctypes.data, offsets.ctypes.data, len(offsets) - 1, names.ctypes.data, name_off.ctypes.data,
                                       quals.ctypes.data if quals is not None else None, C.byref(ln), st.ctypes.data)
        if not p:
            raise RuntimeError("sim_align_core_batch failed")
        try:
            return C.string_at(p, ln.value), st
        finally:
            lib().sim_free(p)

    def align_pe_batch(self, seq, offsets, names, name_off, quals=None, finalize=True, mean=0.0, std_dev=0.0, find_orphan=False, secondary_chains=False):
        """Paired-end: reads 2p / 2p+1 are the mates of pair p; pe_core.h replayed on the host + pe_host.hpp.  finalize=False is the
        learn pass: returns (learn[n_pairs, 4] = aligned, best tot, second tot, dist; stats)."""
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        names = np.ascontiguousarray(names, dtype=np.uint8)
        name_off = np.ascontiguousarray(name_off, dtype=np.uint64)
        if quals is not None:
            quals = np.ascontiguousarray(quals, dtype=np.uint8)
        n_pairs = (len(offsets) - 1) // 2
        ln = C.c_uint64()
        st = np.zeros(5, dtype=np.uint64)
        learn = np.zeros((n_pairs, 4), dtype=np.int64)
        p = lib().sim_align_pe_batch(self.h, seq.ctypes.data, offsets.ctypes.data, n_pairs, names.ctypes.data, name_off.ctypes.data,
                                     quals.ctypes.data if quals is not None else None, int(finalize), float(mean), float(std_dev), int(find_orphan) | (2 if secondary_chains else 0),
                                     learn.ctypes.data, C.byref(ln), st.ctypes.data)
        if not p:
            raise RuntimeError("sim_align_pe_batch failed")
        try:
            return (C.string_at(p, ln.value) if finalize else learn), st
        finally:
            lib().sim_free(p)

    def align_pe_big_batch(self, seq, offsets, names, name_off, quals=None, mean=0.0, std_dev=0.0, find_orphan=False, secondary_chains=False):
        """Every pair through the host pipeline for pairs (pe_big.cpp) with the CPU stand-ins for its DP batches."""
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        names = np.ascontiguousarray(names, dtype=np.uint8)
        name_off = np.ascontiguousarray(name_off, dtype=np.uint64)
        if quals is not None:
            quals = np.ascontiguousarray(quals, dtype=np.uint8)
        ln = C.c_uint64()
        st = np.zeros(5, dtype=np.uint64)
        p = lib().sim_align_pe_big_batch(self.h, seq.ctypes.data, offsets.ctypes.data, (len(offsets) - 1) // 2, names.ctypes.data, name_off.ctypes.data,
                                         quals.ctypes.data if quals is not None else None, float(mean), float(std_dev), int(find_orphan) | (2 if secondary_chains else 0), C.byref(ln), st.ctypes.data)
        if not p:
            raise RuntimeError("sim_align_pe_big_batch failed")
        try:
            return C.string_at(p, ln.value), st
        finally:
            lib().sim_free(p)


class LiftSim:
    """The product's lift tables (lift_build.hpp + lift_core.h) built from seq_starts / w / lifts alone."""

    def __init__(self, seq_starts, w, lifts):
        st = capi.FlatIndexC()
        self._keep = (np.ascontiguousarray(seq_starts, dtype=np.uint64), lifts)
        st.n_seq, st.w = len(seq_starts) - 1, w
        st.seq_starts = self._keep[0].ctypes.data
        if lifts is not None:
            st.lift_second, st.lift_len = lifts.second.ctypes.data, lifts.len.ctypes.data
            st.lift_ins_off, st.lift_ins = lifts.ins_off.ctypes.data, lifts.ins.ctypes.data
            st.lift_del_off, st.lift_del = lifts.del_off.ctypes.data, lifts.dele.ctypes.data
        self.h = lib().liftsim_create(C.byref(st))
        if not self.h:
            raise RuntimeError("lift tables rejected")

    def n_runs(self, seq):
        return int(lib().liftsim_n_runs(self.h, seq))

    def lift(self, pos):
        pos = np.ascontiguousarray(pos, dtype=np.uint64)
        out = np.empty(len(pos), dtype=np.uint64)
        lib().liftsim_lift(self.h, pos.ctypes.data, len(pos), out.ctypes.data)
        return out

    def lift_cigar(self, pos, cigar):
        cigar = np.ascontiguousarray(cigar, dtype=np.uint32)
        cap = 4 * len(cigar) + 4096
        out = np.zeros(cap, dtype=np.uint32)
        n = lib().liftsim_cigar(self.h, int(pos), cigar.ctypes.data, len(cigar), out.ctypes.data, cap)
        assert n >= 0
        return out[:n].copy()

    def __del__(self):
        if getattr(self, "h", None):
            lib().liftsim_destroy(self.h)
            self.h = None
