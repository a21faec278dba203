"""Flat r-index builder (suffix array -> run-length BWT, SA samples, thresholds, LCP samples).

The reference builds these files offline with prefix-free parsing
(pipeline/moni.in:115-212; thirdparty pfp / pfp-thresholds, both absent), which
SURVEY.md §2 marks OUT OF SCOPE.  The hot path still needs an index to run on,
so this module produces the *semantic* content of `<prefix>.thrbv.full.lcp.ms`
(+ `.plain.slp` text, `.ldx` sequence starts) as plain arrays:

  heads[r], starts[r+1]          run-length BWT            (ms_rle_string.hpp:245-303)
  ssa[r], esa[r]                 (SA[run start]-1) mod n, (SA[run end]-1) mod n
                                 = samples_start / samples_last (moni.hpp:148-184)
  thr[r]                         threshold position per run, 0 for the first run
                                 of a letter (thresholds_ds.hpp:413-426)
  slcp[r]                        LCP at each run start      (moni_lcp.hpp:117-145)
  F[256]                         (moni.hpp:253-282)
  text[n-1], seq_starts, names   (.plain.slp / .ldx content)

Everything is expressed with torch tensor ops (sort / gather / scatter-reduce)
so the same code runs on CPU for the small test genomes and on the MI355X for
the 0.8 G-character benchmark text: prefix doubling with radix sorts, LCP from
the stored doubling ranks, thresholds by a segmented arg-min.  torch is used as
a sorting/gather engine here, nothing else.
"""
from __future__ import annotations

import dataclasses
import io
import struct
from typing import List, Optional

import os
import numpy as np
import torch

TERMINATOR = 1           # moni.hpp / r-index: bytes <= 1 in the BWT are stored as 1
MAGIC = b"MONIFLT2"


@dataclasses.dataclass
class Lifts:
    """liftidx::lifts (include/aligner/liftidx.hpp:131-143) as flat arrays, one lift per sequence: `second` (start of the
    target contig in the concatenation), the number of alignment columns and the sorted positions of the ones of the
    levioSAM ins / del bit-vectors (ragged; offsets have n_seq + 1 entries).  The snp vector is not used by this path."""
    second: np.ndarray           # uint64[nseq]
    len: np.ndarray              # uint64[nseq]
    ins_off: np.ndarray          # uint64[nseq+1]
    ins: np.ndarray              # uint64[]
    del_off: np.ndarray          # uint64[nseq+1]
    dele: np.ndarray             # uint64[]

    @staticmethod
    def from_lists(second, length, ins_list, del_list) -> "Lifts":
        u = np.uint64
        io = np.zeros(len(ins_list) + 1, dtype=u)
        do = np.zeros(len(del_list) + 1, dtype=u)
        io[1:] = np.cumsum([len(x) for x in ins_list])
        do[1:] = np.cumsum([len(x) for x in del_list])
        cat = lambda L: (np.concatenate([np.asarray(x, dtype=u) for x in L]) if sum(len(x) for x in L) else np.zeros(0, dtype=u))
        return Lifts(np.asarray(second, dtype=u), np.asarray(length, dtype=u), io, np.ascontiguousarray(cat(ins_list)), do,
                     np.ascontiguousarray(cat(del_list)))

    def ins_of(self, i): return self.ins[int(self.ins_off[i]):int(self.ins_off[i + 1])]
    def del_of(self, i): return self.dele[int(self.del_off[i]):int(self.del_off[i + 1])]


@dataclasses.dataclass
class FlatIndex:
    n: int                       # BWT length = len(text) + 1
    r: int
    w: int
    F: np.ndarray                # uint64[256]
    heads: np.ndarray            # uint8[r]
    starts: np.ndarray           # uint64[r+1], starts[r] = n
    ssa: np.ndarray              # uint64[r]
    esa: np.ndarray              # uint64[r]
    thr: np.ndarray              # uint64[r]
    slcp: np.ndarray             # uint64[r]
    text: np.ndarray             # uint8[n-1]
    seq_starts: np.ndarray       # uint64[nseq+1]
    names: List[str]
    lifts: Optional[Lifts] = None  # None = FASTA-built index (null lifts, liftidx.hpp:150-157)

    # ---- (de)serialisation: one flat little-endian file ----
    def save(self, path: str) -> None:
        names_blob = b"".join(struct.pack("<Q", len(s.encode())) + s.encode() for s in self.names)
        with open(path, "wb") as f:
            f.write(MAGIC)
            f.write(struct.pack("<6Q", self.n, self.r, self.w, len(self.seq_starts) - 1,
                                len(names_blob), 1 if self.lifts is not None else 0))
            def put(a, dt):
                b = np.ascontiguousarray(a, dtype=dt).tobytes()
                f.write(b)
                f.write(b"\0" * ((-len(b)) % 8))
            put(self.F, np.uint64)
            put(self.heads, np.uint8)
            put(self.starts, np.uint64)
            put(self.ssa, np.uint64)
            put(self.esa, np.uint64)
            put(self.thr, np.uint64)
            put(self.slcp, np.uint64)
            put(self.text, np.uint8)
            put(self.seq_starts, np.uint64)
            f.write(names_blob)
            f.write(b"\0" * ((-len(names_blob)) % 8))
            if self.lifts is not None:       # per sequence: second, columns, #ins, #del, then the two lists of ones
                lf = self.lifts
                for i in range(len(self.seq_starts) - 1):
                    a, b = lf.ins_of(i), lf.del_of(i)
                    f.write(struct.pack("<4Q", int(lf.second[i]), int(lf.len[i]), len(a), len(b)))
                    put(a, np.uint64)
                    put(b, np.uint64)

    @staticmethod
    def load(path: str, mmap: bool = False) -> "FlatIndex":
        """mmap: the arrays are read-only views of the mapped file (no private copy of the ~10 GB; processes that load the same file share its pages)"""
        if mmap:
            buf = np.memmap(path, dtype=np.uint8, mode="r")
        else:
            with open(path, "rb") as f:
                buf = f.read()
        if bytes(buf[:8]) != MAGIC:
            raise ValueError("not a MONIFLT2 file: " + path)
        n, r, w, nseq, nblob, has_lifts = struct.unpack_from("<6Q", buf, 8)
        off = 8 + 48
        def get(count, dt):
            nonlocal off
            a = np.frombuffer(buf, dtype=dt, count=count, offset=off)
            if not mmap:
                a = a.copy()
            nb = count * np.dtype(dt).itemsize
            off += nb + ((-nb) % 8)
            return a
        F = get(256, np.uint64)
        heads = get(r, np.uint8)
        starts = get(r + 1, np.uint64)
        ssa = get(r, np.uint64)
        esa = get(r, np.uint64)
        thr = get(r, np.uint64)
        slcp = get(r, np.uint64)
        text = get(n - 1, np.uint8)
        seq_starts = get(nseq + 1, np.uint64)
        names = []
        p = off
        for _ in range(nseq):
            (ln,) = struct.unpack_from("<Q", buf, p)
            names.append(bytes(buf[p + 8:p + 8 + ln]).decode())
            p += 8 + ln
        lifts = None
        if has_lifts:
            off += nblob + ((-nblob) % 8)
            sec, ln_, il, dl = [], [], [], []
            for _ in range(nseq):
                a, b, c, d = struct.unpack_from("<4Q", buf, off)
                off += 32
                sec.append(a); ln_.append(b)
                il.append(get(c, np.uint64)); dl.append(get(d, np.uint64))
            lifts = Lifts.from_lists(sec, ln_, il, dl)
        return FlatIndex(n, r, w, F, heads, starts, ssa, esa, thr, slcp, text, seq_starts, names, lifts)


# --------------------------------------------------------------------------------------
# suffix array by prefix doubling
# --------------------------------------------------------------------------------------

def _group_starts(sorted_keys: torch.Tensor):
    n = sorted_keys.numel()
    flag = torch.ones(n, dtype=torch.bool, device=sorted_keys.device)
    flag[1:] = sorted_keys[1:] != sorted_keys[:-1]
    idx = torch.arange(n, device=sorted_keys.device, dtype=torch.int64)
    gstart = torch.cummax(torch.where(flag, idx, torch.zeros_like(idx)), 0).values
    return flag, gstart


def _rank32(rank: torch.Tensor) -> torch.Tensor:
    """a level's ranks for lcp_from_levels, which only compares them for equality: the low 32 bits as int32 (explicit wrap-around, exact for
    ranks < 2^31), so that a level costs 4 bytes per position for any n < 2^32 (ranks are group starts < n: distinct ranks stay distinct)"""
    return (((rank + (1 << 31)) & 0xFFFFFFFF) - (1 << 31)).to(torch.int32)


def suffix_array(codes: torch.Tensor, bits: int, keep_levels: bool = True, log=None, _force_wide: bool = False, rank64: bool = False):
    """codes: uint8/int64 [n] with a unique smallest 0 at the end.  Returns (sa, key0, k0, levels)
    where levels = [(k, rank_k int32)] for every doubling level (rank_k equal <=> first k
    symbols equal).  The levels keep 32 bits of a rank (n < 2^32) or - rank64, the only form for n >= 2^32: a GRCh38-scale pangenome - the whole
    64-bit rank (8 bytes per position and level instead of 4); the doubling key rank * (n + 1) + rank' fits a signed 64-bit word up to
    n = 3.03e9 - beyond that (_force_wide: always, for tests) a level is two stable sorts instead of one."""
    dev = codes.device
    n = codes.numel()
    if n >= (1 << 32) and not rank64:
        raise ValueError("index_build: n = %d does not fit the 32-bit rank levels of the LCP computation (build with rank64=True / MONI_BUILD_RANK64=1)" % n)
    lvl = (lambda r: r.clone()) if rank64 else _rank32
    wide = _force_wide or (n + 1) * (n + 1) >= (1 << 63)
    k0 = 60 // bits
    c64 = torch.cat([codes.to(torch.int64), torch.zeros(k0, dtype=torch.int64, device=dev)])
    key0 = torch.zeros(n, dtype=torch.int64, device=dev)
    for d in range(k0):
        key0 = (key0 << bits) | c64[d:d + n]
    del c64
    skey, sa = torch.sort(key0)
    flag, gstart = _group_starts(skey)
    del skey
    rank = torch.empty(n, dtype=torch.int64, device=dev)
    rank[sa] = gstart
    levels = [(k0, lvl(rank))] if keep_levels else []
    k = k0
    while not bool(flag.all()):
        if log:
            log("  prefix doubling k=%d, groups=%d / %d" % (k, int(flag.sum()), n))
        r2 = torch.zeros(n, dtype=torch.int64, device=dev)
        if k < n:
            r2[: n - k] = rank[k:] + 1
        if not wide:
            key = rank * (n + 1) + r2
            del r2
            skey, sa = torch.sort(key)
            del key
            flag, gstart = _group_starts(skey)
            del skey
        else:          # (rank, rank') as two keys: stable sort by the second, then by the first
            _, p1 = torch.sort(r2, stable=True)
            _, p2 = torch.sort(rank[p1], stable=True)
            sa = p1[p2]
            del p1, p2
            a, b2 = rank[sa], r2[sa]
            del r2
            n_ = a.numel()
            flag = torch.ones(n_, dtype=torch.bool, device=dev)
            flag[1:] = (a[1:] != a[:-1]) | (b2[1:] != b2[:-1])
            del a, b2
            idx = torch.arange(n_, device=dev, dtype=torch.int64)
            gstart = torch.cummax(torch.where(flag, idx, torch.zeros_like(idx)), 0).values
            del idx
        rank = torch.empty(n, dtype=torch.int64, device=dev)
        rank[sa] = gstart
        k *= 2
        if keep_levels:
            levels.append((k, lvl(rank)))
    return sa, key0, k0, levels


def lcp_from_levels(sa: torch.Tensor, key0: torch.Tensor, k0: int, bits: int, levels) -> torch.Tensor:
    """LCP[j] = lcp(suffix sa[j-1], suffix sa[j]), LCP[0] = 0 (int64 [n])."""
    n = sa.numel()
    dev = sa.device
    x = sa[:-1].clone()
    y = sa[1:].clone()
    l = torch.zeros(n - 1, dtype=torch.int64, device=dev)
    for k, R in reversed(levels):
        ok = (x < n) & (y < n)
        xe = torch.clamp(x, max=n - 1)
        ye = torch.clamp(y, max=n - 1)
        eq = ok & (R[xe] == R[ye])
        step = eq.to(torch.int64) * k
        l += step
        x += step
        y += step
    ok = (x < n) & (y < n)
    xe = torch.clamp(x, max=n - 1)
    ye = torch.clamp(y, max=n - 1)
    xr = key0[xe] ^ key0[ye]
    run = ok.clone()
    mask = (1 << bits) - 1
    for d in range(k0):
        run &= ((xr >> (bits * (k0 - 1 - d))) & mask) == 0
        l += run.to(torch.int64)
    out = torch.zeros(n, dtype=torch.int64, device=dev)
    out[1:] = l
    return out


# --------------------------------------------------------------------------------------
# the builder
# --------------------------------------------------------------------------------------

def build_flat_index(text: np.ndarray, seq_starts: np.ndarray, names: List[str], w: int,
                     device: Optional[str] = None, log=None, lifts: Optional[Lifts] = None) -> FlatIndex:
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    dev = torch.device(device)
    text = np.ascontiguousarray(text, dtype=np.uint8)
    if text.min() < 2:
        raise ValueError("text bytes 0/1 are reserved for the terminator")
    n = int(text.shape[0]) + 1
    present = np.unique(text)
    sigma = len(present) + 1
    bits = max(1, int(np.ceil(np.log2(sigma))))
    b2c = np.zeros(256, dtype=np.uint8)
    b2c[present] = np.arange(1, sigma, dtype=np.uint8)
    c2b = np.zeros(sigma, dtype=np.uint8)
    c2b[0] = TERMINATOR
    c2b[1:] = present
    codes = torch.from_numpy(np.concatenate([b2c[text], np.zeros(1, np.uint8)])).to(dev)
    if log:
        log("suffix array: n=%d sigma=%d bits=%d on %s" % (n, sigma, bits, dev))
    rank64 = n >= (1 << 32) or os.environ.get("MONI_BUILD_RANK64", "") not in ("", "0")      # (the flag: the wide form on any text, e.g. to test it)
    sa, key0, k0, levels = suffix_array(codes, bits, keep_levels=True, log=log, rank64=rank64)
    if log:
        log("lcp from %d levels%s" % (len(levels), " (64-bit ranks)" if rank64 else ""))
    lcp = lcp_from_levels(sa, key0, k0, bits, levels)
    del levels, key0
    # BWT (codes), runs
    prev = sa - 1
    prev[prev < 0] = n - 1
    bwt = codes[prev].to(torch.int64)
    del prev
    rs = torch.ones(n, dtype=torch.bool, device=dev)
    rs[1:] = bwt[1:] != bwt[:-1]
    starts = torch.nonzero(rs).flatten()
    r = int(starts.numel())
    ends = torch.cat([starts[1:] - 1, torch.tensor([n - 1], device=dev)])
    heads_c = bwt[starts]
    ssa = (sa[starts] - 1) % n
    esa = (sa[ends] - 1) % n
    slcp = lcp[starts]
    if log:
        log("r=%d n/r=%.2f; thresholds" % (r, n / r))
    # thresholds: per letter, arg-min of LCP over (end of previous c-run, start of this c-run]
    thr = torch.zeros(r, dtype=torch.int64, device=dev)
    idx = torch.arange(n, dtype=torch.int64, device=dev)
    BIG = torch.iinfo(torch.int64).max
    for c in range(sigma):
        runs_c = torch.nonzero(heads_c == c).flatten()
        rc = int(runs_c.numel())
        if rc <= 1:
            continue
        is_c = bwt == c
        nxt = torch.zeros(n, dtype=torch.bool, device=dev)
        nxt[:-1] = is_c[1:]
        run_end = is_c & ~nxt
        seg = torch.cumsum(run_end.to(torch.int64), 0) - run_end.to(torch.int64)
        del nxt, run_end
        prv = torch.zeros(n, dtype=torch.bool, device=dev)
        prv[1:] = is_c[:-1]
        interior = is_c & prv
        del prv, is_c
        key = torch.where(interior, torch.full_like(idx, BIG), lcp * n + idx)
        del interior
        out = torch.full((rc + 1,), BIG, dtype=torch.int64, device=dev)
        out.scatter_reduce_(0, seg, key, "amin", include_self=True)
        del key, seg
        t = out[:rc] % n
        t[0] = 0
        thr[runs_c] = t
    # F
    counts = torch.bincount(bwt, minlength=sigma).cpu().numpy()
    cnt_b = np.zeros(256, dtype=np.uint64)
    for c in range(sigma):
        cnt_b[c2b[c]] += np.uint64(counts[c])
    F = np.zeros(256, dtype=np.uint64)
    F[1:] = np.cumsum(cnt_b)[:-1]
    starts_np = np.concatenate([starts.cpu().numpy().astype(np.uint64), np.array([n], dtype=np.uint64)])
    return FlatIndex(
        n=n, r=r, w=int(w), F=F,
        heads=c2b[heads_c.cpu().numpy()],
        starts=starts_np,
        ssa=ssa.cpu().numpy().astype(np.uint64),
        esa=esa.cpu().numpy().astype(np.uint64),
        thr=thr.cpu().numpy().astype(np.uint64),
        slcp=slcp.cpu().numpy().astype(np.uint64),
        text=text,
        seq_starts=np.asarray(seq_starts, dtype=np.uint64),
        names=list(names),
        lifts=lifts,
    )


def build_from_pangenome(pg, device: Optional[str] = None, log=None, lifted: bool = True) -> FlatIndex:
    """lifted: carry the pangenome's VCF-style lifts (a `moni build -r ref -v vcf -H12` index: haplotypes lift onto the
    reference contig); False = the FASTA-built form of the same text (null lifts)."""
    lifts = pg.lifts() if (lifted and getattr(pg, "variants", None) is not None) else None
    return build_flat_index(pg.text, pg.seq_starts, pg.names, pg.w, device=device, log=log, lifts=lifts)


# ---- the reference builder's raw per-run files ("raw builder inputs") ------------------------------------
# `moni build` runs bigbwt + pfp-thresholds and then packs their outputs into `<prefix>.thrbv.full.lcp.ms`; the packed
# file needs sdsl/r-index to read, the raw ones are flat arrays of 1- and 5-byte little-endian integers:
#   <prefix>.bwt.heads   1 B per run, bytes <= 1 mean the terminator          (moni.hpp:253-282, ms_rle_string.hpp:245-303)
#   <prefix>.bwt.len     5 B per run
#   <prefix>.ssa / .esa  r x (5 B BWT position, 5 B SA value); used value = SA ? SA - 1 : n - 1   (moni.hpp:148-184)
#   <prefix>.thr_pos     5 B per run, 0 = no threshold                         (thresholds_ds.hpp:393-430)
#   <prefix>.slcp        5 B per entry (r or r + 1 entries)                    (moni_lcp.hpp:117-145)
# plus the text the index was built over and the `.lidx` listing of `name length+w` per sequence (seqidx.hpp, fixture
# data/Chr21.10.lidx).  Reading them gives a FlatIndex without sdsl.  Parity note: no raw files of an upstream build are
# available here, so this reader is pinned only by the reference's loading code cited above and by a write/read round trip.
SSABYTES = 5


def _read5(path: str) -> np.ndarray:
    raw = np.fromfile(path, dtype=np.uint8)
    if raw.size % SSABYTES:
        raise ValueError("invalid file %s: size is not a multiple of %d" % (path, SSABYTES))
    a = np.zeros((raw.size // SSABYTES, 8), dtype=np.uint8)
    a[:, :SSABYTES] = raw.reshape(-1, SSABYTES)
    return a.view("<u8").reshape(-1)


def _write5(path: str, vals) -> None:
    v = np.ascontiguousarray(vals, dtype="<u8")
    if v.size and int(v.max()) >= 1 << 40:
        raise ValueError("value does not fit 5 bytes")
    v.view(np.uint8).reshape(-1, 8)[:, :SSABYTES].tofile(path)


def read_lidx(path: str, w: int):
    """`.lidx`: one `name length+w` line per sequence -> (names, onsets[k+1])."""
    names, on = [], [0]
    for line in open(path).read().splitlines():
        if not line.strip():
            continue
        nm, ln = line.split()
        if int(ln) < w:
            raise ValueError("lidx entry shorter than the separator width")
        names.append(nm)
        on.append(on[-1] + int(ln))
    return names, np.asarray(on, dtype=np.uint64)


def from_raw_files(prefix: str, text_path: str, lidx_path: str, w: int = 10) -> FlatIndex:
    heads = np.fromfile(prefix + ".bwt.heads", dtype=np.uint8).copy()
    heads[heads <= TERMINATOR] = TERMINATOR
    lens = _read5(prefix + ".bwt.len")
    r = int(heads.size)
    if lens.size != r:
        raise ValueError(".bwt.heads and .bwt.len disagree on the number of runs")
    if r and int(lens.min()) == 0:
        raise ValueError("empty run in .bwt.len")
    starts = np.zeros(r + 1, dtype=np.uint64)
    np.cumsum(lens, out=starts[1:])
    n = int(starts[r])

    def samples(path):
        pairs = _read5(path)
        if pairs.size != 2 * r:
            raise ValueError("%s does not hold %d (position, sample) pairs" % (path, r))
        sa = pairs[1::2]
        return np.where(sa > 0, sa - np.uint64(1), np.uint64(n - 1)).astype(np.uint64)

    ssa, esa = samples(prefix + ".ssa"), samples(prefix + ".esa")
    thr = _read5(prefix + ".thr_pos")
    if thr.size != r:
        raise ValueError(".thr_pos does not hold one entry per run")
    slcp = _read5(prefix + ".slcp")
    if slcp.size not in (r, r + 1):
        raise ValueError(".slcp holds %d entries for %d runs" % (slcp.size, r))
    slcp = slcp[:r].copy()
    # F exactly as build_F_ (moni.hpp:253-282)
    cnt = np.zeros(256, dtype=np.uint64)
    np.add.at(cnt, heads, lens)
    F = np.zeros(256, dtype=np.uint64)
    F[1:] = np.cumsum(cnt)[:-1]
    text = np.fromfile(text_path, dtype=np.uint8)
    if text.size != n - 1:
        raise ValueError("text has %d bytes, the BWT %d" % (text.size, n))
    names, on = read_lidx(lidx_path, w)
    if int(on[-1]) + (w - 1 if w else 0) != text.size and int(on[-1]) != text.size:
        raise ValueError(".lidx does not describe this text")
    return FlatIndex(n=n, r=r, w=int(w), F=F, heads=heads, starts=starts, ssa=ssa, esa=esa, thr=thr.astype(np.uint64),
                     slcp=slcp.astype(np.uint64), text=text, seq_starts=on, names=names)


def from_reference_files(ms_path: str, lidx_path: str, text_path: str, w: int = 10) -> FlatIndex:
    """<prefix>.thrbv.full.lcp.ms (moni_lcp::serialize; read by csrc/ms_index_io.hpp through the C ABI) + .lidx + plain text -> FlatIndex
    with null lifts.  (moni_index_load_reference takes the .ldx, lifts included, straight to the device.)"""
    from . import capi
    a = capi.ms_file_read(ms_path)
    text = np.fromfile(text_path, dtype=np.uint8)
    if text.size != a["n"] - 1:
        raise ValueError("text has %d bytes, the BWT %d" % (text.size, a["n"]))
    names, on = read_lidx(lidx_path, w)
    return FlatIndex(n=a["n"], r=a["r"], w=int(w), F=a["F"], heads=a["heads"], starts=a["starts"], ssa=a["ssa"], esa=a["esa"], thr=a["thr"],
                     slcp=a["slcp"], text=text, seq_starts=on, names=names)


def to_raw_files(fi: FlatIndex, prefix: str, text_path: str, lidx_path: str) -> None:
    """The inverse of from_raw_files (tests; also lets upstream's `moni build --no-parse`-style tooling consume our index)."""
    fi.heads.astype(np.uint8).tofile(prefix + ".bwt.heads")
    lens = (fi.starts[1:] - fi.starts[:-1]).astype(np.uint64)
    _write5(prefix + ".bwt.len", lens)
    n = np.uint64(fi.n)

    def pairs(pos, val):
        sa = np.where(val == n - np.uint64(1), np.uint64(0), val + np.uint64(1))      # SA value: the reader takes SA ? SA-1 : n-1
        out = np.empty(2 * len(pos), dtype=np.uint64)
        out[0::2] = pos
        out[1::2] = sa
        return out

    _write5(prefix + ".ssa", pairs(fi.starts[:-1], fi.ssa))
    _write5(prefix + ".esa", pairs(fi.starts[1:] - np.uint64(1), fi.esa))
    _write5(prefix + ".thr_pos", fi.thr)
    _write5(prefix + ".slcp", fi.slcp)
    fi.text.tofile(text_path)
    with open(lidx_path, "w") as f:
        for k, nm in enumerate(fi.names):
            f.write("%s %d\n" % (nm, int(fi.seq_starts[k + 1] - fi.seq_starts[k])))


def _main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="raw r-index files of the reference's builder -> flat index (.mfi) for moni-hip-align")
    ap.add_argument("prefix", help="<prefix>.bwt.heads/.bwt.len/.ssa/.esa/.thr_pos/.slcp")
    ap.add_argument("--text", required=True, help="the text the index was built over (no terminator)")
    ap.add_argument("--lidx", required=True, help="<prefix>.lidx (name, length + w per sequence)")
    ap.add_argument("-w", type=int, default=10)
    ap.add_argument("-o", "--output", required=True)
    a = ap.parse_args(argv)
    fi = from_raw_files(a.prefix, a.text, a.lidx, a.w)
    fi.save(a.output)
    print("n=%d r=%d sequences=%d -> %s" % (fi.n, fi.r, len(fi.names), a.output))


if __name__ == "__main__":
    _main()
