"""Synthetic pangenome + read generator.

The mouse chr19 FASTA is absent from the reference checkout and nothing but
/root/repo travels to the GPU box, so every input is generated from a seed:

  * base genome G0: i.i.d. ACGT of the requested length,
  * haplotypes: G0 + shared variant sites (85 % SNP / 15 % indel, geometric
    indel length mean 3, cap 50), each site carried by a haplotype with the
    site's allele frequency (Beta(0.5, 0.5)),
  * text: every sequence followed by `w` separator bytes (<= 5), the last one
    by w + (w-1), the layout test/src/ldx_slp_test.cpp:101-137 checks,
  * reads: uniform haplotype / position, fair strand, substitution and indel
    errors, no N.

This is test/bench infrastructure, not the product path.
"""
from __future__ import annotations

import dataclasses
from typing import List, Optional

import numpy as np

SEP_SEQ = 5      # separator byte after every sequence (<= 5, never in a read)
SEP_END = 4      # the extra w-1 bytes after the last sequence
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
_COMP[:] = np.arange(256, dtype=np.uint8)
for _a, _b in zip(b"ACGT", b"TGCA"):
    _COMP[_a] = _b


@dataclasses.dataclass
class Pangenome:
    seqs: List[np.ndarray]          # uint8 ASCII, one per sequence
    names: List[str]
    w: int
    # per haplotype (sequences 1..): the variants it carries against sequence 0 as (pos[], kind[], length[]) with kind
    # 0 = SNP, 1 = insertion of `length` bases before reference base pos, 2 = deletion of reference bases [pos, pos + length);
    # None = unrelated sequences (FASTA-style pangenome)
    variants: Optional[List[tuple]] = None

    def lifts(self):
        """levioSAM-style lifts of a `-r ref -v vcf` build (liftidx.hpp:131-143): sequence 0 carries a null lift, every
        haplotype the ins / del column sets of its alignment to sequence 0 (columns = reference bases + inserted bases;
        an insertion's columns precede the reference base it is placed before), all lifting onto sequence 0 (second = 0)."""
        from .index_build import Lifts
        second, length, ins_l, del_l = [0], [len(self.seqs[0])], [np.zeros(0, np.uint64)], [np.zeros(0, np.uint64)]
        for (pos, kind, ln) in self.variants:
            pos = np.asarray(pos, dtype=np.int64); kind = np.asarray(kind); ln = np.asarray(ln, dtype=np.int64)
            is_ins, is_del = kind == 1, kind == 2
            ins_before = np.cumsum(np.where(is_ins, ln, 0)) - np.where(is_ins, ln, 0)      # inserted columns before each event
            col = pos + ins_before
            ins_cols = [np.arange(c, c + l, dtype=np.int64) for c, l in zip(col[is_ins], ln[is_ins])]
            del_cols = [np.arange(c, c + l, dtype=np.int64) for c, l in zip(col[is_del], ln[is_del])]
            ins_l.append(np.concatenate(ins_cols).astype(np.uint64) if ins_cols else np.zeros(0, np.uint64))
            del_l.append(np.concatenate(del_cols).astype(np.uint64) if del_cols else np.zeros(0, np.uint64))
            second.append(0)
            length.append(len(self.seqs[0]) + int(ln[is_ins].sum()))
        return Lifts.from_lists(second, length, ins_l, del_l)

    @property
    def text(self) -> np.ndarray:
        parts = []
        for s in self.seqs:
            parts.append(s)
            parts.append(np.full(self.w, SEP_SEQ, dtype=np.uint8))
        parts.append(np.full(self.w - 1, SEP_END, dtype=np.uint8))
        return np.concatenate(parts)

    @property
    def seq_starts(self) -> np.ndarray:
        """onsets in text coordinates: 0, len0+w, ... , sum(len_i+w)  (k+1 values)."""
        on = [0]
        for s in self.seqs:
            on.append(on[-1] + len(s) + self.w)
        return np.asarray(on, dtype=np.uint64)


def make_pangenome(base_len: int, n_haps: int, seed: int = 19, var_seed: int = 12,
                   site_spacing: int = 1800, w: int = 10,
                   contig: str = "chr19", repeat_frac: float = 0.0, rep_seed: int = 1919, base: np.ndarray = None) -> Pangenome:
    """repeat_frac > 0: interspersed repeats: segments of 300-3000 bp are copied to random places with 5 %
    divergence until that fraction of the base genome is repeat copies, so that MEMs have many occurrences and the phi walks,
    the per-genome cap and the chaining see more than one locus per haplotype."""
    rng = np.random.Generator(np.random.MT19937(seed))
    g0 = _ACGT[rng.integers(0, 4, size=base_len, dtype=np.uint8)]
    if base is not None:                  # a given base genome (tests plant real reads in it)
        g0 = np.ascontiguousarray(base, dtype=np.uint8).copy()
        base_len = len(g0)
    if repeat_frac > 0:
        rr = np.random.Generator(np.random.MT19937(rep_seed))
        code0 = np.full(256, 0, dtype=np.uint8)
        code0[_ACGT] = np.arange(4, dtype=np.uint8)
        covered = 0
        n_fam = max(1, int(base_len * repeat_frac / 1650 / 8))       # families of ~8 copies each
        for _ in range(n_fam):
            ln = int(rr.integers(300, 3001))
            src = int(rr.integers(0, base_len - ln))
            unit = g0[src:src + ln].copy()
            for _c in range(8):
                if covered >= base_len * repeat_frac:
                    break
                dst = int(rr.integers(0, base_len - ln))
                cp = unit.copy()
                k = rr.random(ln) < 0.05
                cp[k] = _ACGT[(code0[cp[k]] + rr.integers(1, 4, size=int(k.sum()), dtype=np.uint8)) & 3]
                g0[dst:dst + ln] = cp
                covered += ln
    seqs = [g0]
    names = [contig]
    vr = np.random.Generator(np.random.MT19937(var_seed))
    n_sites = max(1, base_len // site_spacing) if n_haps > 0 else 0
    # sites: sorted, at least 64 apart so that a deletion (<= 50) never reaches the next one
    pos = np.sort(vr.choice(max(1, (base_len - 128) // 64), size=min(n_sites, max(1, (base_len - 128) // 64)),
                            replace=False)) * 64 + 32
    n_sites = len(pos)
    kind = vr.random(n_sites)           # <0.85 SNP, <0.925 ins, else del
    ilen = np.minimum(vr.geometric(1.0 / 3.0, size=n_sites), 50)
    af = vr.beta(0.5, 0.5, size=n_sites)
    snp_shift = vr.integers(1, 4, size=n_sites)
    ins_bases = _ACGT[vr.integers(0, 4, size=(n_sites, 50), dtype=np.uint8)]
    code = np.full(256, 255, dtype=np.uint8)
    code[_ACGT] = np.arange(4, dtype=np.uint8)
    variants = []
    for h in range(n_haps):
        carry = vr.random(n_sites) < af
        pieces = []
        prev = 0
        vp, vk, vl = [], [], []
        for s in np.nonzero(carry)[0]:
            p = int(pos[s])
            if kind[s] < 0.85:
                pieces.append(g0[prev:p])
                pieces.append(_ACGT[[(int(code[g0[p]]) + int(snp_shift[s])) & 3]])
                prev = p + 1
                vp.append(p); vk.append(0); vl.append(1)
            elif kind[s] < 0.925:
                pieces.append(g0[prev:p])
                pieces.append(ins_bases[s, : int(ilen[s])])
                prev = p
                vp.append(p); vk.append(1); vl.append(int(ilen[s]))
            else:
                pieces.append(g0[prev:p])
                prev = p + int(ilen[s])
                vp.append(p); vk.append(2); vl.append(int(ilen[s]))
        pieces.append(g0[prev:])
        seqs.append(np.concatenate(pieces))
        names.append("S%d_H%d_%s" % (h // 2 + 1, h % 2 + 1, contig))
        variants.append((np.asarray(vp, dtype=np.int64), np.asarray(vk, dtype=np.int8), np.asarray(vl, dtype=np.int64)))
    return Pangenome(seqs=seqs, names=names, w=w, variants=variants)


def revcomp(reads: np.ndarray) -> np.ndarray:
    """Reverse-complement of a [N, L] uint8 matrix (ACGT only complemented)."""
    return _COMP[reads[:, ::-1]]


def make_reads(pg: Pangenome, n_reads: int, read_len: int = 150, seed: int = 150,
               sub_rate: float = 0.01, indel_rate: float = 0.0005,
               n_rate: float = 0.0) -> np.ndarray:
    """[n_reads, read_len] uint8 ASCII reads (fixed length)."""
    rng = np.random.Generator(np.random.MT19937(seed))
    hap = rng.integers(0, len(pg.seqs), size=n_reads)
    out = np.empty((n_reads, read_len), dtype=np.uint8)
    pad = 4
    col = np.arange(read_len + pad)
    for h, s in enumerate(pg.seqs):
        sel = np.nonzero(hap == h)[0]
        if len(sel) == 0:
            continue
        if len(s) < read_len + pad:
            raise ValueError("sequence shorter than a read")
        start = rng.integers(0, len(s) - read_len - pad + 1, size=len(sel))
        win = s[start[:, None] + col[None, :]]          # [k, L+pad]
        # at most one indel per read (rate is per base)
        has = rng.random(len(sel)) < indel_rate * read_len
        ipos = rng.integers(1, read_len - 1, size=len(sel))
        is_ins = rng.random(len(sel)) < 0.5
        base_idx = np.arange(read_len)[None, :].repeat(len(sel), 0)
        # deletion: skip one reference base at ipos
        del_rows = has & ~is_ins
        base_idx[del_rows] += (base_idx[del_rows] >= ipos[del_rows, None])
        # insertion: positions > ipos read one base earlier, ipos gets a random base
        ins_rows = has & is_ins
        base_idx[ins_rows] -= (base_idx[ins_rows] > ipos[ins_rows, None])
        r = np.take_along_axis(win, base_idx, axis=1)
        rnd = _ACGT[rng.integers(0, 4, size=len(sel), dtype=np.uint8)]
        rows = np.nonzero(ins_rows)[0]
        r[rows, ipos[rows]] = rnd[rows]
        out[sel] = r
    # substitutions
    code = np.full(256, 0, dtype=np.uint8)
    code[_ACGT] = np.arange(4, dtype=np.uint8)
    sub = rng.random(out.shape) < sub_rate
    shift = rng.integers(1, 4, size=out.shape, dtype=np.uint8)
    out = np.where(sub, _ACGT[(code[out] + shift) & 3], out)
    if n_rate > 0:
        out = np.where(rng.random(out.shape) < n_rate, np.uint8(ord("N")), out)
    # strand
    rc = rng.random(n_reads) < 0.5
    out[rc] = revcomp(out[rc])
    return np.ascontiguousarray(out)


def make_pairs(pg: Pangenome, n_pairs: int, read_len: int = 150, seed: int = 350, ins_mean: float = 350.0, ins_sd: float = 30.0,
               sub_rate: float = 0.005):
    """FR read pairs: fragments of length N(ins_mean, ins_sd) (at least 2 L + 10) from uniform haplotypes / positions, substitutions at sub_rate,
    mate 1 = the fragment's first L bases, mate 2 = the reverse complement of its last L; half of the fragments come from the other strand (the
    mates swap).  Returns (mates [2 n, L] uint8 interleaved: rows 2p and 2p + 1 are pair p, insert sizes [n])."""
    rng = np.random.Generator(np.random.MT19937(seed))
    L = read_len
    hap = rng.integers(0, len(pg.seqs), size=n_pairs)
    ins = np.maximum(2 * L + 10, rng.normal(ins_mean, ins_sd, size=n_pairs)).astype(np.int64)
    out = np.empty((2 * n_pairs, L), dtype=np.uint8)
    col = np.arange(L)
    for h, s in enumerate(pg.seqs):
        sel = np.nonzero(hap == h)[0]
        if len(sel) == 0:
            continue
        start = (rng.random(len(sel)) * (len(s) - ins[sel])).astype(np.int64)
        a = s[start[:, None] + col[None, :]]
        b = s[(start + ins[sel] - L)[:, None] + col[None, :]]
        out[2 * sel] = a
        out[2 * sel + 1] = revcomp(b)
    code = np.full(256, 0, dtype=np.uint8)
    code[_ACGT] = np.arange(4, dtype=np.uint8)
    sub = rng.random(out.shape) < sub_rate
    shift = rng.integers(1, 4, size=out.shape, dtype=np.uint8)
    out = np.where(sub, _ACGT[(code[out] + shift) & 3], out)
    swap = np.nonzero(rng.random(n_pairs) < 0.5)[0]
    tmp = out[2 * swap].copy(); out[2 * swap] = out[2 * swap + 1]; out[2 * swap + 1] = tmp
    return np.ascontiguousarray(out), ins


def make_pair_names(n_pairs: int, prefix: str = "simulated"):
    """names `simulated.<p>/1`, `simulated.<p>/2` of the interleaved mates, ragged bytes + offsets"""
    names = [("%s.%d/%d" % (prefix, p, k)).encode() for p in range(n_pairs) for k in (1, 2)]
    off = np.zeros(2 * n_pairs + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in names])
    return np.frombuffer(b"".join(names), dtype=np.uint8).copy(), off


READ_BLOCK = 250000


def make_reads_range(pg: Pangenome, lo: int, hi: int, read_len: int = 150, seed: int = 1500, threads: int = 1, **kw) -> np.ndarray:
    """Reads [lo, hi) of an unbounded seeded read set made of blocks of READ_BLOCK reads (block b = make_reads(seed + b)): any
    contiguous range can be generated on its own, so that a rank makes only its shard of a sharded read set.  threads > 1: the
    blocks side by side (same reads: every block has its own generator)."""
    def part(b):
        blk = make_reads(pg, READ_BLOCK, read_len, seed=seed + b, **kw)
        return blk[max(lo, b * READ_BLOCK) - b * READ_BLOCK: min(hi, (b + 1) * READ_BLOCK) - b * READ_BLOCK]
    blocks = range(lo // READ_BLOCK, (max(hi, lo + 1) - 1) // READ_BLOCK + 1)
    if threads > 1 and len(blocks) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(min(threads, len(blocks))) as ex:
            parts = list(ex.map(part, blocks))
    else:
        parts = [part(b) for b in blocks]
    return np.ascontiguousarray(np.concatenate(parts)) if parts else np.zeros((0, read_len), np.uint8)


def make_names_range(lo: int, hi: int, prefix: str = "simulated"):
    """names of reads [lo, hi) of the set (`simulated.<i>`), ragged bytes + offsets"""
    names = [b"%s.%d" % (prefix.encode(), i) for i in range(lo, hi)]
    off = np.zeros(hi - lo + 1, dtype=np.uint64)
    off[1:] = np.cumsum(np.fromiter((len(x) for x in names), dtype=np.int64, count=hi - lo))
    return np.frombuffer(b"".join(names), dtype=np.uint8).copy(), off


def write_fastq(path: str, reads: np.ndarray, prefix: str = "simulated") -> None:
    with open(path, "wb") as f:
        q = b"I" * reads.shape[1]
        for i in range(reads.shape[0]):
            f.write(b"@%s.%d\n" % (prefix.encode(), i))
            f.write(reads[i].tobytes())
            f.write(b"\n+\n")
            f.write(q)
            f.write(b"\n")


def write_fasta(path: str, pg: Pangenome, width: int = 60) -> None:
    with open(path, "wb") as f:
        for name, s in zip(pg.names, pg.seqs):
            f.write(b">" + name.encode() + b"\n")
            b = s.tobytes()
            for i in range(0, len(b), width):
                f.write(b[i:i + width])
                f.write(b"\n")


def make_names(n: int, prefix: str = "simulated"):
    """Read names `simulated.<i>` as ragged bytes + offsets, the layout moni_align_batch takes."""
    names = [("%s.%d" % (prefix, i)).encode() for i in range(n)]
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in names])
    return np.frombuffer(b"".join(names), dtype=np.uint8).copy(), off
