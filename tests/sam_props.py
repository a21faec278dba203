"""Properties of SAM records that can be checked without any code of the restatement or the kernels: CIGAR arithmetic, MD / NM
re-derived from the sequences, AS as the score of the path the OA CIGAR describes, the MAPQ formula (mapq.hpp:146-184) and the
min_score threshold (aligner_ksw2.hpp:394) re-evaluated in Python.  They do not prove parity with upstream; they bound how wrong a
shared misreading of the reference could be."""
import math
import re

import numpy as np

CIG = re.compile(rb"(\d+)([MID])")
COMP = bytes.maketrans(b"ACGTacgt", b"TGCATGCA")


def parse_cigar(c):
    ops = [(int(n), op) for n, op in CIG.findall(c)]
    assert b"".join(b"%d%s" % (n, op) for n, op in ops) == c, c
    return ops


def md_nm(seq: bytes, ref: bytes, ops):
    """write_MD_core's rule spelled from the SAM specification: MD and NM of `seq` against `ref` under the CIGAR."""
    q = t = 0
    run = 0
    nm = 0
    md = []
    for n, op in ops:
        if op == b"M":
            for k in range(n):
                a, b = seq[q + k:q + k + 1].upper(), ref[t + k:t + k + 1].upper()
                a = a if a in (b"A", b"C", b"G", b"T") else b"N"
                b = b if b in (b"A", b"C", b"G", b"T") else b"N"
                if a != b:
                    md.append(b"%d%s" % (run, b)); run = 0; nm += 1
                else:
                    run += 1
            q += n; t += n
        elif op == b"I":
            q += n; nm += n
        else:
            md.append(b"%d^" % run)
            md.append(b"".join((c if c in b"ACGT" else b"N") for c in [ref[t + k:t + k + 1].upper() for k in range(n)]))
            run = 0; t += n; nm += n
    if run > 0:
        md.append(b"%d" % run)
    return b"".join(md), nm


def path_score(seq: bytes, ref: bytes, ops, a=2, b=4, q=4, e=2):
    s = 0
    qi = ti = 0
    for n, op in ops:
        if op == b"M":
            for k in range(n):
                x, y = seq[qi + k:qi + k + 1].upper(), ref[ti + k:ti + k + 1].upper()
                if x not in (b"A", b"C", b"G", b"T") or y not in (b"A", b"C", b"G", b"T"):
                    s -= e
                else:
                    s += a if x == y else -b
            qi += n; ti += n
        elif op == b"I":
            s -= q + n * e; qi += n
        else:
            s -= q + n * e; ti += n
    return s


def mapq_bwa(score, score2, rlen, qlen, min_len=25, a=2, b=4):
    l = max(rlen, qlen)
    sub = score2 if score2 else min_len * a
    if sub >= score:
        return 0
    identity = 1.0 - (l * a - score) / (a + b) / l
    if score == 0:
        mapq = 0
    else:
        tmp = 1.0 if l < 50 else int(math.log(50.0)) / math.log(l)
        tmp *= identity * identity
        mapq = int(6.02 * (score - sub) / a * tmp * tmp + .499)
    mapq = min(60, max(0, mapq))
    return int(mapq * 1.0 + .499)


def check_records(sam: bytes, seqs, names, reads, max_records=None, stride=1):
    """seqs / names: the pangenome's sequences (uint8 arrays) and their names; reads: [N, L] uint8 input reads in SAM order.
    Returns counters of what was checked."""
    by_name = {n.encode(): s.tobytes() for n, s in zip(names, seqs)}
    n_aligned = n_as = n_skipped_as = 0
    lines = [l for l in sam.split(b"\n") if l and not l.startswith(b"@")]
    assert len(lines) == len(reads)
    for i in range(0, len(lines), stride):
        if max_records is not None and n_aligned >= max_records:
            break
        f = lines[i].split(b"\t")
        flag = int(f[1])
        rd = reads[i].tobytes()
        if flag & 4:
            assert f[2] == b"*" and f[5] == b"*" and f[9] == rd
            continue
        n_aligned += 1
        seq = f[9]
        assert seq == (rd[::-1].translate(COMP) if flag & 16 else rd)
        tags = {t[:2]: t[5:] for t in f[11:]}
        AS, NM = int(tags[b"AS"]), int(tags[b"NM"])
        ZS = int(tags.get(b"ZS", b"0"))
        L = len(seq)
        assert 20 + 8 * math.log(L) - 1 < AS <= 2 * L
        oa = tags[b"OA"].rstrip(b";").split(b",")
        o_name, o_pos, o_strand, o_cig, o_mapq, o_nm = oa[0], int(oa[1]), oa[2], oa[3], int(oa[4]), int(oa[5])
        assert o_strand == (b"-" if flag & 16 else b"+") and o_mapq == int(f[4])
        o_ops = parse_cigar(o_cig)
        assert sum(n for n, op in o_ops if op in (b"M", b"I")) == L
        o_ref = by_name[o_name]
        o_span = sum(n for n, op in o_ops if op in (b"M", b"D"))
        assert 1 <= o_pos and o_pos - 1 + o_span <= len(o_ref)                      # inside one sequence (seqidx.hpp:164-167)
        _, nm_o = md_nm(seq, o_ref[o_pos - 1:o_pos - 1 + o_span], o_ops)
        assert nm_o == o_nm, (i, nm_o, o_nm)
        # AS is the score of the path the pangenome-side CIGAR spells (match 2, mismatch -4, gap 4 + 2k, N -2), except where the reference
        # takes a shortcut that is not an affine-gap path: zero-length operations (aligner_ksw2.hpp:2939) and the min(4+2l, 13+l) insertion
        if all(n > 0 for n, _ in o_ops) and all(n <= 9 for n, op in o_ops if op == b"I"):
            assert path_score(seq, o_ref[o_pos - 1:o_pos - 1 + o_span], o_ops) == AS, (i, lines[i][:200])
            n_as += 1
        else:
            n_skipped_as += 1
        if f[2] == b"*":
            continue                                                                # unmapped after lifting
        ops = parse_cigar(f[5])
        assert sum(n for n, op in ops if op in (b"M", b"I")) == L and all(n > 0 for n, _ in ops)
        ref = by_name[f[2]]
        pos = int(f[3])
        span = sum(n for n, op in ops if op in (b"M", b"D"))
        assert 1 <= pos and pos - 1 + span <= len(ref)
        md, nm = md_nm(seq, ref[pos - 1:pos - 1 + span], ops)
        assert nm == NM and md == tags[b"MD"], (i, md, tags[b"MD"], nm, NM)
        assert int(f[4]) == mapq_bwa(AS, ZS, span, L), (i, f[4], AS, ZS, span, L)
    return {"aligned": n_aligned, "as_checked": n_as, "as_skipped": n_skipped_as}
