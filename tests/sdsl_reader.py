"""Readers for the sdsl-serialised pieces of the reference's index files (TEST INFRASTRUCTURE).
Layouts per SURVEY.md App. B, verified byte-exactly against the reference's fixture data/Chr21.10.ldx:
  int_vector<0>      u64 bit_len; u8 width; u64 words[ceil(bit_len/64)]
  bit_vector         u64 bit_len; u64 words[...]
  sd_vector<>        u64 size; u8 wl; int_vector<0> low; bit_vector high; select_support_mcl<1>; select_support_mcl<0>
  select_support_mcl u64 arg_cnt; if arg_cnt: int_vector<0> superblock; bit_vector mini_or_long; per superblock one int_vector<0>
"""
import struct

import numpy as np


class Cursor:
    def __init__(self, buf: bytes, off: int = 0):
        self.b, self.o = buf, off

    def u64(self):
        v = struct.unpack_from("<Q", self.b, self.o)[0]
        self.o += 8
        return v

    def u8(self):
        v = self.b[self.o]
        self.o += 1
        return v

    def words(self, bit_len):
        n = (bit_len + 63) // 64
        a = np.frombuffer(self.b, dtype="<u8", count=n, offset=self.o)
        self.o += 8 * n
        return a


def read_int_vector(c: Cursor):
    bit_len = c.u64()
    width = c.u8()
    w = c.words(bit_len)
    return bit_len, width, w


def read_bit_vector(c: Cursor):
    bit_len = c.u64()
    return bit_len, c.words(bit_len)


def read_select_support_mcl(c: Cursor):
    arg_cnt = c.u64()
    if arg_cnt == 0:
        return 0
    read_int_vector(c)               # superblock
    read_bit_vector(c)               # mini_or_long (may be empty)
    for _ in range((arg_cnt + 4095) >> 12):
        read_int_vector(c)
    return arg_cnt


def unpack(words: np.ndarray, width: int, count: int) -> np.ndarray:
    """count fixed-width little-endian fields from a word array"""
    if count == 0:
        return np.zeros(0, dtype=np.uint64)
    bits = np.unpackbits(words.view(np.uint8), bitorder="little")[: count * width].reshape(count, width)
    weights = (np.uint64(1) << np.arange(width, dtype=np.uint64))
    return (bits.astype(np.uint64) * weights).sum(axis=1).astype(np.uint64)


class SdVector:
    """positions of the ones of an Elias-Fano coded bit-vector"""

    def __init__(self, c: Cursor):
        self.size = c.u64()
        self.wl = c.u8()
        low_bits, low_w, low_words = read_int_vector(c)
        hi_len, hi_words = read_bit_vector(c)
        self.m = read_select_support_mcl(c)          # number of ones
        read_select_support_mcl(c)
        n1 = low_bits // low_w if low_w else 0
        low = unpack(low_words, low_w, n1) if low_w else np.zeros(n1, dtype=np.uint64)
        hb = np.unpackbits(hi_words.view(np.uint8), bitorder="little")[:hi_len]
        one_pos = np.nonzero(hb)[0].astype(np.uint64)                # position of the i-th 1 in `high`
        high = one_pos - np.arange(len(one_pos), dtype=np.uint64)    # number of zeros before it = high part
        assert len(high) == n1, (len(high), n1)
        self.ones = (high << np.uint64(self.wl)) | low
        assert (np.diff(self.ones.astype(np.int64)) > 0).all()

    def rank1(self, i):      # ones in [0, i)
        return int(np.searchsorted(self.ones, np.uint64(i), side="left"))

    def rank0(self, i):
        return i - self.rank1(i)

    def select0(self, k):    # position of the k-th zero, k >= 1
        # zeros before the j-th one: ones[j] - j
        zb = self.ones.astype(np.int64) - np.arange(len(self.ones), dtype=np.int64)
        j = int(np.searchsorted(zb, k, side="left"))   # number of ones before the k-th zero
        return k - 1 + j


class Lift:
    """levioSAM lift::Lift: three sd_vectors over alignment columns (ins, del, snp)"""

    def __init__(self, c: Cursor):
        self.ins, self.dele, self.snp = SdVector(c), SdVector(c), SdVector(c)

    def lift_pos(self, p):   # s2 (haplotype) position -> s1 (reference) position
        return self.ins.rank0(self.dele.select0(p + 1))


def read_ldx_old_layout(buf: bytes):
    """liftidx::load for the layout of data/Chr21.10.ldx (no `w` field after `u`): seqidx.hpp:215-238, liftidx.hpp:131-143"""
    c = Cursor(buf)
    u = c.u64()
    starts = SdVector(c)
    n_names = c.u64()
    names = []
    for _ in range(n_names):
        ln = c.u64()
        names.append(buf[c.o:c.o + ln].decode())
        c.o += ln
    n_lifts = c.u64()
    lifts = []
    for _ in range(n_lifts):
        second = c.u64()
        lifts.append((Lift(c), second))
    return {"u": u, "starts": starts, "names": names, "lifts": lifts, "consumed": c.o}
