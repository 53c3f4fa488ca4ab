"""A Zstandard frame whose literals use a 12-bit Huffman code (the format's maximum; libzstd's encoder stops at 11), built
by hand from RFC 8878 4.2: one Compressed_Literals_Block with a single stream and a direct weight description, no
sequences.  The GPU decoders keep a separate table layout for 12-bit codes (zstd_wg.h: huf_build / huf_run), which no
frame written by libzstd reaches."""
import random
import struct


def make_frame(seed=1, nlit=200):
    rng = random.Random(seed)
    lits = bytes(rng.choice([12] * 8 + [11] * 4 + [10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0]) for _ in range(nlit))
    w = [1, 1] + list(range(2, 13))                      # weights of symbols 0..12: sum of 2^(w-1) = 2^12
    mb = 12
    code, pos = {}, 0
    for ww in range(1, mb + 1):                          # the decoder's table order: weight classes ascending, natural symbol order
        for sy in range(13):
            if w[sy] == ww:
                code[sy] = (pos >> (ww - 1), mb + 1 - ww)
                pos += 1 << (ww - 1)
    assert pos == 1 << mb
    acc, n = 0, 0
    for sy in reversed(lits):                            # last symbol first, LSB-first container, then the end mark
        c, nb = code[sy]
        acc |= c << n
        n += nb
    acc |= 1 << n
    n += 1
    stream = acc.to_bytes((n + 7) // 8, "little")
    listed = w[:12]                                      # the last symbol's weight is implied
    tree = bytes([127 + 12]) + bytes((listed[i] << 4) | listed[i + 1] for i in range(0, 12, 2))
    csize = len(tree) + len(stream)
    assert nlit < 1024 and csize < 1024
    litsec = (2 | (0 << 2) | (nlit << 4) | (csize << 14)).to_bytes(3, "little") + tree + stream
    block = litsec + b"\x00"                             # Number_of_Sequences = 0
    bh = 1 | (2 << 1) | (len(block) << 3)
    fhd = bytes([0x20, nlit]) if nlit < 256 else bytes([0x60]) + struct.pack("<H", nlit - 256)      # single segment, 1- or 2-byte FCS
    frame = struct.pack("<I", 0xFD2FB528) + fhd + bh.to_bytes(3, "little") + block
    return frame, lits
