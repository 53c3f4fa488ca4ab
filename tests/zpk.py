"""Minimal .zpk container reader/writer for the tests (format: reference docs/specs.md, Appendix A of SURVEY.md)."""
import struct


def parse(archive):
    a = bytes(archive)
    assert a[:4] == b"ZPK\x15" and a[6:10] == b"ZPK\x14"
    assert a[-12:-8] == b"ZPK\x12"
    cdr = struct.unpack_from("<Q", a, len(a) - 8)[0]
    assert a[cdr:cdr + 4] == b"ZPK\x13"
    count, block = struct.unpack_from("<QQ", a, cdr + 4)
    p = cdr + 20
    out = []
    for _ in range(count):
        (nl,) = struct.unpack_from("<H", a, p)
        name = a[p + 2:p + 2 + nl].decode()
        off, cs, us, h, m = struct.unpack_from("<QQQQB", a, p + 2 + nl)
        out.append(dict(filename=name, offset=off, comp_size=cs, uncomp_size=us, hash=h, method=m))
        p += 2 + nl + 33
    return out


def assemble(payloads, entries):
    out = bytearray(b"ZPK\x15" + struct.pack("<H", 1) + b"ZPK\x14")
    for p in payloads:
        out += p
    cdr_off = len(out)
    body = bytearray()
    for (name, off, cs, us, h, m) in entries:
        nb = name.encode()
        body += struct.pack("<H", len(nb)) + nb + struct.pack("<QQQQB", off, cs, us, h, m)
    out += b"ZPK\x13" + struct.pack("<QQ", len(entries), len(body)) + body
    out += b"ZPK\x12" + struct.pack("<Q", cdr_off)
    return bytes(out)
