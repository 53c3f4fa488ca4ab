#!/usr/bin/env python3
"""Apply INTEGRATION.md section B to a SCRATCH COPY of the reference's lib/zpack_read.c and lib/zpack_write.c.

Reads the sources where they lie under /root/reference/lib, writes the patched copies ONLY under oracle/_ref/patched/ (git-ignored,
like every build product of the reference).  What changes: zpack_read_file and zpack_compress_file are replaced by the bodies in
oracle/integration/patch_{read,write}.inc.c (this repository's code: one call into the codec C-ABI each), the XXH3 of
zpack_add_written_file_entry becomes the codec's hash, and zpack_close_reader / zpack_close_writer drop the codec.  Everything else —
container parsing and emission, streaming, raw copies — stays the reference's own code, compiled unchanged.
TEST INFRASTRUCTURE: tests/test_gpu_integration_patch.py runs the reference's own test flows through the result."""
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(HERE), "_ref", "patched")


def replace_function(text, signature_regex, replacement):
    m = re.search(signature_regex, text, re.M)
    assert m, signature_regex
    end = text.index("\n}\n", m.start()) + 3
    return text[:m.start()] + replacement + text[end:]


def main():
    os.makedirs(OUT, exist_ok=True)
    rd = open(os.path.join(REF, "lib", "zpack_read.c")).read()
    rd = replace_function(rd, r"^int zpack_read_file\(zpack_reader\* reader, zpack_file_entry\* entry, zpack_u8\* buffer, size_t max_size, void\* dctx\)\n\{",
                          open(os.path.join(HERE, "patch_read.inc.c")).read())
    assert "memset(reader, 0, sizeof(zpack_reader));" in rd
    rd = rd.replace("memset(reader, 0, sizeof(zpack_reader));", "zpk_patch_drop(reader);\n    memset(reader, 0, sizeof(zpack_reader));")
    open(os.path.join(OUT, "zpack_read.c"), "w").write(rd)

    wr = open(os.path.join(REF, "lib", "zpack_write.c")).read()
    wr = replace_function(wr, r"^static int zpack_compress_file\(zpack_writer\* writer, zpack_u8\* buffer, size_t capacity,\n",
                          open(os.path.join(HERE, "patch_write.inc.c")).read())
    assert "entry->hash = XXH3_64bits(file->buffer, file->size);" in wr
    wr = wr.replace("entry->hash = XXH3_64bits(file->buffer, file->size);", "entry->hash = zpk_patch_hash;")
    assert "memset(writer, 0, sizeof(zpack_writer));" in wr
    wr = wr.replace("memset(writer, 0, sizeof(zpack_writer));", "zpk_patch_wdrop(writer);\n    memset(writer, 0, sizeof(zpack_writer));")
    open(os.path.join(OUT, "zpack_write.c"), "w").write(wr)
    print("patched copies under", OUT)


if __name__ == "__main__":
    main()
