#!/bin/bash
# CPU, AddressSanitizer + UBSan: the HOST code that walks untrusted frame headers for the frame-parallel and block-parallel readers
# (walk_lz4_frames / walk_zstd_frames / walk_lz4_single / walk_zstd_single / zpj_parse_block in zpack_amd/csrc/zpk_codec.hip, lifted out of
# the source as it is — no copy to drift) on mutated frames:
# no read outside the entry, and every accepted plan tiles its entry exactly.  tools/hostfuzz/run.sh [iterations]
set -e
cd "$(dirname "$0")/../.."
mkdir -p /tmp/zpk_hostfuzz
python3 - <<'PY'
import re
s = open("zpack_amd/csrc/zpk_codec.hip").read()
# the walkers as they are in the source (no copy to drift): frame sequences + one LZ4 frame; then one Zstandard frame
a = s.index("struct BigSub {"); b = s.index("// The common second half of the block-parallel readers")
c = s.index("// Bytes an FSE table description (RFC 8878 4.1.1) takes"); d = s.index("// -> ZPK_OK with redo = 0: the entry is decoded, its XXH3 is the expected one")
l4 = open("zpack_amd/csrc/lz4_pj.h").read(); zs = open("zpack_amd/csrc/zstd_pj.h").read()
types = "#define PJ_BLOCK 65536u\n#define ZPJ_BLOCK (128u << 10)\n#define ZPJ_NONE 0xFFFFFFFFu\n#define ZPK_PJ_MIN_BLOCKS 4u\n#define ZPK_ZPJ_MIN_BLOCKS 2u\n"
types += re.search(r"struct PjBlock \{[^}]*\};", l4).group(0) + "\n" + re.search(r"struct ZpjBlock \{.*?\n\};", zs, re.S).group(0) + "\n"
t = open("tools/hostfuzz/walk_fuzz_main.cpp.in").read().replace("/*@TYPES@*/", types).replace("/*@WALKERS@*/", s[a:b] + s[c:d])
open("/tmp/zpk_hostfuzz/walk_fuzz.cpp", "w").write(t)
PY
g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -std=c++17 -o /tmp/zpk_hostfuzz/walk_fuzz /tmp/zpk_hostfuzz/walk_fuzz.cpp
/tmp/zpk_hostfuzz/walk_fuzz ${1:-3000000}
