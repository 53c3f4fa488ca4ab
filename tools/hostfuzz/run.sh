#!/bin/bash
# CPU, AddressSanitizer + UBSan: the HOST code that walks untrusted frame headers for the frame-parallel reader (walk_lz4_frames /
# walk_zstd_frames in zpack_amd/csrc/zpk_codec.hip, lifted out of the source as it is — no copy to drift) on mutated frame sequences:
# no read outside the entry, and every accepted plan tiles its entry exactly.  tools/hostfuzz/run.sh [iterations]
set -e
cd "$(dirname "$0")/../.."
mkdir -p /tmp/zpk_hostfuzz
python3 - <<'PY'
s = open("zpack_amd/csrc/zpk_codec.hip").read()
a = s.index("struct BigSub {"); b = s.index("// the frames of entries [g0, g1) of `be` as one device batch")
t = open("tools/hostfuzz/walk_fuzz_main.cpp.in").read().replace("/*@WALKERS@*/", s[a:b])
open("/tmp/zpk_hostfuzz/walk_fuzz.cpp", "w").write(t)
PY
g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -std=c++17 -o /tmp/zpk_hostfuzz/walk_fuzz /tmp/zpk_hostfuzz/walk_fuzz.cpp
/tmp/zpk_hostfuzz/walk_fuzz ${1:-3000000}
