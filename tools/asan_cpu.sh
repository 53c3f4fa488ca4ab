#!/bin/bash
# CPU-only: the host library (zpack_amd/host/*.c) and the oracle built with AddressSanitizer + UBSan, the not-gpu tests run against them.
# (GPU ASan is not available on the pool; the device code is covered by the parity tests and the fuzzers instead.)
set -e
cd "$(dirname "$0")/.."
tmp=$(mktemp -d)
pre="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
cp zpack_amd/libzpack_amd.so $tmp/libzpack_amd.so; cp oracle/liboracle.so $tmp/liboracle.so
restore() { cp $tmp/libzpack_amd.so zpack_amd/libzpack_amd.so; cp $tmp/liboracle.so oracle/liboracle.so; rm -rf $tmp; }
trap restore EXIT
gcc -O1 -g -fPIC -shared -std=c11 -Wall -Wextra -D_FILE_OFFSET_BITS=64 -D_POSIX_C_SOURCE=200809L -fvisibility=hidden -fsanitize=address,undefined \
    -fno-omit-frame-pointer -Iinclude -o zpack_amd/libzpack_amd.so zpack_amd/host/*.c -Lzpack_amd -lzpk_codec -Wl,-rpath,$PWD/zpack_amd
make -s -C oracle -B liboracle.so CFLAGS="-O1 -g -fPIC -Wall -Wextra -std=c11 -fsanitize=address,undefined -fno-omit-frame-pointer"
LD_PRELOAD="$pre" ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 python -m pytest tests/test_cdr_cpu.py tests/test_abi_cpu.py tests/test_oracle_golden.py -x -q
