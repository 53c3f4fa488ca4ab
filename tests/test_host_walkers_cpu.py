"""CPU: the host code that walks untrusted frame and block headers for the frame-parallel and block-parallel readers and for the bounded
stream steps (walk_lz4_frames, walk_zstd_frames, walk_lz4_single, walk_zstd_single, zpj_parse_block, zpj_parse_frame_header, zpj_ncount_len in
zpack_amd/csrc/zpk_codec.hip — lifted out of the source at run time, no copy to drift) under AddressSanitizer + UBSan on mutated frames:
no read outside the entry, every accepted plan tiles its entry, every accepted block lies inside it (tools/hostfuzz; a longer run is
profiles/r05/r05_hostfuzz_asan_ubsan.txt)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++ with the sanitizer runtimes")
def test_host_walkers_under_asan_ubsan():
    p = subprocess.run(["bash", os.path.join(ROOT, "tools", "hostfuzz", "run.sh"), "60000"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-2000:])
    assert "every accepted block inside its entry" in p.stdout and "every accepted plan tiles its entry exactly" in p.stdout, p.stdout[-1000:]
