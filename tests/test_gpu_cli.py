"""Batch-aware `t` and `x`: zpack_amd/zpk-batch over zpack_read_files_packed prints what the reference's
command_test / extract_files_i print (programs/commands.c:706-773, :413-487) — same lines, same per-entry verdicts — here checked
against the verdicts the COMPILED REFERENCE library gives for the same (damaged) archive, formatted the way commands.c formats them."""
import os
import subprocess

import numpy as np
import pytest

import zpack_amd
from benchdata import datagen as dg
from tests import zpk
from tests._libs import ZPackAPI, have_ref, ref

pytestmark = pytest.mark.gpu
EXE = os.path.join(os.path.dirname(zpack_amd.ZPACK_SO), "zpk-batch")


def _damaged_archive(tmp_path, stop_at=None):
    Z = ZPackAPI(zpack_amd.ZPACK_SO)
    rng = np.random.default_rng(3)
    files = [("dir%d/sub/f%03d.txt" % (i % 3, i), dg.fill(i % 4, 91, i, int(rng.integers(1, 200000))).tobytes()) for i in range(40)]
    files[7] = ("../../escape/../evil.txt", files[7][1])          # must land inside the output directory
    arcs = []
    for method, level in ((2, 0), (1, 3)):
        arcs.append(bytearray(Z.write_archive(files, method, level)))
    a = arcs[0]
    ents = zpk.parse(a)
    # entry 5: wrong hash in the CDR; entry 11: one payload byte flipped (hash mismatch or decode error, whatever the reference says)
    cdr = a.rfind(b"ZPK\x13")
    pos = cdr + 20
    for i, e in enumerate(ents):
        nl = int.from_bytes(a[pos:pos + 2], "little")
        if i == 5:
            a[pos + 2 + nl + 24] ^= 0xFF
        pos += 2 + nl + 33
    a[ents[11]["offset"] + ents[11]["comp_size"] // 2] ^= 0x5A
    if stop_at is not None:                                       # a frame that no longer decodes: magic destroyed
        a[ents[stop_at]["offset"]] ^= 0xFF
    p = tmp_path / "damaged.zpk"
    p.write_bytes(bytes(a))
    return str(p), bytes(a), files


def _reference_verdicts(arc, files):
    R = ref()
    rc, r, keep = R.open_memory(arc)
    assert rc == 0
    out = []
    for i, (name, data) in enumerate(files):
        rc, got = R.read_file(r, i, len(data))
        out.append((rc, got))
    R.close_reader(r)
    return out


@pytest.mark.skipif(not have_ref(), reason="needs the compiled reference for the expected verdicts")
@pytest.mark.parametrize("stop_at", [None, 23])
def test_t_prints_what_the_reference_prints(tmp_path, stop_at):
    path, arc, files = _damaged_archive(tmp_path, stop_at)
    verdicts = _reference_verdicts(arc, files)
    want = ["-- Reading archive: %s" % path, "-- Found %d files" % len(files), "-- Testing files..."]
    corrupt, status = 0, 0
    for (name, _), (rc, _) in zip(files, verdicts):
        want.append("  %s" % name)
        if rc == 15:
            want.append("-- File is corrupted!"); corrupt += 1
        elif rc != 0:
            want.append('Error: Failed to decompress "%s" (error %d)' % (name, rc)); status = 1
            break
    if status == 0:
        want += ["-- Done.", "-- Corrupted files: %d/%d" % (corrupt, len(files))]
    p = subprocess.run([EXE, "t", path], capture_output=True, text=True, timeout=300)
    assert p.stdout.splitlines() == want, p.stdout[-2000:] + p.stderr[-500:]
    assert p.returncode == status
    assert corrupt >= 1 and (stop_at is None or status == 1)


@pytest.mark.skipif(not have_ref(), reason="needs the compiled reference for the expected verdicts")
def test_x_writes_the_files_and_counts_errors(tmp_path):
    path, arc, files = _damaged_archive(tmp_path, stop_at=23)
    verdicts = _reference_verdicts(arc, files)
    out = tmp_path / "out"
    p = subprocess.run([EXE, "x", path, "-o", str(out)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-1000:]
    lines = p.stdout.splitlines()
    errors = 0
    for (name, data), (rc, got) in zip(files, verdicts):
        safe = "/".join(c for c in name.replace("\\", "/").split("/") if c not in ("", ".", ".."))
        f = out / safe
        assert os.path.realpath(f).startswith(os.path.realpath(out) + os.sep)
        if rc in (0, 15):                                        # a corrupted file is still written, with a warning (commands.c:366-368)
            assert f.read_bytes() == got[:len(data)], name
            assert ("Warning: File is corrupted (file hash mismatch)" in lines[lines.index("  %s" % name) + 1]) == (rc == 15)
        else:
            errors += 1
            assert 'Error: Failed to extract "%s" (error %d)' % (name, rc) in lines
            assert not f.exists()
    assert lines[-1] == "-- Done." and (errors == 0 or "-- Errors: %d" % errors in lines)
    assert errors == 1
