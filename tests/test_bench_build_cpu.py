"""CPU checks of the measurement / build plumbing that the driver depends on: `bench.py --gpus N` starts its own ranks as a CHILD
process before anything touches torch or HIP, and `zpack_amd.build` rebuilds exactly when the inputs of a target changed (content
stamps: mtimes do not survive a push to a fresh box)."""
import os
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_bench_spawns_ranks_as_a_child_process(monkeypatch):
    import bench
    calls = {}

    def fake_run(cmd, env=None, stdout=None, text=None):
        calls["cmd"], calls["env"] = cmd, env
        return types.SimpleNamespace(returncode=0, stdout='noise\n{"metric": "x", "n_gpus": 2}\n')

    import subprocess
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(bench, "_visible_gpus", lambda: 1)
    assert "torch" not in bench.spawn_ranks.__code__.co_names            # the parent never imports torch: no HIP before the spawn
    rc = bench.spawn_ranks(2, ["--gpus", "2", "--workload", "c4_mixed"])
    cmd = calls["cmd"]
    assert rc == 0 and cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert "--master-addr" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "2", "--workload", "c4_mixed"] and os.path.basename(cmd[-5]) == "bench.py"
    assert calls["env"]["ZPK_BENCH_REHEARSAL"] == "1"                     # fewer GPUs than ranks: all ranks share cuda:0 over gloo
    assert calls["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # enough GPUs: the real thing, no rehearsal flag
    monkeypatch.setattr(bench, "_visible_gpus", lambda: 8)
    monkeypatch.delenv("ZPK_BENCH_REHEARSAL", raising=False)
    assert bench.spawn_ranks(8, ["--gpus", "8"]) == 0 and "ZPK_BENCH_REHEARSAL" not in calls["env"]
    # a rehearsal is limited to 4 ranks on one card (the GPU boxes allow few processes on a card)
    monkeypatch.setattr(bench, "_visible_gpus", lambda: 1)
    assert bench.spawn_ranks(8, ["--gpus", "8"]) == 2


def test_bench_default_scaling_by_workload():
    src = open(os.path.join(ROOT, "bench.py")).read()
    # c4_mixed is ONE archive sharded over the ranks (125 000 entries per GPU; --entries / --scaling strong fix the total instead)
    assert 'one_archive = args.workload == "c4_mixed" or args.scaling == "strong"' in src
    assert 'n_total = w["n"] if fixed_total else w["n"] * world' in src and "first=lo" in src       # every rank builds only its slice
    assert 'if args.gpus > 1 and "WORLD_SIZE" not in os.environ:' in src


def test_build_stamps_follow_the_contents(tmp_path):
    from zpack_amd import build as b
    src = tmp_path / "a.c"
    src.write_text("int x;\n")
    target = str(tmp_path / "t.so")
    dig = b._stale(target, [str(src)], "flags")
    assert dig is not None                                                # no target yet
    open(target, "w").write("bin")
    b._stamp(target, dig)
    assert b._stale(target, [str(src)], "flags") is None                  # up to date, whatever the mtimes say
    os.utime(str(src), (1, 1))
    assert b._stale(target, [str(src)], "flags") is None
    src.write_text("int y;\n")
    assert b._stale(target, [str(src)], "flags") is not None              # contents changed
    src.write_text("int x;\n")
    assert b._stale(target, [str(src)], "other flags") is not None        # flags changed
