"""Safety net of the N > 1 path, the parts that need no GPU: the launcher's watchdog (`bench._watch`:
first failing rank stops the others, overall deadline), the fresh-child retry on the host-driven loop after a stalled captured loop
(`bench.launch_ranks`, `bench.supervise_rank`: status 86 / marker file -> RDO_DP_GRAPH=0, never a re-exec), and the heartbeat wait of
the captured data-parallel loop (`UnitEngine._hb_wait`: progress -> returns, none for DP_STALL_S -> DpStallError)."""
import os
import subprocess
import sys
import threading
import time
import types

import pytest
import torch

import bench


def _sleeper(seconds):
    return subprocess.Popen([sys.executable, "-c", f"import time; time.sleep({seconds})"])


def test_watch_stops_the_other_ranks_when_one_fails():
    bad = subprocess.Popen([sys.executable, "-c", f"import sys; sys.exit({bench.STALL_STATUS})"])
    slow = _sleeper(120)
    t0 = time.monotonic()
    rc = bench._watch([bad, slow], time.monotonic() + 60, "rank process")
    assert rc == bench.STALL_STATUS and time.monotonic() - t0 < 20
    assert slow.wait(timeout=10) != 0                     # terminated, not left waiting in a collective


def test_watch_deadline_ends_a_run_that_hangs():
    slow = _sleeper(120)
    t0 = time.monotonic()
    rc = bench._watch([slow], time.monotonic() + 0.5, "rank process")
    assert rc == 124 and time.monotonic() - t0 < 20
    assert slow.poll() is not None


STUB = r"""
import os, sys
log = os.environ["STUB_LOG"]
with open(log, "a") as f:
    f.write(f"{os.environ['RANK']} {os.environ.get('RDO_DP_GRAPH', '1')} {os.environ['MASTER_PORT']} {os.environ.get('RDO_BENCH_CHILD')}\n")
if os.environ.get("RDO_DP_GRAPH", "1") == "1" and os.environ["STUB_MODE"] == "stall":
    if os.environ["RANK"] == "0":
        import time; time.sleep(1.0)          # (the other rank is up and has written its line before this one reports the stall)
        open(f"/tmp/rdo_bench_stall_{os.environ['MASTER_PORT']}", "w").close()
        sys.exit(86)
    import time; time.sleep(120)          # the other rank hangs in its collective until the launcher ends it
if os.environ["STUB_MODE"] == "crash":
    sys.exit(3)
sys.exit(0)
"""


@pytest.fixture
def stub(tmp_path, monkeypatch):
    log = tmp_path / "stub.log"
    monkeypatch.setattr(bench, "_rank_cmd", lambda: [sys.executable, "-c", STUB])
    monkeypatch.setenv("STUB_LOG", str(log))
    monkeypatch.setenv("RDO_BENCH_SHARE_GPU", "1")
    monkeypatch.delenv("RDO_DP_GRAPH", raising=False)
    return log


def _rows(log):
    return [l.split() for l in open(log).read().splitlines()]


def test_launcher_retries_on_the_host_loop_after_a_stall(stub, monkeypatch):
    monkeypatch.setenv("STUB_MODE", "stall")
    rc = bench.launch_ranks(types.SimpleNamespace(gpus=2))
    rows = _rows(stub)
    assert rc == 0
    first, second = [r for r in rows if r[1] == "1"], [r for r in rows if r[1] == "0"]
    assert sorted(r[0] for r in first) == ["0", "1"] and sorted(r[0] for r in second) == ["0", "1"]     # both ranks, twice
    assert {r[2] for r in first} != {r[2] for r in second}              # a fresh rendezvous port for the fresh children
    assert all(r[3] == "1" for r in rows)                               # children are marked: they never become supervisors themselves
    assert not os.path.exists(f"/tmp/rdo_bench_stall_{first[0][2]}")


def test_launcher_does_not_retry_an_ordinary_failure(stub, monkeypatch):
    monkeypatch.setenv("STUB_MODE", "crash")
    rc = bench.launch_ranks(types.SimpleNamespace(gpus=2))
    rows = _rows(stub)
    assert rc == 3 and 1 <= len(rows) <= 2 and all(r[1] == "1" for r in rows)      # one attempt only (a rank may be stopped before it logs)


def test_supervisor_retries_its_rank_after_a_stall(stub, monkeypatch):
    """torch.distributed.run's contract: RANK / WORLD_SIZE / MASTER_* in the environment; every rank's supervisor decides by itself."""
    monkeypatch.setenv("STUB_MODE", "stall")
    for k, v in dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29871").items():
        monkeypatch.setenv(k, v)
    rc = bench.supervise_rank()
    rows = _rows(stub)
    assert rc == 0 and [r[1] for r in rows] == ["1", "0"] and rows[0][2] == "29871" and rows[1][2] == str(29871 + 101)


def test_heartbeat_wait_returns_on_progress_and_raises_on_a_stall():
    from quantization.engine import DpStallError, UnitEngine
    eng = types.SimpleNamespace(_hb_host=torch.zeros(1, dtype=torch.int64), _hb_issued=5, DP_STALL_S=0.3, _rank=lambda: 0)

    def tick():
        for i in range(1, 6):
            time.sleep(0.05)
            eng._hb_host[0] = i
    th = threading.Thread(target=tick)
    th.start()
    UnitEngine._hb_wait(eng, 5)                          # progress every 50 ms: no stall although the whole wait exceeds DP_STALL_S / 2
    th.join()
    eng._hb_issued = 9
    t0 = time.monotonic()
    with pytest.raises(DpStallError):
        UnitEngine._hb_wait(eng, 9)
    assert 0.25 < time.monotonic() - t0 < 5


def test_supervisor_takes_its_rank_down_with_it(tmp_path):
    """torch.distributed.run stops its workers -- here the supervisors -- with SIGTERM (and SIGKILL after a grace period): the actual
    rank, a child of the supervisor, must not survive either (signal forwarding; PR_SET_PDEATHSIG for the SIGKILL case)."""
    import signal
    import psutil
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for sig in (signal.SIGTERM, signal.SIGKILL):
        code = ("import sys, os; sys.path.insert(0, %r); import bench; "
                "bench._rank_cmd = lambda: [sys.executable, '-c', 'import time; time.sleep(120)']; "
                "os.environ.update(RANK='0', WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT='29872'); "
                "sys.exit(bench.supervise_rank())") % root
        sup = subprocess.Popen([sys.executable, "-c", code])
        kids, t0 = [], time.time()
        while not kids and time.time() - t0 < 60:
            kids = psutil.Process(sup.pid).children()
            time.sleep(0.1)
        assert len(kids) == 1
        os.kill(sup.pid, sig)
        sup.wait(timeout=30)
        t0 = time.time()
        while time.time() - t0 < 15 and kids[0].is_running() and kids[0].status() != psutil.STATUS_ZOMBIE:
            time.sleep(0.1)
        assert not (kids[0].is_running() and kids[0].status() != psutil.STATUS_ZOMBIE), f"the rank survived its supervisor ({sig!r})"
