"""bench.py's N-rank code on ONE GPU (missing 1 / next 2): the rank > 0 paths -- JSON on rank 0 only, MAX all-reduce of
the timed region, barriers, per-unit data-parallel probe, teardown -- and the launcher's watchdog run before the driver's first
8-GPU run does.  Two fresh child processes share cuda:0 and talk over gloo (RDO_BENCH_BACKEND=gloo + RDO_BENCH_SHARE_GPU=1, both
test-only switches: RCCL refuses two ranks on one device); the engines, plans and the bucket sequence are the ones RCCL runs."""
import json
import os
import signal
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--images", "8", "--no-extras", "--no-cpu-baseline", "--sustain-steps", "0",
        "--recon-iters", "0"]


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(RDO_BENCH_BACKEND="gloo", RDO_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
    return env


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _check_line(stdout):
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len(lines) == 1, stdout                       # exactly ONE line on stdout, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["parallelism"] == "dp2" and d["config"]["units"] == 29 and d["config"]["batch_per_gpu"] == 4
    # value = the image-iterations ALL ranks processed / the slowest rank's time
    assert abs(d["value"] - 29 * 4 * 3 * 2 / (d["ms_per_step"] * 3 / 1e3)) <= 1e-3 * d["value"]
    assert d["config"]["dp_loop"] == "host" and d["dp_graph"] is False
    dp = d["dp"]
    assert dp["backend"] == "gloo" and len(dp["units"]) == 29
    for name, u in dp["units"].items():
        assert u["iter_us"] > 0 and u["allreduce_us"] > 0 and u["bucket_kb"] > 0, name
    assert dp["units"]["g_a.1"]["collectives_per_iter"] == 2 and dp["units"]["h_a.0"]["collectives_per_iter"] == 1
    return d


def test_two_ranks_launched_like_the_driver_does():
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment (torch.distributed.run's contract), two children."""
    port = _free_port()
    procs = []
    for r in range(2):
        env = _env(RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + ARGS, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], [o[1][-2000:] for o in outs]
    _check_line(outs[0][0])
    assert outs[1][0].strip() == ""                      # rank 1 prints nothing on stdout


def test_launcher_spawns_ranks_and_reports_one_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + ARGS, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    _check_line(r.stdout)


def test_launcher_stops_everything_when_a_rank_dies():
    """A rank killed mid-run: the parent terminates the other rank (it would wait in a collective for ever) and exits non-zero --
    a watchdog over fresh child processes, never a re-exec -- and no JSON line is printed for a run that did not finish."""
    import psutil
    import tempfile
    with tempfile.TemporaryFile("w+") as errf:
        parent = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + ARGS, env=_env(), stdout=subprocess.PIPE, stderr=errf,
                                  text=True)
        try:
            kids, t0 = [], time.time()
            while len(kids) < 2 and time.time() - t0 < 120:
                kids = psutil.Process(parent.pid).children()
                time.sleep(0.2)
            assert len(kids) == 2

            def err_text():
                errf.seek(0)
                return errf.read()
            t0 = time.time()
            while "caches built" not in err_text() and time.time() - t0 < 600:      # rank 0's log line: both ranks are up, past the
                assert parent.poll() is None, err_text()[-3000:]                     # process group's creation, in front of the loops
                time.sleep(0.2)
            assert "caches built" in err_text()
            os.kill(kids[1].pid, signal.SIGKILL)
            out, _ = parent.communicate(timeout=180)
            err = err_text()
        finally:
            if parent.poll() is None:
                parent.kill()
    assert parent.returncode != 0
    assert out.strip() == ""
    assert "stopping the other ranks" in err
    time.sleep(1.0)
    assert not any(k.is_running() and k.status() != psutil.STATUS_ZOMBIE for k in kids)
