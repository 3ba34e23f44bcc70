"""Data-parallel PRODUCT engine, world size 2 (SURVEY 8e identity: N ranks on shards == 1 rank on the concatenated mini-batch).

Two processes share cuda:0 and talk over the `gloo` backend with CUDA tensors (RCCL refuses two ranks on one device; the GPU box
has one GPU).  Each runs `UnitEngine` with world = 2 on its half of the caches and its row of the global mini-batch:
plan A (forward/backward + chained gradients of the early ops) -> async all-reduce of the front of the bucket, overlapped with
plan A2 (the last weight gradient) -> all-reduce of the rest -> plan B (apply with 1/world, Adam, next soft weights).  With the
QDrop counter running over the GLOBAL mini-batch (`batch_offset`) the two ranks draw exactly the mask a single process draws, so
the trained alphas must agree with the 1-rank engine to fp32 reduction-order noise."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
SEED = 1005
ITERS = 8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gidx(iters):
    return np.array([[i % 3, 3 + (2 * i) % 3] for i in range(iters)], dtype=np.int64)   # one image from each rank's shard


def _rank(rank, world, port, golden, tag, kind, overlap, out_q, break_capture=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "rdo-ptq_amd"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from helpers import nhwc, product_unit
        from quantization import dp
        from quantization.engine import UnitEngine
        fx = np.load(golden)
        n = fx[f"{tag}/inp_q"].shape[0]
        lo, hi = dp.shard_range(n, rank, world)
        idx = torch.from_numpy(_gidx(ITERS)[:, rank:rank + 1] - lo)
        unit, k, mods = product_unit(fx, tag, kind)
        eng = UnitEngine(k, mods, nhwc(fx[f"{tag}/inp_q"][lo:hi]), nhwc(fx[f"{tag}/inp_fp"][lo:hi]), nhwc(fx[f"{tag}/out"][lo:hi]),
                         batch_size=1, iters=ITERS, input_prob=0.5, seed=SEED, idx_table=idx, batch_offset=rank, dp_overlap=overlap)
        assert eng.world == 2 and eng.split
        assert (eng.plan_a2 is not None) == (overlap and kind != "layer")
        if break_capture:
            # the capture path with gloo standing in for RCCL: rank 0's capture raises, rank 1's "succeeds" (a stand-in object: gloo
            # collectives cannot be captured) -- the MIN agreement must put BOTH ranks on the host loop with the same iteration count
            class _Fake:
                def replay(self):
                    raise AssertionError("a graph was replayed although another rank's capture failed")

            def capture(self):
                if rank == 0:
                    raise RuntimeError("capture refused (injected)")
                return _Fake()
            UnitEngine.DP_GRAPH_BACKENDS = ("nccl", "gloo")
            UnitEngine._dp_capture = capture
        eng.run()
        torch.cuda.synchronize()
        total = eng.logs()[0]
        if break_capture:
            assert eng.dp_path == "host" and eng._dp_graph is None and eng._dp_graph_failed and eng.dp_fallbacks == 1
            assert eng._done == ITERS and int(eng._it2.max().item()) == ITERS     # (hand-over: the counter's shadow word is ahead by one)
        if rank == 0:
            out_q.put(({n_: eng.alpha_of(n_).cpu().numpy() for n_ in eng.ops}, total.numpy()))
        dist.barrier()
    except BaseException as e:          # the parent must not wait out its queue timeout for a rank that failed
        out_q.put(("error", f"rank {rank}: {e!r}"))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("tag,kind,overlap,break_capture", [("g_a.1", "rb", True, False), ("g_a.1", "rb", False, False), ("g_a.0", "rbws", True, False),
                                                            ("g_s.1", "rbu", True, False), ("g_a.6", "layer", True, False),
                                                            ("g_a.1", "rb", True, True)])
def test_two_rank_engine_equals_single_rank(golden_dir, tag, kind, overlap, break_capture):
    """break_capture: one rank's graph capture of the data-parallel iteration fails: both ranks must
    finish on the host loop, having run the same number of iterations, with the single-rank result."""
    from helpers import nhwc, product_unit
    from quantization.engine import UnitEngine
    golden = os.path.join(golden_dir, "recon_toy.npz")
    fx = np.load(golden)
    unit, k, mods = product_unit(fx, tag, kind)
    ref = UnitEngine(k, mods, nhwc(fx[f"{tag}/inp_q"]), nhwc(fx[f"{tag}/inp_fp"]), nhwc(fx[f"{tag}/out"]), batch_size=2,
                     iters=ITERS, input_prob=0.5, seed=SEED, idx_table=torch.from_numpy(_gidx(ITERS)))
    assert ref.world == 1 and not ref.split
    ref.run()
    torch.cuda.synchronize()
    ref_alpha = {n_: ref.alpha_of(n_).cpu().numpy() for n_ in ref.ops}
    ref_total = ref.logs()[0].numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, golden, tag, kind, overlap, q, break_capture)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        got_alpha, got_total = q.get(timeout=180)
        assert not isinstance(got_alpha, str), got_total
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    # the per-iteration loss of the 2-rank run is the mean of the ranks' local losses = the global mini-batch loss
    np.testing.assert_allclose(got_total, ref_total, rtol=2e-5, atol=1e-7)
    for n_, a in ref_alpha.items():
        np.testing.assert_allclose(got_alpha[n_], a, rtol=0, atol=2e-6)
