"""Two ranks sharing cuda:0 over `gloo` (the box has one GPU; RCCL needs one GPU per rank): exercises the
multi-rank code of GraphedTrainingStep (three hipGraph segments + eager all-reduce of flat gradient
buffers) and of the eager hook-driven ddp.GradSync path on the real HIP kernels."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, mode, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import unet_bssfp_amd as M
        from unet_bssfp_amd import ddp
        from unet_bssfp_amd.gan import GraphedTrainingStep, bSSFPToDWITensorModel, synthetic_batch
        dev = "cuda:0"
        torch.cuda.set_device(0)
        torch.manual_seed(10 + rank)                       # ranks start different: rank 0 is broadcast
        gen, discr = M.Generator("bssfp", dropout=0.0), M.Discriminator("bssfp")
        model = bSSFPToDWITensorModel("bssfp", gen=gen.to(dev), discr=discr.to(dev)).train()
        batch = synthetic_batch(2, 32, seed=70 + rank, device=dev)
        if mode == "graph":
            ddp.broadcast_module_state(model.gen, 0)
            ddp.broadcast_module_state(model.discr, 0)
            step = GraphedTrainingStep(model, batch, warmup=2)
            step()
            step()
        else:
            ddp.attach(model)
            for i in range(4):
                model.training_step(batch, i)
        torch.cuda.synchronize()
        digest = torch.stack([p.detach().double().sum() for p in model.parameters()]).cpu()
        gathered = [torch.zeros_like(digest) for _ in range(world)]
        dist.all_gather(gathered, digest)
        same = all(torch.equal(g, gathered[0]) for g in gathered)
        finite = bool(torch.isfinite(digest).all())
        q.put((rank, "ok" if (same and finite) else f"FAIL same={same} finite={finite}", digest.tolist()))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, f"FAIL: {e!r} {traceback.format_exc()}", None))
    finally:
        dist.destroy_process_group()


def _run(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    return res


def test_graph_segments_with_eager_allreduce_two_ranks(hip):
    res = _run("graph")
    assert all(r[1] == "ok" for r in res), [r[1] for r in res]


def test_graph_and_hook_paths_agree_two_ranks(hip):
    """Same 4 steps through (a) hipGraph segments + flat all-reduce and (b) eager GradSync hooks."""
    a, b = _run("graph"), _run("eager")
    assert all(r[1] == "ok" for r in a + b), [r[1] for r in a + b]
    da, db = torch.tensor(a[0][2]), torch.tensor(b[0][2])
    torch.testing.assert_close(da, db, rtol=1e-6, atol=1e-6)
