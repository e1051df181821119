"""Two ranks sharing cuda:0 over `gloo` (the box has one GPU; RCCL needs one GPU per rank): exercises the
multi-rank code on the real HIP kernels -- GraphedTrainingStep (five hipGraph segments cut where a gradient bucket is
complete, the bucket's all-reduce launched between them) and the eager path (ddp.attach: gradsink.GradBuckets launch a
bucket's all-reduce from the backward function that completes it).  RCCL itself runs only in the driver's scaling bench."""
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
        if mode.startswith("eager_ygrad"):
            # a target that requires grad: Discriminator.forward_pair falls back to two calls (as it does for extents that
            # are not multiples of 32, where only the forward pass of the strided layers exists)
            batch["dwi-tensor_orig"]["data"].requires_grad_(True)
        if mode == "eager_ygrad_nopair":
            model.pair_discriminator_calls = False
        extra = None
        if mode == "graph":
            ddp.broadcast_module_state(model.gen, 0)
            ddp.broadcast_module_state(model.discr, 0)
            step = GraphedTrainingStep(model, batch, warmup=2)
            step()
            step()
            extra = step.launch_log
        elif mode == "order":
            # enqueue order of one eager step: every gradient kernel launch and every all-reduce launch, in host order
            # (= stream order: the collective waits for what was enqueued before it and overlaps what comes after)
            from unet_bssfp_amd import gradsink, ops
            log = []
            real_wgrad, real_launch = ops.conv_wgrad, gradsink.GradBuckets.launch
            ops.conv_wgrad = lambda *a, **k: (log.append(("wgrad", None)), real_wgrad(*a, **k))[1]
            def launch(self, b):
                log.append(("launch", ("gen" if self is model.sinks_gen else "discr", b)))
                return real_launch(self, b)
            gradsink.GradBuckets.launch = launch
            ddp.attach(model)
            model.training_step(batch, 0)
            ops.conv_wgrad, gradsink.GradBuckets.launch = real_wgrad, real_launch
            extra = log
        elif mode == "gen_only":
            # BASELINE.json configs[1] on an attached model: a backward loop other than training_step must still average
            # the gradients over the ranks (the ranks draw different batches: without the exchange they drift apart)
            ddp.attach(model)
            for i in range(3):
                model.generator_only_step(batch, i)
            extra = list(model.sinks_gen.launch_order)
        elif mode.startswith("eager_ygrad"):
            ddp.attach(model)
            for i in range(2):
                model.training_step(batch, i)
            extra = [model.sinks_discr.uses]
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
        q.put((rank, "ok" if (same and finite) else f"FAIL same={same} finite={finite}", digest.tolist(), extra))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, f"FAIL: {e!r} {traceback.format_exc()}", None, None))
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
    # the late buckets are exchanged between the two backward stages of their phase
    assert res[0][3] == [("allreduce gen[0]", "before gen backward stage 2"), ("allreduce discr[0]", "before discr backward stage 2")]


def test_allreduce_of_bucket0_is_enqueued_before_the_last_backward_kernels(hip):
    """src/train.py:30-32 (DDP overlaps the gradient all-reduce with backward): in the eager step the all-reduce of each
    network's bucket 0 (the layers whose gradients are ready first) must be enqueued BEFORE the weight-gradient kernels of
    the remaining layers -- stream order decides what it can overlap -- and bucket 1 after them."""
    res = _run("order")
    assert all(r[1] == "ok" for r in res), [r[1] for r in res]
    for r in res:
        log = r[3]
        pos = {e[1]: i for i, e in enumerate(log) if e[0] == "launch"}
        wg = [i for i, e in enumerate(log) if e[0] == "wgrad"]
        assert set(pos) == {("gen", 0), ("gen", 1), ("discr", 0), ("discr", 1)}, pos
        gen_wg = [i for i in wg if i < pos[("gen", 1)]]
        discr_wg = [i for i in wg if pos[("gen", 1)] < i < pos[("discr", 1)]]
        # weight-gradient kernels follow the launch of bucket 0 in both phases: there is backward work left to overlap
        assert pos[("gen", 0)] < gen_wg[-1] and sum(i > pos[("gen", 0)] for i in gen_wg) >= 4, (pos, gen_wg)
        assert pos[("discr", 0)] < discr_wg[-1] and sum(i > pos[("discr", 0)] for i in discr_wg) >= 2, (pos, discr_wg)


def test_graph_and_hook_paths_agree_two_ranks(hip):
    """Same 4 steps through (a) hipGraph segments with the exchanges between them and (b) the eager step whose backward
    functions launch the exchanges: identical kernels, identical buckets -> identical parameters."""
    a, b = _run("graph"), _run("eager")
    assert all(r[1] == "ok" for r in a + b), [r[1] for r in a + b]
    da, db = torch.tensor(a[0][2]), torch.tensor(b[0][2])
    torch.testing.assert_close(da, db, rtol=1e-6, atol=1e-6)


def test_generator_only_loop_on_an_attached_model_two_ranks(hip):
    res = _run("gen_only")
    assert all(r[1] == "ok" for r in res), [r[1] for r in res]
    assert all(sorted(r[3]) == [0, 1] for r in res), [r[3] for r in res]      # both generator buckets were exchanged


def test_zero_grad_on_sink_parameters_fails_loudly(hip):
    """optimizer.zero_grad() (set_to_none=True by default) drops the permanent .grad views of the gradient sinks: the next
    backward must raise a clear error instead of handing the kernels a null pointer."""
    import unet_bssfp_amd as M
    from unet_bssfp_amd.gan import bSSFPToDWITensorModel, synthetic_batch
    torch.manual_seed(0)
    model = bSSFPToDWITensorModel("bssfp", gen=M.Generator("bssfp", dropout=0.0).to("cuda:0"),
                                  discr=M.Discriminator("bssfp").to("cuda:0")).train()
    batch = synthetic_batch(2, 32, seed=1, device="cuda:0")
    model.generator_only_step(batch, 0)
    model.optimizers()[0].zero_grad()
    with pytest.raises(RuntimeError, match="gradient bucket"):
        model.generator_only_step(batch, 1)


def test_discriminator_phase_as_two_calls_announces_two_contributions_two_ranks(hip):
    """when ``Discriminator.forward_pair`` falls back to two calls (an input that requires grad; extents that are
    not multiples of 32), every discriminator parameter receives TWO gradient contributions in the phase; the eagerly
    attached buckets must be told so (an all-reduce launched after the first call's contributions would miss the second
    call's).  The default path must equal the explicitly unpaired one bit for bit, on every rank."""
    a, b = _run("eager_ygrad"), _run("eager_ygrad_nopair")
    assert all(r[1] == "ok" for r in a + b), [r[1] for r in a + b]
    assert all(r[3] == [2] for r in a + b), [r[3] for r in a + b]
    assert a[0][2] == b[0][2]
