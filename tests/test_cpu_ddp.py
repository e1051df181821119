"""world_size-2 `gloo` tests of the data-parallel path on CPU: bucketed gradient all-reduce from
grad-ready hooks (ddp.GradSync), parameter/buffer broadcast, the single stacked log reduce, and the
whole GAN step (harness + sync) against a single-process run on the concatenated batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker_sync(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    from unet_bssfp_amd import ddp
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)                      # different init per rank on purpose
        net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.BatchNorm1d(16), torch.nn.Linear(16, 4),
                                  torch.nn.Linear(4, 4))   # last layer stays unused -> never fires
        ddp.broadcast_module_state(net, 0)
        w0 = net[0].weight.detach().clone()
        gathered = [torch.zeros_like(w0) for _ in range(world)]
        dist.all_gather(gathered, w0)
        assert all(torch.equal(g, gathered[0]) for g in gathered), "broadcast did not equalise parameters"
        sync = ddp.GradSync(list(net.parameters()), bucket_mb=0.0005)      # tiny buckets -> several of them
        assert len(sync.buckets) >= 3
        g = torch.Generator().manual_seed(7)
        xs = torch.rand(world, 6, 8, generator=g)
        out = net[2](net[1](net[0](xs[rank])))
        out.pow(2).sum().backward()
        sync.finish()
        local = [None if p.grad is None else p.grad.detach().clone() for p in net.parameters()]
        # reference: average of the per-rank gradients, computed by brute force
        for p, mine in zip(net.parameters(), local):
            if mine is None:
                continue
            assert p.grad.shape == p.shape
        net.zero_grad()
        ref = []
        for r in range(world):
            net.zero_grad()
            o = net[2](net[1](net[0](xs[r])))
            o.pow(2).sum().backward()
            ref.append([None if p.grad is None else p.grad.detach().clone() for p in net.parameters()])
        for i, mine in enumerate(local):
            if mine is None:
                assert ref[0][i] is None
                continue
            avg = sum(ref[r][i] for r in range(world)) / world
            torch.testing.assert_close(mine, avg, rtol=1e-5, atol=1e-7)
        # gradient accumulation: two backward() calls in one phase -- the second accumulation arrives after some
        # buckets have already been exchanged; finish() must deliver the average of the SUMMED local gradients
        net.zero_grad()
        for rep in range(2):
            o = net[2](net[1](net[0](xs[rank] * (rep + 1))))
            o.pow(2).sum().backward()
        sync.finish()
        got = [None if p.grad is None else p.grad.detach().clone() for p in net.parameters()]
        acc = []
        for r in range(world):
            net.zero_grad()
            for rep in range(2):
                o = net[2](net[1](net[0](xs[r] * (rep + 1))))
                o.pow(2).sum().backward()
            acc.append([None if p.grad is None else p.grad.detach().clone() for p in net.parameters()])
        for i, mine in enumerate(got):
            if mine is not None:
                torch.testing.assert_close(mine, sum(acc[r][i] for r in range(world)) / world, rtol=1e-5, atol=1e-7)
        logs = ddp.reduce_logs(torch.tensor([float(rank), 2.0]))
        torch.testing.assert_close(logs, torch.tensor([(world - 1) / 2.0, 2.0]))
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, f"FAIL: {e!r}"))
    finally:
        dist.destroy_process_group()


def _worker_gan(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    from oracle import unet_ref as R
    from unet_bssfp_amd import ddp
    from unet_bssfp_amd.gan import bSSFPToDWITensorModel, synthetic_batch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        torch.manual_seed(rank)                            # ranks start different; attach() broadcasts rank 0
        gen = R.RefGenerator("bssfp", dropout=0.0).train()
        discr = R.RefDiscriminator("bssfp").train()
        model = bSSFPToDWITensorModel("bssfp", gen=gen, discr=discr, optimizer_class=torch.optim.SGD, lr=1e-3)
        ddp.attach(model)
        unused = {id(p) for p in gen.blocks["dwi-tensor"].parameters()} | {id(p) for p in discr.d1["t1w"].parameters()}
        for sync in (model.grad_sync_gen, model.grad_sync_discr):
            for b in sync.buckets:
                assert not any(id(p) in unused for p in b.params), "unused modality head must be excluded statically"
        batch = synthetic_batch(2, 32, seed=50 + rank)
        model.training_step(batch, 0)
        logs = ddp.reduce_logs(model.stacked_logs())
        digest = torch.stack([p.detach().double().sum() for p in model.parameters()])
        gathered = [torch.zeros_like(digest) for _ in range(world)]
        dist.all_gather(gathered, digest)
        assert all(torch.allclose(g, gathered[0], rtol=0, atol=0) for g in gathered), "ranks diverged after the step"
        q.put((rank, "ok", logs.tolist(), digest.tolist()))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, f"FAIL: {e!r} {traceback.format_exc()}", None, None))
    finally:
        dist.destroy_process_group()


def _worker_gen_only(rank, world, port, q):
    """BASELINE.json configs[1] on an attached model: generator_only_step must average the gradients over the ranks (the
    ranks draw different batches: without the exchange their parameters drift apart)."""
    import sys
    sys.path.insert(0, ROOT)
    from oracle import unet_ref as R
    from unet_bssfp_amd import ddp
    from unet_bssfp_amd.gan import bSSFPToDWITensorModel, synthetic_batch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        torch.manual_seed(rank)
        gen = R.RefGenerator("bssfp", dropout=0.0).train()
        discr = R.RefDiscriminator("bssfp").train()
        model = bSSFPToDWITensorModel("bssfp", gen=gen, discr=discr, optimizer_class=torch.optim.SGD, lr=1e-3)
        ddp.attach(model)
        batch = synthetic_batch(1, 32, seed=50 + rank)
        for i in range(2):
            model.generator_only_step(batch, i)
        digest = torch.stack([p.detach().double().sum() for p in model.gen.parameters()])
        gathered = [torch.zeros_like(digest) for _ in range(world)]
        dist.all_gather(gathered, digest)
        assert all(torch.equal(g, gathered[0]) for g in gathered), "ranks diverged in the generator-only loop"
        q.put((rank, "ok", digest.tolist()))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, f"FAIL: {e!r} {traceback.format_exc()}", None))
    finally:
        dist.destroy_process_group()


def _run(worker, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    return sorted(res)


def test_gradsync_buckets_average_and_broadcast_gloo():
    res = _run(_worker_sync)
    assert all(r[1] == "ok" for r in res), res


def test_gan_step_data_parallel_matches_single_process_gloo():
    """2 ranks x (N=2 per rank), SGD so that the update is linear in the averaged gradient.  The
    generator (InstanceNorm: per-sample statistics) must equal the single-process step whose loss is
    the mean of the two per-rank losses; ranks must stay bit-identical to each other."""
    import sys
    sys.path.insert(0, ROOT)
    from oracle import unet_ref as R
    res = _run(_worker_gan)
    assert all(r[1] == "ok" for r in res), [r[1] for r in res]
    assert res[0][3] == res[1][3]
    # single-process emulation: same init (rank 0's), per-rank batches, averaged gradients, per-rank BN
    torch.manual_seed(0)
    gen = R.RefGenerator("bssfp", dropout=0.0).train()
    discr = R.RefDiscriminator("bssfp").train()
    for p in discr.parameters():
        p.requires_grad_(False)
    grads = None
    adv_l1 = []
    import copy
    for r in range(2):
        d_r = copy.deepcopy(discr)                        # BatchNorm buffers are per rank
        x, y = R.synthetic_batch(2, 32, seed=50 + r)
        y_hat = gen(x)
        logits = d_r(x, y_hat)
        adv = torch.nn.functional.binary_cross_entropy_with_logits(logits, torch.ones_like(logits))
        l1 = torch.nn.functional.l1_loss(y_hat, y)
        adv_l1.append((adv.item(), l1.item()))
        gen.zero_grad()
        (adv + l1 * 100.0).backward()
        g = [None if p.grad is None else p.grad.clone() for p in gen.parameters()]
        grads = g if grads is None else [None if a is None else a + b for a, b in zip(grads, g)]
    logs = res[0][2]
    assert logs[0] == pytest.approx(sum(a for a, _ in adv_l1) / 2, rel=1e-5)          # gen_loss_adversarial
    assert logs[1] == pytest.approx(sum(b for _, b in adv_l1) / 2, rel=1e-5)          # gen_loss_recon_L1


def test_generator_only_loop_data_parallel_gloo():
    res = _run(_worker_gen_only)
    assert all(r[1] == "ok" for r in res), [r[1] for r in res]
    assert res[0][2] == res[1][2]


def test_gradient_buckets_exchange_only_when_asked_to():
    """gradsink.GradBuckets starts collectives only for callers that attach the model (distributed=True): a process group
    that merely EXISTS must not make a never-attached model all-reduce (its parameters were never broadcast, other ranks
    may not be training)."""
    import sys
    sys.path.insert(0, ROOT)
    from unet_bssfp_amd.gradsink import GradBuckets, sink_grad, sink_of
    port = _free_port()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        ps = [torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(5))]
        local = GradBuckets([ps], None, uses_per_phase=1)
        assert local.world == 1 and not local.exchange
        local.begin_phase()
        assert local.fresh(ps[0]) and sink_of(ps[0]) is local and sink_grad(ps[0]).shape == ps[0].shape
        local.written(ps[0]); local.written(ps[1])
        assert local.complete() and local.launch_order == []
        local.finish()
        local.detach()
        forced = GradBuckets([ps], None, uses_per_phase=1, distributed=True, force_collectives=True)
        assert forced.world == 1 and forced.exchange                 # one rank, real all-reduce (rehearsal)
        forced.begin_phase()
        sink_grad(ps[0]).fill_(2.0); sink_grad(ps[1]).fill_(3.0)
        forced.written(ps[0]); forced.written(ps[1])
        assert forced.launch_order == [0]
        forced.finish()
        assert float(ps[0].grad.sum()) == 24.0 and float(ps[1].grad.sum()) == 15.0      # averaged over one rank: unchanged
        ps[0].grad = None
        with pytest.raises(RuntimeError, match="gradient bucket"):
            sink_grad(ps[0])
    finally:
        dist.destroy_process_group()


def _worker_flat_buffers(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    from types import SimpleNamespace
    from oracle import unet_ref as R
    from unet_bssfp_amd import ddp
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(5)
        model = SimpleNamespace(gen=R.RefGenerator("bssfp", dropout=0.0).train(), discr=R.RefDiscriminator("bssfp").train())
        n_before = {k: v.clone() for k, v in model.discr.state_dict().items()}
        flats = ddp.flatten_buffers(model)
        assert len(flats) == 2                                # f32 running statistics, int64 batch counters
        assert ddp.flatten_buffers(model) is flats             # idempotent
        # the state dict is unchanged (keys, values, the PatchGAN's double registration d1.* / blocks.*) and the buffers are views
        after = model.discr.state_dict()
        assert list(after) == list(n_before) and all(torch.equal(after[k], n_before[k]) for k in after)
        rm = model.discr.d2.bn.running_mean
        f32 = [f for (dt, _), f in flats.items() if dt == torch.float32][0]
        assert rm.data_ptr() >= f32.data_ptr() and rm.data_ptr() < f32.data_ptr() + f32.numel() * 4
        # ranks drift apart (per-rank batch statistics), a forward pass writes THROUGH the views, one broadcast per dtype re-aligns
        g = torch.Generator().manual_seed(50 + rank)
        x, y = torch.rand(2, 24, 32, 32, 32, generator=g), torch.rand(2, 6, 32, 32, 32, generator=g)
        with torch.no_grad():
            model.discr(x, model.gen(x))
            model.discr(x, y)
        assert int(model.discr.d2.bn.num_batches_tracked) == 2 and float(rm.abs().sum()) > 0
        mine = torch.cat([f.double().reshape(-1) for f in flats.values()])
        calls = []
        real = dist.broadcast
        dist.broadcast = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        try:
            ddp.broadcast_buffers(model, every=1, step=0)
        finally:
            dist.broadcast = real
        assert len(calls) == 2, calls                          # one collective per dtype, no per-tensor traffic
        got = torch.cat([f.double().reshape(-1) for f in flats.values()])
        gathered = [torch.zeros_like(got) for _ in range(world)]
        dist.all_gather(gathered, got)
        same = all(torch.equal(t, gathered[0]) for t in gathered)
        moved = rank == 0 or not torch.equal(mine, got)
        # ... and the modules see the broadcast values (eval-mode forward equal on both ranks)
        xe = torch.rand(2, 24, 32, 32, 32, generator=torch.Generator().manual_seed(9))
        model.gen.eval(); model.discr.eval()
        with torch.no_grad():
            out = model.discr(xe, torch.zeros(2, 6, 32, 32, 32)).double().reshape(-1)
        outs = [torch.zeros_like(out) for _ in range(world)]
        dist.all_gather(outs, out)
        # (parameters were seeded identically on both ranks, so equal buffers <=> equal eval outputs)
        q.put((rank, "ok" if (same and moved and torch.equal(outs[0], outs[1])) else f"FAIL same={same} moved={moved}", None))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, f"FAIL: {e!r} {traceback.format_exc()}", None))
    finally:
        dist.destroy_process_group()


def test_flat_buffer_storage_broadcasts_batchnorm_state_in_one_collective_per_dtype():
    """ddp.flatten_buffers: the BatchNorm buffers of both networks as views of one flat tensor per dtype -- DDP's per-forward
    buffer broadcast (src/train.py:30) becomes one collective per dtype without gather / scatter copies."""
    res = _run(_worker_flat_buffers)
    assert all(r[1] == "ok" for r in res), res
