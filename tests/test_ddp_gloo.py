"""CPU, world_size 2 (gloo): the bucketed gradient reducer averages gradients across ranks bucket by bucket, honours
no_sync() gradient accumulation, and every rank ends with identical values."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rga3.parallel.ddp import GradBucketReducer

    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4), torch.nn.Linear(4, 3))
    red = GradBucketReducer(net.parameters(), bucket_mb=0.0003)  # ~300 B buckets -> several buckets
    assert len(red.buckets) >= 3
    xs = [torch.randn(5, 8, generator=torch.Generator().manual_seed(10 * r + m)) for r in range(world) for m in range(2)]
    my = xs[2 * rank: 2 * rank + 2]
    red.begin_step()
    with red.no_sync():
        red.begin_micro_step()
        net(my[0]).pow(2).sum().backward()
    red.begin_micro_step()
    net(my[1]).pow(2).sum().backward()
    red.finish()
    got = [red.grad_view(p).clone() for p in net.parameters()]
    red.remove()
    # reference: average over ranks of the per-rank accumulated gradients
    ref = [torch.zeros_like(p) for p in net.parameters()]
    for r in range(world):
        net.zero_grad()
        for m in range(2):
            net(xs[2 * r + m]).pow(2).sum().backward()
        for a, p in zip(ref, net.parameters()):
            a += p.grad / world
    ok = all(torch.allclose(a, b, atol=1e-5) for a, b in zip(got, ref))
    # a parameter the loss never reaches (its bucket never counts down during backward) + accumulation: the bucket is still exchanged in
    # finish(), the used parameters accumulate over both micro-steps, the unused one reads as zeros -- also on a second optimizer step
    torch.manual_seed(1)
    used, unused = torch.nn.Linear(8, 4), torch.nn.Linear(8, 4)
    ref2 = [torch.zeros_like(p) for p in used.parameters()]          # reference first: no reducer hooks attached yet
    for r in range(world):
        used.zero_grad()
        for m in range(2):
            used(xs[2 * r + m]).pow(2).sum().backward()
        for a, p in zip(ref2, used.parameters()):
            a += p.grad / world
    used.zero_grad()
    red2 = GradBucketReducer(list(used.parameters()) + list(unused.parameters()), bucket_mb=1.0)   # ONE bucket holding all four tensors
    for step in range(2):
        red2.begin_step()
        with red2.no_sync():
            red2.begin_micro_step()
            used(my[0]).pow(2).sum().backward()
        red2.begin_micro_step()
        used(my[1]).pow(2).sum().backward()
        red2.finish()
        ok = ok and all(torch.allclose(red2.grad_view(p), a, atol=1e-5) for p, a in zip(used.parameters(), ref2))
        ok = ok and all(float(red2.grad_view(p).abs().max()) == 0.0 for p in unused.parameters())
    red2.remove()
    q.put((rank, ok, [g.sum().item() for g in got]))
    dist.destroy_process_group()


def test_bucketed_allreduce_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] == res[1][2]  # identical on both ranks


def _worker_divergent(rank, world, port, q):
    """the set of parameters that receive a gradient differs between ranks (a sample without [SEG] skips SAM2 on one rank only).
    Three equal-size buckets; rank 1 never reaches parameter b.  Launch order must stay bucket 0, 1, 2 on both ranks."""
    sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rga3.parallel.ddp import GradBucketReducer

    torch.manual_seed(0)
    a, b, c = (torch.nn.Parameter(torch.randn(64)) for _ in range(3))
    red = GradBucketReducer([a, b, c], bucket_mb=64 * 4 / (1 << 20))   # one parameter per bucket, all the same size
    assert len(red.buckets) == 3
    x = torch.randn(64, generator=torch.Generator().manual_seed(5 + rank))
    oks = []
    for step in range(2):
        red.begin_step()
        red.begin_micro_step()
        loss = (a * x).sum() * 1.0 + (c * x * 3.0).sum()
        if rank == 0:
            loss = loss + (b * x * 2.0).sum()
        loss.backward()
        red.finish()
        xs = [torch.randn(64, generator=torch.Generator().manual_seed(5 + r)) for r in range(world)]
        ref_a = sum(xs) / world
        ref_b = 2.0 * xs[0] / world              # only rank 0 contributes
        ref_c = 3.0 * sum(xs) / world
        oks.append(torch.allclose(red.grad_view(a), ref_a, atol=1e-6) and torch.allclose(red.grad_view(b), ref_b, atol=1e-6)
                   and torch.allclose(red.grad_view(c), ref_c, atol=1e-6))
    q.put((rank, all(oks), [red.grad_view(p).sum().item() for p in (a, b, c)]))
    red.remove()
    dist.destroy_process_group()


def _worker_sparse(rank, world, port, q):
    """Sparse row exchange of an embedding-table gradient (announce in forward, add in backward, all-gather in finish) against the dense average;
    overlapping and disjoint ids, gradient accumulation over two micro-steps, two optimizer steps (rows of step 1 must be gone in step 2)."""
    sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np
    from rga3.parallel.ddp import GradBucketReducer, sparse_sink_for

    V, H = 50, 16
    table = torch.nn.Parameter(torch.zeros(V, H))
    w = torch.nn.Parameter(torch.ones(4))
    ok, sums = True, []
    # announce_cap 8192: every rank's ids travel with the forward-time announcement (the union of the touched rows is taken on the host, no device-side unique);
    # announce_cap 8: the unions (up to 16 ids) outgrow it and the step falls back to the id exchange at finish().  Same results either way.
    for acap in (8192, 8):
        table.data.zero_()
        red = GradBucketReducer([table, w], bucket_mb=1.0, sparse_params=[table], announce_cap=acap)
        assert sparse_sink_for(table) is red and len(red.buckets) == 1
        for step in range(2):
            red.begin_step()
            dense_ref = torch.zeros(world, V, H)
            for mi in range(2):
                for r in range(world):   # every rank can rebuild everyone's contribution
                    g = torch.Generator().manual_seed(100 * step + 10 * r + mi)
                    ids = np.unique(torch.randint(0, V if r == 0 else V // 2, (5 + 3 * r,), generator=g).numpy()).astype(np.int64)
                    rows = torch.randn(len(ids), H, generator=g)
                    dense_ref[r][torch.from_numpy(ids)] += rows
                    if r == rank:
                        mine = (ids, rows)
                ctx = red.no_sync() if mi == 0 else __import__("contextlib").nullcontext()
                with ctx:
                    red.begin_micro_step()
                    red.announce_sparse(table, mine[0])
                    (w * (rank + 1.0)).sum().backward()
                    red.add_sparse(table, torch.from_numpy(mine[0]), mine[1])
            red.finish()
            ref = dense_ref.mean(0)
            ok = ok and torch.allclose(red.grad_view(table), ref, atol=1e-6)
            ok = ok and torch.allclose(red.grad_view(w), torch.full((4,), 2.0 * (1 + world) / 2), atol=1e-6)   # two micro-steps, mean over ranks of (rank + 1)
            touched = torch.nonzero(dense_ref.abs().sum((0, 2)) > 0).flatten()      # the union of every rank's rows: what the optimizer's row mask / norm must follow
            ok = ok and torch.equal(red._sp[id(table)]["last_ids"].cpu(), touched)
        sums += [red.grad_view(table).sum().item(), red.sparse_bytes_last]
        red.remove()
    ok = ok and sums[0] == sums[2]       # announced and fallback paths: the same bits
    q.put((rank, bool(ok), sums))
    dist.destroy_process_group()


def _run(worker):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + ((os.getpid() * 7 + hash(worker.__name__)) % 2000)
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    return res


def test_rank_divergent_unused_parameter_keeps_bucket_order():
    res = _run(_worker_divergent)
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] == res[1][2]


def test_sparse_row_exchange_two_ranks():
    res = _run(_worker_sparse)
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] == res[1][2]   # bitwise-identical sums on both ranks


def test_sparse_parameter_refuses_dense_gradient():
    """with tied word embeddings the LM head's dense dW lands in the table's .grad; a table registered as sparse would silently drop it.  The reducer
    refuses it loudly, and sparse_candidates() does not offer a tied table in the first place."""
    import pytest
    from rga3.parallel.ddp import GradBucketReducer, sparse_candidates

    table = torch.nn.Parameter(torch.zeros(10, 8))
    w = torch.nn.Parameter(torch.ones(4))
    red = GradBucketReducer([table, w], bucket_mb=1.0, sparse_params=[table])
    red.begin_step()
    with pytest.raises(RuntimeError, match="dense gradient"):
        (table.sum() + w.sum()).backward()
    red.remove()

    class M(torch.nn.Module):
        def __init__(self, tied):
            super().__init__()
            self.emb = torch.nn.Embedding(10, 8)
            self.head = torch.nn.Linear(8, 10, bias=False)
            if tied:
                self.head.weight = self.emb.weight

        def get_input_embeddings(self):
            return self.emb

        def get_output_embeddings(self):
            return self.head

    assert sparse_candidates(M(True)) == []
    m = M(False)
    assert sparse_candidates(m) == [m.emb.weight] or sparse_candidates(m)[0] is m.emb.weight


def test_gradient_produced_inside_its_bucket_slice():
    """dense_grad_out_for: a producer (the LM head's dW = dlogits^T h, 1.09 GB at the 7B size) writes a parameter's gradient straight into the reducer's bucket slice and
    returns that tensor; the reducer recognises it and moves nothing.  First gradient of an optimizer step only: a second micro-step gets None and is ADDED as usual
    (reference: DeepSpeed's gradient accumulation over micro-steps, train_joint.py:325-346, 521-535)."""
    from rga3.parallel.ddp import GradBucketReducer, dense_grad_out_for

    p = torch.nn.Parameter(torch.zeros(6, 8))
    q = torch.nn.Parameter(torch.zeros(5))
    red = GradBucketReducer([p, q], bucket_mb=1.0)
    g1, g2 = torch.arange(48.0).view(6, 8), torch.ones(6, 8) * 0.5
    seen = []

    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, w, g):
            ctx.g = g
            return w.sum() * 0.0

        @staticmethod
        def backward(ctx, _):
            out = dense_grad_out_for(p)
            seen.append(out is not None)
            if out is None:
                return ctx.g.clone(), None
            out.copy_(ctx.g)
            return out, None

    red.begin_step()
    (Fn.apply(p, g1) + q.sum()).backward()
    assert seen == [True] and p.grad is None
    assert torch.equal(red.grad_view(p), g1)
    red.begin_micro_step()
    (Fn.apply(p, g2) + q.sum()).backward()
    assert seen == [True, False]
    red.finish()
    assert torch.equal(red.grad_view(p), g1 + g2) and torch.equal(red.grad_view(q), torch.full((5,), 2.0))
    red.begin_step()
    assert dense_grad_out_for(p) is not None        # a new optimizer step starts from an empty slice again
    # a REMOVED reducer no longer offers its bucket (the producer would write every later dW into the same dead slice and autograd's accumulation would
    # add that tensor to itself: p.grad = 2 * dW_new instead of dW_old + dW_new)
    red.remove()
    assert dense_grad_out_for(p) is None
    (Fn.apply(p, g1) + q.sum()).backward()
    (Fn.apply(p, g2) + q.sum()).backward()
    assert seen[-2:] == [False, False]
    assert torch.equal(p.grad, g1 + g2)
    # a second reducer over the same parameters takes the sink over; removing it leaves no entry behind
    red2 = GradBucketReducer([p, q], bucket_mb=1.0)
    assert dense_grad_out_for(p) is not None and dense_grad_out_for(p).data_ptr() == red2.grad_view(p).data_ptr()
    red2.remove()
    assert dense_grad_out_for(p) is None
