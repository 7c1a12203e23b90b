"""The MULTI-RANK training step on the GPU with no unexpected device -> host wait.

RCCL refuses two ranks on one device and the test box has one GPU, so two processes share cuda:0 and exchange through gloo -- the reducer code path of an N > 1
run (bucket all-reduce, forward-time announcement of the embedding rows, row all-gather, union of the touched rows, row-masked AdamW) with only the transport
swapped.  After a warm-up step each rank runs two optimizer steps of the tiny Qwen2.5-VL (LoRA r8 on q / v, lm_head, embed_tokens; a different batch per rank
and step) under torch.cuda.set_sync_debug_mode("error"): any synchronising call outside the host-staged gloo transfers (which RCCL does not have) raises.
The replicas must stay bit-identical (reference: DeepSpeed's data-parallel step, train_joint.py:325-346, 500-535)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    try:
        for p in (ROOT, os.path.join(ROOT, "rga3-release_amd")):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import numpy as np

        from rga3.model.qwen2_5_vl import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration
        from rga3.model.qwen_train import add_lora
        from rga3.parallel.ddp import FusedAdamW, GradBucketReducer, sparse_candidates
        from rga3.utils.staging import dict_to_cuda
        from tests.qwen_tiny import det_params, gold, product_cfg_kwargs

        dev = torch.device("cuda:0")
        torch.manual_seed(0)          # replicas start from the same LoRA factors (a trainer seeds or broadcasts; reference train_joint.py seeds per run)
        g = gold()
        m = Qwen2_5_VLForConditionalGeneration(Qwen2_5_VLConfig(**product_cfg_kwargs()))
        m.load_state_dict(det_params(g, bf16_round=False), strict=True)
        m = m.to(torch.bfloat16).to(dev)
        add_lora(m, r=8, alpha=16, dropout=0.0, exclude=("visual",))
        m.train()
        for n, p in m.named_parameters():
            p.requires_grad_(("lora_" in n) or n in ("lm_head.weight", "model.embed_tokens.weight"))
        with torch.no_grad():
            for n, p in m.named_parameters():
                if "lora_B" in n:
                    p.normal_(0.0, 0.01, generator=torch.Generator(device=dev).manual_seed(7))     # same on every rank
        trainables = [p for p in m.parameters() if p.requires_grad]
        sparse = sparse_candidates(m)
        assert len(sparse) == 1
        red = GradBucketReducer(trainables, bucket_mb=0.05, sparse_params=sparse)      # several dense buckets + the sparse table
        opt = FusedAdamW.for_reducer(red, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.0, max_grad_norm=1.0)

        def batch(step):
            gen = torch.Generator().manual_seed(1000 * step + 17 * rank)
            ids = torch.randint(1, 300, (1, 48), generator=gen)
            labels = torch.full_like(ids, -100)
            labels[:, -8:] = ids[:, -8:]
            return dict_to_cuda(dict(input_ids=ids, attention_mask=torch.ones_like(ids), labels=labels), dev)

        def step(i):
            red.begin_step()
            red.begin_micro_step()
            out = m(**batch(i))
            out.loss.backward()
            red.finish()
            opt.step(red.grad_view, red.flat_grads())
            return out.loss

        step(0)                                  # builds every cached table / workspace, tunes the GEMM shapes
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")
        try:
            l1 = step(1)
            l2 = step(2)
        finally:
            torch.cuda.set_sync_debug_mode("default")
        torch.cuda.synchronize()
        sums = torch.stack([p.detach().float().sum() for p in trainables] + [p.detach().float().abs().sum() for p in trainables]).cpu()
        allsums = [torch.empty_like(sums) for _ in range(world)]
        dist.all_gather(allsums, sums)
        same = all(torch.equal(allsums[0], s) for s in allsums[1:])
        q.put((rank, True, bool(same), bool(torch.isfinite(l1).item() and torch.isfinite(l2).item()), ""))
        dist.destroy_process_group()
    except Exception as e:   # noqa: BLE001
        import traceback
        q.put((rank, False, False, False, traceback.format_exc()[-1500:]))


def test_two_ranks_one_gpu_step_without_unexpected_sync():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 13) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    for rank, ok, same, finite, err in res:
        assert ok, f"rank {rank}: {err}"
        assert finite
        assert same, "replicas diverged: the exchanged gradients / row unions differ between ranks"
