"""GPU parity: the product Qwen2.5-VL host modules on HIP kernels vs the fp32 oracle (same bf16-rounded weights)
and vs the transformers-5.15 golden vectors.  Tolerances: rel-L2 <= 2e-2 on hidden/logits (bf16 pipeline vs fp32),
loss <= 1e-2 relative; position ids / greedy tokens bit-exact."""
import numpy as np
import pytest
import torch

from oracle import qwen25vl as Q
from oracle.detweights import det_tensor
from tests.qwen_tiny import det_params, gold, oracle_cfg, product_cfg_kwargs, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model(dev):
    from rga3.model.qwen2_5_vl import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration

    g = gold()
    m = Qwen2_5_VLForConditionalGeneration(Qwen2_5_VLConfig(**product_cfg_kwargs()))
    missing, unexpected = m.load_state_dict(det_params(g, bf16_round=False), strict=True)
    return m.to(torch.bfloat16).to(dev).eval()


@pytest.mark.parametrize("key", ["a", "b"])
def test_vit_parity(model, dev, key):
    g = gold()
    grid = g[f"g4_{key}_grid"]
    px = det_tensor(f"pixel_values_{key}", (int(np.prod(grid[0])), 1176), 1.0, seed=5).to(torch.bfloat16)
    with torch.no_grad():
        out = model.visual(px.to(dev), grid)
    ref = Q.vit_forward(det_params(g), px.float(), grid, oracle_cfg())
    assert rel_l2(out, ref) < 2e-2
    assert rel_l2(out, torch.from_numpy(g[f"vit_{key}_pooler"])) < 3e-2  # vs HF fp32 (weights unrounded)


def test_full_forward_parity(model, dev):
    g = gold()
    px = torch.cat([det_tensor("pixel_values_full0", (192, 1176), 1.0, seed=5), det_tensor("pixel_values_full1", (192, 1176), 1.0, seed=6)], 0).to(torch.bfloat16)
    ids, am, labels = (torch.from_numpy(g[k]) for k in ("full_input_ids", "full_attention_mask", "full_labels"))
    with torch.no_grad():
        out = model(input_ids=ids.to(dev), attention_mask=am.to(dev), labels=labels.to(dev), pixel_values_videos=px.to(dev),
                    video_grid_thw=torch.from_numpy(g["full_grid"]), second_per_grid_ts=torch.tensor([1.0, 1.0]), output_hidden_states=True)
    ref = Q.forward(det_params(g), oracle_cfg(), ids, am, labels=labels, pixel_values_videos=px.float(), video_grid_thw=g["full_grid"],
                    second_per_grid_ts=np.array([1.0, 1.0]))
    m = am.bool()
    assert rel_l2(out.hidden_states[-1][m], ref["hidden"][m]) < 2e-2
    assert rel_l2(out.logits[m], ref["logits"][m]) < 2e-2
    assert abs(out.loss.item() - ref["loss"].item()) / ref["loss"].item() < 1e-2
    assert abs(out.loss.item() - float(g["full_loss"])) / float(g["full_loss"]) < 2e-2
    assert out.logits[~m].abs().sum().item() == 0  # pad rows are not computed (packed tokens)


def test_rmsnorm_fold_equals_unfolded_route(model, dev):
    """The RMSNorm-folded route of the inference forward (producer leaves row sums of squares, consumer scales its accumulators; rga3_gemm_rms_bf16) against the
    route with stand-alone norm launches on the same weights: ViT features and logits within bf16 noise of each other (both are within 2e-2 of the oracle:
    test_vit_parity / test_full_forward_parity run with the fold on), bit-identical run to run."""
    import rga3.model.qwen2_5_vl as QM

    g = gold()
    px = torch.cat([det_tensor("pixel_values_full0", (192, 1176), 1.0, seed=5), det_tensor("pixel_values_full1", (192, 1176), 1.0, seed=6)], 0).to(torch.bfloat16)
    ids, am = (torch.from_numpy(g[k]) for k in ("full_input_ids", "full_attention_mask"))

    def run():
        with torch.no_grad():
            vit = model.visual(px.to(dev), g["full_grid"])
            out = model(input_ids=ids.to(dev), attention_mask=am.to(dev), pixel_values_videos=px.to(dev), video_grid_thw=torch.from_numpy(g["full_grid"]),
                        second_per_grid_ts=torch.tensor([1.0, 1.0]))
        return vit, out.logits

    assert QM.rms_fold_enabled()
    calls = []
    real = QM.ops.gemm
    QM.ops.gemm = lambda *a, **k: (calls.append(("rms_in" in k and k["rms_in"] is not None, k.get("rms_out") is not None)), real(*a, **k))[1]
    try:
        v1, l1 = run()
    finally:
        QM.ops.gemm = real
    assert sum(c[0] for c in calls) >= 4 and sum(c[1] for c in calls) >= 4, "the folded route did not run"
    v1b, l1b = run()
    assert torch.equal(v1, v1b) and torch.equal(l1, l1b)
    QM.set_rms_fold(False)
    try:
        v0, l0 = run()
    finally:
        QM.set_rms_fold(True)
    m = am.bool()
    assert rel_l2(v1, v0.float().cpu()) < 2e-2          # two bf16 pipelines with different rounding points: each is within 2e-2 of the fp32 oracle
    assert rel_l2(l1[m], l0[m].float().cpu()) < 2e-2


def test_kv_cache_decode_parity(model, dev):
    """Prefill + teacher-forced single-token decode steps through the KV cache reproduce the oracle's full-sequence
    logits (rel-L2 <= 2e-2) and its greedy token wherever the oracle's top-1/top-2 margin exceeds the bf16 noise."""
    from rga3.model.qwen2_5_vl import KVCache

    g = gold()
    full = torch.from_numpy(g["gen_output_ids"])            # prompt + 6 tokens generated by HF greedy decode
    S0 = g["gen_input_ids"].shape[1]
    px = det_tensor("pixel_values_full0", (192, 1176), 1.0, seed=5).to(torch.bfloat16)
    grid, spg = torch.tensor([[2, 8, 12]]), torch.tensor([1.0])
    ref = Q.forward(det_params(g), oracle_cfg(), full, torch.ones_like(full), pixel_values_videos=px.float(), video_grid_thw=grid.numpy(),
                    second_per_grid_ts=np.array([1.0]))["logits"][0]
    c = model.config
    cache = KVCache(c.num_hidden_layers, 1, full.shape[1] + 1, c.num_key_value_heads, c.head_dim, dev, torch.bfloat16)
    with torch.no_grad():
        out = model(input_ids=full[:, :S0].to(dev), attention_mask=torch.ones(1, S0, dtype=torch.long, device=dev), past_key_values=cache,
                    pixel_values_videos=px.to(dev), video_grid_thw=grid, second_per_grid_ts=spg)
        rows = [out.logits[0, -1].float().cpu()]
        for t in range(S0, full.shape[1] - 1):
            am = torch.ones(1, t + 1, dtype=torch.long, device=dev)
            out = model(input_ids=full[:, t:t + 1].to(dev), attention_mask=am, past_key_values=cache)
            rows.append(out.logits[0, -1].float().cpu())
    got = torch.stack(rows)
    want = ref[S0 - 1: full.shape[1] - 1]
    assert rel_l2(got, want) < 2e-2
    top2 = want.topk(2, dim=-1).values
    decided = (top2[:, 0] - top2[:, 1]) > 0.05 * want.abs().max()
    assert torch.equal(got.argmax(-1)[decided], want.argmax(-1)[decided])
    assert decided[0] and int(got[0].argmax()) == int(g["gen_output_ids"][0, S0])  # first generated token matches HF


def test_generate_api(model, dev):
    g = gold()
    ids = torch.from_numpy(g["gen_input_ids"])
    px = det_tensor("pixel_values_full0", (192, 1176), 1.0, seed=5).to(torch.bfloat16)
    seq = model.generate(input_ids=ids.to(dev), attention_mask=torch.ones_like(ids).to(dev), pixel_values_videos=px.to(dev),
                         video_grid_thw=torch.tensor([[2, 8, 12]]), second_per_grid_ts=torch.tensor([1.0]), max_new_tokens=6)
    assert seq.shape == (1, ids.shape[1] + 6) and torch.equal(seq[:, : ids.shape[1]].cpu(), ids)
    assert int(seq[0, ids.shape[1]]) == int(g["gen_output_ids"][0, ids.shape[1]])


def test_generate_graph_decode_equals_eager(model, dev):
    """generate() for one sequence replays a captured hipGraph per decode step (device-side cache length / position); the token ids must equal
    the eager step-by-step loop exactly, including when EOS arrives between two host checks."""
    torch.manual_seed(11)
    ids = torch.randint(10, 300, (1, 37), device=dev)
    with torch.no_grad():
        a = model.generate(input_ids=ids, max_new_tokens=40, do_sample=False, eos_token_id=-1, decode_graph=False)
        b = model.generate(input_ids=ids, max_new_tokens=40, do_sample=False, eos_token_id=-1)
        assert a.shape == (1, 77) and torch.equal(a, b)
        eos = int(a[0, 37 + 11])                      # make an early new token the EOS: both paths must stop at its first occurrence
        first = int((a[0, 37:] == eos).nonzero()[0])
        c = model.generate(input_ids=ids, max_new_tokens=40, do_sample=False, eos_token_id=eos, decode_graph=False)
        d = model.generate(input_ids=ids, max_new_tokens=40, do_sample=False, eos_token_id=eos)
        assert c.shape[1] == 37 + first + 1 and torch.equal(c, d) and torch.equal(c, a[:, :c.shape[1]])
