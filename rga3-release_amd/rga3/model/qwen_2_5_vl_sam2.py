"""UniGR joint model — drop-in for reference model/qwen_2_5_vl_sam2.py (same class / method names, kwargs, return
types and dict keys: SURVEY.md 8(b)).  ``from model.qwen_2_5_vl_sam2 import UniGRConfig, UniGRModel`` works when
``rga3-release_amd/rga3`` is on sys.path.

The arithmetic runs on the rga3 HIP kernels (rga3.hip.ops); output-preserving waste of the reference is removed
(each item covered by tests/test_unigr_gpu.py):
  * text_hidden_fcs is evaluated only on the rows that are gathered afterwards (reference: all S positions, :215-218);
  * the argmax-IoU candidate is selected before the 1024^2 upsample (reference sam2.py:3388-3402);
  * samples without a [SEG] token skip SAM2 entirely in training (their loss slice is empty, reference :289-290);
  * no torch.cuda.empty_cache() per forward (reference :313).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from ..hip import autograd as AG
from ..hip import ops
from ..utils.staging import host_of, upload
from .qwen2_5_vl import CausalLMOutput, Linear, Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration
from .sam2 import SAM2


def _mask_sums(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """per-mask {sum bce, sum sigmoid*t, sum sigmoid, sum t} via the HIP reduction kernel; pred/target [n, h, w]."""
    return ops.bce_dice_sums(pred.float().contiguous(), target.float().contiguous())


def dice_loss(inputs: torch.Tensor, targets: torch.Tensor, num_masks: float, scale=1000, eps=1e-6):
    """reference model/qwen_2_5_vl_sam2.py:17-40"""
    if inputs.shape[0] == 0:
        return inputs.sum() * 0.0
    s = _mask_sums(inputs, targets)
    numerator = 2 * (s[:, 1] / scale)
    denominator = s[:, 2] / scale + s[:, 3] / scale
    loss = 1 - (numerator + eps) / (denominator + eps)
    return loss.sum() / (num_masks + 1e-8)


def sigmoid_ce_loss(inputs: torch.Tensor, targets: torch.Tensor, num_masks: float):
    """reference model/qwen_2_5_vl_sam2.py:43-60"""
    if inputs.shape[0] == 0:
        return inputs.sum() * 0.0
    s = _mask_sums(inputs, targets)
    hw = inputs[0].numel()
    return (s[:, 0] / hw).sum() / (num_masks + 1e-8)


class UniGRConfig(Qwen2_5_VLConfig):
    """reference :82-101"""

    def __init__(self, train_mask_decoder=False, out_dim=256, ce_loss_weight=.0, dice_loss_weight=.0, bce_loss_weight=.0,
                 seg_token_idx=0, sam_pretrained=None, **kwargs):
        self.train_mask_decoder = train_mask_decoder
        self.out_dim = out_dim
        self.ce_loss_weight = ce_loss_weight
        self.dice_loss_weight = dice_loss_weight
        self.bce_loss_weight = bce_loss_weight
        self.seg_token_idx = seg_token_idx
        self.sam_pretrained = sam_pretrained
        self.sam_config = kwargs.pop("sam_config", None)  # optional SAM2 size overrides (tests use tiny SAM2s)
        super().__init__(**kwargs)


class UniGRModel(Qwen2_5_VLForConditionalGeneration):
    config_class = UniGRConfig

    def __init__(self, config):
        super().__init__(config)
        if not config.train_mask_decoder:  # inference mode (reference :111-115)
            self.initialize_sam_modules(config)

    @classmethod
    def from_pretrained(cls, path, config=None, torch_dtype=None, device=None, strict=True, **unused):
        """HF-style constructor used by the reference's entry points (train_joint.py:176-184, evaluation/*/inference_*.py):
        builds the model from `config` (or <path>/config.json) directly in `torch_dtype` on `device`, then copies the checkpoint under
        `path` shard by shard (rga3.utils.checkpoint).  Keyword arguments that only steer HF internals (attn_implementation,
        low_cpu_mem_usage, device_map, use_cache, ...) are accepted and ignored: attention is this library's own kernel."""
        from ..utils import checkpoint as CK

        cfg = config if config is not None else UniGRConfig.from_pretrained(path)
        old = torch.get_default_dtype()
        if torch_dtype is not None:
            torch.set_default_dtype(torch_dtype)
        try:
            with torch.device(device if device is not None else "cpu"):
                model = cls(cfg)
        finally:
            torch.set_default_dtype(old)
        CK.load_checkpoint(model, path, strict=strict)
        return model

    def save_pretrained(self, path, state_dict=None, max_shard_size=5 * 2**30, **unused):
        """HF layout (sharded safetensors + index + config.json); `state_dict=` as merge_lora_weights_and_save_hf_model.py:134 passes it."""
        from ..utils import checkpoint as CK

        CK.save_checkpoint(state_dict if state_dict is not None else self, path, max_shard_bytes=max_shard_size, config=self.config)

    def merge_and_unload(self):
        """PEFT's call on the wrapped model (merge_lora_weights_and_save_hf_model.py:133): fold the LoRA factors into q_proj / v_proj."""
        from ..utils import checkpoint as CK

        CK.merge_lora_(self)
        return self

    def initialize_sam_modules(self, config):
        """reference :117-140"""
        self.grounding_encoder = SAM2(ckpt_path=config.sam_pretrained, **(getattr(config, "sam_config", None) or {}))
        self.grounding_encoder.sam2_model.requires_grad_(False)
        if config.train_mask_decoder:
            self.grounding_encoder.sam2_model.sam_mask_decoder.train()
            self.grounding_encoder.sam2_model.sam_mask_decoder.requires_grad_(True)
        else:
            self.grounding_encoder.sam2_model.sam_mask_decoder.eval()
        in_dim, out_dim = config.hidden_size, config.out_dim
        text_fc = [Linear(in_dim, in_dim), nn.ReLU(inplace=True), Linear(in_dim, out_dim), nn.Dropout(0.0)]
        self.text_hidden_fcs = nn.ModuleList([nn.Sequential(*text_fc)])
        self.text_hidden_fcs.train()
        for p in self.text_hidden_fcs.parameters():
            p.requires_grad = True
        ref = self.lm_head.weight
        self.grounding_encoder.to(device=ref.device, dtype=ref.dtype)
        self.text_hidden_fcs.to(device=ref.device, dtype=ref.dtype)

    def forward(self, **kwargs):
        """reference :143-146"""
        if "past_key_values" in kwargs:
            return super().forward(**kwargs)
        return self.model_forward(**kwargs)

    def prefetch_next(self, **next_batch):
        """Tell the model which batch comes NEXT (its pixel tensors: `pixel_values_videos` / `video_grid_thw`, `pixel_values` / `image_grid_thw`).  The training
        forward then launches the frozen vision tower for it on a side stream right after this step's SAM2 image encoder -- where the launch-bound mask path begins and
        most CUs fall idle (Qwen2_5_VLForConditionalGeneration.prefetch_vision).  Optional; a forward without it computes the features itself, bit-identically."""
        self.__dict__["_next_pixels"] = {k: next_batch.get(k) for k in ("pixel_values", "image_grid_thw", "pixel_values_videos", "video_grid_thw")}

    def _launch_prefetch(self):
        nxt = self.__dict__.pop("_next_pixels", None)
        if nxt is not None:
            self.prefetch_vision(**nxt)

    # -- frozen SAM2 image encoder, one step ahead --------------------------------------------------------------------------
    def prefetch_sam(self, images_sam, after=None):
        """(after: a trainable parameter -- the launch is DEFERRED to the moment that parameter's gradient has been accumulated in the running / next backward, once.  The
        encoder opens with its bandwidth-bound stages (1 M tokens x 144 channels per frame batch): started a few decoder layers before the end of backward they run
        beside matrix-bound dX products, and the matrix-bound stage 3 then meets the bandwidth-bound optimizer.)
        Run the FROZEN SAM2 image encoder (Hiera-L trunk + FPN: a third of a training step, all matrix work) on the NEXT batch's `images_sam` now, on a side
        stream, and keep the result for the forward that receives this same tensor.  Meant to be called right before the optimizer step: AdamW streams 28 bytes per
        trainable element at the HBM roofline and leaves the matrix pipe idle (5.9 ms per step at 1.15 B elements), the encoder is the opposite -- the two overlap,
        and the rest of the encoder runs beside the next forward's products.  It reads no parameter the optimizer updates (reference qwen_2_5_vl_sam2.py:121 freezes
        the grounding encoder; conv_s0 / conv_s1 of the trainable mask decoder are applied in the forward itself), every step still runs exactly one encoder pass per
        sample, and the features are bit-identical to the ones computed in line (same kernels).  Optional: a forward without a matching prefetch computes them itself."""
        gm = self.grounding_encoder
        enc = gm.sam2_model.image_encoder
        if images_sam is None or not images_sam.is_cuda or images_sam.requires_grad or any(p_.requires_grad for p_ in enc.parameters()):
            return
        if after is not None:
            # at most ONE pending hook on the model: a new request replaces the sample of a hook that never fired (backward skipped, parameter frozen
            # later, micro-step without its gradient) instead of stacking another closure over a stale tensor; the hook reads the latest request when it fires.
            pend = self.__dict__.get("_pf_sam_pending")
            if pend is not None and pend["param"] is after:
                pend["images"] = images_sam
                return
            self._drop_pending_sam_prefetch()
            pend = self.__dict__["_pf_sam_pending"] = {"param": after, "images": images_sam}

            def fire(_p):
                cur = self.__dict__.get("_pf_sam_pending")
                if cur is None or cur is not pend:
                    return
                self._drop_pending_sam_prefetch()          # one shot
                self.prefetch_sam(cur["images"])

            pend["handle"] = after.register_post_accumulate_grad_hook(fire)
            return
        self._drop_pending_sam_prefetch()
        st = self.__dict__.get("_pf_sam_stream")
        if st is None:
            st = self.__dict__["_pf_sam_stream"] = torch.cuda.Stream(device=images_sam.device)
        cache = self.__dict__.setdefault("_pf_sam_cache", {})
        cache.clear()
        cur = torch.cuda.current_stream(images_sam.device)
        ready = torch.cuda.Event()
        ready.record(cur)
        for i in range(images_sam.shape[0]):
            img = images_sam[i]
            with torch.cuda.stream(st), torch.no_grad():
                st.wait_event(ready)
                images_sam.record_stream(st)
                lv = gm.sam2_model.encode_frozen(img)
                done = torch.cuda.Event()
                done.record(st)
            cache[img.data_ptr()] = ((images_sam._version, tuple(img.shape), img.dtype), images_sam, lv, done)   # holds images_sam: its address cannot be recycled meanwhile

    def _drop_pending_sam_prefetch(self):
        pend = self.__dict__.pop("_pf_sam_pending", None)
        if pend is not None and pend.get("handle") is not None:
            pend["handle"].remove()

    def _prefetched_sam(self, images_sam, i):
        pend = self.__dict__.get("_pf_sam_pending")
        if pend is not None and pend["images"] is images_sam:
            self._drop_pending_sam_prefetch()     # its gradient never arrived and the sample is being forwarded now: the request is moot
        cache = self.__dict__.get("_pf_sam_cache")
        img = images_sam[i]
        hit = cache.pop(img.data_ptr(), None) if cache else None
        if hit is None or hit[0] != (images_sam._version, tuple(img.shape), img.dtype):
            return None
        _, _, lv, done = hit
        cur = torch.cuda.current_stream(images_sam.device)
        cur.wait_event(done)
        for t, _, _ in lv:
            t.record_stream(cur)              # allocated on the side stream, consumed (and later freed) on this one
        return lv

    # ---- helpers ------------------------------------------------------------------------------------------
    def _seg_embeddings(self, hidden_last: torch.Tensor, seg_token_mask_np: np.ndarray, pl=None):
        """text_hidden_fcs on the gathered rows (value-identical to MLP-then-gather, reference :215-218)."""
        B, S, H = hidden_last.shape
        where = np.flatnonzero(seg_token_mask_np.reshape(-1))
        counts = seg_token_mask_np.sum(-1).astype(np.int64)
        if where.size == 0:
            return torch.zeros((0, self.config.out_dim), dtype=hidden_last.dtype, device=hidden_last.device), counts
        pl = pl if pl is not None else {}
        if "seg_rows" not in pl:
            pl["seg_rows"] = upload(where, hidden_last.device)
        idx = pl["seg_rows"]
        fc = self.text_hidden_fcs[0]
        if torch.is_grad_enabled():
            rows = AG.GatherRowsFn.apply(hidden_last.reshape(B * S, H), idx)
            return AG.linear(AG.linear(rows, fc[0].weight, fc[0].bias, None, "relu"), fc[2].weight, fc[2].bias), counts
        rows = ops.gather_rows(hidden_last.reshape(B * S, H), idx)
        return fc[2](fc[0](rows, act="relu")), counts

    @staticmethod
    def _shifted_seg_mask(ids_np: np.ndarray, seg_idx: int) -> np.ndarray:
        """(ids == seg)[:, 1:] with a False column appended: the position that PREDICTS [SEG] (reference :209-210, :343-344)."""
        m = ids_np == seg_idx
        return np.concatenate([m[:, 1:], np.zeros_like(m[:, :1])], axis=1)

    # ---- training / validation forward ---------------------------------------------------------------------
    def model_forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None, labels=None,
                      use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None, pixel_values=None,
                      pixel_values_videos=None, image_grid_thw=None, video_grid_thw=None, rope_deltas=None, cache_position=None,
                      second_per_grid_ts=None, images_sam=None, offset=None, masks_list=None, label_list=None, resize_list=None,
                      inference: bool = False, **kwargs):
        """reference :149-321"""
        batch_size, num_frames_sam = images_sam.shape[:2]
        device = images_sam.device
        assert batch_size == len(offset) - 1
        output = super().forward(input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids, labels=labels,
                                 output_hidden_states=True, pixel_values=pixel_values, pixel_values_videos=pixel_values_videos,
                                 image_grid_thw=image_grid_thw, video_grid_thw=video_grid_thw, second_per_grid_ts=second_per_grid_ts)
        ce_loss = output.loss * self.config.ce_loss_weight
        pl = self.__dict__.get("_last_plan") or {}   # host plan of the forward just run: holds the labels' host copy (no second device -> host read)
        labels_np = pl["labels_np"] if pl.get("labels_np") is not None else host_of(labels)
        seg_mask = self._shifted_seg_mask(labels_np, self.config.seg_token_idx)
        pred_embeddings, counts = self._seg_embeddings(output.hidden_states[-1], seg_mask, pl)
        seg_token_offset = np.concatenate([[0], np.cumsum(counts)])[np.asarray(host_of(offset))]
        gm = self.grounding_encoder
        out_dim = self.config.out_dim

        def sample_embedding(i):
            a, b = int(seg_token_offset[i]), int(seg_token_offset[i + 1])
            if a == b:
                return torch.zeros((1, out_dim), device=device, dtype=images_sam.dtype)
            return pred_embeddings[a:b]

        if inference:
            pred_masks = []
            for i in range(batch_size):
                e = sample_embedding(i)
                sess = gm.get_sam2_embeddings(images_sam[i])
                masks = gm.language_embd_inference(sess, [e] * num_frames_sam)  # [T, n_obj, S, S]
                h, w = label_list[i].shape
                masks = ops.bilinear(masks[:, 0].contiguous(), (h, w))
                pred_masks.append(masks > 0)  # sigmoid(x) > 0.5
            return {"pred_masks": pred_masks, "gt_masks": masks_list}

        has_seg = np.diff(seg_token_offset) > 0
        mask_bce_loss = torch.zeros((), device=device)
        mask_dice_loss = torch.zeros((), device=device)
        num_masks = 0
        for i in range(batch_size):
            gt_mask = masks_list[i]
            if not has_seg[i]:
                assert gt_mask.shape[0] == 0, f"gt_mask.shape: {gt_mask.shape}, pred_mask.shape: (0, ...)"
                continue  # empty slice: contributes 0 to both losses and to num_masks (reference :289-305)
            e = sample_embedding(i)
            assert e.shape[0] == 1, "one [SEG] per sample on the training path (reference sam2.py:3356 assert)"
            st = gm.get_sam2_embeddings_train(images_sam[i], frozen=self._prefetched_sam(images_sam, i))
            self._launch_prefetch()      # next batch's frozen ViT (if the trainer announced it) goes out beside the mask decoder / its backward
            _, high = gm.inject_language_embd_train(st, e[None].expand(num_frames_sam, -1, -1))
            grad = torch.is_grad_enabled()
            pred = AG.BilinearFn.apply(high[:, 0].contiguous(), tuple(label_list[i].shape), None) if grad else ops.bilinear(high[:, 0].contiguous(), tuple(label_list[i].shape))
            assert gt_mask.shape[0] == pred.shape[0], "gt_mask.shape: {}, pred_mask.shape: {}".format(gt_mask.shape, pred.shape)
            n = gt_mask.shape[0]
            if grad:   # sum over masks of (mean BCE, dice) == sigmoid_ce_loss * n, dice_loss * n of the reference (:297-304)
                b_, d_ = AG.MaskLossFn.apply(pred, gt_mask.to(device))
                mask_bce_loss, mask_dice_loss = mask_bce_loss + b_, mask_dice_loss + d_
            else:
                mask_bce_loss = mask_bce_loss + sigmoid_ce_loss(pred, gt_mask.to(device), num_masks=n) * n
                mask_dice_loss = mask_dice_loss + dice_loss(pred, gt_mask.to(device), num_masks=n) * n
            num_masks += n
        self._launch_prefetch()          # (a batch without [SEG]: no SAM2 pass happened above)
        mask_bce_loss = self.config.bce_loss_weight * mask_bce_loss / (num_masks + 1e-8)
        mask_dice_loss = self.config.dice_loss_weight * mask_dice_loss / (num_masks + 1e-8)
        mask_loss = mask_bce_loss + mask_dice_loss
        loss = ce_loss + mask_loss
        return {"loss": loss, "ce_loss": ce_loss, "mask_bce_loss": mask_bce_loss, "mask_dice_loss": mask_dice_loss, "mask_loss": mask_loss}

    # ---- inference ---------------------------------------------------------------------------------------------
    def evaluate(self, input_ids=None, attention_mask=None, pixel_values=None, pixel_values_videos=None, image_grid_thw=None,
                 video_grid_thw=None, second_per_grid_ts=None, images_sam=None, resize_list=None, original_size_list=None):
        """reference :325-393 -> (hf_output, [bool masks [T, h, w] per [SEG]])"""
        with torch.no_grad():
            assert images_sam.shape[0] == 1
            seg_mask = self._shifted_seg_mask(host_of(input_ids), self.config.seg_token_idx)
            output = super().forward(input_ids=input_ids, attention_mask=attention_mask, pixel_values=pixel_values,
                                     pixel_values_videos=pixel_values_videos, image_grid_thw=image_grid_thw, video_grid_thw=video_grid_thw,
                                     second_per_grid_ts=second_per_grid_ts, output_hidden_states=True)
            pred_embeddings, counts = self._seg_embeddings(output.hidden_states[-1], seg_mask)
            seg_token_offset = np.concatenate([[0], np.cumsum(counts)])
            per_sample = [pred_embeddings[int(seg_token_offset[i]):int(seg_token_offset[i + 1])] for i in range(len(seg_token_offset) - 1)]
            pred_masks = []
            gm = self.grounding_encoder
            sess = None
            for i, e in enumerate(per_sample):
                # one image-encoder pass over the clip serves every [SEG] of the sample
                sess = gm.get_sam2_embeddings(images_sam[0]) if sess is None else type(sess)(gm.sam2_model, images_sam[0], feats=sess._ensure_feats())
                masks = gm.language_embd_inference(sess, [e] * images_sam.shape[1])
                h, w = original_size_list[i]
                masks = ops.bilinear(masks[:, 0].contiguous(), (int(h), int(w)))
                pred_masks.append(masks > 0)
        ops.poll_gemm_health()
        return output, pred_masks
