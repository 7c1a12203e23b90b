"""Generates tests/golden/sam2_multiobj.npz: the REFERENCE's own multi-object paths at n_obj = 2 on the tiny predictor and clip of make_sam2_fixtures.py (same
weights: name-derived + the fitted read-out stored in sam2_tiny.npz).  Build container only:  python tests/golden/make_sam2_multiobj_fixtures.py

  A  SAM2.language_embd_inference(state, [[e_obj0, e_obj1]] * T)  (/root/reference/model/sam2.py:378-404): every object prompted on every frame, then
     propagate_in_video; each yield is [n_obj, 1, S, S] (:4049-4132 through _get_orig_video_res_output) and the wrapper concatenates the yields on dim 0 ->
     [T * n_obj, 1, S, S], FRAME-major (frame 0 obj 0, frame 0 obj 1, frame 1 obj 0, ...);
  B  add_language_embd(state, 0, 100, e0) + add_language_embd(state, 0, 101, e1), then propagate_in_video: the memory path with a batch of two objects (the per-frame
     inference runs with batch_size = n_obj on shared image features, :3977-4047, :3630-3747) -> masks [T * n_obj, 1, S, S] frame-major, object pointers [T, n_obj, C].
The second object's prompt is the negated first (a different, deterministic embedding), so the two objects' masks differ.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_sam2_fixtures as M  # noqa: E402  (sets up the reference import shims)

from oracle.detweights import det_tensor  # noqa: E402
from tests.sam2_tiny import gold, images, lang  # noqa: E402


def second_prompt(T):
    return det_tensor("lang_embd_obj1", (T, 1, 256), 4.0, seed=9).to(torch.bfloat16).float()     # 4 x the scale of object 0: masks that differ visibly


def main():
    G = gold()
    fitted = {k[5:]: torch.from_numpy(G[k]) for k in G.files if k.startswith("fit::")}
    wrap, _ = M.build_tiny_predictor(overrides=fitted)
    pred = wrap.sam2_model
    imgs, e0 = images(), lang()
    T = imgs.shape[0]
    e1 = second_prompt(T)
    out = {}
    with torch.no_grad():
        counts, hs = M.count_calls(pred)
        state = wrap.get_sam2_embeddings(imgs)
        state["device"] = state["storage_device"] = torch.device("cpu")
        masks = wrap.language_embd_inference(state, [torch.cat([e0[t], e1[t]], 0) for t in range(T)])      # language_embd[t][obj] -> [256]
        out["A_masks"] = masks.numpy()
        out["A_counts"] = np.array([counts[k] for k in ("enc", "memattn", "memenc", "dec")])
        for h in hs:
            h.remove()
        counts, hs = M.count_calls(pred)
        state = pred.init_state(imgs)
        state["device"] = state["storage_device"] = torch.device("cpu")
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            pred.add_language_embd(state, 0, 100, e0[0][None], inference=True)
            pred.add_language_embd(state, 0, 101, e1[0][None], inference=True)
            res, ids = [], []
            for t, obj_ids, m in pred.propagate_in_video(state):
                res.append(m)
                ids.append(list(obj_ids))
        out["B_masks"] = torch.cat(res, 0).numpy()
        out["B_obj_ids"] = np.array(ids)
        out["B_counts"] = np.array([counts[k] for k in ("enc", "memattn", "memenc", "dec")])
        od = state["output_dict"]
        out["B_obj_ptrs"] = np.stack([od["cond_frame_outputs" if t == 0 else "non_cond_frame_outputs"][t]["obj_ptr"].numpy() for t in range(T)])
        for h in hs:
            h.remove()
    for k in ("A_masks", "B_masks"):
        m = torch.from_numpy(out[k]).reshape(T, 2, 128, 128)
        print(k, out[k].shape, "margin obj0 / obj1:", [[round(float((x.abs() > 0.05 * x.abs().max()).float().mean()), 3) for x in m[:, o]] for o in (0, 1)],
              "obj0 vs obj1 sign agreement:", [round(float(((a > 0) == (b > 0)).float().mean()), 3) for a, b in zip(m[:, 0], m[:, 1])])
    print("counts A", out["A_counts"], "B", out["B_counts"], "obj ids", out["B_obj_ids"][0], "ptrs", out["B_obj_ptrs"].shape)
    np.savez_compressed(os.path.join(M.OUT, "sam2_multiobj.npz"), **out)
    print("wrote sam2_multiobj.npz")


if __name__ == "__main__":
    main()
