"""Generates tests/golden/sam2_reverse.npz: the REFERENCE's own SAM2VideoPredictor.propagate_in_video(start_frame_idx, max_frame_num_to_track, reverse) and
track_in_reverse memory selection (/root/reference/model/sam2.py:4049-4132, :2829-2928) on the tiny predictor and clip of make_sam2_fixtures.py (same weights:
name-derived + the fitted read-out stored in sam2_tiny.npz).  Build container only:  python tests/golden/make_sam2_reverse_fixtures.py

Cases (5-frame clip, one language prompt):
  A  prompt on frame 2; propagate forward (frames 2, 3, 4); then propagate reverse from frame 2 (frames 2, 1, 0) on the SAME state -- the reverse pass attends to the
     memories the forward pass left on frames 3 and 4 (reference :2866-2893: frame_idx + t_rel, whatever pass produced it);
  B  prompt on frame 4; reverse with max_frame_num_to_track = 2 (frames 4, 3, 2);
  C  prompt on frame 1; forward with start_frame_idx = 1, max_frame_num_to_track = 2 (frames 1, 2, 3);
  D  prompt on frame 0; reverse from frame 0: the reference yields nothing (:4087-4090).
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_sam2_fixtures as M  # noqa: E402  (sets up the reference import shims)

from tests.sam2_tiny import gold, images, lang  # noqa: E402


def main():
    G = gold()
    fitted = {k[5:]: torch.from_numpy(G[k]) for k in G.files if k.startswith("fit::")}
    wrap, _ = M.build_tiny_predictor(overrides=fitted)
    pred = wrap.sam2_model
    imgs, emb = images(), lang()
    out = {}

    def run(prompt_frame, passes):
        state = pred.init_state(imgs)
        state["device"] = state["storage_device"] = torch.device("cpu")
        res = []
        with torch.no_grad(), torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            pred.add_language_embd(state, prompt_frame, 100, emb[0][None], inference=True)
            for kw in passes:
                fr, ms = [], []
                for t, _, m in pred.propagate_in_video(state, **kw):
                    fr.append(t)
                    ms.append(m)
                res.append((fr, torch.cat(ms, 0) if ms else torch.zeros(0, 1, 128, 128)))
        od = state["output_dict"]
        ptrs = {t: o["obj_ptr"] for k in ("cond_frame_outputs", "non_cond_frame_outputs") for t, o in od[k].items()}
        return res, ptrs

    (fa, ra), pa = run(2, [dict(), dict(start_frame_idx=2, reverse=True)])
    out["A_fwd_frames"], out["A_fwd_masks"] = np.array(fa[0]), fa[1].numpy()
    out["A_rev_frames"], out["A_rev_masks"] = np.array(ra[0]), ra[1].numpy()
    out["A_obj_ptrs"] = np.stack([pa[t].numpy() for t in range(5)])
    (rb,), _ = run(4, [dict(reverse=True, max_frame_num_to_track=2)])
    out["B_frames"], out["B_masks"] = np.array(rb[0]), rb[1].numpy()
    (rc,), _ = run(1, [dict(start_frame_idx=1, max_frame_num_to_track=2)])
    out["C_frames"], out["C_masks"] = np.array(rc[0]), rc[1].numpy()
    (rd,), _ = run(0, [dict(reverse=True)])
    out["D_frames"] = np.array(rd[0], dtype=np.int64)
    for k in ("A_fwd", "A_rev", "B", "C"):
        m = torch.from_numpy(out[k + "_masks"]).reshape(-1, 128, 128)
        print(k, out[k + "_frames"].tolist(), "margin:", [round(float((x.abs() > 0.05 * x.abs().max()).float().mean()), 3) for x in m])
    print("D", out["D_frames"].tolist())
    np.savez_compressed(os.path.join(M.OUT, "sam2_reverse.npz"), **out)
    print("wrote sam2_reverse.npz")


if __name__ == "__main__":
    main()
