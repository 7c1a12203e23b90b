from rga3.model.qwen_2_5_vl_sam2 import UniGRConfig, UniGRModel, dice_loss, sigmoid_ce_loss  # noqa: F401
