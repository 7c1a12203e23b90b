from rga3.model.sam2 import SAM2, SAM2VideoPredictor, VideoSession, load_sam2_checkpoint  # noqa: F401
