"""Drop-in module tree with the reference's import paths (model.qwen_2_5_vl_sam2, model.sam2, model.STOM).

Put ``rga3-release_amd/dropin`` first on sys.path and the reference's entry points keep their import lines:
``from model.qwen_2_5_vl_sam2 import UniGRConfig, UniGRModel`` (train_joint.py:23, app.py:17).
"""
import os
import sys

_PKG = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)
