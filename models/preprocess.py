"""``models.preprocess`` of the reference (models/preprocess.py) served by the MI355X implementation."""
from passt_amd.preprocess import AugmentMelSTFT, model_ing  # noqa: F401
