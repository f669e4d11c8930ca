"""passt_amd -- the kkoutini/PaSST training hot path, hand-written for MI355X (gfx950).

Public surface mirrors the reference's two model modules:
  passt_amd.passt       <-> models/passt.py       (PaSST, get_model, get_model_passt, ...)
  passt_amd.preprocess  <-> models/preprocess.py  (AugmentMelSTFT)
plus the caller glue of ex_audioset.py's training step (passt_amd.train) and the data-parallel
gradient reducer (passt_amd.ddp).  All compute goes through libpasst_amd.so (include/passt_amd.h).
"""
from .passt import PaSST, get_model, get_model_passt, get_ensemble_model, EnsembelerModel  # noqa: F401
from .preprocess import AugmentMelSTFT  # noqa: F401

__version__ = "0.1.0"
