"""``models.passt`` of the reference (models/passt.py) served by the MI355X implementation."""
from passt_amd.passt import *  # noqa: F401,F403
from passt_amd.passt import (EnsembelerModel, PaSST, fix_embedding_layer, get_ensemble_model, get_model,  # noqa: F401
                             get_model_passt, lighten_model, model_ing)
