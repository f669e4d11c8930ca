"""``models.passt`` of the reference (models/passt.py) served by the MI355X implementation.

Drop-in module path: ``DynamicIngredient("models.passt.model_ing", ...)`` (ex_audioset.py:61-70) resolves here when
this repository is on ``sys.path`` instead of the reference.  ``models/`` deliberately has no ``__init__.py``: like
the reference's own ``models/`` it is a namespace package, so whichever tree comes first on ``sys.path`` wins.
"""
from passt_amd.passt import *  # noqa: F401,F403
from passt_amd.passt import (EnsembelerModel, PaSST, fix_embedding_layer, get_ensemble_model, get_model,  # noqa: F401
                             get_model_passt, lighten_model, model_ing)
