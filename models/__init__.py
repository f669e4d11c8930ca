"""Drop-in module path of the reference: ``DynamicIngredient("models.passt.model_ing", ...)`` and
``("models.preprocess.model_ing", instance_cmd="AugmentMelSTFT")`` (ex_audioset.py:61-70) resolve here when this
repository is on ``sys.path`` instead of the reference's own ``models/`` package.  Thin re-exports only."""
