"""Comparison baselines of the reference (uappgd.py, fast_uap.py) are out of the ADiL hot-path scope
(SURVEY.md §2 rows 5-6).  The names stay importable so `from attacks import ADILR, UAPPGD, FastUAP, ADIL`
(attacks/__init__.py:1-5) keeps working; constructing one raises."""
from .base import Attack


class UAPPGD(Attack):
    def __init__(self, model, *args, **kwargs):
        raise NotImplementedError("UAPPGD (uappgd.py) is a comparison baseline outside the ADiL hot path")


class FastUAP(Attack):
    def __init__(self, model, *args, **kwargs):
        raise NotImplementedError("FastUAP (fast_uap.py) is a comparison baseline outside the ADiL hot path")
