"""Same export list as the reference's attacks/__init__.py:1-5."""
from .adil_regularized import ADILR
from .baselines import UAPPGD, FastUAP
from .adil import ADIL, Attack_dict_model

__all__ = ["ADILR", "UAPPGD", "FastUAP", "ADIL", "Attack_dict_model"]
