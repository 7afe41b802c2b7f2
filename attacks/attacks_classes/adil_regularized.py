"""Drop-in for the reference's attacks/attacks_classes/adil_regularized.py import path."""
from dl_attack_on_imagenet_amd.attacks.adil_regularized import (ADILR, Attack_dict_model, adil,  # noqa: F401
                                                                learn_coding_vectors, sadil, sadil_updated)
