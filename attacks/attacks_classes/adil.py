"""Drop-in for the reference's attacks/attacks_classes/adil.py import path."""
from dl_attack_on_imagenet_amd.attacks.adil import ADIL, Attack_dict_model  # noqa: F401
