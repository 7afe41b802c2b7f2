"""Drop-in for the reference's attacks/utils.py import path."""
from dl_attack_on_imagenet_amd.attacks.utils import *  # noqa: F401,F403
from dl_attack_on_imagenet_amd.attacks.utils import (Attack, QuickAttackDataset, clamp_image, compute_fooling_rate,  # noqa: F401
                                                     constraint_dict, get_prox_l1, get_slices, get_target,
                                                     project_onto_l1_ball)
