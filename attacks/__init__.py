"""Drop-in import surface of the reference: `from attacks import ADILR, UAPPGD, FastUAP, ADIL`
(reference attacks/__init__.py:1-5).  Implementation: dl_attack_on_imagenet_amd.attacks."""
from dl_attack_on_imagenet_amd.attacks import ADILR, UAPPGD, FastUAP, ADIL, Attack_dict_model  # noqa: F401
