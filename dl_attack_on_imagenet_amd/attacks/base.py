"""Minimal stand-in for torchattacks.attack.Attack, the third-party base class the reference
subclasses (attacks/utils.py:4, adil.py:38,68).  Only the surface the ADiL path uses:
`.model`, `.device`, `._targeted`, `.attack` (name) and `__call__` -> eval() + forward()."""
import torch


class Attack:
    def __init__(self, name, model):
        self.attack = name
        self.model = model
        self.model_name = type(model).__name__
        try:
            self.device = next(model.parameters()).device
        except StopIteration:
            self.device = torch.device("cpu")
        self._targeted = False

    def forward(self, *inputs):
        raise NotImplementedError

    def __call__(self, *inputs, **kwargs):
        self.model.eval()
        return self.forward(*inputs, **kwargs)

    def __str__(self):
        public = {k: v for k, v in self.__dict__.items()
                  if not k.startswith("_") and k not in ("model", "dictionary", "data_train", "data_val")
                  and isinstance(v, (int, float, str, bool, type(None)))}
        return f"{type(self).__name__}(" + ", ".join(f"{k}={v}" for k, v in public.items()) + ")"
