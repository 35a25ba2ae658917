"""Base class of the inference-only modules: a torch.nn.Module (so it nests, `.eval()`s and prints
like the reference's modules) that is born in eval mode and refuses train mode."""
import torch


class InferenceModule(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.training = False

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("glass_amd builds the inference hot path only; training is out of scope")
        return super().train(False)
