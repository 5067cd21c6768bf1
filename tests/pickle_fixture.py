"""Stand-ins used by the checkpoint-converter tests: a wrapper with the shape of LightningLite's `_LiteModule`
(pytorch_lightning/lite/wrappers.py in 1.8.x keeps the user's module in `_forward_module`, so every state_dict key
of a training-time pickle starts with `_forward_module.`) and a tiny module to wrap.  Pickles name classes by module
path, hence this importable file."""
import torch


class _LiteModule(torch.nn.Module):
    def __init__(self, forward_module):
        super().__init__()
        self._forward_module = forward_module

    def forward(self, *a, **k):
        return self._forward_module(*a, **k)


class TinyHead(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.linear = torch.nn.Linear(8, 4)
        self.norm = torch.nn.BatchNorm1d(4)

    def forward(self, x):
        return self.norm(self.linear(x))
