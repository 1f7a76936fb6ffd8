"""Host-side distribution wrappers returned by `VIMAPolicy.forward_action_decoder`.

Same interface as the reference's `vima/nn/action_decoder/dists.py:7-28`: a `MultiCategorical` holds one
`Categorical` per action dimension (logits are normalised by `torch.distributions.Categorical`, i.e.
`logits - logsumexp`), `.mode()` is the per-dimension argmax stacked on the last axis. The logits themselves
come from the HIP action-head kernels; this is only the container the eval loop expects.
"""
from __future__ import annotations

import torch

__all__ = ["Categorical", "MultiCategorical"]


class Categorical(torch.distributions.Categorical):
    def mode(self):
        return self.logits.argmax(dim=-1)


class MultiCategorical(torch.distributions.Distribution):
    def __init__(self, logits: torch.Tensor, action_dims):
        if logits.dim() < 2:
            raise AssertionError(tuple(logits.shape))
        super().__init__(batch_shape=logits.shape[:-1], validate_args=False)
        self._action_dims = tuple(action_dims)
        if logits.size(-1) != sum(self._action_dims):
            raise AssertionError(f"sum of action dims {self._action_dims} != {logits.size(-1)}")
        self.raw_logits = logits
        self._dists = [Categorical(logits=s) for s in torch.split(logits, list(self._action_dims), dim=-1)]

    def mode(self):
        return torch.stack([torch.argmax(d.probs, dim=-1) for d in self._dists], dim=-1)

    def log_prob(self, actions):
        return torch.stack([d.log_prob(a) for d, a in zip(self._dists, torch.unbind(actions, dim=-1))], dim=-1).sum(-1)

    def entropy(self):
        return torch.stack([d.entropy() for d in self._dists], dim=-1).sum(-1)

    def sample(self, sample_shape=torch.Size()):
        return torch.stack([d.sample(sample_shape) for d in self._dists], dim=-1)
