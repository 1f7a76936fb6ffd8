"""Host-side containers of the policy API surface."""
from __future__ import annotations


class MapDict(dict):
    """Nested dict exposing `.map_structure(func=)` like the reference's DataDict
    (vima/utils.py:495-508); `forward_obs_token` calls it (vima_policy.py:246)."""

    def map_structure(self, func):
        def rec(x):
            if isinstance(x, dict):
                return MapDict({k: rec(v) for k, v in x.items()})
            return func(x)
        return rec(self)

    def to(self, device):
        return self.map_structure(lambda x: x.to(device))
