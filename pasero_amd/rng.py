"""Dropout RNG bookkeeping.  Masks are Philox4x32-10 functions of (seed, offset, element index) computed inside the
kernels; this module only hands out a fresh `offset` per dropout call site so that masks are independent across
layers and steps, and the backward pass can regenerate the forward mask from the same pair."""
import torch

_seed = None
_offset = 0


def manual_seed(seed: int) -> None:
    global _seed, _offset
    _seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    _offset = 0


def next_offset():
    """-> (seed, offset).  Seeded lazily from torch's global seed (so `torch.manual_seed` / Pasero's
    `utils.set_random_seed` control dropout here too)."""
    global _seed, _offset
    if _seed is None:
        manual_seed(torch.initial_seed())
    _offset += 1
    return _seed, _offset


def get_state():
    return _seed, _offset


def set_state(state) -> None:
    global _seed, _offset
    _seed, _offset = state
