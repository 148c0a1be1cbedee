"""Deterministic, machine-independent weights for parity tests.

Every tensor of a state_dict is filled from a numpy PCG64 stream seeded by (seed, crc32(key)),
so the value of a tensor depends only on its key name and shape -- not on construction order or
on torch's RNG.  ``oracle/gen_golden.py`` loads these into the *reference* model to produce the
golden outputs; the tests load the same values into the build's model.
"""
import zlib

import numpy as np
import torch


def fill_state_dict(sd, seed=0):
    """Returns a new dict with the same keys/shapes/dtypes as ``sd`` and deterministic values."""
    out = {}
    for k, v in sd.items():
        rng = np.random.default_rng([int(seed), zlib.crc32(k.encode())])
        shape = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros_like(v)
            continue
        if k.endswith(".filter"):
            out[k] = v.clone()
            continue
        if len(shape) == 4:                       # conv weight: xavier-normal scale
            fan_in = shape[1] * shape[2] * shape[3]
            fan_out = shape[0] * shape[2] * shape[3]
            a = rng.standard_normal(shape) * np.sqrt(2.0 / (fan_in + fan_out))
        elif k.endswith("running_mean"):
            a = 0.1 * rng.standard_normal(shape)
        elif k.endswith("running_var"):
            a = 1.0 + 0.2 * rng.random(shape)
        elif k.endswith(".w"):                     # soft-max temperature
            a = 1.0 + 0.3 * rng.standard_normal(shape)
        elif k.endswith(".weight"):                # norm gamma
            a = 1.0 + 0.1 * rng.standard_normal(shape)
        else:                                      # conv bias / norm beta
            a = 0.05 * rng.standard_normal(shape)
        out[k] = torch.from_numpy(np.asarray(a, dtype=np.float32)).to(v.dtype).reshape(shape)
    return out
