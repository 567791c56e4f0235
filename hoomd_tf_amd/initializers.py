"""Keras-style initialisers for the pair-MLP weights (tf.keras.layers.Dense defaults:
glorot-uniform kernel, zero bias -- the defaults example 08 / NlistNN use,
build_examples.py:199-205)."""
import math

import numpy as np


def glorot_uniform(rng, fan_in, fan_out, dtype=np.float32):
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=(fan_in, fan_out)).astype(dtype)


def mlp_params(seed=3, K=32, H1=64, H2=64, dtype=np.float32, bias_scale=0.0):
    """RBF(K) -> H1 -> H2 -> 1 weights, numpy default_rng(seed)."""
    rng = np.random.default_rng(seed)
    p = {
        "W1": glorot_uniform(rng, K, H1, dtype), "b1": np.zeros(H1, dtype),
        "W2": glorot_uniform(rng, H1, H2, dtype), "b2": np.zeros(H2, dtype),
        "W3": glorot_uniform(rng, H2, 1, dtype), "b3": np.zeros(1, dtype),
    }
    if bias_scale:
        for k in ("b1", "b2", "b3"):
            p[k] = (bias_scale * rng.standard_normal(p[k].shape)).astype(dtype)
    return p
