"""shared helpers for the test-suite (golden fixtures, model construction)"""
import json
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_gold(name):
    g = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    return g


def gold_opt(g, key="opt_json"):
    return json.loads(bytes(g[key]).decode())


def gold_json(g, key):
    return json.loads(bytes(g[key]).decode())


def t(a, device=None):
    x = torch.from_numpy(np.asarray(a))
    return x.to(device) if device is not None else x


def gold_state(g, prefix):
    return {k[len(prefix):]: t(g[k]) for k in g.files if k.startswith(prefix)}


def gold_batch(g, device=None):
    b = {}
    feats = []
    i = 0
    while f"in.feats{i}" in g.files:
        feats.append(t(g[f"in.feats{i}"], device))
        i += 1
    b["feats"] = feats
    for k in g.files:
        if k.startswith("in.") and not k.startswith("in.feats"):
            b[k[3:]] = t(g[k], device)
    return b


def maxerr(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())
