"""Shared test helpers: golden loading, case -> config, synthetic weights."""
import json
import os

import numpy as np
import torch

from oracle import synth
from yolov6_amd.configs import get_config

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# golden case -> built-in config name (tests must not read /root/reference)
CASE_CONFIG = {"tiny": "yolov6s", "n": "yolov6n", "s": "yolov6s", "s_qa_tiny": "yolov6s_qa", "l6_tiny": "yolov6l6",
               "m_tiny": "yolov6m", "s_mbla_tiny": "yolov6s_mbla", "n6": "yolov6n6", "m6_tiny": "yolov6m6",
               "t_pan": "yolov6t", "s_csp_pan_tiny": "yolov6s_csp", "n6_pan": "yolov6n6",
               "n_base": "yolov6n_base", "s_base_tiny": "yolov6s_base", "s_qav1_tiny": "yolov6s_qa"}


def case_meta(case):
    with open(os.path.join(GOLDEN, f"keys_{case}.json")) as f:
        return json.load(f)


def case_config(case):
    meta = case_meta(case)
    cfg = get_config(CASE_CONFIG[case])
    for k, v in meta["overrides"].items():      # dotted keys reach nested entries ("neck.type")
        if k == "training_mode":
            cfg.training_mode = v
            continue
        node = cfg.model
        *path, leaf = k.split(".")
        for part in path:
            node = node[part]
        node[leaf] = v
    assert cfg.training_mode == meta["training_mode"]
    return cfg, meta


def case_golden(case):
    return np.load(os.path.join(GOLDEN, f"model_{case}.npz"))


def synth_sd_from_keys(keys, seed=0):
    tmpl = {k: torch.empty(shape, dtype=torch.int64 if k.endswith("num_batches_tracked") else torch.float32)
            for k, shape in keys.items()}
    return synth.synth_state_dict(tmpl, seed)


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))
