"""ProteInfer TF-weights pickle -> torch state_dict (weight-layout contract of
protnote/utils/proteinfer.py:7-41): the pickle is an ordered {tf_variable_name: ndarray}; the global
step becomes every BatchNorm's num_batches_tracked (inserted right after its moving_variance); entries are
matched to the torch state_dict POSITIONALLY; arrays with ndim >= 2 get all axes reversed
(TF [k, Cin, Cout] -> torch [Cout, Cin, k], [in, out] -> [out, in])."""
import pickle

import numpy as np
import torch

GLOBAL_STEP = "inferrer/global_step:0"


def ordered_tf_variables(tf_weights: dict) -> list:
    step = np.array(tf_weights[GLOBAL_STEP])
    out = []
    for name, value in tf_weights.items():
        if name == GLOBAL_STEP:
            continue
        out.append((name, value))
        if "batch_normalization" in name and "moving_variance" in name:
            out.append(("/".join(name.split("/")[:-1] + ["num_batches_tracked:0"]), step))
    return out


def transfer_tf_weights_to_torch(torch_model: torch.nn.Module, tf_weights_path: str):
    with open(tf_weights_path, "rb") as f:
        tf_weights = pickle.load(f)
    state = torch_model.state_dict()
    new_state = dict(state)
    for (name, param), (tf_name, value) in zip(state.items(), ordered_tf_variables(tf_weights)):
        value = np.asarray(value)
        if value.ndim >= 2:
            value = np.transpose(value, tuple(reversed(range(value.ndim))))
        if tuple(value.shape) != tuple(param.shape):
            raise AssertionError(f"{name} and {tf_name} don't have the same shape")
        new_state[name] = torch.from_numpy(np.ascontiguousarray(value))
    torch_model.load_state_dict(new_state)
