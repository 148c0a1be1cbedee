"""Checkpoint interchange with the reference.

Same file format and function names as /root/reference/utils.py:302-314: ``torch.save({"state_dict", "seed",
"model_param"})``; ``load_model`` = ``torch.load(map_location='cpu')`` + ``load_state_dict`` (+ optional ``.eval()``).
A ``.pt`` written by the reference's train.py loads here with strict=True and vice versa, because the state_dict keys,
order and shapes are identical (tests/test_boundary_cpu.py).
"""
import torch


def save_model(model, path, seed=None, model_param=None):
    data = {"state_dict": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, "seed": seed, "model_param": model_param}
    torch.save(data, path)


def load_model(model, path, eval_mode=False):
    data = torch.load(path, map_location="cpu")
    model.load_state_dict(data["state_dict"])
    if eval_mode:
        model.eval()
    return data.get("seed"), data.get("model_param")
