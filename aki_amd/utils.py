"""Small helpers mirrored from src/utils.py (getattr_recursive :13-24, setattr_recursive :27-35, num_params :54-59,
stack_with_padding :62-96, stack_with_padding_2D_attention :99-108).  The padding/stacking itself happens inside
the splice kernel on the product path; these versions exist for callers that use them directly."""
import torch


def getattr_recursive(obj, att):
    if att == "":
        return obj
    i = att.find(".")
    if i < 0:
        return getattr(obj, att)
    return getattr_recursive(getattr(obj, att[:i]), att[i + 1:])


def setattr_recursive(obj, att, val):
    if "." in att:
        obj = getattr_recursive(obj, ".".join(att.split(".")[:-1]))
    setattr(obj, att.split(".")[-1], val)


def num_params(module, filter_to_trainable=False):
    if filter_to_trainable:
        return sum(p.numel() for p in module.parameters() if p.requires_grad)
    return sum(p.numel() for p in module.parameters())


def stack_with_padding(list_of_tensors, padding_value=0, padding_side="right"):
    max_tokens = max(t.size(0) for t in list_of_tensors)
    out = []
    for t in list_of_tensors:
        pad = torch.full((max_tokens - t.size(0), *t.shape[1:]), padding_value, dtype=t.dtype, device=t.device)
        out.append(torch.cat((t, pad), dim=0) if padding_side == "right" else torch.cat((pad, t), dim=0))
    return torch.stack(out)


def stack_with_padding_2D_attention(list_of_tensors):
    max_size = max(t.size(1) for t in list_of_tensors)
    out = []
    for t in list_of_tensors:
        a = t.shape[-1]
        out.append(torch.nn.functional.pad(t, (0, max_size - a, 0, max_size - a)))
    return torch.stack(out)
