"""F2: checkpoint I/O in the reference's format and the inference/export loop.

Format (the commented torch.save at main_frame_val_text_missing.py:375,384 and the load at
main_frame_val_text_missing_inference.py:341): a dict {'epoch', 'state_dict', 'optimizer'} where
`state_dict` is that of the get_models wrapper (keys 'model.<name>', optionally with a leading 'module.'
from DataParallel) and `optimizer` is torch.optim.Adam's state_dict over model.parameters().
The published 49 MB file is exactly this: 4 268 884 fp32 parameters + two Adam moments.
"""
import numpy as np
import torch

from ._lib import SdumcError


def _net(model):
    return model.model if hasattr(model, "model") and not hasattr(model, "_flat") else model


def adam_state_from_flat(net, adam_m, adam_v, step, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5):
    """torch.optim.Adam.state_dict() equivalent for the fused step's flat moment buffers: parameter ids follow
    named_parameters() order; parameters that never receive a gradient have no state entry (like torch)."""
    lay = net._layout
    state, ids = {}, []
    for i, name in enumerate(net._pnames):
        ids.append(i)
        off, shape, live = lay.entries[name]
        if not live:
            continue
        n = int(np.prod(shape))
        state[i] = {"step": torch.tensor(float(step)),
                    "exp_avg": adam_m[off:off + n].view(shape).detach().cpu().clone(),
                    "exp_avg_sq": adam_v[off:off + n].view(shape).detach().cpu().clone()}
    group = {"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": weight_decay, "amsgrad": False,
             "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
             "decoupled_weight_decay": False, "params": ids}
    return {"state": state, "param_groups": [group]}


def flat_from_adam_state(net, opt_state, device):
    """Inverse of adam_state_from_flat: (adam_m, adam_v, step) flat buffers [live]; hand them to
    TrainStep / FusedTrainer / DataParallelStep .load_optimizer_state(m, v, step) to continue the run."""
    lay = net._layout
    m = torch.zeros(lay.live, device=device)
    v = torch.zeros(lay.live, device=device)
    step = 0.0
    for i, name in enumerate(net._pnames):
        st = opt_state["state"].get(i)
        off, shape, live = lay.entries[name]
        if st is None or not live:
            continue
        n = int(np.prod(shape))
        m[off:off + n] = st["exp_avg"].reshape(-1).to(device)
        v[off:off + n] = st["exp_avg_sq"].reshape(-1).to(device)
        step = max(step, float(st["step"]))
    return m, v, step


def save_checkpoint(path, model, optimizer_state, epoch):
    """`model`: the get_models wrapper (keys get the 'model.' prefix) or the bare network."""
    sd = model.state_dict()
    if not hasattr(model, "model") or hasattr(model, "_flat"):
        sd = {"model." + k: v for k, v in sd.items()}
    torch.save({"epoch": epoch, "state_dict": {k: v.detach().cpu() for k, v in sd.items()}, "optimizer": optimizer_state},
               path)


def load_checkpoint(path, model, strict=False, legacy_pickle=False):
    """Loads like the reference's inference script: strips a leading 'module.', strict=False by default.
    The format holds only tensors and plain containers, so it is read with torch's restricted unpickler
    (weights_only=True); `legacy_pickle=True` opts into full pickle for files that carry other Python objects --
    only for files you trust (the reference's own torch.load, ..._inference.py:341, executes arbitrary pickles)."""
    ck = torch.load(path, map_location="cpu", weights_only=not legacy_pickle)
    if "state_dict" not in ck:
        raise SdumcError(f"{path}: not a reference-format checkpoint (keys {list(ck)})")
    sd = {k.replace("module.", ""): v for k, v in ck["state_dict"].items()}
    if hasattr(model, "_flat"):                                   # bare network: drop the wrapper prefix
        sd = {(k[len("model."):] if k.startswith("model.") else k): v for k, v in sd.items()}
    missing, unexpected = model.load_state_dict(sd, strict=strict)
    return ck.get("epoch"), ck.get("optimizer"), missing, unexpected


@torch.no_grad()
def run_inference(model, batches):
    """The eval/export loop of main_frame_val_text_missing_inference.py:100-215: both streams under no_grad,
    predictions and the four embeddings of each stream concatenated on the host.  `batches` yields the reference's
    batch tuples (CPU or device tensors)."""
    model.eval()
    dev = next(model.parameters()).device
    cols = {k: [] for k in ("val_preds_full", "val_preds_missing", "val_labels", "full_rep", "missing_rep", "full_rnc",
                            "missing_rnc", "text_rep_query_full", "text_rep_query_missing", "text_rep_full",
                            "text_rep_missing")}
    names = []
    for data in batches:
        b = data[0]
        audio, text, video, feat4 = (b[k].to(dev) for k in ("audios", "texts", "videos", "feat4s"))
        y0, e0 = model([audio, text, video, False])
        y1, e1 = model([audio, feat4, video, True])
        for key, t in (("val_preds_full", y0), ("val_preds_missing", y1), ("val_labels", data[-2].float()),
                       ("full_rep", e0[0]), ("missing_rep", e1[0]), ("full_rnc", e0[1]), ("missing_rnc", e1[1]),
                       ("text_rep_query_full", e0[2]), ("text_rep_query_missing", e1[2]),
                       ("text_rep_full", e0[3]), ("text_rep_missing", e1[3])):
            cols[key].append(t.detach().cpu().numpy())
        names += list(data[-1])
    out = {k: np.concatenate(v, axis=0) for k, v in cols.items()}
    out["names"] = names
    out["val_mse"] = float(np.mean((out["val_labels"].reshape(-1) - out["val_preds_full"].reshape(-1)) ** 2))
    out["val_mse_missing"] = float(np.mean((out["val_labels"].reshape(-1) - out["val_preds_missing"].reshape(-1)) ** 2))
    return out
