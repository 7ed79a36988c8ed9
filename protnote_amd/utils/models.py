"""On-disk formats around the hot path, so real ProtNote artefacts drop in:

  * checkpoint dict {epoch, model_state_dict, optimizer_state_dict, best_val_metric}
    (protnote/utils/models.py:304-321 save_checkpoint, :324-374 load_model incl. the DDP 'module.' prefix rule)
  * cached label embeddings: `<name>.pt` = f32 tensor [N_desc, d] and `<name>_index.pt` = pandas DataFrame with
    columns id, description_type, description, token_count (bin/generate_label_embeddings.py:123-164,
    protnote/data/datasets.py:115-127) -> rows filtered/sorted by protnote_amd.data.labels.
"""
from collections import OrderedDict

import numpy as np
import torch

from ..data import labels as LB


def save_checkpoint(model, optimizer, epoch, best_val_metric, model_path):
    torch.save({"epoch": epoch, "model_state_dict": model.state_dict(),
                "optimizer_state_dict": optimizer.state_dict() if optimizer is not None else {},
                "best_val_metric": best_val_metric}, model_path)


def strip_ddp_prefix(state_dict):
    """Keys saved from a DDP-wrapped model start with 'module.' (reference load_model :352-358)."""
    keys = list(state_dict.keys())
    if keys and keys[0].startswith("module."):
        return OrderedDict((k[7:], v) for k, v in state_dict.items())
    return state_dict


# Full unpickling of a checkpoint executes whatever the file contains.  It is what the reference does (torch.load without
# weights_only, utils/models.py:347), but here it is OPT-IN: set this flag (or PN_TRUST_CHECKPOINT=1) for files you trust.
TRUST_PICKLED_CHECKPOINTS = False


def _numpy_scalar_globals():
    """The few numpy types a metric scalar drags into an otherwise tensor-only checkpoint (`best_val_metric` as np.float64)."""
    out = [np.dtype, np.float64, np.float32, np.int64]
    for mod in ("numpy._core.multiarray", "numpy.core.multiarray"):
        try:
            out.append(__import__(mod, fromlist=["scalar"]).scalar)
        except (ImportError, AttributeError):
            pass
    out += [type(np.dtype(np.float64)), type(np.dtype(np.float32)), type(np.dtype(np.int64))]
    return out


def _load_checkpoint_file(path: str, map_location):
    """A checkpoint written by the reference's save_checkpoint (utils/models.py:304-321) holds tensors, ints and floats only, so
    it loads under torch's restricted unpickler (weights_only=True: no code execution; numpy scalar types are allow-listed for
    a `best_val_metric` that arrives as np.float64).  A file that needs more than that is REFUSED unless the caller opted into
    full unpickling (TRUST_PICKLED_CHECKPOINTS / PN_TRUST_CHECKPOINT=1) - a pickle that fails the restricted unpickler is
    exactly what a malicious one looks like."""
    import os
    import pickle

    try:
        with torch.serialization.safe_globals(_numpy_scalar_globals()):
            return torch.load(path, map_location=map_location, weights_only=True)
    except (pickle.UnpicklingError, RuntimeError, AttributeError) as exc:
        if not (TRUST_PICKLED_CHECKPOINTS or os.environ.get("PN_TRUST_CHECKPOINT") == "1"):
            raise RuntimeError(f"{path}: not loadable with torch's restricted unpickler ({str(exc)[:160]}).  Full unpickling "
                               "executes code from the file; if you trust it, set protnote_amd.utils.models."
                               "TRUST_PICKLED_CHECKPOINTS = True (or PN_TRUST_CHECKPOINT=1) and load again") from exc
        import warnings

        warnings.warn(f"{path}: full unpickling (trusted by the caller) - this executes whatever the file contains", stacklevel=3)
        return torch.load(path, map_location=map_location, weights_only=False)


def load_checkpoint_into(model, checkpoint_path: str, map_location="cpu", strict: bool = True):
    """Load `model_state_dict` of a reference-format checkpoint into a protnote_amd (or reference) model.
    Returns the rest of the checkpoint (epoch, optimizer_state_dict, best_val_metric).
    Loaded with torch's restricted unpickler first (_load_checkpoint_file)."""
    ckpt = _load_checkpoint_file(checkpoint_path, map_location)
    model.load_state_dict(strip_ddp_prefix(ckpt["model_state_dict"]), strict=strict)
    return {k: v for k, v in ckpt.items() if k != "model_state_dict"}


def load_model(trainer, checkpoint_path: str, rank: int, from_checkpoint=False):
    """Twin of the reference's load_model (utils/models.py:324-374; called by bin/main.py:521-527): the checkpoint's
    model_state_dict ('module.' prefix of a DDP-wrapped writer stripped, :352-358) goes into the trainer's model; with
    `from_checkpoint` the optimiser state (by parameter id: FusedClipAdam / FusedClipSGD take torch's layout), the epoch
    (as `starting_epoch` and `epoch`) and `best_val_metric` are restored too.  Tensors saved on cuda:0 land on this
    rank's device (:347)."""
    map_location = {"cuda:0": f"cuda:{rank}"} if torch.cuda.is_available() else "cpu"
    ckpt = _load_checkpoint_file(checkpoint_path, map_location)
    model = trainer._get_model() if hasattr(trainer, "_get_model") else getattr(trainer.model, "module", trainer.model)
    model.load_state_dict(strip_ddp_prefix(ckpt["model_state_dict"]))
    opt = getattr(trainer, "optimizer", None)
    if opt is not None and hasattr(opt, "repack"):
        opt.repack()  # load_state_dict copied INTO the flat views; make sure nothing was re-assigned
    if "optimizer_state_dict" in ckpt and from_checkpoint and opt is not None:  # an evaluation-only trainer has none
        opt.load_state_dict(ckpt["optimizer_state_dict"])
    if "epoch" in ckpt and from_checkpoint:
        trainer.starting_epoch = ckpt["epoch"]
        trainer.epoch = trainer.starting_epoch
    if "best_val_metric" in ckpt and from_checkpoint:
        trainer.best_val_metric = ckpt["best_val_metric"]


def index_path_for(embedding_path: str) -> str:
    """`a/b/emb.pt` -> `a/b/emb_index.pt` - same split-on-first-dot rule as the reference (datasets.py:115-118;
    any other '.' in the path breaks it there too), but done on the file name only."""
    head, dot, ext = embedding_path.rpartition(".")
    return f"{head}_index.{ext}" if dot else embedding_path + "_index"


def load_label_embedding_cache(embedding_path: str, label_vocabulary, descriptions=("name", "label")):
    """-> (embeddings [M, d] f32 in vocabulary order with each label's descriptions on consecutive rows,
           token_counts [M] i64, descriptions_per_label) for inference-time ensembling (ProtNote.py:313-322)."""
    emb = torch.load(embedding_path, map_location="cpu", weights_only=False)
    index = torch.load(index_path_for(embedding_path), map_location="cpu", weights_only=False)
    ids = index["id"].astype(str).tolist()
    types = index["description_type"].astype(str).tolist()
    kept, span = LB.embedding_row_index(ids, types, list(label_vocabulary), list(descriptions))
    missing = [l for l in label_vocabulary if l not in span]
    if missing:
        raise KeyError(f"{len(missing)} labels have no cached embedding, e.g. {missing[:3]}")
    rows = LB.sorted_embedding_rows(span, list(label_vocabulary))
    counts = np.asarray(index["token_count"].values)[kept][rows]
    per_label = {span[l][1] - span[l][0] + 1 for l in label_vocabulary}
    return (emb[torch.from_numpy(kept[rows])].float().contiguous(), torch.from_numpy(counts.astype(np.int64)),
            per_label.pop() if len(per_label) == 1 else None)
