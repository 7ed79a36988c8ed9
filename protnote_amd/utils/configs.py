"""Config surface of the reference (protnote/utils/configs.py + configs/base_config.yaml): YAML load, the
`--override KEY VALUE ...` rule (python-literal values, null/true/false spelled the YAML way), the mapping
from config keys to the model constructors that bin/main.py:383-446 performs, and the naming rule of the cached
label-embedding files.  No path prefixing or loggers (I/O side, out of scope)."""
import os
from ast import literal_eval
from pathlib import Path

import yaml


def get_project_root():
    """Reference configs.py:268-270: the directory that holds `configs/`, `data/` and `outputs/` - there the checkout the
    package lives in.  Here: $PROTNOTE_PROJECT_ROOT when set (point it at the ProtNote checkout whose configs / data this
    drop-in should serve), else the directory above the package."""
    env = os.environ.get("PROTNOTE_PROJECT_ROOT")
    return Path(env).resolve() if env else Path(__file__).resolve().parent.parent.parent


def update_config_paths(config, project_root):
    """Reference configs.py:272-280: data paths -> <root>/data/<value>, output paths -> <root>/outputs/<value>."""
    for key, value in config["paths"].get("data_paths", {}).items():
        config["paths"]["data_paths"][key] = project_root / "data" / value
    for key, value in config["paths"].get("output_paths", {}).items():
        config["paths"]["output_paths"][key] = project_root / "outputs" / value
    return config


def load_config(config_file: str = "base_config.yaml"):
    """Reference configs.py:282-290, as every script calls it (`config, project_root = load_config()`): reads
    <project_root>/configs/<config_file> (an absolute `config_file` wins, as pathlib joins it there too), prefixes the data /
    output paths and returns (config, project_root)."""
    project_root = get_project_root()
    with open(project_root / "configs" / config_file) as f:
        config = yaml.safe_load(f)
    return update_config_paths(config, project_root), project_root


def try_literal_eval(val):
    """Reference configs.py:38-48: 'null'/'true'/'false' are recognised, everything else goes through
    ast.literal_eval and falls back to the raw string."""
    if isinstance(val, str):
        low = val.lower()
        if low == "null":
            return None
        if low == "true":
            return True
        if low == "false":
            return False
    try:
        return literal_eval(val)
    except (ValueError, SyntaxError):
        return val


def override_config(config: dict, overrides) -> dict:
    """Reference configs.py:51-71: flat `KEY VALUE KEY VALUE ...`; a key must already exist in params or paths."""
    if overrides is None:
        return config
    if len(overrides) % 2 != 0:
        raise ValueError("Overrides must be provided as key-value pairs.")
    for key, value in zip(overrides[::2], overrides[1::2]):
        found = False
        for section in ("params", "embed_sequences_params"):
            if key in config.get(section, {}):
                config[section][key] = try_literal_eval(value)
                found = True
        for section in config.get("paths", {}).values():
            if isinstance(section, dict) and key in section:
                section[key] = value
                found = True
        if not found:
            raise KeyError(f"Key '{key}' not found in the 'params' or 'paths' section of the config.")
    return config


_LABEL_ENCODER_NICKNAMES = {"microsoft/biogpt": "BioGPT", "intfloat/e5-large-v2": "E5",
                            "intfloat/multilingual-e5-large-instruct": "E5_multiling_inst"}


def generate_label_embedding_path(params: dict, base_label_embedding_path: str) -> str:
    """Name of the file that caches the label embeddings of one (label encoder, pooling method) pair - reference
    utils/configs.py:74-107, used by get_setup (:249-251) and by the producer bin/generate_label_embeddings.py:85-92:
    `dir/<first>_<rest>.<ext>` -> `dir/<first>_<Nick>_<rest>_<pool>.<ext>` where the file name is cut at its FIRST dot
    and at underscores (`frozen_label_embeddings.pt` -> `frozen_BioGPT_label_embeddings_mean.pt`).  Unsupported
    LABEL_ENCODER_CHECKPOINT values fail the same assertion."""
    assert params["LABEL_ENCODER_CHECKPOINT"] in _LABEL_ENCODER_NICKNAMES, "Model not supported"
    parts = base_label_embedding_path.split("/")
    name = parts[-1].split(".")
    words = name[0].split("_")
    stem = "_".join([words[0], _LABEL_ENCODER_NICKNAMES[params["LABEL_ENCODER_CHECKPOINT"]]] + words[1:])
    parts[-1] = stem + "_" + params["LABEL_EMBEDDING_POOLING_METHOD"] + "." + name[1]
    return "/".join(parts)


def build_models(config: dict, num_labels: int = None, label_encoder=None, feature_fusion: str = None):
    """ProteInfer + ProtNote exactly as bin/main.py:396-446 builds them from `params` /
    `embed_sequences_params` (random-initialised encoder; weights then come from load_state_dict)."""
    import torch

    from ..models.ProtNote import ProtNote
    from ..models.protein_encoders import ProteInfer

    p, e = config["params"], config["embed_sequences_params"]
    enc = ProteInfer(num_labels=num_labels or e["PROTEINFER_NUM_GO_LABELS"], input_channels=e["INPUT_CHANNELS"],
                     output_channels=e["OUTPUT_CHANNELS"], kernel_size=e["KERNEL_SIZE"], activation=torch.nn.ReLU,
                     dilation_base=e["DILATION_BASE"], num_resnet_blocks=e["NUM_RESNET_BLOCKS"],
                     bottleneck_factor=e["BOTTLENECK_FACTOR"])
    model = ProtNote(
        protein_embedding_dim=p["PROTEIN_EMBEDDING_DIM"], label_embedding_dim=p["LABEL_EMBEDDING_DIM"],
        latent_dim=p["LATENT_EMBEDDING_DIM"], label_embedding_pooling_method=p["LABEL_EMBEDDING_POOLING_METHOD"],
        sequence_embedding_dropout=p["SEQUENCE_EMBEDDING_DROPOUT"], label_embedding_dropout=p["LABEL_EMBEDDING_DROPOUT"],
        label_embedding_noising_alpha=p["LABEL_EMBEDDING_NOISING_ALPHA"], label_encoder=label_encoder,
        sequence_encoder=enc, inference_descriptions_per_label=len(p["INFERENCE_GO_DESCRIPTIONS"].split("+")),
        output_mlp_hidden_dim_scale_factor=p["OUTPUT_MLP_HIDDEN_DIM_SCALE_FACTOR"],
        output_mlp_num_layers=p["OUTPUT_MLP_NUM_LAYERS"], output_neuron_bias=None,
        outout_mlp_add_batchnorm=p["OUTPUT_MLP_BATCHNORM"], residual_connection=p["RESIDUAL_CONNECTION"],
        projection_head_num_layers=p["PROJECTION_HEAD_NUM_LAYERS"], dropout=p["OUTPUT_MLP_DROPOUT"],
        projection_head_hidden_dim_scale_factor=p["PROJECTION_HEAD_HIDDEN_DIM_SCALE_FACTOR"],
        label_encoder_num_trainable_layers=p["LABEL_ENCODER_NUM_TRAINABLE_LAYERS"],
        train_sequence_encoder=p["TRAIN_SEQUENCE_ENCODER"], feature_fusion=feature_fusion or p["FEATURE_FUSION"],
        temperature=p["SUPCON_TEMP"])
    return enc, model


def build_training(config: dict, model, world_size: int = 1):
    """Loss, optimiser and epoch driver from the `params` section, as ProtNoteTrainer.__init__ / _set_optimizer read it
    (ProtNoteTrainer.py:89-245): LOSS_FN (+ FOCAL_LOSS_*, BCE_POS_WEIGHT, LABEL_SMOOTHING), OPTIMIZER (Adam | AdamW | SGD;
    WEIGHT_DECAY for AdamW and SGD), LEARNING_RATE, CLIP_VALUE (null = no clipping), GRADIENT_ACCUMULATION_STEPS,
    DECISION_TH, TRAIN_SEQUENCE_ENCODER, TRAIN_PROJECTION_HEAD (False freezes output_layer.* only, as in the reference).
    Returns (loss_fn, optimizer, trainer); `trainer.evaluate(loader, estimate_map=params['ESTIMATE_MAP'])`."""
    import torch

    from ..models.ProtNoteTrainer import Trainer
    from ..models.train_path import trainable_parameters
    from .losses import get_loss
    from .optim import FusedClipAdam, FusedClipSGD

    p = config["params"]
    if p.get("SYNC_BN", False) and world_size > 1:
        # bin/main.py:449-450 converts to SyncBatchNorm; here the statistics of a layer are reduced inside one C call,
        # so the switch is library-wide: the per-rank column sums are all-reduced through a callback
        from .distributed import enable_sync_batchnorm

        if not enable_sync_batchnorm():
            raise RuntimeError("SYNC_BN: True with world_size > 1 needs an initialised torch.distributed process group")
    loss_fn = get_loss(config, bce_pos_weight=torch.tensor(float(p.get("BCE_POS_WEIGHT", 1))))
    name = p.get("OPTIMIZER", "Adam")
    if name not in ("Adam", "AdamW", "SGD"):
        raise ValueError("Unsupported optimizer name")  # ProtNoteTrainer.py:244-245
    params = list(trainable_parameters(model))  # heads (+ raw_attn_scorer with LABEL_EMBEDDING_POOLING_METHOD: all)
    train_enc = bool(p.get("TRAIN_SEQUENCE_ENCODER", False))
    if train_enc:
        params += list(model.sequence_encoder.trunk_parameters())
    # requires_grad as _set_optimizer leaves it (ProtNoteTrainer.py:210-226) - that loop only ever clears the flag - then
    # the ids torch.optim.Adam would give the parameters: position in named_parameters() filtered by requires_grad
    # (sequence_encoder - classifier included - before W_p, W_l, raw_attn_scorer, output_layer), so optimizer_state_dict
    # of a reference checkpoint loads by id.
    # TRAIN_PROJECTION_HEAD: False freezes exactly what the reference freezes: every output_layer.* parameter (:221-222).
    # Its other test, name.startswith("W_p.weight") / ("W_l.weight") (:216-219), matches no parameter of the model (the
    # names are W_p.0.weight ...), so W_p and W_l KEEP training - reproduced as is.
    train_head = bool(p.get("TRAIN_PROJECTION_HEAD", True))
    for n_, q_ in model.named_parameters():
        if n_.startswith("sequence_encoder") and not train_enc:
            q_.requires_grad = False
        if (n_.startswith("W_p.weight") or n_.startswith("W_l.weight")) and not train_head:
            q_.requires_grad = False
        if n_.startswith("output_layer") and not train_head:
            q_.requires_grad = False
    params = [q_ for q_ in params if q_.requires_grad]  # the frozen stacks cost nothing: no gradient GEMMs, no state
    order = {id(q_): k for k, q_ in enumerate(q_ for _, q_ in model.named_parameters() if q_.requires_grad)}
    clip = p.get("CLIP_VALUE", 1)
    common = dict(lr=p.get("LEARNING_RATE", 3e-4), max_norm=None if clip is None else float(clip),
                  param_ids=[order[id(q_)] for q_ in params], n_param_ids=len(order))
    if name == "SGD":  # torch.optim.SGD(trainable, lr, weight_decay=WEIGHT_DECAY), :238-243
        opt = FusedClipSGD(params, weight_decay=p.get("WEIGHT_DECAY", 0.0), **common)
    else:              # Adam ignores WEIGHT_DECAY (:230-231), AdamW decouples it (:232-237)
        opt = FusedClipAdam(params, weight_decay=p.get("WEIGHT_DECAY", 0.0) if name == "AdamW" else 0.0, **common)
    th = p.get("DECISION_TH", 0.5)
    trainer = Trainer(model, loss_fn, opt, world_size=world_size, threshold=0.5 if th is None else th,
                      gradient_accumulation_steps=p.get("GRADIENT_ACCUMULATION_STEPS", 1))
    return loss_fn, opt, trainer
