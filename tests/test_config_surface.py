"""The base_config.yaml surface the hot path depends on: key names -> constructor kwargs, --override rule."""
import pytest
import yaml

from protnote_amd.utils import configs as CF

CFG = yaml.safe_load("""
params:
  LEARNING_RATE: 0.0003
  PROTEIN_EMBEDDING_DIM: 1100
  LABEL_EMBEDDING_DIM: 1024
  LATENT_EMBEDDING_DIM: 1024
  OUTPUT_MLP_HIDDEN_DIM_SCALE_FACTOR: 3
  OUTPUT_MLP_NUM_LAYERS: 3
  OUTPUT_MLP_BATCHNORM: True
  RESIDUAL_CONNECTION: False
  OUTPUT_MLP_DROPOUT: 0.0
  LABEL_EMBEDDING_DROPOUT: 0.0
  SEQUENCE_EMBEDDING_DROPOUT: 0.0
  PROJECTION_HEAD_NUM_LAYERS: 4
  PROJECTION_HEAD_HIDDEN_DIM_SCALE_FACTOR: 3
  FEATURE_FUSION: concatenation
  LABEL_EMBEDDING_POOLING_METHOD: mean
  LABEL_EMBEDDING_NOISING_ALPHA: 20.0
  LABEL_ENCODER_NUM_TRAINABLE_LAYERS: 0
  TRAIN_SEQUENCE_ENCODER: False
  INFERENCE_GO_DESCRIPTIONS: name+label
  SUPCON_TEMP: 0.07
  CLIP_VALUE: 1
  LOSS_FN: FocalLoss
embed_sequences_params:
  INPUT_CHANNELS: 20
  OUTPUT_CHANNELS: 1100
  KERNEL_SIZE: 9
  DILATION_BASE: 3
  NUM_RESNET_BLOCKS: 5
  BOTTLENECK_FACTOR: 0.5
  PROTEINFER_NUM_GO_LABELS: 32102
paths:
  data_paths:
    TRAIN_DATA_PATH: a.fasta
""")


def test_override_rule():
    import copy

    c = CF.override_config(copy.deepcopy(CFG), ["CLIP_VALUE", "null", "LOSS_FN", "BCE", "OUTPUT_MLP_BATCHNORM",
                                                "false", "LEARNING_RATE", "1e-3", "TRAIN_DATA_PATH", "b.fasta"])
    assert c["params"]["CLIP_VALUE"] is None and c["params"]["LOSS_FN"] == "BCE"
    assert c["params"]["OUTPUT_MLP_BATCHNORM"] is False and c["params"]["LEARNING_RATE"] == 1e-3
    assert c["paths"]["data_paths"]["TRAIN_DATA_PATH"] == "b.fasta"
    with pytest.raises(KeyError):
        CF.override_config(copy.deepcopy(CFG), ["NOT_A_KEY", "1"])
    with pytest.raises(ValueError):
        CF.override_config(copy.deepcopy(CFG), ["CLIP_VALUE"])


def test_build_models_matches_reference_parameter_counts():
    import copy

    cfg = copy.deepcopy(CFG)
    cfg["embed_sequences_params"]["PROTEINFER_NUM_GO_LABELS"] = 8  # keep the unused classifier small
    enc, model = CF.build_models(cfg)
    named = dict(model.named_parameters())
    # SURVEY 2.2 K16 [probed on the reference]: 25 417 728 W_p + 25 184 256 W_l + 25 187 329 output_layer
    count = lambda pre: sum(v.numel() for k, v in named.items() if k.startswith(pre))
    assert count("W_p.") == 25_417_728 and count("W_l.") == 25_184_256 and count("output_layer.") == 25_187_329
    trunk = sum(v.numel() for k, v in named.items() if k.startswith("sequence_encoder.") and "output_layer" not in k)
    assert trunk == 30_473_850
    assert model.inference_descriptions_per_label == 2 and model.temperature == 0.07
    keys = set(model.state_dict())
    for k in ("W_p.0.weight", "W_p.1.running_mean", "W_p.12.weight", "W_l.9.num_batches_tracked",
              "output_layer.0.weight", "output_layer.9.bias", "output_layer.11.weight", "output_layer.11.bias",
              "sequence_encoder.conv1.weight", "sequence_encoder.resnet_blocks.4.bn_activation_2.0.running_var",
              "sequence_encoder.resnet_blocks.0.masked_conv1.bias", "sequence_encoder.output_layer.weight"):
        assert k in keys, k


def test_output_mlp_num_layers_one_builds_the_reference_key_layout(golden_dir):
    """`OUTPUT_MLP_NUM_LAYERS: 1` (configs/base_config.yaml:34; get_mlp, ProtNote.py:337-378): Linear(2d, h, no bias), BatchNorm,
    ReLU, Linear(h, 1) - the key set of a state dict the REFERENCE built with that setting (config_holes_1layer_*.npz)."""
    import copy

    import numpy as np
    import os

    cfg = copy.deepcopy(CFG)
    cfg["params"]["OUTPUT_MLP_NUM_LAYERS"] = 1
    cfg["embed_sequences_params"]["PROTEINFER_NUM_GO_LABELS"] = 8
    _, model = CF.build_models(cfg)
    keys = {k for k in model.state_dict() if k.startswith("output_layer.")}
    g = np.load(os.path.join(golden_dir, "config_holes_1layer_concatenation.npz"))
    ref = {k[3:] for k in g.files if k.startswith("sd/output_layer.")}
    assert keys == ref == {"output_layer.0.weight", "output_layer.1.weight", "output_layer.1.bias", "output_layer.1.running_mean",
                           "output_layer.1.running_var", "output_layer.1.num_batches_tracked", "output_layer.3.weight",
                           "output_layer.3.bias"}
    assert tuple(model.output_layer[0].weight.shape) == (3072, 2048) and tuple(model.output_layer[3].weight.shape) == (1, 3072)
    hd, _ = model._pair_desc()
    assert hd.nlayers == 1 and hd.h == 3072


def test_build_training_rejects_unknown_optimizer():
    import copy

    cfg = copy.deepcopy(CFG)
    # Adam / AdamW / SGD are the reference's choices; anything else is its ValueError (ProtNoteTrainer.py:244-245)
    cfg["params"].update(OPTIMIZER="RMSprop", FOCAL_LOSS_GAMMA=2, FOCAL_LOSS_ALPHA=-1, LABEL_SMOOTHING=0.0)
    with pytest.raises(ValueError, match="Unsupported optimizer name"):
        CF.build_training(cfg, model=None)


REAL_CONFIG = "/root/reference/configs/base_config.yaml"


@pytest.mark.skipif(not __import__("os").path.exists(REAL_CONFIG),
                    reason="the reference checkout only exists in the build container")
def test_real_base_config_builds_and_embedded_copy_agrees():
    """The reference's own configs/base_config.yaml (read in place - never copied into the repo): every key the
    embedded test config above uses exists there with the same value, `--override` works on it, and
    build_models / build_training accept it unmodified."""
    import copy

    real, root = CF.load_config(REAL_CONFIG)  # absolute path: taken as is, like the reference's pathlib join
    assert str(real["paths"]["data_paths"]["TRAIN_DATA_PATH"]).startswith(str(root / "data"))
    for section in ("params", "embed_sequences_params"):
        for k, v in CFG[section].items():
            assert k in real[section], (section, k)
            assert real[section][k] == v, (section, k, real[section][k], v)
    cfg = CF.override_config(copy.deepcopy(real), ["PROTEINFER_NUM_GO_LABELS", "8", "LOSS_FN", "BCE"])
    enc, model = CF.build_models(cfg)
    assert sum(p.numel() for p in model.W_p.parameters()) == 25_417_728
    assert model.feature_fusion == real["params"]["FEATURE_FUSION"]
    assert model.label_embedding_noising_alpha == real["params"]["LABEL_EMBEDDING_NOISING_ALPHA"]
