"""The drop-in boundary is the reference's Python API: every twin must take the arguments the reference's callers
pass.  This compares `inspect.signature` of each twin with the REFERENCE's own object - parameter names, order, kinds and
defaults - mechanically, so a drift cannot go unnoticed (VERDICT r05 found four by hand).  Build-container only: the
reference is imported in a child process (its absent third-party imports stubbed exactly as tests/golden/make_golden.py
stubs them), never in this one."""
import importlib
import inspect
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PROTNOTE_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "protnote")),
                                reason="the reference checkout only exists in the build container")

# reference object ("module:attribute path"); the twin lives at the same path under protnote_amd
NAMES = [
    "protnote.models.ProtNote:ProtNote.__init__", "protnote.models.ProtNote:ProtNote.forward",
    "protnote.models.ProtNote:ProtNote.additive_attention", "protnote.models.ProtNote:ProtNote._get_joint_embeddings",
    "protnote.models.ProtNote:ProtNote._get_concatenated_features_dim", "protnote.models.ProtNote:get_mlp",
    "protnote.models.protein_encoders:ProteInfer.__init__", "protnote.models.protein_encoders:ProteInfer.get_embeddings",
    "protnote.models.protein_encoders:ProteInfer.forward", "protnote.models.protein_encoders:ProteInfer.from_pretrained",
    "protnote.models.protein_encoders:MaskedConv1D.__init__", "protnote.models.protein_encoders:MaskedConv1D.forward",
    "protnote.models.protein_encoders:Residual.__init__", "protnote.models.protein_encoders:Residual.forward",
    "protnote.utils.losses:get_loss", "protnote.utils.losses:FocalLoss.__init__", "protnote.utils.losses:FocalLoss.forward",
    "protnote.utils.losses:RGDBCE.__init__", "protnote.utils.losses:RGDBCE.forward", "protnote.utils.losses:CBLoss.__init__",
    "protnote.utils.losses:CBLoss.forward", "protnote.utils.losses:WeightedBCE.__init__",
    "protnote.utils.losses:WeightedBCE.forward", "protnote.utils.losses:BatchWeightedBCE.__init__",
    "protnote.utils.losses:BatchWeightedBCE.forward", "protnote.utils.losses:SupCon.__init__",
    "protnote.utils.losses:SupCon.forward",
    "protnote.data.samplers:GeneralDistributedSampler.__init__", "protnote.data.samplers:DistributedWeightedSampler.__init__",
    "protnote.data.samplers:GridBatchSampler.__init__", "protnote.data.collators:collate_variable_sequence_length",
    "protnote.utils.models:save_checkpoint", "protnote.utils.models:load_model",
    "protnote.utils.configs:load_config", "protnote.utils.configs:override_config",
    "protnote.utils.configs:generate_label_embedding_path", "protnote.utils.configs:try_literal_eval",
    "protnote.utils.configs:get_project_root", "protnote.utils.configs:update_config_paths",
    "protnote.utils.proteinfer:transfer_tf_weights_to_torch",
    "protnote.models.ProtNoteTrainer:calculate_tp_fn_fp", "protnote.models.ProtNoteTrainer:calculate_f1",
    "protnote.models.ProtNoteTrainer:calculate_f1_micro",
]

_DUMP = r'''
import sys, json, inspect, importlib
sys.path.insert(0, sys.argv[2]); sys.path.insert(0, sys.argv[3])
import make_golden as MG
MG.install_stubs()
def sig(obj):
    return [[p.name, p.kind.name, None if p.default is inspect._empty else repr(p.default)]
            for p in inspect.signature(obj).parameters.values()]
res = {}
for n in json.loads(sys.argv[1]):
    mod, _, attr = n.partition(":")
    o = importlib.import_module(mod)
    for a in attr.split("."):
        o = getattr(o, a)
    res[n] = sig(o)
print("SIGNATURES " + json.dumps(res))
'''


def _sig(obj):
    return [[p.name, p.kind.name, None if p.default is inspect._empty else repr(p.default)]
            for p in inspect.signature(obj).parameters.values()]


@pytest.fixture(scope="module")
def reference_signatures():
    out = subprocess.run([sys.executable, "-c", _DUMP, json.dumps(NAMES), os.path.join(ROOT, "tests", "golden"), ROOT],
                         capture_output=True, text=True, timeout=600, env={**os.environ, "PROTNOTE_REFERENCE": REF})
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("SIGNATURES ")]
    assert lines, out.stderr[-2000:]
    return json.loads(lines[-1][len("SIGNATURES "):])


@pytest.mark.parametrize("name", NAMES)
def test_twin_signature_equals_reference(reference_signatures, name):
    mod, _, attr = name.partition(":")
    obj = importlib.import_module(mod.replace("protnote", "protnote_amd", 1))
    for a in attr.split("."):
        obj = getattr(obj, a)
    assert _sig(obj) == reference_signatures[name], name


def test_sampler_defaults_resolve_through_torch_distributed():
    """`None` world size / rank resolve exactly as in the reference (samplers.py:15-37 via torch's DistributedSampler,
    :66-74): from the process group - and fail the way torch.distributed fails when there is none."""
    import torch.distributed as dist

    from protnote_amd.data.samplers import DistributedWeightedSampler, GeneralDistributedSampler

    assert not dist.is_initialized()
    with pytest.raises((RuntimeError, ValueError)):
        DistributedWeightedSampler([0.1] * 10)
    with pytest.raises((RuntimeError, ValueError)):
        GeneralDistributedSampler(list(range(10)))
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29577", rank=0, world_size=1)
    try:
        s = DistributedWeightedSampler([0.1] * 10)
        assert (s.world_size, s.rank, len(s)) == (1, 0, 10)
        g = GeneralDistributedSampler(list(range(10)))
        assert (g.num_replicas, g.rank) == (1, 0) and list(g) == list(range(10))
    finally:
        dist.destroy_process_group()
    with pytest.raises(ValueError, match="Invalid rank"):
        GeneralDistributedSampler(list(range(10)), num_replicas=2, rank=2)
