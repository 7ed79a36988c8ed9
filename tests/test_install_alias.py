"""protnote_amd.install_as_protnote(): the `sys.modules` aliasing of INTEGRATION.md section 1 as one call, so that the
reference's own import lines (bin/main.py:9-11) yield the twins.  Run in child processes: the aliases must not leak into
the other tests' interpreter."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code, extra_path=None):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([ROOT] + ([extra_path] if extra_path else []))
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd="/tmp")


def test_reference_import_lines_yield_the_twins():
    r = _run("""
import protnote_amd
names = protnote_amd.install_as_protnote()
from protnote.models.ProtNote import ProtNote
from protnote.models.protein_encoders import ProteInfer
from protnote.utils.losses import get_loss
from protnote.utils.models import load_model, save_checkpoint
from protnote.data.collators import collate_variable_sequence_length
from protnote.data.samplers import GridBatchSampler
import protnote.utils.configs as cfg
import protnote_amd.models.ProtNote as T1, protnote_amd.models.protein_encoders as T2, protnote_amd.utils.losses as T3
assert ProtNote is T1.ProtNote and ProteInfer is T2.ProteInfer and get_loss is T3.get_loss
assert cfg.load_config.__module__ == "protnote_amd.utils.configs"
assert "protnote.models.ProtNote" in names and len(names) >= 12
m = ProtNote(protein_embedding_dim=16, label_embedding_dim=8, latent_dim=8, output_mlp_hidden_dim_scale_factor=2,
             output_mlp_num_layers=2, projection_head_num_layers=2)
assert "output_layer.0.weight" in m.state_dict()
print("ALIAS-OK")
""")
    assert "ALIAS-OK" in r.stdout, r.stderr[-2000:]


def test_a_real_protnote_package_is_not_shadowed_silently(tmp_path):
    pkg = tmp_path / "protnote"
    pkg.mkdir()
    (pkg / "__init__.py").write_text("REAL = True\n")
    r = _run("""
import protnote_amd
try:
    protnote_amd.install_as_protnote()
except RuntimeError as e:
    assert "importable" in str(e)
    print("REFUSED")
protnote_amd.install_as_protnote(force=True)
import protnote
assert protnote is protnote_amd
print("FORCED")
""", extra_path=str(tmp_path))
    assert "REFUSED" in r.stdout and "FORCED" in r.stdout, r.stderr[-2000:]
