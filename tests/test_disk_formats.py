"""Reference on-disk formats: checkpoint dict (incl. DDP 'module.' prefix), cached label-embedding pair."""
import os

import numpy as np
import pandas as pd
import torch

from protnote_amd.models.ProtNote import ProtNote
from protnote_amd.models.protein_encoders import ProteInfer
from protnote_amd.utils import models as M


def _small():
    enc = ProteInfer(5, 20, 8, 9, torch.nn.ReLU, 3, 1, 0.5)
    return ProtNote(protein_embedding_dim=8, label_embedding_dim=8, latent_dim=4, sequence_encoder=enc,
                    output_mlp_hidden_dim_scale_factor=2, output_mlp_num_layers=2, projection_head_num_layers=2,
                    projection_head_hidden_dim_scale_factor=2)


def test_checkpoint_roundtrip_and_ddp_prefix(tmp_path):
    a, b = _small(), _small()
    path = os.path.join(tmp_path, "ckpt.pt")
    M.save_checkpoint(a, None, epoch=3, best_val_metric=0.5, model_path=path)
    rest = M.load_checkpoint_into(b, path)
    assert rest["epoch"] == 3 and rest["best_val_metric"] == 0.5
    for (k, v), (k2, v2) in zip(a.state_dict().items(), b.state_dict().items()):
        assert k == k2 and torch.equal(v, v2)
    # a checkpoint written from a DDP-wrapped reference model
    ddp = {"epoch": 1, "best_val_metric": 0.1, "optimizer_state_dict": {},
           "model_state_dict": {"module." + k: v for k, v in a.state_dict().items()}}
    torch.save(ddp, path)
    c = _small()
    M.load_checkpoint_into(c, path)
    assert all(torch.equal(v, c.state_dict()[k]) for k, v in a.state_dict().items())


def test_label_embedding_cache(tmp_path):
    rows = []
    for gid in ["GO:3", "GO:1", "GO:9", "GO:2"]:
        for dt in ["name", "label", "synonym_exact"]:
            rows.append({"id": gid, "description_type": dt, "description": f"{gid}/{dt}", "token_count": len(rows) + 1})
    index = pd.DataFrame(rows)
    emb = torch.arange(len(rows), dtype=torch.float32)[:, None].repeat(1, 4)
    ep = os.path.join(tmp_path, "emb_BioGPT.pt")
    torch.save(emb, ep)
    torch.save(index, M.index_path_for(ep))
    assert M.index_path_for(ep).endswith("emb_BioGPT_index.pt")
    e, counts, per = M.load_label_embedding_cache(ep, ["GO:1", "GO:2", "GO:3"], ("name", "label"))
    # vocabulary order, name+label of each label on consecutive rows (rows 3,4 | 9,10 | 0,1 of the cache)
    assert e[:, 0].tolist() == [3, 4, 9, 10, 0, 1] and counts.tolist() == [4, 5, 10, 11, 1, 2] and per == 2


def test_tf_weight_transfer_matches_reference(golden_dir):
    """ProteInfer.from_pretrained on a synthetic TF-variable pickle vs the state dict the reference's
    transfer_tf_weights_to_torch produced from the same pickle (tests/golden/make_golden.py)."""
    g = np.load(os.path.join(golden_dir, "tf_weights_small_expected.npz"))
    cfg = {k[4:]: g[k].item() for k in g.files if k.startswith("cfg_")}
    model = ProteInfer.from_pretrained(os.path.join(golden_dir, "tf_weights_small.pkl"), activation=torch.nn.ReLU,
                                       **cfg)
    sd = model.state_dict()
    keys = [k[3:] for k in g.files if k.startswith("sd/")]
    assert sorted(keys) == sorted(sd.keys())
    for k in keys:
        assert np.array_equal(sd[k].numpy(), g["sd/" + k]), k
    assert int(sd["resnet_blocks.1.bn_activation_2.0.num_batches_tracked"]) == 12345
