"""Host-side input layout and integer label bookkeeping vs vectors produced by the reference's own collator
and ProteinDataset (tests/golden/make_golden.py).  Integer work: bit-exact."""
import os

import numpy as np
import torch

from protnote_amd.data import labels as LB
from protnote_amd.data.collators import collate_variable_sequence_length


def test_collator_layout(golden_dir):
    g = np.load(os.path.join(golden_dir, "collator.npz"))
    lab = torch.from_numpy(g["in/label_embeddings"])
    cnt = torch.from_numpy(g["in/label_token_counts"])
    batch = []
    for i in range(4):
        ids = torch.from_numpy(g[f"in/ids{i}"])
        batch.append({"sequence_onehots": torch.nn.functional.one_hot(ids, 20).T.float(), "sequence_id": f"P{i}",
                      "sequence_length": torch.tensor(len(ids)), "label_multihots": torch.from_numpy(g[f"in/multihots{i}"]),
                      "label_embeddings": lab, "label_token_counts": cnt})
    out = collate_variable_sequence_length(batch)
    for k in ("sequence_onehots", "sequence_lengths", "label_embeddings", "label_token_counts", "label_multihots"):
        assert np.array_equal(out[k].numpy(), g["out/" + k]), k
        assert str(out[k].dtype) == str(g["out_dtype/" + k]), k
    assert out["sequence_ids"] == [f"P{i}" for i in range(4)]
    # sub-sampling variants keep embeddings and multihots aligned
    sub = collate_variable_sequence_length(batch, label_sample_size=3)
    assert np.array_equal(sub["label_embeddings"].numpy(), g["out/label_embeddings"][:3])
    assert np.array_equal(sub["label_multihots"].numpy(), g["out/label_multihots"][:, :3])
    inb = collate_variable_sequence_length(batch, in_batch_sampling=True)
    keep = g["out/label_multihots"].sum(0) > 0
    assert np.array_equal(inb["label_multihots"].numpy(), g["out/label_multihots"][:, keep])


def test_label_bookkeeping_bit_exact(golden_dir):
    g = np.load(os.path.join(golden_dir, "bookkeeping.npz"))
    data = [(s, i, l.split(" ")) for s, i, l in zip(g["fasta_seqs"], g["fasta_ids"], g["fasta_labels"])]
    idx_ids, idx_types = list(g["index_id"]), list(g["index_type"])
    tok = g["index_token_count"]
    for dtype, descr in (("test", ["name", "label"]), ("train", ["name", "label", "synonym_exact"])):
        p = dtype + "/"
        recs = LB.deduplicate(data)
        if dtype == "train":  # MAX_SEQUENCE_LENGTH only trims the train set (datasets.py:162-168)
            recs = [r for r in recs if len(r[0]) <= 15]
        assert [r[1] for r in recs] == list(g[p + "kept_ids"])
        voc = LB.generate_vocabularies(recs)
        assert voc["label_vocab"] == list(g[p + "label_vocabulary"])
        assert voc["amino_acid_vocab"] == list(g[p + "amino_acid_vocabulary"])
        kept, span = LB.embedding_row_index(idx_ids, idx_types, voc["label_vocab"], descr)
        assert np.array_equal(kept, g[p + "filtered_rows"])
        assert [span[l][0] for l in voc["label_vocab"]] == list(g[p + "min_idx"])
        assert [span[l][1] for l in voc["label_vocab"]] == list(g[p + "max_idx"])
        rows = LB.sorted_embedding_rows(span, voc["label_vocab"])
        assert np.array_equal(kept[rows], g[p + "sorted_rows"])
        assert np.array_equal(tok[kept][rows], g[p + "sorted_token_counts"])
        np.random.seed(123)
        srows = LB.sampled_embedding_rows(span, voc["label_vocab"])
        assert np.array_equal(kept[srows], g[p + "sampled_rows_seed123"])
        assert np.array_equal(tok[kept][srows], g[p + "sampled_token_counts_seed123"])
        l2i, _ = LB.get_vocab_mappings(voc["label_vocab"])
        a2i, _ = LB.get_vocab_mappings(voc["amino_acid_vocab"])
        for i, (seq, sid, labs) in enumerate(recs):
            assert np.array_equal(LB.label_multihot(labs, l2i).numpy(), g[p + f"ex{i}/multihots"])
            assert np.array_equal(LB.sequence_onehot(seq, a2i).numpy(), g[p + f"ex{i}/onehots"])
            assert int(g[p + f"ex{i}/length"]) == len(seq)


def test_length_bucket_sampler():
    from protnote_amd.data.samplers import LengthBucketBatchSampler

    lens = [5, 130, 2000, 128, 129, 3000, 256, 257, 1, 700]
    s = LengthBucketBatchSampler(lens, batch_size=2)
    batches = list(s)
    assert sorted(i for b in batches for i in b) == list(range(len(lens)))  # a partition
    for b in batches:
        assert len({s.bucket_of(lens[i]) for i in b}) == 1 and len(b) <= 2
    assert s.bucket_of(128) == 0 and s.bucket_of(129) == 1 and s.bucket_of(3000) == 4
    a = list(LengthBucketBatchSampler(lens, 2, shuffle=True, seed=3, rank=0, world_size=2))
    b = list(LengthBucketBatchSampler(lens, 2, shuffle=True, seed=3, rank=1, world_size=2))
    assert sorted(i for x in a + b for i in x) == list(range(len(lens)))  # ranks partition the epoch


def test_distributed_weighted_sampler_streams(golden_dir):
    """Bit-identical index streams vs the reference sampler (2 ranks x 2 epochs, with / without replacement)."""
    from protnote_amd.data.samplers import DistributedWeightedSampler

    g = np.load(os.path.join(golden_dir, "samplers.npz"))
    weights = torch.from_numpy(g["weights"])
    for repl in (True, False):
        for world in (1, 2):
            for rank in range(world):
                w = weights if repl else torch.cat([weights, weights[:8]])
                s = DistributedWeightedSampler(w, world_size=world, rank=rank, replacement=repl)
                if not repl:
                    s.num_samples = 30 // world
                    s.total_size = s.num_samples * world
                for epoch in (0, 1):
                    s.set_epoch(epoch)
                    assert list(iter(s)) == g[f"repl{int(repl)}/w{world}/r{rank}/e{epoch}"].tolist()


def test_grid_and_general_distributed_sampler_streams(golden_dir):
    """GridBatchSampler / GeneralDistributedSampler (protnote/data/samplers.py:15-63,127-224) against streams generated
    by the reference classes: same `random` seed -> the same grid cells in the same order over two epochs (drop_last and
    shuffle_grid on / off), the same rank shards; and the collator's grid path picks the cell's label batch."""
    import random

    from protnote_amd.data.collators import collate_variable_sequence_length
    from protnote_amd.data.samplers import GeneralDistributedSampler, GridBatchSampler

    g = np.load(os.path.join(golden_dir, "grid_samplers.npz"))
    obs = [5, 3, 8, 0, 9, 1, 7, 2, 6, 4, 10]
    for drop in (False, True):
        for shuf in (True, False):
            random.seed(5)
            s = GridBatchSampler(obs, 4, drop, num_labels=7, labels_batch_size=3, shuffle_grid=shuf)
            assert len(s) == int(g[f"grid/drop{int(drop)}/shuf{int(shuf)}/len"])
            for epoch in (0, 1):
                cells = list(iter(s))
                ro, rl = g[f"grid/drop{int(drop)}/shuf{int(shuf)}/e{epoch}/obs"], g[f"grid/drop{int(drop)}/shuf{int(shuf)}/e{epoch}/labels"]
                assert len(cells) == len(ro)
                for k, cell in enumerate(cells):
                    assert [c[0] for c in cell] == [v for v in ro[k].tolist() if v >= 0]
                    assert all(c[1] == [v for v in rl[k].tolist() if v >= 0] for c in cell)
    stream = [11, 4, 7, 0, 2, 9, 5, 13, 1, 8]
    for drop in (False, True):
        for rank in range(3):
            d = GeneralDistributedSampler(stream, num_replicas=3, rank=rank, drop_last=drop)
            assert list(iter(d)) == g[f"general/drop{int(drop)}/r{rank}"].tolist()
            assert len(d) == int(g[f"general/drop{int(drop)}/r{rank}/len"])
    # a grid cell through the collator: items carry label_idxs = the cell's label batch (datasets.py:412-423)
    random.seed(1)
    cell = next(iter(GridBatchSampler([0, 1, 2], 2, False, num_labels=5, labels_batch_size=2)))
    emb = torch.arange(5 * 3, dtype=torch.float32).reshape(5, 3)
    items = [{"sequence_onehots": torch.eye(20)[:, :4], "sequence_id": f"s{i}", "sequence_length": torch.tensor(4),
              "label_multihots": torch.tensor([1, 0, 1, 0, 1]), "label_embeddings": emb,
              "label_idxs": torch.tensor(lab), "label_token_counts": torch.arange(5)} for i, lab in cell]
    out = collate_variable_sequence_length(items, label_sample_size=2, grid_sampler=True)
    lab = torch.tensor(cell[0][1])
    assert torch.equal(out["label_embeddings"], emb[lab]) and torch.equal(out["label_multihots"][0], items[0]["label_multihots"][lab])
