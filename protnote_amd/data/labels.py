"""Label-index bookkeeping of the reference data pipeline, as pure functions (integer work, bit-exact):

  * vocabularies            - protnote/utils/data.py:123-151 (sorted unique ids / labels / residues)
  * vocab mappings          - protnote/utils/data.py:116-120
  * multihot / one-hot      - protnote/data/datasets.py:353-377 (process_example)
  * embedding-row bookkeeping - protnote/data/datasets.py:269-343: filter the cached-embedding index by
    description type and label vocabulary, first/last row per label, rows in vocabulary order ("sorted",
    which makes consecutive rows belong to one label - the layout ProtNote's inference-time ensembling
    reshape relies on, ProtNote.py:313-322), and one randomly sampled row per label (train augmentation).
"""
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np
import torch


def deduplicate(data: List[Tuple[str, str, List[str]]]) -> List[Tuple[str, str, List[str]]]:
    """Keep the first record of every distinct sequence (datasets.py:150-158)."""
    seen, out = set(), []
    for rec in data:
        if rec[0] not in seen:
            seen.add(rec[0])
            out.append(rec)
    return out


def generate_vocabularies(data: Iterable[Tuple[str, str, List[str]]]) -> Dict[str, List[str]]:
    aa, lab, ids = set(), set(), set()
    for seq, sid, labels in data:
        ids.add(sid)
        lab.update(labels)
        aa.update(seq)
    return {"amino_acid_vocab": sorted(aa), "label_vocab": sorted(lab), "sequence_id_vocab": sorted(ids)}


def get_vocab_mappings(vocabulary: Sequence[str]):
    assert len(vocabulary) == len(set(vocabulary)), "items in vocabulary must be unique"
    term2int = {t: i for i, t in enumerate(vocabulary)}
    return term2int, {i: t for t, i in term2int.items()}


def label_multihot(labels: Sequence[str], label2int: Dict[str, int]) -> torch.Tensor:
    """int64 [N]: one_hot(ints).sum(0) - repeated labels count twice, exactly like the reference."""
    out = torch.zeros(len(label2int), dtype=torch.int64)
    for l in labels:
        out[label2int[l]] += 1
    return out


def sequence_onehot(sequence: str, aminoacid2int: Dict[str, int]) -> torch.Tensor:
    """int64 [A, L] one-hot (datasets.py:369-371)."""
    ids = torch.tensor([aminoacid2int[a] for a in sequence], dtype=torch.long)
    out = torch.zeros(len(aminoacid2int), len(sequence), dtype=torch.int64)
    out[ids, torch.arange(len(sequence))] = 1
    return out


def embedding_row_index(index_ids: Sequence[str], index_types: Sequence[str], label_vocabulary: Sequence[str],
                        descriptions_considered: Sequence[str]):
    """-> (kept_rows int64 [M] into the cached embedding matrix, {label: (min_idx, max_idx)} into the KEPT rows)."""
    vocab, types = set(label_vocabulary), set(descriptions_considered)
    kept = [i for i, (g, t) in enumerate(zip(index_ids, index_types)) if t in types and g in vocab]
    span: Dict[str, Tuple[int, int]] = {}
    for new_i, old_i in enumerate(kept):
        g = index_ids[old_i]
        lo, hi = span.get(g, (new_i, new_i))
        span[g] = (min(lo, new_i), max(hi, new_i))
    return np.asarray(kept, dtype=np.int64), span


def sorted_embedding_rows(span: Dict[str, Tuple[int, int]], label_vocabulary: Sequence[str]) -> np.ndarray:
    """Rows [min..max] of every label, in vocabulary order (datasets.py:327-343)."""
    rows: List[int] = []
    for g in label_vocabulary:
        lo, hi = span[g]
        rows.extend(range(lo, hi + 1))
    return np.asarray(rows, dtype=np.int64)


def sampled_embedding_rows(span: Dict[str, Tuple[int, int]], label_vocabulary: Sequence[str]) -> np.ndarray:
    """One np.random.randint(min, max+1) draw per label in vocabulary order (datasets.py:311-325); consumes the
    global numpy RNG stream exactly like the reference."""
    return np.asarray([np.random.randint(low=span[g][0], high=span[g][1] + 1) for g in label_vocabulary],
                      dtype=np.int64)
