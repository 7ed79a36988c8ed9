"""Length-bucketed batching for variable-length inference (BASELINE configs[4]): sequences are
grouped by padding bucket so a batch never mixes a 40-residue and a 2000-residue protein; eval-mode logits are
invariant to the padded length, so this is parity-safe and only saves encoder work."""
import random
from typing import Iterator, List, Sequence

DEFAULT_BUCKETS = (128, 256, 512, 1024, 2048)


class LengthBucketBatchSampler:
    """Yields lists of dataset indices; every batch's sequences share one bucket (the last bucket is open-ended)."""

    def __init__(self, lengths: Sequence[int], batch_size: int, buckets: Sequence[int] = DEFAULT_BUCKETS,
                 shuffle: bool = False, seed: int = 0, rank: int = 0, world_size: int = 1):
        self.lengths, self.batch_size, self.buckets = list(lengths), int(batch_size), tuple(sorted(buckets))
        self.shuffle, self.seed, self.rank, self.world_size = shuffle, seed, rank, world_size
        self.epoch = 0

    def bucket_of(self, n: int) -> int:
        for k, b in enumerate(self.buckets):
            if n <= b:
                return k
        return len(self.buckets) - 1

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def _batches(self) -> List[List[int]]:
        groups = [[] for _ in self.buckets]
        for i, n in enumerate(self.lengths):
            groups[self.bucket_of(n)].append(i)
        rng = random.Random(self.seed + self.epoch)
        out = []
        for grp in groups:
            if self.shuffle:
                rng.shuffle(grp)
            out += [grp[s:s + self.batch_size] for s in range(0, len(grp), self.batch_size)]
        if self.shuffle:
            rng.shuffle(out)
        return out[self.rank::self.world_size]  # rank-strided, like the reference samplers (samplers.py:61,111)

    def __iter__(self) -> Iterator[List[int]]:
        return iter(self._batches())

    def __len__(self) -> int:
        return len(self._batches())


class DistributedWeightedSampler:
    """Index stream of the reference's DistributedWeightedSampler (protnote/data/samplers.py:66-124): every epoch
    draws floor(N / world) * world indices from torch.multinomial(weights) on a CPU generator seeded with the
    epoch, keeps the rank-strided slice and shuffles it with the same generator - bit-identical streams."""

    def __init__(self, weights, world_size: int = 1, rank: int = 0, replacement: bool = True):
        import math

        import torch

        self.weights = weights if isinstance(weights, torch.Tensor) else torch.tensor(weights, dtype=torch.double)
        self.world_size, self.rank, self.replacement, self.epoch = world_size, rank, replacement, 0
        self.num_samples = int(math.floor(len(self.weights) * 1.0 / world_size))
        self.total_size = self.num_samples * world_size

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def __len__(self) -> int:
        return self.num_samples

    def __iter__(self):
        import torch

        g = torch.Generator()
        g.manual_seed(self.epoch)
        if not self.replacement:
            assert len(self.weights) > self.total_size, \
                "When sampling without replacement, number of samples to draw must be less than the dataset size"
        idx = torch.multinomial(self.weights, self.total_size, replacement=self.replacement, generator=g)
        mine = idx[self.rank:self.total_size:self.world_size]
        return iter(mine[torch.randperm(len(mine), generator=g)].tolist())
