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


def _resolve_world(world_size, rank):
    """`None` defaults of the reference's samplers (samplers.py:15-37 via torch's DistributedSampler, :66-74): the process
    group's world size / this process's rank; RuntimeError when torch.distributed is not available, and whatever
    torch.distributed raises when no process group has been initialised."""
    if world_size is None or rank is None:
        import torch.distributed as dist

        if not dist.is_available():
            raise RuntimeError("Requires distributed package to be available")
        if world_size is None:
            world_size = dist.get_world_size()
        if rank is None:
            rank = dist.get_rank()
    return int(world_size), int(rank)


class DistributedWeightedSampler:
    """Index stream of the reference's DistributedWeightedSampler (protnote/data/samplers.py:66-124): every epoch
    draws floor(N / world) * world indices from torch.multinomial(weights) on a CPU generator seeded with the
    epoch, keeps the rank-strided slice and shuffles it with the same generator - bit-identical streams."""

    def __init__(self, weights, world_size=None, rank=None, replacement=True):
        import math

        import torch

        world_size, rank = _resolve_world(world_size, rank)  # None -> torch.distributed, as the reference (:69-74)
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


class GeneralDistributedSampler:
    """Rank shard of ANY sampler's index stream (reference protnote/data/samplers.py:15-63: a DistributedSampler with
    shuffle=False over `list(sampler)`): the stream is padded by repeating its head to a multiple of the world size
    (or cut to one with drop_last) and rank r keeps indices r, r + world, ..."""

    def __init__(self, sampler, num_replicas=None, rank=None, seed: int = 0, drop_last: bool = False):
        import math

        # None -> torch.distributed's world, as torch's DistributedSampler (the reference's base class, :15-37) resolves it
        num_replicas, rank = _resolve_world(num_replicas, rank)
        if rank >= num_replicas or rank < 0:
            raise ValueError(f"Invalid rank {rank}, rank should be in the interval [0, {num_replicas - 1}]")
        n = len(sampler)
        assert n > num_replicas, "Total samples must be > num replicas"
        self.sampler, self.num_replicas, self.rank, self.seed, self.drop_last = sampler, num_replicas, rank, seed, drop_last
        self.epoch = 0
        if drop_last and n % num_replicas != 0:  # torch DistributedSampler: drop the tail
            self.num_samples = math.ceil((n - num_replicas) / num_replicas)
        else:
            self.num_samples = math.ceil(n / num_replicas)
        self.total_size = self.num_samples * num_replicas

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def __len__(self) -> int:
        return self.num_samples

    def __iter__(self):
        import math

        import torch

        torch.manual_seed(self.epoch + self.seed)  # as the reference does: the wrapped sampler may draw from torch's RNG
        indices = list(self.sampler)
        if not self.drop_last:
            pad = self.total_size - len(indices)
            if pad <= len(indices):
                indices += indices[:pad]
            else:
                indices += (indices * math.ceil(pad / len(indices)))[:pad]
        else:
            indices = indices[:self.total_size]
        assert len(indices) == self.total_size
        return iter(indices[self.rank:self.total_size:self.num_replicas])


class GridBatchSampler:
    """Batches over the (observation batch) x (label batch) grid (reference protnote/data/samplers.py:127-224, the
    `grid_sampler` option of create_multiple_loaders, datasets.py:609-618): every epoch shuffles the label indices with
    Python's `random`, cuts observations and labels into batches, takes their product - optionally shuffled - and yields,
    per grid cell, the list [(observation index, label batch), ...].  A dataset item built from such an index carries
    `label_idxs` = the label batch (datasets.py:412-423), which the collator's grid path reads from batch[0]
    (collators.py:56-70 / protnote_amd.data.collators.sample_label_indices).  Same draws from `random` in the same order
    as the reference: identical streams for identical seeds."""

    def __init__(self, observation_sampler, observations_batch_size: int, drop_last_observation_batch: bool,
                 num_labels: int, labels_batch_size: int, shuffle_grid: bool = True):
        self.observation_sampler = observation_sampler
        self.observations_batch_size = observations_batch_size
        self.drop_last_observation_batch = drop_last_observation_batch
        self.num_labels, self.labels_batch_size, self.shuffle_grid = num_labels, labels_batch_size, shuffle_grid
        self.labels_idxs = list(range(num_labels))
        n_lab = -(-num_labels // labels_batch_size)
        n_obs = (len(observation_sampler) // observations_batch_size if drop_last_observation_batch
                 else -(-len(observation_sampler) // observations_batch_size))
        self.total_num_batches = int(n_lab * n_obs)

    def __len__(self) -> int:
        return self.total_num_batches

    def get_label_batches(self):
        return [self.labels_idxs[i:i + self.labels_batch_size] for i in range(0, self.num_labels, self.labels_batch_size)]

    def get_observation_batches(self):
        idx = list(self.observation_sampler)
        bs = self.observations_batch_size
        full = len(idx) // bs * bs
        batches = [idx[i:i + bs] for i in range(0, full, bs)]
        if not self.drop_last_observation_batch and full < len(idx):
            batches.append(idx[full:])
        return batches

    def __iter__(self):
        from itertools import product

        random.shuffle(self.labels_idxs)
        cells = list(product(self.get_observation_batches(), self.get_label_batches()))
        if self.shuffle_grid:
            random.shuffle(cells)
        for observation_batch, label_batch in cells:
            yield list(product(observation_batch, [label_batch]))
