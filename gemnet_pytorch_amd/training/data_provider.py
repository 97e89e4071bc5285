"""Train/val/test split + infinite batch iterators (counterpart of gemnet/training/data_provider.py:25-174).

The DataContainer batches internally (its __getitem__ takes a list of molecule ids and runs the native index
builder once for the whole batch), so the DataLoader is driven by a BatchSampler used as *sampler*."""
import functools

import numpy as np
import torch
from torch.utils.data import DataLoader, Subset
from torch.utils.data.sampler import BatchSampler, SequentialSampler, SubsetRandomSampler


def collate(batch, target_keys):
    batch = batch[0]
    inputs = {k: v for k, v in batch.items() if k not in target_keys}
    targets = {k: v for k, v in batch.items() if k in target_keys}
    return inputs, targets


class DataProvider:
    def __init__(self, data_container, ntrain, nval, batch_size=1, seed=None, random_split=False, shuffle=True,
                 sample_with_replacement=False, split=None, **kwargs):
        self.kwargs = kwargs
        self.data_container = data_container
        self._ndata = len(data_container)
        self.batch_size, self.seed = batch_size, seed
        self.random_split, self.shuffle, self.sample_with_replacement = random_split, shuffle, sample_with_replacement
        self._random_state = np.random.RandomState(seed=seed)
        if split is None:
            all_idx = np.arange(self._ndata)
            if random_split:
                all_idx = self._random_state.permutation(all_idx)
            if sample_with_replacement:
                all_idx = self._random_state.choice(all_idx, self._ndata, replace=True)
            self.idx = {"train": all_idx[:ntrain], "val": all_idx[ntrain:ntrain + nval],
                        "test": all_idx[ntrain + nval:]}
        else:
            if isinstance(split, str):
                assert split.endswith(".npz"), "'split' has to be a .npz file if 'split' is of type str"
                split = np.load(split)
            elif not isinstance(split, dict):
                raise TypeError("'split' has to be either of type str or dict if not None.")
            self.idx = {k: np.array(split[k]) for k in ("train", "val", "test")}
        self.nsamples = {k: len(v) for k, v in self.idx.items()}

    def save_split(self, path):
        assert isinstance(path, str) and path.endswith(".npz"), "'path' has to end with .npz"
        np.savez(path, **self.idx)

    def get_dataset(self, split, batch_size=None):
        assert split in self.idx
        batch_size = batch_size or self.batch_size
        indices = self.idx[split]
        if self.shuffle and split == "train":
            gen = torch.Generator()
            if self.seed is not None:
                gen.manual_seed(self.seed)
            sampler, dataset = SubsetRandomSampler(indices, gen), self.data_container
        else:
            dataset = Subset(self.data_container, indices)
            sampler = SequentialSampler(dataset)
        loader = DataLoader(dataset, sampler=BatchSampler(sampler, batch_size=batch_size, drop_last=False),
                            collate_fn=functools.partial(collate, target_keys=self.data_container.targets),
                            pin_memory=torch.cuda.is_available(), **self.kwargs)

        def forever():
            while True:
                yield from loader

        return forever()
