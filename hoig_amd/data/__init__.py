"""The reference's ``data`` package for the hot path's input side (HOIG_HOv3/data/__init__.py:4-54): the same two
names, ``CustomDatasetDataLoader(opt, is_for_train, use_ddp)`` with ``load_data()`` / ``load_sampler()`` / ``__len__`` and
``DatasetFactory.get_by_name``, so that train_ddp.py:33-41 and eval.py keep their lines.

What differs is WHERE a sample's pixels are worked on.  The reference's workers decode, resize, warp, scale and normalise every frame
on the host and re-parse the object mesh text for every item (hov3_dataset.py:215-250).  Here a worker only decodes (PIL) and reads
the annotation; the batch of 8-bit frames goes to the GPU from pinned memory and ``DeviceStage`` (device_stage.py) does the rest there
in a handful of launches -- the reference's arithmetic, integer for integer (hoig_amd/csrc/data_prep.hip) -- with the object meshes
parsed once and kept on the device.  ``load_data()`` yields the reference's batch dict with device tensors; ``Trainer.set_input``
takes it as it is."""
import torch.utils.data

from .device_stage import DeviceStage, collate_raw


class _DeviceBatches(object):
    """The iterable ``load_data()`` returns: a torch DataLoader of raw host batches, finished on the device one batch ahead."""

    def __init__(self, loader, stage):
        self._loader, self._stage = loader, stage
        self.dataset, self.batch_size = loader.dataset, loader.batch_size

    def __len__(self):
        return len(self._loader)

    def __iter__(self):
        pending = None
        for raw in self._loader:
            nxt = self._stage.submit(raw)              # H2D + kernels of batch i+1 on the side stream ...
            if pending is not None:
                yield self._stage.finish(pending)      # ... while the caller trains on batch i
            pending = nxt
        if pending is not None:
            yield self._stage.finish(pending)


class CustomDatasetDataLoader(object):
    def __init__(self, opt, is_for_train=True, use_ddp=False, device=None):
        self._opt = opt
        self._is_for_train = is_for_train
        self._num_threds = opt.n_threads_train if is_for_train else opt.n_threads_test
        self._device = device
        self._create_dataset(use_ddp=use_ddp)

    def _create_dataset(self, use_ddp=False):
        self._dataset = DatasetFactory.get_by_name(self._opt.dataset_mode, self._opt, self._is_for_train)
        pin = torch.cuda.is_available()
        if use_ddp:                                                     # data/__init__.py:12-20
            self._sampler = torch.utils.data.distributed.DistributedSampler(self._dataset)
            self._dataloader = torch.utils.data.DataLoader(
                self._dataset, batch_size=self._opt.batch_size, shuffle=False, num_workers=int(self._num_threds),
                sampler=self._sampler, drop_last=True, collate_fn=collate_raw, pin_memory=pin)
        else:                                                           # :21-29
            self._sampler = None
            self._dataloader = torch.utils.data.DataLoader(
                self._dataset, batch_size=self._opt.batch_size, shuffle=not self._opt.serial_batches,
                num_workers=int(self._num_threds), drop_last=False, collate_fn=collate_raw, pin_memory=pin)
        self._stage = None                      # (made by load_data(): the host half alone needs no GPU)

    def load_data(self):
        if self._stage is None:
            self._stage = DeviceStage(self._dataset, device=self._device)
        return _DeviceBatches(self._dataloader, self._stage)

    def load_raw_data(self):
        """The host half alone: raw batches (decoded 8-bit frames + annotations) as the workers deliver them."""
        return self._dataloader

    def load_sampler(self):
        return self._sampler

    def __len__(self):
        return len(self._dataset)


class DatasetFactory(object):
    @staticmethod
    def get_by_name(dataset_name, opt, is_for_train):
        if dataset_name == 'hov3':
            from .hov3_dataset import HOv3Dataset
            dataset = HOv3Dataset(opt, is_for_train)
        elif dataset_name == 'ycb':                                         # the HOIG_DexYCB copy's factory (its data/__init__.py:45-47)
            from .ycb_dataset import YCBDataset
            dataset = YCBDataset(opt, is_for_train)
        else:
            raise ValueError("Dataset [%s] not recognized." % dataset_name)
        print('Dataset {} was created'.format(dataset.name))
        return dataset
