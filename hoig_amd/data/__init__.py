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
    """``CustomDatasetDataLoader(opt, is_for_train, use_ddp)`` as train_ddp.py:33-41 / eval.py construct it.  ``opt`` fields read:
    ``dataset_mode``, ``batch_size``, ``serial_batches``, ``n_threads_train`` / ``n_threads_test`` (+ the dataset's own)."""

    def __init__(self, opt, is_for_train=True, use_ddp=False, device=None):
        self._dataset = DatasetFactory.get_by_name(opt.dataset_mode, opt, is_for_train)
        self._opt = opt
        self._device, self._stage = device, None          # (the stage is made by load_data(): the host half alone needs no GPU)
        # One rank of a data-parallel job sees its own shard in the sampler's order and drops the ragged last batch, so that every
        # rank runs the same number of steps; a single process shuffles unless opt.serial_batches and keeps the short batch.
        self._sampler = torch.utils.data.distributed.DistributedSampler(self._dataset) if use_ddp else None
        self._dataloader = torch.utils.data.DataLoader(
            self._dataset, batch_size=opt.batch_size, sampler=self._sampler, drop_last=bool(use_ddp),
            shuffle=False if use_ddp else not opt.serial_batches,
            num_workers=int(opt.n_threads_train if is_for_train else opt.n_threads_test),
            collate_fn=collate_raw, pin_memory=torch.cuda.is_available())

    def load_data(self):
        if self._stage is None:
            self._stage = DeviceStage(self._dataset, device=self._device, prepare=self._raw_stage())
        return _DeviceBatches(self._dataloader, self._stage)

    def _raw_stage(self):
        """The raw-batch stage of Trainer.set_input (HandRecoveryFlow.forward, models/trainer.py:46-145, and the assignment of its
        outputs, :346-362) as the loader's last device step, one batch ahead -- when the options carry what it needs (opt.mano_model,
        opt.object_assets: train_ddp.py passes ONE opt to the loader and to the model) and opt.loader_prepares is not False.  The
        batch keeps its raw entries; Trainer.set_input takes the prepared ones."""
        opt = self._opt
        if not getattr(opt, 'loader_prepares', True) or getattr(opt, 'mano_model', None) is None or not getattr(opt, 'object_assets', None):
            return None
        from .. import input_prep as IP
        from ..hand_recovery import HandRecoveryFlow
        flow = HandRecoveryFlow(opt, device=self._device)

        def prepare(b):
            import torch
            with torch.no_grad():
                out = flow(b['imageA'], b['imageB'], b['manoA'], b['manoB'])
                return IP.to_prepared(out, b['imageA'].float(), b['imageB'].float(), b.get('maskA'), b.get('maskB'))
        return prepare

    def load_raw_data(self):
        """The host half alone: raw batches (decoded 8-bit frames + annotations) as the workers deliver them."""
        return self._dataloader

    def load_sampler(self):
        return self._sampler

    def __len__(self):
        return len(self._dataset)


def _hov3(opt, is_for_train):
    from .hov3_dataset import HOv3Dataset
    return HOv3Dataset(opt, is_for_train)


def _ycb(opt, is_for_train):
    from .ycb_dataset import YCBDataset
    return YCBDataset(opt, is_for_train)


class DatasetFactory(object):
    # --dataset_mode of the HOIG_HOv3 copy / of the HOIG_DexYCB copy (each copy's data/__init__.py knows only its own)
    _makers = {'hov3': _hov3, 'ycb': _ycb}

    @staticmethod
    def get_by_name(dataset_name, opt, is_for_train):
        make = DatasetFactory._makers.get(dataset_name)
        if make is None:
            raise ValueError("Dataset [%s] not recognized." % dataset_name)
        dataset = make(opt, is_for_train)
        print('Dataset {} was created'.format(dataset.name))
        return dataset
