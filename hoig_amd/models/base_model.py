"""What the reference's drivers expect of any model object (models/base_model.py:7-147), written for the flat-buffer networks.

Only the NAMES are contract here: ``name`` / ``is_train``, the per-iteration verbs a driver calls, and the checkpoint helpers
``_save_network / _load_network / _save_optimizer / _load_optimizer / _load_params``.  Checkpoint files keep the reference's
naming scheme and payload -- ``net_epoch_<label>_id_<G|D>.pth`` is ``torch.save(state_dict)`` of NCHW fp32 tensors under the
reference's parameter names, ``opt_epoch_<label>_id_<G|D>.pth`` a ``torch.optim.Adam``-layout state -- so runs can be resumed
across the two implementations (hoig_amd/nn.py produces both layouts from the flat buffers).
"""
import os

import torch

_DDP_PREFIX = 'module.'
# verbs a concrete model has to provide itself (the reference's stubs `assert False`; a typed error names the class instead)
_REQUIRED = ('set_input', 'set_train', 'set_eval', 'forward', 'test', 'optimize_parameters', 'save', 'load')


class CheckpointDir(object):
    """<checkpoints_dir>/<name>/{net,opt}_epoch_<label>_id_<ident>.pth"""

    def __init__(self, root):
        self.root = root

    def file(self, kind, label, ident):
        return os.path.join(self.root, '{}_epoch_{}_id_{}.pth'.format(kind, label, ident))

    def write(self, payload, kind, label, ident):
        os.makedirs(self.root, exist_ok=True)
        path = self.file(kind, label, ident)
        torch.save(payload, path)
        return path

    def read(self, path):
        if not os.path.isfile(path):
            raise FileNotFoundError('checkpoint %s does not exist (nothing has been trained / saved under %s yet)'
                                    % (path, self.root))
        return torch.load(path, map_location='cpu')


def strip_ddp_prefix(state):
    """Keys saved through a DistributedDataParallel wrapper carry 'module.'; a bare network wants them without
    (base_model.py:108-116 strips seven characters off every key that mentions 'module')."""
    return type(state)((k[len(_DDP_PREFIX):] if k.startswith(_DDP_PREFIX) else k, v) for k, v in state.items())


def _missing(verb):
    def stub(self, *args, **kwargs):
        raise NotImplementedError('%s.%s()' % (type(self).__name__, verb))
    stub.__name__ = verb
    return stub


class BaseModel(object):
    def __init__(self, opt, use_ddp=False):
        self._name = type(self).__name__
        self._opt = opt
        self._use_ddp = bool(use_ddp)
        self._is_train = bool(opt.is_train)
        self._gpu_ids = getattr(opt, 'gpu_ids', None)
        self._save_dir = os.path.join(opt.checkpoints_dir, opt.name)
        self._ckpt = CheckpointDir(self._save_dir)
        cond_nc = getattr(opt, 'cond_nc', 2)
        self._G_cond_nc, self._D_cond_nc = cond_nc, cond_nc

    name = property(lambda self: self._name)
    is_train = property(lambda self: self._is_train)

    # reporting hooks: empty unless the model has something to say
    def get_image_paths(self):
        return {}

    get_current_visuals = get_current_errors = get_current_scalars = get_image_paths

    def update_learning_rate(self):
        return None

    # ---- checkpoints
    def _save_network(self, network, network_label, epoch_label):
        print('saved net: %s' % self._ckpt.write(network.state_dict(), 'net', epoch_label, network_label))

    def _save_optimizer(self, optimizer, optimizer_label, epoch_label):
        self._ckpt.write(optimizer.state_dict(), 'opt', epoch_label, optimizer_label)

    def _load_network(self, network, network_label, epoch_label, need_module=False):
        self._load_params(network, self._ckpt.file('net', epoch_label, network_label), need_module)

    def _load_params(self, network, load_path, need_module=False):
        """need_module=True hands the file's keys to the network untouched (the HOIG_DexYCB copy loads into its DDP wrappers
        that way, HOIG_DexYCB/models/trainer.py:562,566); otherwise a 'module.' prefix is removed first."""
        state = self._ckpt.read(load_path)
        # (a BARE network always gets the prefix removed: need_module describes how the DexYCB copy feeds its DDP wrappers, and a
        # single-GPU run of that copy must still read the checkpoints a DDP run saved)
        wrapped = hasattr(network, 'module')
        network.load_state_dict(state if (need_module and wrapped) else strip_ddp_prefix(state))
        print('Loading net: %s' % load_path)

    def _load_optimizer(self, optimizer, optimizer_label, epoch_label):
        path = self._ckpt.file('opt', epoch_label, optimizer_label)
        optimizer.load_state_dict(self._ckpt.read(path))
        print('loaded optimizer: %s' % path)

    def print_network(self, network):
        print(network)
        print('Total number of parameters: %d' % sum(p.numel() for p in network.parameters()))


for _verb in _REQUIRED:
    setattr(BaseModel, _verb, _missing(_verb))
del _verb
