"""Model API base + checkpoint helpers (reference: models/base_model.py:7-147).  File names and formats are the
reference's: ``net_epoch_<label>_id_<G|D>.pth`` = torch.save(state_dict) with NCHW fp32 tensors under the
reference parameter names, ``opt_epoch_<label>_id_<G|D>.pth`` = torch.optim.Adam-layout state."""
import os
from collections import OrderedDict

import torch


class BaseModel(object):
    def __init__(self, opt, use_ddp=False):
        self._name = 'BaseModel'
        self._opt = opt
        self._gpu_ids = opt.gpu_ids
        self._is_train = opt.is_train
        self._use_ddp = use_ddp
        self._save_dir = os.path.join(opt.checkpoints_dir, opt.name)
        self._G_cond_nc = self._D_cond_nc = getattr(opt, 'cond_nc', 2)

    @property
    def name(self):
        return self._name

    @property
    def is_train(self):
        return self._is_train

    def set_input(self, input):
        assert False, "set_input not implemented"

    def set_train(self):
        assert False, "set_train not implemented"

    def set_eval(self):
        assert False, "set_eval not implemented"

    def forward(self, *input):
        assert False, "forward not implemented"

    def test(self):
        assert False, "test not implemented"

    def get_image_paths(self):
        return {}

    def optimize_parameters(self):
        assert False, "optimize_parameters not implemented"

    def get_current_visuals(self):
        return {}

    def get_current_errors(self):
        return {}

    def get_current_scalars(self):
        return {}

    def save(self, label):
        assert False, "save not implemented"

    def load(self):
        assert False, "load not implemented"

    def update_learning_rate(self):
        pass

    # ---- checkpoint files -------------------------------------------------------------------
    def _path(self, kind, label, ident):
        return os.path.join(self._save_dir, '%s_epoch_%s_id_%s.pth' % (kind, label, ident))

    def _save_optimizer(self, optimizer, optimizer_label, epoch_label):
        os.makedirs(self._save_dir, exist_ok=True)
        torch.save(optimizer.state_dict(), self._path('opt', epoch_label, optimizer_label))

    def _load_optimizer(self, optimizer, optimizer_label, epoch_label):
        load_path = self._path('opt', epoch_label, optimizer_label)
        assert os.path.exists(load_path), 'Weights file not found. %s ' \
                                          'Have you trained a model!? We are not providing one' % load_path
        optimizer.load_state_dict(torch.load(load_path, map_location='cpu'))
        print('loaded optimizer: %s' % load_path)

    def _save_network(self, network, network_label, epoch_label):
        os.makedirs(self._save_dir, exist_ok=True)
        save_path = self._path('net', epoch_label, network_label)
        torch.save(network.state_dict(), save_path)
        print('saved net: %s' % save_path)

    def _load_network(self, network, network_label, epoch_label, need_module=False):
        self._load_params(network, self._path('net', epoch_label, network_label), need_module)

    def _load_params(self, network, load_path, need_module=False):
        assert os.path.exists(load_path), \
            'Weights file not found. Have you trained a model!? We are not providing one %s' % load_path
        save_data = torch.load(load_path, map_location='cpu')
        if need_module:
            network.load_state_dict(save_data)
        else:
            state_dict = OrderedDict()
            for k, v in save_data.items():
                state_dict[k[7:] if 'module' in k else k] = v      # strip DDP's 'module.' (base_model.py:108-116)
            network.load_state_dict(state_dict)
        print('Loading net: %s' % load_path)

    def print_network(self, network):
        num_params = sum(p.numel() for p in network.parameters())
        print(network)
        print('Total number of parameters: %d' % num_params)
