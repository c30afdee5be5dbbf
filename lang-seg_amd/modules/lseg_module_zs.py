"""LSegModuleZS -- the reference's zero-shot module surface (modules/lseg_module_zs.py:20-73), network side only.

`LSegModuleZS(data_path, dataset, batch_size, base_lr, max_epochs, **kwargs)` builds `self.net = LSegNetZS(label_list=...)`
from `label_files/fewshot_<dataset>.txt` (get_labels, :60-71) and forwards `(x, class_info)` to it
(lsegmentation_module_zs.py:82-83).  The few-shot episode loaders / Evaluator of lsegmentation_module_zs.py and
fewshot_data/ are host-side data plumbing outside the hot path and are not mirrored (SURVEY.md §8 out of scope).
"""
import os

import torch

from .models.lseg_net_zs import LSegNetZS, LSegRNNetZS

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class LSegModuleZS(torch.nn.Module):
    def __init__(self, data_path, dataset, batch_size, base_lr, max_epochs, **kwargs):
        super().__init__()
        self.data_path, self.dataset = data_path, dataset
        self.batch_size, self.base_lr, self.max_epochs = batch_size, base_lr, max_epochs
        self.other_kwargs = kwargs
        label_list = self.get_labels(dataset)
        self.len_dataloader = len(label_list)
        use_pretrained = kwargs.get("use_pretrained", True) in ("True", True)
        if kwargs.get("backbone", "clip_vitl16_384") in ["clip_resnet101"]:
            self.net = LSegRNNetZS()
        else:
            self.net = LSegNetZS(label_list=label_list, backbone=kwargs.get("backbone", "clip_vitl16_384"),
                                 features=kwargs.get("num_features", 256), aux=kwargs.get("aux", False),
                                 use_pretrained=use_pretrained, arch_option=kwargs.get("arch_option", 0),
                                 block_depth=kwargs.get("block_depth", 0), activation=kwargs.get("activation", "lrelu"))

    def get_labels(self, dataset):                      # lseg_module_zs.py:60-71
        path = "label_files/fewshot_{}.txt".format(dataset)
        if not os.path.exists(path):
            path = os.path.join(_HERE, path)            # in-tree copy when not run from the reference's CWD
        assert os.path.exists(path), "*** Error : {} not exist !!!".format(path)
        with open(path, "r") as f:
            return [line.strip() for line in f.readlines()]

    def forward(self, x, class_info):                   # lsegmentation_module_zs.py:82-83
        return self.net(x, class_info)
