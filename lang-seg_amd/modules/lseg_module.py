"""LSegModule -- drop-in for the reference's modules/lseg_module.py:23-183: same constructor
signature, attributes (base_size, crop_size, _up_kwargs, mean, std, val_transform, num_classes,
net) and CLI flags; `self.net` is the HIP-engine-backed LSegNet."""
import os
from argparse import ArgumentParser

from .lsegmentation_module import LSegmentationModule
from .models.lseg_net import LSegNet

try:
    from encoding.models.sseg.base import up_kwargs
except ImportError:
    up_kwargs = {"mode": "bilinear", "align_corners": True}      # [3P] encoding, SURVEY App. A.3

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class LSegModule(LSegmentationModule):
    def __init__(self, data_path, dataset, batch_size, base_lr, max_epochs, **kwargs):
        super().__init__(data_path, dataset, batch_size, base_lr, max_epochs, **kwargs)
        if dataset == "citys":
            self.base_size, self.crop_size = 2048, 768
        else:
            self.base_size, self.crop_size = 520, 480
        norm_mean = [0.5, 0.5, 0.5]
        norm_std = [0.5, 0.5, 0.5]
        print("** Use norm {}, {} as the mean and std **".format(norm_mean, norm_std))
        try:
            import torchvision.transforms as transforms
            tf = [transforms.ToTensor(), transforms.Normalize(norm_mean, norm_std)]
            self.train_transform = transforms.Compose(tf)
            self.val_transform = transforms.Compose(tf)
        except ImportError:
            self.train_transform = self.val_transform = None
        # datasets come from PyTorch-Encoding (host-side, out of scope): built only when importable
        self.trainset = self.valset = None
        try:
            from data import get_dataset      # the reference's data/__init__.py shim
            self.trainset = get_dataset(dataset, root=data_path, split="train", mode="train",
                                        transform=self.train_transform, base_size=self.base_size,
                                        crop_size=self.crop_size)
            self.valset = get_dataset(dataset, root=data_path, split="val", mode="val",
                                      transform=self.val_transform, base_size=self.base_size,
                                      crop_size=self.crop_size)
        except Exception:
            pass
        labels = self.get_labels("ade20k")
        self.num_classes = self.nclass = len(labels)
        self.net = LSegNet(
            labels=labels,
            backbone=kwargs["backbone"],
            features=kwargs["num_features"],
            crop_size=self.crop_size,
            arch_option=kwargs["arch_option"],
            block_depth=kwargs["block_depth"],
            activation=kwargs["activation"],
        )
        self.net.pretrained.model.patch_embed.img_size = (self.crop_size, self.crop_size)
        self._up_kwargs = up_kwargs
        self.mean = norm_mean
        self.std = norm_std
        self.criterion = self.get_criterion(**{"se_loss": False, "aux": False, "se_weight": 0.2,
                                                "aux_weight": 0.2, "ignore_index": -1, **kwargs})

    def get_labels(self, dataset):
        """lseg_module.py:97-109 (path relative to CWD first, then the packaged copy)."""
        labels = []
        path = "label_files/{}_objectInfo150.txt".format(dataset)
        if not os.path.exists(path):
            path = os.path.join(_PKG_ROOT, path)
        assert os.path.exists(path), "*** Error : {} not exist !!!".format(path)
        with open(path, "r") as f:
            for line in f.readlines():
                labels.append(line.strip().split(",")[-1].split(";")[0])
        if dataset in ["ade20k"]:
            labels = labels[1:]
        return labels

    @staticmethod
    def add_model_specific_args(parent_parser):
        parser = LSegmentationModule.add_model_specific_args(parent_parser)
        parser = ArgumentParser(parents=[parser])
        parser.add_argument("--backbone", type=str, default="clip_vitl16_384", help="backbone network")
        parser.add_argument("--num_features", type=int, default=256,
                            help="number of featurs that go from encoder to decoder")
        parser.add_argument("--dropout", type=float, default=0.1, help="dropout rate")
        parser.add_argument("--finetune_weights", type=str, help="load weights to finetune from")
        parser.add_argument("--no-scaleinv", default=True, action="store_false", help="turn off scaleinv layers")
        parser.add_argument("--no-batchnorm", default=False, action="store_true", help="turn off batchnorm")
        parser.add_argument("--widehead", default=False, action="store_true", help="wider output head")
        parser.add_argument("--widehead_hr", default=False, action="store_true", help="wider output head")
        parser.add_argument("--arch_option", type=int, default=0, help="which kind of architecture to be used")
        parser.add_argument("--block_depth", type=int, default=0, help="how many blocks should be used")
        parser.add_argument("--activation", choices=["lrelu", "tanh"], default="lrelu",
                            help="use which activation to activate the block")
        return parser
