"""LSegmentationModule -- mirror of the reference's modules/lsegmentation_module.py:26-304
(Lightning base: forward / evaluate / evaluate_random, optimiser param groups + poly LR,
criterion).  pytorch_lightning and PyTorch-Encoding are used when importable; when they are
absent (this build environment) the class degrades to a plain nn.Module with the same
attributes so the forward/evaluate surface (what test_lseg.py and the evaluators call) works.
"""
from argparse import ArgumentParser

import torch
import torch.nn as nn

try:
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except ImportError:          # no Lightning: keep the nn.Module protocol the evaluators rely on
    pl = None

    class _Base(nn.Module):
        def log(self, *a, **k):
            pass

try:
    from encoding.nn import SegmentationLosses
    from encoding.utils import SegmentationMetric
except ImportError:
    SegmentationLosses = SegmentationMetric = None

# The metric step after the forward runs on the device (lseg_hip.metrics -> lseg_op_seg_stats): one pass over the
# [B,K,H,W] scores instead of the reference's argmax + numpy histograms on the host
# ([3P] encoding.utils.batch_pix_accuracy / batch_intersection_union, lsegmentation_module.py:49-50,59-60).
from lseg_hip.metrics import batch_pix_accuracy, batch_intersection_union      # noqa: E402


class LSegmentationModule(_Base):
    def __init__(self, data_path, dataset, batch_size, base_lr, max_epochs, **kwargs):
        super().__init__()
        self.data_path = data_path
        self.batch_size = batch_size
        self.base_lr = base_lr / 16 * batch_size             # lsegmentation_module.py:32
        self.lr = self.base_lr
        self.epochs = max_epochs
        self.other_kwargs = kwargs
        self.enabled = False                                  # AMP explicitly off (:37)

    def forward(self, x):
        return self.net(x)

    def evaluate(self, x, target=None):                       # :43-52
        if target is not None and hasattr(self.net, "forward_metrics"):
            # metric-only call: the counts come from the engine's low-resolution logits through the x2 bilinear on the fly
            # (lseg_forward_stats) -- the [B,K,H,W] logits (138 MB per image) are not even written
            r = self.net.forward_metrics(x, target, ignore_index=self.other_kwargs.get("ignore_index", -1))
            return r["correct"], r["labeled"], r["area_inter"].numpy(), r["area_union"].numpy()
        pred = self.net.forward(x)
        if isinstance(pred, (tuple, list)):
            pred = pred[0]
        if target is None:
            return pred
        correct, labeled = batch_pix_accuracy(pred.data, target.data)
        inter, union = batch_intersection_union(pred.data, target.data, self.nclass)
        return correct, labeled, inter, union

    def evaluate_random(self, x, labelset, target=None):      # :54-63
        pred = self.net.forward(x, labelset)
        if isinstance(pred, (tuple, list)):
            pred = pred[0]
        if target is None:
            return pred
        correct, labeled = batch_pix_accuracy(pred.data, target.data)
        inter, union = batch_intersection_union(pred.data, target.data, self.nclass)
        return correct, labeled, inter, union

    def training_step(self, batch, batch_nb):                 # :66-81 (autocast/GradScaler are disabled there: self.enabled = False)
        img, target = batch
        out = self(img)                                       # train-mode engine forward; autograd node = lseg_backward
        multi_loss = isinstance(out, tuple)
        if not hasattr(self, "criterion"):
            self.criterion = self.get_criterion(**self.other_kwargs)
        loss = self.criterion(*out, target) if multi_loss else self.criterion(out, target)
        final_output = out[0] if multi_loss else out
        train_pred, train_gt = self._filter_invalid(final_output, target)
        if train_gt.nelement() != 0 and hasattr(self, "train_accuracy"):
            self.train_accuracy(train_pred, train_gt)
        self.log("train_loss", loss)
        return loss

    def native_training_step(self, img, target, lr_scale=1.0):
        """The same step without autograd in the loop (the fast path bench.py --train measures): engine forward + fused
        upsample/CE/backward + bucketed RCCL all-reduce + fused SGD, through lseg_hip.train.DataParallelTrainer."""
        from lseg_hip.train import DataParallelTrainer
        net = self.net
        B, _, H, W = img.shape
        eng = net._engine(B, H, W, net.text.shape[0], img.device)
        if getattr(self, "_trainer", None) is None or self._trainer.eng is not eng:
            eng.set_tokens(net.text)
            self._trainer = DataParallelTrainer(eng, dict(net.state_dict()), sync_bn=True)
        wd = self.other_kwargs.get("weight_decay", 1e-4)
        loss = self._trainer.step(img.float(), target, self.base_lr * lr_scale, self.base_lr * 10 * lr_scale, 0.9, wd,
                                  self.other_kwargs.get("ignore_index", -1))
        return loss                                           # the fused SGD updated the parameters' storage in place

    def _filter_invalid(self, pred, target):                  # :114-117
        valid = target != self.other_kwargs["ignore_index"]
        _, mx = torch.max(pred, dim=1)
        return mx[valid], target[valid]

    def configure_optimizers(self):                           # :119-175 (same param groups / poly LR)
        params_list = [{"params": self.net.pretrained.parameters(), "lr": self.base_lr}]
        if hasattr(self.net, "scratch"):
            params_list.append({"params": self.net.scratch.parameters(), "lr": self.base_lr * 10})
        opt = torch.optim.SGD(params_list, lr=self.base_lr, momentum=0.9,
                              weight_decay=self.other_kwargs.get("weight_decay", 1e-4))
        sch = torch.optim.lr_scheduler.LambdaLR(opt, lambda x: pow(1.0 - x / self.epochs, 0.9))
        return [opt], [sch]

    def get_criterion(self, **kwargs):                        # :236-244
        if SegmentationLosses is None:
            return nn.CrossEntropyLoss(ignore_index=kwargs.get("ignore_index", -1))
        return SegmentationLosses(se_loss=kwargs["se_loss"], aux=kwargs["aux"], nclass=self.num_classes,
                                  se_weight=kwargs["se_weight"], aux_weight=kwargs["aux_weight"],
                                  ignore_index=kwargs["ignore_index"])

    @staticmethod
    def add_model_specific_args(parent_parser):               # :246-304
        parser = ArgumentParser(parents=[parent_parser], add_help=False)
        parser.add_argument("--data_path", type=str, help="path where dataset is stored")
        parser.add_argument("--dataset", default="ade20k", help="dataset to train on")
        parser.add_argument("--batch_size", type=int, default=16, help="size of the batches")
        parser.add_argument("--base_lr", type=float, default=0.004, help="learning rate")
        parser.add_argument("--momentum", type=float, default=0.9, help="SGD momentum")
        parser.add_argument("--weight_decay", type=float, default=1e-4, help="weight_decay")
        parser.add_argument("--aux", action="store_true", default=False, help="Auxilary Loss")
        parser.add_argument("--aux-weight", type=float, default=0.2, help="Auxilary loss weight (default: 0.2)")
        parser.add_argument("--se-loss", action="store_true", default=False, help="Semantic Encoding Loss SE-loss")
        parser.add_argument("--se-weight", type=float, default=0.2, help="SE-loss weight (default: 0.2)")
        parser.add_argument("--midasproto", action="store_true", default=False, help="midasprotocol")
        parser.add_argument("--ignore_index", type=int, default=-1, help="numeric value of ignore label in gt")
        parser.add_argument("--augment", action="store_true", default=False, help="Use extended augmentations")
        return parser
