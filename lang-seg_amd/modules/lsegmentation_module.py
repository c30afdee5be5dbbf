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


class EngineSGD(torch.optim.SGD):
    """The optimizer `configure_optimizers` returns (lsegmentation_module.py:119-175: SGD, momentum 0.9, weight decay, two parameter
    groups -- pretrained at base_lr, scratch at 10 x base_lr -- under a poly LambdaLR).  A regular torch.optim.SGD for every caller
    (param_groups, lr schedulers, closures, add_param_group); when the gradients of this step were produced by the HIP engine
    (`training_step` -> `loss.backward()`), `step()` runs the engine's fused lseg_sgd_step instead of torch's foreach kernels: one launch
    over the fp32 masters that also refreshes the engine's bf16 operand copies -- no re-bind, no re-pack of 344 M parameters.
    Momentum lives in the engine; state_dict() / load_state_dict() move it to / from torch's `momentum_buffer` layout, so Lightning
    checkpoints (`optimizer_states`) keep working.  zero_grad() only flags the engine (the next backward overwrites its buckets)."""

    def __init__(self, params, net=None, **kw):
        super().__init__(params, **kw)
        self._net = net
        self._pending_momentum = None

    def _engine(self):
        net = self._net
        if net is None:
            return None
        for eng in net._engines.values():
            ts = getattr(eng, "_ts", None)
            if ts is not None and ts.fresh:
                return eng
        return None

    def _fusable(self, eng=None):
        """The fused lseg_sgd_step assigns the two learning rates by KEY PREFIX (pretrained.* / scratch.*): it may only replace torch's
        step when the optimizer's groups are exactly those two sets, every parameter of them trainable, equal hyper-parameters otherwise.
        A caller who regrouped, froze or added parameters gets torch.optim.SGD."""
        g = self.param_groups
        if len(g) != 2:
            return False
        same = all(g[0][k] == g[1][k] for k in ("momentum", "weight_decay", "dampening", "nesterov"))
        if not (same and g[0]["dampening"] == 0 and not g[0]["nesterov"] and not g[0].get("maximize", False)):
            return False
        if eng is None:
            return True
        # the key holds everything the answer depends on: WHICH tensors sit in each group (a swap within equal-sized groups) and whether each
        # is trainable (a later p.requires_grad_(False) must send the step back to torch: the fused step would keep applying weight
        # decay and momentum to a frozen tensor) -- two tuple builds over ~370 parameters per step, ~50 us
        key = (id(eng), tuple(tuple(id(p) for p in x["params"]) for x in g), tuple(p.requires_grad for x in g for p in x["params"]))
        if getattr(self, "_fusable_cache", (None, None))[0] != key:
            names = {id(p): k for k, p in self._net.named_parameters()}
            groups = [{names.get(id(p)) for p in x["params"]} for x in g]
            want0 = {k for k in eng.grads if k.startswith("pretrained.")}
            want1 = {k for k in eng.grads if k.startswith("scratch.")}
            trainable = all(p.requires_grad for x in g for p in x["params"])
            # parameters the backward never reaches (pretrained.model.norm / head, refinenet4.resConfUnit1: DDP's find_unused_parameters
            # in the reference) sit in the groups without a gradient: torch skips them, and so does the engine
            ok = trainable and want0 <= groups[0] and want1 <= groups[1] and not (groups[0] & want1) and not (groups[1] & want0)
            ok = ok and all(k is None or k in eng.grads or not k.startswith(("pretrained.", "scratch.")) or self._never_has_grad(k)
                            for k in (groups[0] | groups[1]))
            self._fusable_cache = (key, ok)
        return self._fusable_cache[1]

    @staticmethod
    def _never_has_grad(key):
        return key.startswith(("pretrained.model.norm.", "pretrained.model.head.")) or ".refinenet4.resConfUnit1." in key

    def _momentum_owner(self):
        """The ONE training engine that holds the momentum buffers (the first that took a fused step)."""
        for e in (self._net._engines.values() if self._net is not None else []):
            ts = getattr(e, "_ts", None)
            if ts is not None and ts.sgd_steps > 0:
                return e
        return None

    def _sync_momentum_to_torch(self):
        """Before torch's own step takes over after fused steps: the engine's momentum becomes torch's `momentum_buffer` state (and the
        engine forgets it), so the two paths never run on different optimizer states."""
        owner = self._momentum_owner()
        if owner is None:
            return
        params = {k: p for k, p in self._net.named_parameters()}
        for k in owner.grads:
            p = params.get(k)
            if p is not None:
                self.state[p]["momentum_buffer"] = owner.get_momentum(k).to(p.device)
        torch.cuda.current_stream(owner.device).synchronize()
        owner._ts.sgd_steps = 0
        owner.mark_sgd_initialized(False)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:                               # Lightning hands training_step + backward over as the closure
            with torch.enable_grad():
                loss = closure()
        eng = self._engine()
        if eng is None or self._net.autograd_grads or not self._fusable(eng):
            self._sync_momentum_to_torch()
            super().step()
            return loss
        ts = eng._ts
        owner = self._momentum_owner()
        if owner is not None and owner is not eng:
            # a second training engine (another crop size / batch path) takes a fused step: the momentum moves with the optimizer, it
            # is not re-started per engine
            self._push_momentum(eng, {k: owner.get_momentum(k) for k in owner.grads})
            torch.cuda.current_stream(owner.device).synchronize()
            ts.sgd_steps = owner._ts.sgd_steps
            owner._ts.sgd_steps = 0
            owner.mark_sgd_initialized(False)
        if self._pending_momentum is not None:                # restored from a checkpoint: push it into the engine once
            self._push_momentum(eng, self._pending_momentum)
            self._pending_momentum = None
        elif ts.sgd_steps == 0 and any("momentum_buffer" in st for st in self.state.values()):
            # torch steps came first (fallback path): their momentum seeds the engine's
            params = {id(p): k for k, p in self._net.named_parameters()}
            self._push_momentum(eng, {params[id(p)]: st["momentum_buffer"] for p, st in self.state.items()
                                      if "momentum_buffer" in st and id(p) in params})
            self.state.clear()
        g = self.param_groups
        eng.sgd_step(g[0]["lr"], g[1]["lr"], g[0]["momentum"], g[0]["weight_decay"])
        ts.sgd_steps += 1
        ts.fresh = False
        self._net.invalidate_engines(except_=eng)
        return loss

    def zero_grad(self, set_to_none=True):
        eng_any = False
        if self._net is not None and not self._net.autograd_grads:
            for eng in self._net._engines.values():
                ts = getattr(eng, "_ts", None)
                if ts is not None:
                    ts.lazy_zero, ts.fresh = True, False
                    eng_any = True
        if not eng_any:
            super().zero_grad(set_to_none=set_to_none)
            return
        # the engine's buckets are overwritten by the next backward (no memset of 1.4 GB); what a hook or a clip could still see through
        # .grad would be last step's values, so the references are dropped like torch's set_to_none (the backward re-attaches the views)
        for grp in self.param_groups:
            for p in grp["params"]:
                p.grad = None

    # ---- checkpoint compatibility with torch.optim.SGD ------------------------------------------------------------------------
    def _index_keys(self):
        """optimizer parameter index (torch's state_dict numbering) -> state-dict key of the network."""
        names = {id(p): k for k, p in self._net.named_parameters()}
        out, i = {}, 0
        for grp in self.param_groups:
            for p in grp["params"]:
                out[i] = names.get(id(p))
                i += 1
        return out

    def state_dict(self):
        sd = super().state_dict()
        eng = next((e for e in (self._net._engines.values() if self._net is not None else []) if getattr(e, "_ts", None) is not None
                    and e._ts.sgd_steps > 0), None)
        if eng is not None:
            for i, k in self._index_keys().items():
                if k in eng.grads:
                    sd["state"][i] = {"momentum_buffer": eng.get_momentum(k)}
            torch.cuda.current_stream(eng.device).synchronize()
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        keys = self._index_keys()
        mom = {keys[i]: st["momentum_buffer"] for i, st in state_dict.get("state", {}).items()
               if isinstance(st, dict) and st.get("momentum_buffer") is not None and keys.get(i)}
        self._pending_momentum = mom or None

    @staticmethod
    def _push_momentum(eng, mom):
        for k, v in mom.items():
            if k in eng.grads:
                eng.set_momentum(k, v)
        eng.mark_sgd_initialized(True)


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

    def _fused_criterion(self):
        """ignore_index when `self.criterion` is the plain cross-entropy the fused loss implements -- nn.CrossEntropyLoss / [3P] encoding
        SegmentationLosses(se_loss=False, aux=False), mean over pixels != ignore_index, no class weights, no label smoothing -- else None."""
        if not hasattr(self, "criterion"):
            self.criterion = self.get_criterion(**self.other_kwargs)
        c = self.criterion
        if not isinstance(c, nn.CrossEntropyLoss) or c.weight is not None or c.reduction != "mean" or getattr(c, "label_smoothing", 0.0) != 0.0:
            return None
        if getattr(c, "se_loss", False) or getattr(c, "aux", False):
            return None
        return int(c.ignore_index)

    def training_step(self, batch, batch_nb):                 # :66-81 (autocast/GradScaler are disabled there: self.enabled = False)
        img, target = batch
        ignore = self._fused_criterion()
        if ignore is not None and hasattr(self.net, "forward_loss") and not self.other_kwargs.get("materialize_logits", False):
            # `out = self(img); loss = self.criterion(out, target)` as ONE autograd node on the engine: no [B,K,H,W] logits, the loss
            # value and the train-accuracy counts come from the low-resolution logits through the x2 bilinear (lseg_train_loss)
            loss = self.net.forward_loss(img, target, ignore_index=ignore)
            counts = self.net._last_train_counts              # int64[2] on the device: correct, labeled (== _filter_invalid + Accuracy)
            self._train_counts = counts if getattr(self, "_train_counts", None) is None else self._train_counts + counts
            self.log("train_loss", loss)
            return loss
        out = self(img)                                       # train-mode engine forward; autograd node = lseg_backward(dlogits)
        multi_loss = isinstance(out, tuple)
        loss = self.criterion(*out, target) if multi_loss else self.criterion(out, target)
        final_output = out[0] if multi_loss else out
        train_pred, train_gt = self._filter_invalid(final_output, target)
        if train_gt.nelement() != 0 and hasattr(self, "train_accuracy"):
            self.train_accuracy(train_pred, train_gt)
        self.log("train_loss", loss)
        return loss

    def training_epoch_end(self, outs):                       # :83-84
        c = getattr(self, "_train_counts", None)
        if c is not None:                                     # fused path: Accuracy().compute() == correct / labeled over the epoch
            c = c.cpu()
            self.log("train_acc_epoch", float(c[0]) / max(1, int(c[1])))
            self._train_counts = None
        elif hasattr(self, "train_accuracy"):
            self.log("train_acc_epoch", self.train_accuracy.compute())

    def validation_step(self, batch, batch_nb):               # :86-105 -- loss, pixAcc, IoU counts from ONE engine pass, no logits
        img, target = batch
        ignore = self._fused_criterion()
        if ignore is None or not hasattr(self.net, "forward_metrics"):
            raise NotImplementedError("validation_step needs the plain cross-entropy criterion (se_loss / aux are never enabled by the reference's scripts)")
        r = self.net.forward_metrics(img, target, ignore_index=ignore)
        if not hasattr(self, "val_iou"):
            from lseg_hip.metrics import SegmentationMetric as DeviceMetric
            self.val_iou = DeviceMetric(self.nclass)
        self.val_iou._accumulate(r["correct"], r["labeled"], r["area_inter"], r["area_union"])
        pixAcc, iou = self.val_iou.get()
        self.log("val_loss_step", r["nll_sum"] / max(1, r["nll_count"]))
        self.log("pix_acc_step", pixAcc)
        self.log("val_acc_step", r["correct"] / max(1, r["labeled"]))
        self.log("val_iou", iou)

    def validation_epoch_end(self, outs):                     # :107-112
        if not hasattr(self, "val_iou"):
            return
        pixAcc, iou = self.val_iou.get()
        self.log("val_acc_epoch", pixAcc)
        self.log("val_iou_epoch", iou)
        self.log("pix_acc_epoch", pixAcc)
        self.val_iou.reset()

    def native_training_step(self, img, target, lr_scale=1.0):
        """One whole step without autograd or torch.optim in the loop: engine forward + fused upsample/CE/backward + bucketed gradient
        all-reduce (when torch.distributed is up) + fused SGD at base_lr * lr_scale.  Same engine, buckets and exchange as
        training_step -> loss.backward() -> EngineSGD.step(); returns the local mean cross-entropy (device tensor, no sync)."""
        net = self.net
        ignore = self.other_kwargs.get("ignore_index", -1)
        eng, keys, _ = net._train_inputs(img, "")
        eng.forward(img.float(), want_logits=False)
        ts = eng._ts
        try:
            loss = eng.backward(target=target.to(eng.device, torch.int64), ignore_index=ignore, accumulate=False)
            if ts.exchange is not None:
                ts.exchange.finish()
        except BaseException:
            if ts.exchange is not None:
                ts.exchange.abort()
            raise
        wd = self.other_kwargs.get("weight_decay", 1e-4)
        eng.sgd_step(self.base_lr * lr_scale, self.base_lr * 10 * lr_scale, 0.9, wd)
        ts.sgd_steps += 1
        ts.lazy_zero, ts.fresh = True, False
        net.invalidate_engines(except_=eng)
        return loss

    def _filter_invalid(self, pred, target):                  # :114-117
        valid = target != self.other_kwargs["ignore_index"]
        _, mx = torch.max(pred, dim=1)
        return mx[valid], target[valid]

    def configure_optimizers(self):                           # :119-175 (same param groups / poly LR)
        params_list = [{"params": self.net.pretrained.parameters(), "lr": self.base_lr}]
        if hasattr(self.net, "scratch"):
            params_list.append({"params": self.net.scratch.parameters(), "lr": self.base_lr * 10})
        # torch.optim.SGD whose step() is the engine's fused lseg_sgd_step when this step's gradients came from the engine
        opt = EngineSGD(params_list, net=self.net, lr=self.base_lr, momentum=0.9,
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
