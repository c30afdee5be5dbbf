"""Shared helpers of the training-step parity tests (tests/test_gpu_train.py, tests/test_zz_gpu_random_net_gradients.py,
tests/gpu_train_spread.py).  Test infrastructure: the oracle is the checker here, never the product."""
import torch

from lseg_hip.engine import HipEngine
from oracle.lseg_oracle import lseg_forward


def target_map(B, H, W, K, seed):                       # == oracle/make_ref_train_golden.synthetic_target
    g = torch.Generator().manual_seed(1000 + seed)
    t = torch.randint(0, K, (B, H, W), generator=g)
    t[torch.rand((B, H, W), generator=g) < 0.2] = -1
    return t


def engine_step(cfg, sd, x, target, tok, accumulate=False, eng=None, **engine_kw):
    B, _, H, W = x.shape
    sd_dev = {k: v.cuda() for k, v in sd.items()}
    if eng is None:
        eng = HipEngine(cfg, H, W, max_batch=B, max_labels=tok.shape[0], **engine_kw)
        eng.load_state_dict(sd_dev)
        eng.set_tokens(tok)
        eng.enable_training(sd_dev)
    out = eng.forward(x.cuda())
    loss = eng.backward(target=target.cuda(), ignore_index=-1, accumulate=accumulate)
    torch.cuda.synchronize()
    return eng, out, loss, sd_dev


def rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-20)).item()


def cosine(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-300))


def oracle_backward(sd, x, tok, cfg, dlogits):
    """oracle.lseg_forward in train mode under autograd with a GIVEN d(logits): isolates the backward arithmetic from the loss's
    sensitivity to the forward's rounding (softmax over logits scaled by 14.3: a 0.1 logit error moves a probability by ~10 %)."""
    bn_stats = ("running_mean", "running_var", "num_batches_tracked")
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()
              if v.is_floating_point() and not k.endswith(bn_stats) and not k.startswith("clip_pretrained.")}
    full = dict(sd)
    full.update(leaves)
    out = lseg_forward(full, x, tok, cfg, bn_train=True)
    out.backward(dlogits)
    return out.detach(), {k: v.grad for k, v in leaves.items() if v.grad is not None}


def away_from_the_relu_kinks(sd, cfg):
    """Shift the DPT head of the seeded net so that every ReLU input (layerN_rn outputs, fusion sums, bn1 outputs) is positive:
    a ReLU whose bf16 input has the other sign than the fp32 one contributes a full-magnitude gradient error, which on the
    zero-centred random net (1-4 % of the elements flip) hides everything else.  With the kinks out of reach the comparison
    measures the backward arithmetic and its wiring."""
    sd = dict(sd)
    for k in list(sd):
        if ".bn1.bias" in k or ".bn2.bias" in k or k.endswith("out_conv.bias"):
            sd[k] = sd[k] + 4.0
    for l in range(4):
        a = f"pretrained.act_postprocess{l + 1}."
        bk = a + ("4.bias" if a + "4.bias" in sd else "3.bias")
        sd[bk] = sd[bk] + 4.0
        wk = f"scratch.layer{l + 1}_rn.weight"
        sd[wk] = sd[wk] + 5.0 / (36.0 * cfg.reassemble[l])
    return sd
