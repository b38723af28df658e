"""Fine-tuning step with the contract of the reference's trainer (lib/train/trainers/trainer.py:44-63,
lib/train/losses/enerf.py:16-56, lib/train/optimizer.py:12-28): loss = sum_i loss_weight[i] * MSE(rgb_level{i},
rgb_{i}) [+ 0.01 * VGG perceptual loss when torchvision VGG16 weights are available -- they are not
offline, so the perceptual term is off and reported as such], Adam(lr 5e-4, eps 1e-8), gradient value
clipping at 40, exponential LR decay (gamma 0.5 every 50 epochs).  Works under torch DDP
(gradient all-reduce over RCCL) unchanged: every op is a regular autograd Function."""
import sys

import torch
import torch.nn as nn
from torch.optim._functional import adam as _adam_update

from . import convnet
from .config import cfg


class NetworkWrapper(nn.Module):
    """lib/train/losses/enerf.py:7-56: returns (output, loss, scalar_stats, image_stats) like the reference's wrapper,
    with the reference's stat names (color_mse_i, psnr_i, perceptual_loss_i, loss).

    `perceptual`: a callable (pred (n,3,h,w), target (n,3,h,w)) -> scalar standing in for the reference's
    VGGPerceptualLoss (torchvision VGG16 weights, not available offline).  With `perceptual=None` the 0.01 *
    perceptual term of the levels that ask for it (train_img / num_patchs > 0) is OFF; that is logged once and
    reported in every step's stats as `perceptual_loss_i = nan` -- the loss then differs from the reference's by
    exactly that term."""

    def __init__(self, net, train_loader=None, perceptual=None):
        super().__init__()
        self.net = net
        self.perceptual = perceptual
        self._warned = False

    def _perceptual_inputs(self, batch, out, i):
        cc = cfg.enerf.cas_config
        if cc.train_img[i]:
            B, S, C, H, W = batch["src_inps"].shape
            rs = cc.render_scale[i]
            h, w = int(H * rs), int(W * rs)
            return (out[f"rgb_level{i}"].reshape(B, h, w, 3).permute(0, 3, 1, 2),
                    batch[f"rgb_{i}"].reshape(B, h, w, 3).permute(0, 3, 1, 2))
        ps, nr, npatch = cc.patch_size[i], cc.num_rays[i], cc.num_patchs[i]
        cut = slice(nr, nr + npatch * ps * ps)
        return (out[f"rgb_level{i}"][:, cut].reshape(-1, ps, ps, 3).permute(0, 3, 1, 2),
                batch[f"rgb_{i}"][:, cut].reshape(-1, ps, ps, 3).permute(0, 3, 1, 2))

    def forward(self, batch):
        out = self.net(batch)
        cc = cfg.enerf.cas_config
        loss = 0
        stats = {}
        for i in range(cc.num):
            key = f"rgb_level{i}"
            if key not in out:
                continue
            mse = ((out[key] - batch[f"rgb_{i}"]) ** 2).mean()
            stats[f"color_mse_{i}"] = mse.detach()
            stats[f"psnr_{i}"] = -10.0 * torch.log10(mse.detach())
            loss = loss + cc.loss_weight[i] * mse
            wants = bool(cc.get("train_img", [False] * cc.num)[i]) or cc.get("num_patchs", [0] * cc.num)[i] > 0
            if wants and self.training:
                if self.perceptual is not None:
                    p = self.perceptual(*self._perceptual_inputs(batch, out, i))
                    loss = loss + 0.01 * p * cc.loss_weight[i]
                    stats[f"perceptual_loss_{i}"] = p.detach()
                else:
                    if not self._warned:
                        print(f"[train] level {i}: the 0.01 * VGG perceptual term of lib/train/losses/enerf.py:33-52 is OFF "
                              "(no VGG16 weights offline; pass NetworkWrapper(..., perceptual=fn) to enable it)", file=sys.stderr)
                        self._warned = True
                    stats[f"perceptual_loss_{i}"] = torch.full((), float("nan"), device=mse.device)
        stats["loss"] = loss.detach() if torch.is_tensor(loss) else loss
        return out, loss, stats, {}


class GroupedAdam(torch.optim.Adam):
    """torch.optim.Adam with the reference's one-param-group-per-parameter layout (lib/train/optimizer.py:18-21,
    so optimizer state dicts stay interchangeable with its checkpoints), stepping all groups that share their
    hyper-parameters in ONE multi-tensor update instead of one per group: 115 groups x ~9 foreach launches per step
    (5 ms of 3 us kernels and 15 ms of host time at 512x640) become ~9 launches.  The arithmetic is torch's own
    `torch.optim._functional.adam`, called once per bucket with the concatenated tensor lists: bit-identical results."""

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        buckets = {}
        for group in self.param_groups:
            key = (group["lr"], tuple(group["betas"]), group["eps"], group["weight_decay"], group["amsgrad"],
                   group["maximize"], group["foreach"], group["capturable"], group["differentiable"], group["fused"],
                   group["decoupled_weight_decay"])
            if any(torch.is_tensor(k) for k in key[:4]):       # tensor hyper-parameters: leave to torch
                return super().step()
            b = buckets.setdefault(key, (group, [], [], [], [], [], [], [False]))
            b[7][0] |= bool(self._init_group(group, *b[1:7]))
        for group, params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps, cplx in buckets.values():
            if not params:
                continue
            beta1, beta2 = group["betas"]
            _adam_update(params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps, amsgrad=group["amsgrad"],
                                  has_complex=cplx[0], beta1=beta1, beta2=beta2, lr=group["lr"],
                                  weight_decay=group["weight_decay"], eps=group["eps"], maximize=group["maximize"],
                                  foreach=group["foreach"], capturable=group["capturable"],
                                  differentiable=group["differentiable"], fused=group["fused"],
                                  grad_scale=getattr(self, "grad_scale", None), found_inf=getattr(self, "found_inf", None),
                                  decoupled_weight_decay=group["decoupled_weight_decay"])
        return loss


def make_optimizer(net, lr=5e-4, weight_decay=0.0, eps=1e-8):
    """lib/train/optimizer.py:12-28 with cfg.train.optim = 'adam' (every shipped config): one param group per
    parameter, in named_parameters() order."""
    params = [{"params": [p], "lr": lr, "weight_decay": weight_decay, "eps": eps} for _, p in net.named_parameters() if p.requires_grad]
    return GroupedAdam(params, lr=lr, weight_decay=weight_decay, eps=eps)


def make_lr_scheduler(optimizer, gamma=0.5, decay_epochs=50):
    return torch.optim.lr_scheduler.LambdaLR(optimizer, lambda epoch: gamma ** (epoch / decay_epochs))


def train_step(wrapper, optimizer, batch, clip=40.0):
    out, loss, stats, _ = wrapper(batch)
    loss = loss.mean()
    optimizer.zero_grad()
    loss.backward()
    torch.nn.utils.clip_grad_value_(wrapper.parameters(), clip)
    optimizer.step()
    return loss.detach(), stats


class GraphedTrainStep:
    """`train_step` with forward + loss + backward of a step replayed as ONE HIP graph (trainer.py:44-63 is then three
    host calls per step instead of ~900 eager launches; gradient clipping and the optimiser stay eager: Adam's step
    counters live on the host).

    Every kernel of the training path goes to the current stream and nothing on it reads the device from the host, so the
    step captures unmodified.  A captured step is specialised to the shapes of the batch's tensors and -- for the
    K-volume networks -- to the targets named in batch['meta'] (the triplets view_selection.json picks for them are
    baked into the launches; other networks' batches are keyed by shapes alone); a batch with another key is captured
    separately (up to `max_graphs`, each replay binds ITS gradient tensors to `p.grad` before clip + Adam), the first
    `eager_steps` steps with a key run eagerly (allocator pools, packed weights, MIOpen's solver picks settle there).
    Batches that arrive as new tensors are copied into the captured buffers.  Parameters, batch-norm statistics and
    gradients are updated in place, so the graph always reads the current ones.  Not under DDP (the bucketed all-reduce
    hooks are host callbacks): a DistributedDataParallel wrapper falls back to `train_step`.

    returns (loss, stats) like train_step; both are the graph's static tensors (valid until the next step)."""

    def __init__(self, wrapper, optimizer, clip=40.0, eager_steps=3, max_graphs=4):
        self.wrapper, self.optimizer, self.clip = wrapper, optimizer, clip
        self.eager_steps, self.max_graphs = eager_steps, max_graphs
        self.entries, self.seen = {}, {}
        self.stats = {"eager": 0, "captures": 0, "replays": 0, "copies": 0}
        self.disabled = isinstance(wrapper, nn.parallel.DistributedDataParallel)

    def _key(self, batch):
        # the targets named in batch['meta'] select cost-volume triplets that are baked into the captured launches -- of
        # the K-volume networks only (view_selection_outputs); plain ENeRF batches of one shape share one graph
        targets = ()
        net = getattr(self.wrapper, "net", None)
        if getattr(net, "view_selection_outputs", None) is not None:
            meta = batch.get("meta") or {}
            targets = tuple(f"{s}_{v}" for s, v in zip(meta.get("scene", ()), meta.get("tar_view", ())))
        shapes = tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(batch.items())
                       if torch.is_tensor(v) and not getattr(v, "_bmv_built_rays", False))
        return shapes, targets

    def _finish(self):
        torch.nn.utils.clip_grad_value_(self.wrapper.parameters(), self.clip)
        self.optimizer.step()

    def _fwd_bwd(self, batch):
        out, loss, stats, _ = self.wrapper(batch)
        loss = loss.mean()
        loss.backward()
        return loss.detach(), stats

    def _capture(self, key, batch):
        import gc
        if len(self.entries) >= self.max_graphs:
            del self.entries[min(self.entries, key=lambda k: self.entries[k]["hits"])]
        static = {k: v for k, v in batch.items() if not getattr(v, "_bmv_built_rays", False)}
        # gradients become allocations of the graph's private pool: every replay writes the same tensors, the optimiser
        # reads them after the replay
        self.optimizer.zero_grad(set_to_none=True)
        gc.collect()
        from . import ktimer
        graph = torch.cuda.CUDAGraph()
        was, ktimer.enabled = ktimer.enabled, False       # per-kernel event brackets are host-side objects: not in a graph
        convnet.clear_pack_cache()       # (weight packs cached by an earlier capture live in THAT graph's pool)
        try:
            with torch.cuda.graph(graph):
                loss, stats = self._fwd_bwd(dict(static))
        finally:
            ktimer.enabled = was
            convnet.clear_pack_cache()   # ... and this capture's must not be handed to an eager step or another capture
        # the gradient tensors this graph writes (allocations of ITS pool).  Any other step -- an eager one, another key's
        # graph -- rebinds p.grad (zero_grad(set_to_none=True)): every replay binds these back before clip + Adam read
        # p.grad, and un-binds the parameters this graph leaves without a gradient
        params = [p for p in self.wrapper.parameters()]
        e = {"graph": graph, "static": static, "loss": loss, "stats": stats, "hits": 0,
             "grads": [(p, p.grad) for p in params]}
        self.entries[key] = e
        self.stats["captures"] += 1
        return e

    def __call__(self, batch):
        if self.disabled or not self.wrapper.training:
            self.stats["eager"] += 1
            return train_step(self.wrapper, self.optimizer, batch, self.clip)
        key = self._key(batch)
        e = self.entries.get(key)
        if e is None:
            n = self.seen.get(key, 0)
            if len(self.seen) > 64:
                self.seen.clear()
            self.seen[key] = n + 1
            if n < self.eager_steps:
                self.stats["eager"] += 1
                # (set_to_none: a later capture must not find gradients allocated outside its pool)
                self.optimizer.zero_grad(set_to_none=True)
                res = self._fwd_bwd(batch)
                self._finish()
                return res
            e = self._capture(key, batch)
        else:
            for k, v in batch.items():
                s = e["static"].get(k)
                if torch.is_tensor(v) and s is not None and s is not v and not getattr(v, "_bmv_built_rays", False) \
                        and not (s.data_ptr() == v.data_ptr() and s.stride() == v.stride()):
                    s.copy_(v)
                    self.stats["copies"] += 1
        e["graph"].replay()
        convnet.clear_pack_cache()       # the replay stepped the parameters without bumping their version counters
        for p, g in e["grads"]:
            p.grad = g
        e["hits"] += 1
        self.stats["replays"] += 1
        self._finish()
        return e["loss"], e["stats"]
