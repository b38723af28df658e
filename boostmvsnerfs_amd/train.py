"""Fine-tuning step with the contract of the reference's trainer (lib/train/trainers/trainer.py:44-63,
lib/train/losses/enerf.py:16-56, lib/train/optimizer.py:12-28): loss = sum_i loss_weight[i] * MSE(rgb_level{i},
rgb_{i}) [+ 0.01 * VGG perceptual loss when torchvision VGG16 weights are available -- they are not
offline, so the perceptual term is off and reported as such], Adam(lr 5e-4, eps 1e-8), gradient value
clipping at 40, exponential LR decay (gamma 0.5 every 50 epochs).  Works under torch DDP
(gradient all-reduce over RCCL) unchanged: every op is a regular autograd Function."""
import torch
import torch.nn as nn

from .config import cfg


class NetworkWrapper(nn.Module):
    def __init__(self, net):
        super().__init__()
        self.net = net

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
            stats[f"mse_level{i}"] = mse.detach()
            stats[f"psnr_level{i}"] = -10.0 * torch.log10(mse.detach())
            loss = loss + cc.loss_weight[i] * mse
        stats["loss"] = loss.detach()
        return out, loss, stats


def make_optimizer(net, lr=5e-4, weight_decay=0.0, eps=1e-8):
    params = [{"params": [p], "lr": lr, "weight_decay": weight_decay} for _, p in net.named_parameters() if p.requires_grad]
    return torch.optim.Adam(params, lr=lr, weight_decay=weight_decay, eps=eps)


def make_lr_scheduler(optimizer, gamma=0.5, decay_epochs=50):
    return torch.optim.lr_scheduler.LambdaLR(optimizer, lambda epoch: gamma ** (epoch / decay_epochs))


def train_step(wrapper, optimizer, batch, clip=40.0):
    out, loss, stats = wrapper(batch)
    loss = loss.mean()
    optimizer.zero_grad()
    loss.backward()
    torch.nn.utils.clip_grad_value_(wrapper.parameters(), clip)
    optimizer.step()
    return loss.detach(), stats
