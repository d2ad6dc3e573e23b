"""PolyWarmupAdamW -- counterpart of the reference's utils/optimizer.py:3-33 (AdamW with a linear-warm-up / poly-decay LR
recomputed inside step()).  Host-side schedule logic; the parameter update itself is torch.optim.AdamW (the training step,
BASELINE config 5, is not on the round-1 path)."""
import torch


class PolyWarmupAdamW(torch.optim.AdamW):
    def __init__(self, params, lr, weight_decay, betas, warmup_iter=None, max_iter=None, warmup_ratio=None, power=None):
        super().__init__(params, lr=lr, betas=betas, weight_decay=weight_decay, eps=1e-8)
        self.global_step = 0
        self.warmup_iter = warmup_iter
        self.warmup_ratio = warmup_ratio
        self.max_iter = max_iter
        self.power = power
        self.__init_lr = [group['lr'] for group in self.param_groups]

    def lr_mult(self):
        if self.global_step < self.warmup_iter:
            return 1 - (1 - self.global_step / self.warmup_iter) * (1 - self.warmup_ratio)
        if self.global_step < self.max_iter:
            return (1 - self.global_step / self.max_iter) ** self.power
        return None

    def step(self, closure=None):
        m = self.lr_mult()
        if m is not None:
            for i in range(len(self.param_groups)):
                self.param_groups[i]['lr'] = self.__init_lr[i] * m
        super().step(closure)
        self.global_step += 1
