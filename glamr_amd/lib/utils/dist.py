"""Diagonal Gaussian of the two CVAEs -- the object `forward(data)` leaves under data['q_z_dist'] / data['p_z_dist']
(reference interface: lib/utils/dist.py:8-52, class Normal).  The parameters are produced by the HIP networks; this holder only
carries them with the reference's attribute and method names (mu, logvar, sigma, rsample / sample, mode, kl)."""
import torch


class Normal:
    def __init__(self, mu=None, logvar=None, params=None):
        if params is not None:
            mu, logvar = torch.chunk(params, chunks=2, dim=-1)
        if mu is None or logvar is None:
            raise ValueError('Normal needs mu and logvar (or params)')
        self.mu, self.logvar = mu, logvar
        self.sigma = torch.exp(0.5 * logvar)

    def rsample(self, eps=None):
        eps = torch.randn_like(self.sigma) if eps is None else eps
        return self.mu + eps * self.sigma

    def sample(self, eps=None):
        return self.rsample(eps)

    def mode(self):
        return self.mu

    def kl(self, p=None):
        """KL(self || p); p = None means the standard normal (dist.py:27-35)."""
        if p is None:
            return -0.5 * (1 + self.logvar - self.mu.pow(2) - self.logvar.exp())
        a = (self.mu - p.mu) / (p.sigma + 1e-8)
        b = self.sigma / (p.sigma + 1e-8)
        return 0.5 * (a * a + b * b) - 0.5 - torch.log(b)

    @classmethod
    def stack(cls, arr, dim=0):
        return cls(torch.stack([x.mu for x in arr], dim=dim), torch.stack([x.logvar for x in arr], dim=dim))

    @classmethod
    def cat(cls, arr, dim=0):
        return cls(torch.cat([x.mu for x in arr], dim=dim), torch.cat([x.logvar for x in arr], dim=dim))
