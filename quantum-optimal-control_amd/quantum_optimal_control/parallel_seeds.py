"""Seed-parallel outer loop: independent random-restart control sets sharded over the GPUs of one node.

The reference optimises exactly one control set per Grape() call and draws its initial guess from NumPy's global RNG
(core/system_parameters.py:272-284); random restarts are an embarrassingly-parallel loop around it.  Here the
restarts become the leading `n_seeds` dimension of one engine per GPU (one process per GPU), block-partitioned over
ranks.  Seeds never interact, so there is NO data-path collective: the only exchange is one all-gather of the
per-seed scalars (fidelity, iterations) at the end -- RCCL over xGMI when the process group backend is "nccl",
gloo in the CPU tests -- followed by an optional broadcast of the winner's controls.
"""
import numpy as np


class SeedShard(object):
    """Block partition of `total_seeds` over `world` ranks: rank r owns [first, first + count)."""

    def __init__(self, total_seeds, rank=0, world=1):
        if not (0 <= rank < world):
            raise ValueError('rank %d outside world %d' % (rank, world))
        self.total, self.rank, self.world = int(total_seeds), int(rank), int(world)
        base, extra = divmod(self.total, self.world)
        self.counts = [base + (1 if r < extra else 0) for r in range(self.world)]
        self.firsts = [int(np.sum(self.counts[:r])) for r in range(self.world)]
        self.first, self.count = self.firsts[self.rank], self.counts[self.rank]

    def _device(self, dist):
        import torch
        if dist is not None and dist.get_backend() == 'nccl':
            return torch.device('cuda', torch.cuda.current_device())
        return torch.device('cpu')

    def all_gather(self, local_values, dist=None):
        """Gather one float64 per seed from every rank, returned in global seed order (length total_seeds)."""
        local_values = np.asarray(local_values, dtype=np.float64).reshape(-1)
        assert local_values.shape[0] == self.count
        if dist is None or self.world == 1:
            return local_values.copy()
        import torch
        dev = self._device(dist)
        width = max(self.counts)
        buf = torch.zeros(width, dtype=torch.float64, device=dev)
        buf[:self.count] = torch.from_numpy(local_values).to(dev)
        out = [torch.zeros(width, dtype=torch.float64, device=dev) for _ in range(self.world)]
        dist.all_gather(out, buf)
        return np.concatenate([out[r][:self.counts[r]].cpu().numpy() for r in range(self.world)])

    def owner_of(self, seed):
        for r in range(self.world):
            if self.firsts[r] <= seed < self.firsts[r] + self.counts[r]:
                return r
        raise IndexError(seed)

    def broadcast_from_owner(self, seed, local_array_fn, shape, dist=None):
        """Broadcast a float64 array (e.g. the winner's uks) from the rank that owns `seed` to all ranks."""
        owner = self.owner_of(seed)
        if dist is None or self.world == 1:
            return np.asarray(local_array_fn(seed - self.first), dtype=np.float64).reshape(shape)
        import torch
        dev = self._device(dist)
        if self.rank == owner:
            t = torch.from_numpy(np.ascontiguousarray(local_array_fn(seed - self.first), dtype=np.float64)).reshape(shape).to(dev)
        else:
            t = torch.zeros(shape, dtype=torch.float64, device=dev)
        dist.broadcast(t, src=owner)
        return t.cpu().numpy()


def restart_guesses(k, steps, first_seed, count, base_seed=1000):
    """Initial optimisation variables N(0, 1/sqrt(steps)) (same law as system_parameters.py:279-282), one
    independent, reproducible stream per global seed index."""
    return np.stack([np.random.default_rng(base_seed + first_seed + i).normal(0, 1. / np.sqrt(steps), (k, steps))
                     for i in range(count)]) if count > 0 else np.zeros((0, k, steps))


def select_best(fidelities):
    """Index of the best seed (ties -> lowest index, deterministic on every rank)."""
    return int(np.argmax(np.asarray(fidelities)))
