"""
    Batch-sharded multi-GPU inference: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm).

    The reference is a single-process library with no distributed code at all; inference on image i
    depends only on the weights, so the path shards by images with NO data-path collective between layers:
      1. once  : rank 0's weights are broadcast to every rank (flat buckets; ResNet-50 = 102 MB fp32 in 4 buckets - large
                 messages, because xGMI is point-to-point and ring collectives are per-link bound);
      2. / step: rank r runs images [r*B/P, (r+1)*B/P) of the global batch on its own GPU;
      3. / step: an all-gather (or gather to rank 0) of the fp32 logits, 4 KB per image.
    The compute callable is injected, so the same plumbing is exercised on CPU with gloo in the tests.
"""

__all__ = ['shard_range', 'broadcast_module_state', 'ShardedInference']

import torch
import torch.distributed as dist

_BUCKET_BYTES = 32 << 20


def shard_range(total: int, rank: int, world: int):
    """Contiguous, near-even split of `total` images: the first `total % world` ranks take one extra."""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def broadcast_module_state(module: torch.nn.Module, src: int = 0, group=None, bucket_bytes: int = _BUCKET_BYTES):
    """Make every rank's parameters and buffers equal to rank `src`'s, with a few large broadcasts per dtype."""
    by_dtype = {}
    for _, t in sorted(module.state_dict().items()):
        by_dtype.setdefault(t.dtype, []).append(t)
    n_msgs = 0
    for dtype, tensors in by_dtype.items():
        bucket, size = [], 0
        for t in tensors + [None]:
            if t is not None and (size == 0 or size + t.numel() * t.element_size() <= bucket_bytes):
                bucket.append(t)
                size += t.numel() * t.element_size()
                continue
            flat = torch.cat([b.detach().reshape(-1) for b in bucket])
            dist.broadcast(flat, src=src, group=group)
            off = 0
            with torch.no_grad():
                for b in bucket:
                    b.copy_(flat[off:off + b.numel()].view_as(b))
                    off += b.numel()
            n_msgs += 1
            bucket, size = ([t], t.numel() * t.element_size()) if t is not None else ([], 0)
    return n_msgs


class ShardedInference(object):
    """
    forward_fn: callable(x_local) -> logits_local [n_local, num_classes] (e.g. a pytorchcv_amd net on this rank's GPU).
    """
    def __init__(self, forward_fn, group=None):
        self.forward_fn = forward_fn
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def local_slice(self, global_batch: int):
        return shard_range(global_batch, self.rank, self.world)

    def run_local(self, x_local: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            return self.forward_fn(x_local)

    def gather_all(self, y_local: torch.Tensor) -> torch.Tensor:
        """Equal shards: one all-gather collective, issued even at world size 1 (so that a single-GPU rehearsal runs RCCL)."""
        out = torch.empty((self.world * y_local.shape[0],) + tuple(y_local.shape[1:]), dtype=y_local.dtype, device=y_local.device)
        dist.all_gather_into_tensor(out, y_local.contiguous(), group=self.group)
        return out

    def gather(self, y_local: torch.Tensor, global_batch: int | None = None) -> torch.Tensor:
        """All ranks receive the logits of the whole batch, in image order."""
        if self.world == 1:
            return y_local
        counts = [shard_range(global_batch, r, self.world) for r in range(self.world)] if global_batch is not None else None
        if counts is None or len({b - a for a, b in counts}) == 1:
            out = torch.empty((self.world * y_local.shape[0],) + tuple(y_local.shape[1:]), dtype=y_local.dtype,
                              device=y_local.device)
            dist.all_gather_into_tensor(out, y_local.contiguous(), group=self.group)
            return out
        # uneven shards: pad every rank's block to the largest one, one all-gather, then drop the padding rows
        biggest = max(b - a for a, b in counts)
        padded = torch.zeros((biggest,) + tuple(y_local.shape[1:]), dtype=y_local.dtype, device=y_local.device)
        padded[:y_local.shape[0]] = y_local
        out = torch.empty((self.world * biggest,) + tuple(y_local.shape[1:]), dtype=y_local.dtype, device=y_local.device)
        dist.all_gather_into_tensor(out, padded, group=self.group)
        return torch.cat([out[r * biggest:r * biggest + (b - a)] for r, (a, b) in enumerate(counts)])

    def __call__(self, x_global_or_local: torch.Tensor, already_sharded: bool = False) -> torch.Tensor:
        if already_sharded:
            return self.gather(self.run_local(x_global_or_local))
        n = x_global_or_local.shape[0]
        a, b = self.local_slice(n)
        return self.gather(self.run_local(x_global_or_local[a:b]), global_batch=n)
