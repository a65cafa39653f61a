"""
    Batch-sharded multi-GPU inference: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm).

    The reference is a single-process library with no distributed code at all; inference on image i
    depends only on the weights, so the path shards by images with NO data-path collective between layers:
      1. once  : rank 0's weights are broadcast to every rank in flat buckets of up to 32 MB (large messages, because xGMI is
                 point-to-point and ring collectives are per-link bound): either the module state (`broadcast_module_state`,
                 ResNet-50 = 102 MB fp32, every rank then packs for itself) or, for serving, what the kernels actually read -
                 the PACKED 16-bit weight arenas and the folded fp32 scale/shift arrays (`broadcast_packed_state`, ResNet-50 =
                 51 MB, nothing is re-packed on the receiving ranks);
      2. / step: rank r runs images [r*B/P, (r+1)*B/P) of the global batch on its own GPU;
      3. / step: an all-gather (or gather to rank 0) of the fp32 logits, 4 KB per image.
    The compute callable is injected, so the same plumbing is exercised on CPU with gloo in the tests.
"""

__all__ = ['shard_range', 'broadcast_module_state', 'broadcast_packed_state', 'packed_state_tensors', 'replicas_agree', 'mark_local_state_authoritative', 'ShardedInference']

import torch
import torch.distributed as dist

_BUCKET_BYTES = 32 << 20


def shard_range(total: int, rank: int, world: int):
    """Contiguous, near-even split of `total` images: the first `total % world` ranks take one extra."""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def _broadcast_buckets(tensors, src, group, bucket_bytes):
    """`tensors` (same order and shapes on every rank) become equal to rank `src`'s, a few large broadcasts per dtype."""
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    n_msgs = n_bytes = 0
    for dtype, group_tensors in by_dtype.items():
        bucket, size = [], 0
        for t in group_tensors + [None]:
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
            n_bytes += size
            bucket, size = ([t], t.numel() * t.element_size()) if t is not None else ([], 0)
    return n_msgs, n_bytes


def broadcast_module_state(module: torch.nn.Module, src: int = 0, group=None, bucket_bytes: int = _BUCKET_BYTES):
    """Make every rank's parameters and buffers equal to rank `src`'s, with a few large broadcasts per dtype. Runners that held
    packed state received from another rank (`broadcast_packed_state`) re-pack from the now-authoritative local tensors."""
    n = _broadcast_buckets([t for _, t in sorted(module.state_dict().items())], src, group, bucket_bytes)[0]
    for r in _runners(module):
        r.local_state_restored()
    return n


def mark_local_state_authoritative(module: torch.nn.Module):
    """After every rank has loaded the same fp32 state by itself (load_state_dict from the same file): forget that packed state
    was once received from another rank, re-pack from the local tensors on next use."""
    for r in _runners(module):
        r.local_state_restored()


def _runners_of(m):
    from . import engine
    for attr in sorted(vars(m)):
        if not attr.startswith("_pcv"):
            continue
        v = vars(m)[attr]
        items = [v] if not isinstance(v, (dict, list, tuple)) else (
            [v[k] for k in sorted(v, key=repr)] if isinstance(v, dict) else list(v))
        for r in items:
            if isinstance(r, (engine.ConvRunner, engine.BnActRunner)):
                yield r


def _runners(module: torch.nn.Module):
    """Every ConvRunner / BnActRunner hanging off `module`'s tree, in module order."""
    for _, m in sorted(module.named_modules()):
        for r in _runners_of(m):
            yield r


def packed_state_tensors(module: torch.nn.Module):
    """What the kernels read at inference time, in module order: (tensors, refresh hooks).
      * per convolution / linear layer: the packed weight blob in the compute type and the folded fp32 scale / shift
        (`engine.ConvRunner`), per stand-alone BatchNorm the folded scale / shift (`engine.BnActRunner`);
      * the fp32 source tensors of everything that is read UNPACKED at run time: modules without a runner (SE excitation
        matrices and biases), and convolutions whose weights are folded into an SE squeeze map (`ConvRunner.squeezed_excite`
        re-derives that map from the fp32 weights: its cache is dropped by the hook so that it is rebuilt from the received
        tensors).
    Every runner must already be built (one forward of the net on any input of the serving shape does that)."""
    from . import engine
    tensors, hooks = [], []
    owned = set()

    for _, m in sorted(module.named_modules()):
        for r in _runners_of(m):
            if r.scale is None or r.shift is None:
                raise RuntimeError("packed state requested before the first forward built every runner")
            if getattr(r, "packed", None) is not None:
                tensors.append(r.packed)
            tensors += [r.scale, r.shift]
            if getattr(r, "_sq_key", None) is not None:              # SE squeeze folded through this convolution: fp32 too
                hooks.append(lambda r=r: setattr(r, "_sq_key", None))
                continue
            src = r._sources() if isinstance(r, engine.ConvRunner) else (r.bn.weight, r.bn.bias, r.bn.running_mean, r.bn.running_var)
            owned.update(id(t) for t in src if t is not None)
    for _, t in sorted(module.state_dict(keep_vars=True).items()):
        if id(t) not in owned and t.dtype.is_floating_point:
            tensors.append(t.data)
    return tensors, hooks


def broadcast_packed_state(module: torch.nn.Module, src: int = 0, group=None, bucket_bytes: int = _BUCKET_BYTES):
    """Serving-time weight distribution: rank `src`'s packed inference state (see `packed_state_tensors`) overwrites every
    other rank's. Half the bytes of the fp32 state for a 16-bit net, and the receiving ranks do not pack. After it, the fp32
    parameters of convolutions on ranks != src are NOT those of `src` (only their packed form is), so the receiving runners are
    marked (`ConvRunner.adopt_foreign_state`): running them in a configuration that needs a re-pack - another dtype, channel pitch
    or padding parity, touched parameters - raises until `broadcast_module_state` (or a load_state_dict on every rank followed by
    it) has made the fp32 state authoritative again.
    Returns (messages, bytes)."""
    tensors, hooks = packed_state_tensors(module)
    out = _broadcast_buckets(tensors, src, group, bucket_bytes)
    for h in hooks:
        h()
    if (dist.get_rank(group) if dist.is_initialized() else 0) != src:
        # the receivers' packed state no longer derives from their own fp32 parameters: a later key change (another input shape
        # that flips a TF-"same" padding parity, another dtype or channel pitch) must not silently re-pack from them
        for r in _runners(module):
            r.adopt_foreign_state()
    return out


def replicas_agree(values: torch.Tensor, group=None) -> bool:
    """True when `values` (any small tensor: logits of a common input, a checksum) is identical on every rank: all-reduce MIN
    and MAX of its float64 image and compare. What bench.py runs after the weight broadcast, so that a tensor the broadcast
    missed shows up as an error instead of plausible logits on ranks != 0."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return True
    v = values.detach().double().reshape(-1)
    v = torch.nan_to_num(v, nan=1.0e300, posinf=1.0e301, neginf=-1.0e301)
    lo, hi = v.clone(), v.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    return bool(torch.equal(lo, hi))


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
