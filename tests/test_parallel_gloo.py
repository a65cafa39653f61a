"""CPU, world_size 2, gloo: the batch-sharded multi-GPU plumbing (weight broadcast, contiguous image shards, logits
all-gather) gives exactly the single-process result. The compute callable is the oracle's CPU forward (the product's HIP
forward needs a GPU); what is under test is pytorchcv_amd.parallel."""

import os
import sys
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir, uneven):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from pytorchcv_amd.model_provider import get_model
    from pytorchcv_amd.parallel import ShardedInference, broadcast_module_state, shard_range
    from pytorchcv_amd.synth import synth_state_dict, synth_input
    from oracle import refnet
    net = get_model("resnet10").eval()
    if rank == 0:
        net.load_state_dict(synth_state_dict(net.state_dict(), seed=5))      # other ranks keep their random init
    n_msgs = broadcast_module_state(net, src=0)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    batch = 5 if uneven else 4
    x = synth_input(batch, 3, 64, 64, seed=9)

    def fwd(xl):
        # resnet10 body on a 64x64 input would not reach 7x7; use the conv trunk digest as "logits"
        taps = {}
        try:
            refnet.resnet_forward(sd, xl, blocks=10, taps=taps)
        except RuntimeError:
            pass
        return taps["stage4"].mean(dim=(2, 3))

    runner = ShardedInference(fwd)
    y = runner(x)
    a, b = shard_range(batch, rank, world)
    torch.save(dict(y=y, lo=a, hi=b, msgs=n_msgs, sd_sum=float(sum(v.double().sum() for v in sd.values()))),
               os.path.join(out_dir, "r{}.pt".format(rank)))
    dist.destroy_process_group()


@pytest.mark.parametrize("uneven", [False, True])
def test_sharded_inference_equals_single_process(tmp_path, uneven):
    sys.path.insert(0, ROOT)
    world, port = 2, 29500 + (os.getpid() % 2000) + (1 if uneven else 0)
    mp.spawn(_worker, args=(world, port, str(tmp_path), uneven), nprocs=world, join=True)
    r0 = torch.load(os.path.join(str(tmp_path), "r0.pt"))
    r1 = torch.load(os.path.join(str(tmp_path), "r1.pt"))
    assert torch.equal(r0["y"], r1["y"])                       # every rank holds the whole batch, in image order
    assert r0["sd_sum"] == r1["sd_sum"]                        # broadcast made the replicas identical
    assert r0["msgs"] >= 1
    assert (r0["lo"], r0["hi"], r1["lo"], r1["hi"]) == ((0, 3, 3, 5) if uneven else (0, 2, 2, 4))
    # single-process reference
    from pytorchcv_amd.model_provider import get_model
    from pytorchcv_amd.synth import synth_state_dict, synth_input
    from oracle import refnet
    net = get_model("resnet10").eval()
    sd = synth_state_dict(net.state_dict(), seed=5)
    x = synth_input(5 if uneven else 4, 3, 64, 64, seed=9)
    taps = {}
    try:
        refnet.resnet_forward(sd, x, blocks=10, taps=taps)
    except RuntimeError:
        pass
    ref = taps["stage4"].mean(dim=(2, 3))
    assert r0["y"].shape == ref.shape
    assert float((r0["y"] - ref).abs().max()) <= 1e-5          # per-image results do not depend on the sharding


def test_shard_range_partitions_exactly():
    from pytorchcv_amd.parallel import shard_range
    for total in (0, 1, 7, 8, 2048, 2050):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _bucket_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pytorchcv_amd import parallel
    g = torch.Generator().manual_seed(100 + rank)                 # every rank starts with different contents
    tensors = [torch.randn(n, generator=g) for n in (7, 1000, 3, 50000)] + \
              [torch.randint(0, 255, (n,), generator=g, dtype=torch.uint8) for n in (17, 40000, 5)]
    msgs, nbytes = parallel._broadcast_buckets(tensors, 0, None, bucket_bytes=64 << 10)      # 64 KB buckets: several messages
    torch.save(dict(tensors=tensors, msgs=msgs, nbytes=nbytes), os.path.join(out_dir, "b{}.pt".format(rank)))
    dist.destroy_process_group()


def test_bucketed_broadcast_mixed_dtypes(tmp_path):
    """The flat-bucket broadcast behind broadcast_module_state / broadcast_packed_state: mixed dtypes (the packed arenas are
    uint8 blobs, the folded BN constants fp32), tensors larger than a bucket, several messages - every rank ends up with rank
    0's contents and the byte count is the sum of the tensors."""
    world, port = 2, 31500 + (os.getpid() % 2000)
    mp.spawn(_bucket_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(str(tmp_path), "b0.pt"))
    r1 = torch.load(os.path.join(str(tmp_path), "b1.pt"))
    g = torch.Generator().manual_seed(100)
    want = [torch.randn(n, generator=g) for n in (7, 1000, 3, 50000)] + \
           [torch.randint(0, 255, (n,), generator=g, dtype=torch.uint8) for n in (17, 40000, 5)]
    for a, b, w in zip(r0["tensors"], r1["tensors"], want):
        assert torch.equal(a, w) and torch.equal(b, w)
    assert r0["nbytes"] == r1["nbytes"] == sum(t.numel() * t.element_size() for t in want)
    assert r0["msgs"] == r1["msgs"] == 3        # fp32: {7, 1000, 3} + {50000 alone: larger than a bucket}; uint8: one bucket
