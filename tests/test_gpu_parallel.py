"""GPU, world size 1, RCCL: the multi-GPU code path of bench.py / pytorchcv_amd.parallel rehearsed on the one GPU a test box has -
process group on backend "nccl" (= RCCL), weight broadcast (module state and packed state), hipGraph forward, logits all-gather.
Nothing here measures scaling (no multi-GPU node is available to the tests); it proves the collectives are issued on the
right tensors and change nothing at world size 1, and that the packed broadcast really carries what the kernels read."""

import os
import pytest
import torch
import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl_group(cuda_device):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 200))
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=cuda_device)
    yield dist
    dist.destroy_process_group()


def _net(name, dev):
    import pytorchcv_amd
    from pytorchcv_amd.model_provider import get_model
    net = get_model(name).eval()
    net.load_state_dict(util.model_state(name, net.state_dict()), strict=True)
    return pytorchcv_amd.set_compute_dtype(net.to(dev), "bf16")


def test_rccl_rehearsal_broadcast_graph_allgather(rccl_group, cuda_device):
    from pytorchcv_amd.parallel import ShardedInference, broadcast_module_state
    from pytorchcv_amd.graph import capture
    net = _net("resnet18", cuda_device)
    _, ids = util.model_golden("resnet18")
    x = util.images(ids).to(cuda_device).repeat(16, 1, 1, 1).contiguous()            # 64 images: two graph lanes
    with torch.no_grad():
        y_local = net(x).clone()
        assert broadcast_module_state(net, src=0) >= 1                                # RCCL broadcast of the fp32 state
        g = capture(net, x)                                                           # re-packs nothing: same tensors
        runner = ShardedInference(g)
        y = runner.gather_all(runner.run_local(x))                                    # all_gather_into_tensor on RCCL
    torch.cuda.synchronize()
    assert y.shape == y_local.shape and torch.equal(y, y_local)
    assert torch.equal(runner(x), y_local)                                            # shard + run + gather in one call


@pytest.mark.parametrize("name", ["resnet50", "seresnet50", "preresnet18"])
def test_packed_state_broadcast_carries_what_the_kernels_read(name, rccl_group, cuda_device):
    """broadcast_packed_state: the packed arenas + folded constants (+ fp32 tensors read unpacked: SE matrices) are the
    complete inference state. A second net whose fp32 convolution weights are garbage gives the first net's logits once it
    holds the first net's packed state; the bytes on the wire are about half the fp32 state for a bf16 net."""
    from pytorchcv_amd import parallel
    import pytorchcv_amd
    from pytorchcv_amd.model_provider import get_model
    a = _net(name, cuda_device)
    b = pytorchcv_amd.set_compute_dtype(get_model(name).eval().to(cuda_device), "bf16")  # random init: different weights
    _, ids = util.model_golden(name)
    x = util.images(ids).to(cuda_device)
    with torch.no_grad():
        ya = a(x).clone()
        yb0 = b(x).clone()                                                            # builds b's runners (packs garbage)
    assert not torch.equal(ya, yb0)
    ta, _ = parallel.packed_state_tensors(a)
    tb, hooks = parallel.packed_state_tensors(b)
    assert [(t.shape, t.dtype) for t in ta] == [(t.shape, t.dtype) for t in tb]
    with torch.no_grad():
        for s, d in zip(ta, tb):                                                      # what a broadcast from a's rank would deliver
            d.copy_(s)
        for h in hooks:
            h()
        yb = b(x).clone()
    assert torch.equal(ya, yb)
    msgs, nbytes = parallel.broadcast_packed_state(b, src=0)                          # and the collective itself, on RCCL
    fp32_state = sum(t.numel() * t.element_size() for t in a.state_dict().values())
    assert msgs >= 1 and nbytes == sum(t.numel() * t.element_size() for t in tb)
    if name == "resnet50":
        # 51 MB of 16-bit packed state + 28 MB of second copies in MFMA-fragment order for d3i / d1i (the 256- / 512-channel 3x3 and the
        # 1024- / 2048-channel 1x1 layers) vs 102 MB of fp32
        assert 0.45 * fp32_state < nbytes < 0.85 * fp32_state, (nbytes, fp32_state)
    with torch.no_grad():
        assert torch.equal(b(x), ya)


def test_c_abi_rccl_helpers_world_size_1(cuda_device):
    """pcv_rccl_broadcast / pcv_rccl_allgather (include/pcv_amd.h: the two collectives of the batch-sharded path for a C caller) on
    a 1-rank communicator made with RCCL's own C API - the library resolves ncclBroadcast / ncclAllGather from the librccl this
    process has. At world size 1 both are identities: what is checked is that the entry points reach RCCL with the right
    arguments (a grouped broadcast of several buffers, byte counts), return 0, and leave the data intact; bad arguments are
    refused before RCCL is called."""
    import ctypes
    from pytorchcv_amd import _lib
    rccl = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]
    uid, comm = UniqueId(), ctypes.c_void_p()
    rccl.ncclGetUniqueId.argtypes = [ctypes.POINTER(UniqueId)]
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    torch.cuda.set_device(cuda_device)
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0 and comm.value
    try:
        L, ctx = _lib.lib(), _lib.ctx_for(0)
        assert L.pcv_rccl_available() == 1
        st = ctypes.c_void_p(torch.cuda.current_stream(cuda_device).cuda_stream)
        g = torch.Generator().manual_seed(3)
        bufs = [torch.randint(0, 255, (n,), generator=g, dtype=torch.uint8).to(cuda_device) for n in (4096, 17, 1 << 20)] + \
               [torch.randn(1000, generator=g).to(cuda_device)]
        want = [b.clone() for b in bufs]
        ptrs = (ctypes.c_void_p * len(bufs))(*[b.data_ptr() for b in bufs])
        sizes = (ctypes.c_size_t * len(bufs))(*[b.numel() * b.element_size() for b in bufs])
        assert L.pcv_rccl_broadcast(ctx, comm, ptrs, sizes, len(bufs), 0, st) == 0, L.pcv_last_error(ctx)
        y = torch.randn((4, 1000), generator=g).to(cuda_device)
        out = torch.empty_like(y)
        assert L.pcv_rccl_allgather(ctx, comm, ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(out.data_ptr()), y.numel() * 4, st) == 0, \
            L.pcv_last_error(ctx)
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(bufs, want)) and torch.equal(out, y)
        assert L.pcv_rccl_broadcast(ctx, None, ptrs, sizes, len(bufs), 0, st) == -1
        assert L.pcv_rccl_broadcast(ctx, comm, ptrs, sizes, 0, 0, st) == -1
        assert L.pcv_rccl_allgather(ctx, comm, None, ctypes.c_void_p(out.data_ptr()), 16, st) == -1
    finally:
        rccl.ncclCommDestroy(comm)


def test_bench_two_ranks_share_this_gpu_end_to_end(cuda_device):
    """`python bench.py --gpus 2` as the driver's multi-GPU tier runs it - a plain process that starts its own ranks - with the REAL
    product path in both ranks (PCV_BENCH_REHEARSE=1: both on cuda:0, the collectives through gloo, because a test box has one GPU):
    rank 0 loads the weights, rank 1 keeps its random init and must receive the packed arenas; the replica check (exit 5) and the
    row-by-row parity check (exit 3) are live. One JSON line, `n_gpus` 2, marked as a rehearsal. Three processes on the card."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(PCV_BENCH_REHEARSE="1", PYTHONPATH=root)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "resnet18_bs256", "--batch", "32",
                        "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--inflight", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 64 and "rehearsal" in out
    assert out["config"]["launch"].startswith("2 hipGraphs in flight")      # the all-gather of every step rides on its slot's stream
    wb = out["weights_broadcast"]
    assert wb and wb["messages"] >= 1 and wb["bytes"] > 1e6 and wb["replicas_agree"] is True and wb["seconds"] > 0
    assert out["config"]["launcher_choice"] == "flags"


def test_bench_two_gpus_real_rccl(cuda_device):
    """`python bench.py --gpus 2` with one GPU per rank and RCCL (backend "nccl") for the weight broadcast, the replica check and the
    per-step all-gather - the driver's multi-GPU tier at its smallest size. Runs where the box has two GPUs; skipped on the one-GPU
    test boxes (the same path runs there as a rehearsal, above). Weak and strong (fixed global batch) scaling lines."""
    import json
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (one rank per GPU over RCCL / xGMI)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "PCV_BENCH_REHEARSE")}
    env.update(PYTHONPATH=root, PCV_BENCH_TIMEOUT_S="500")
    for extra, batch in ((["--batch", "32"], 64), (["--scaling", "strong", "--global-batch", "64"], 64)):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "resnet18_bs256", "--steps", "3",
                            "--warmup", "1", "--no-cpu-baseline"] + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
        assert out["n_gpus"] == 2 and out["config"]["global_batch"] == batch and "rehearsal" not in out
        assert out["config"]["launcher_choice"].startswith("fixed for multi-rank")
        assert out["weights_broadcast"]["replicas_agree"] is True and out["weights_broadcast"]["seconds"] > 0
