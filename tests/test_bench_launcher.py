"""CPU: `bench.py --gpus N` starting its own ranks (no torch.distributed.run wrapper), rehearsed at world size 2 with gloo and a
stand-in forward (PCV_BENCH_STUB=1: the product path has no CPU form and is not what is under test). What is checked is the
contract of the launcher path: the parent relays exactly one JSON line from rank 0, n_gpus is the process group's world size,
the weight broadcast ran and the replicas agreed, a failing rank fails the whole run with ITS exit code; and that a runner
whose packed state arrived by broadcast refuses to re-pack from its own (different) fp32 parameters."""

import os
import sys
import json
import subprocess
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(extra_env, *argv, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PCV_BENCH_STUB="1", OMP_NUM_THREADS="2")
    env.update(extra_env)
    return subprocess.run([sys.executable, BENCH] + list(argv), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                          timeout=timeout)


def test_self_launch_world_size_2():
    p = _run({}, "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "16")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "the parent relays rank 0's ONE JSON line: {}".format(lines)
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 32 and out["config"]["per_gpu_batch"] == 16
    assert out["value"] > 0 and out["ms_per_step"] > 0
    wb = out["weights_broadcast"]
    assert wb["messages"] >= 1 and wb["bytes"] == (3 * 8 * 8 * 1000 + 1000) * 4 and wb["replicas_agree"] is True
    assert "NOT a measurement" in out["stub"] and out["roofline"] is None and out["cpu_baseline"] is None


def test_strong_scaling_splits_a_fixed_global_batch():
    """--scaling strong: BASELINE config 5 literally (a fixed global batch over the ranks); the line says so."""
    p = _run({}, "--gpus", "2", "--steps", "2", "--warmup", "1", "--scaling", "strong", "--global-batch", "24")
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["scaling"] == "strong" and out["config"]["global_batch"] == 24 and out["config"]["per_gpu_batch"] == 12
    assert out["weights_broadcast"]["seconds"] >= 0
    p = _run({}, "--gpus", "2", "--steps", "2", "--warmup", "1", "--scaling", "strong", "--global-batch", "25")
    assert p.returncode == 2 and "does not split" in p.stderr


def test_stuck_ranks_are_terminated_after_the_wall_clock_limit():
    """a rank that never returns (RCCL init, a collective) must not hang the caller: PCV_BENCH_TIMEOUT_S."""
    p = _run({"PCV_BENCH_STUB_HANG_RANK": "1", "PCV_BENCH_TIMEOUT_S": "20"}, "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8",
             timeout=120)
    assert p.returncode == 124, (p.returncode, p.stderr[-1000:])
    assert "PCV_BENCH_TIMEOUT_S" in p.stderr and p.stdout.strip() == ""


def test_launcher_counts_gpus_without_the_runtime(monkeypatch):
    """The launcher parent stays GPU-free: devices come from the visibility variables or the KFD topology, never from torch.cuda."""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")
    assert bench.visible_gpus() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpus() == 0
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES", raising=False)
    n = bench.visible_gpus()
    assert n is None or n >= 0                                   # (no KFD in the build container: None; on a GPU box: the GPUs)
    import ast
    import inspect
    for fn in (bench.launch_ranks, bench.visible_gpus):           # no `<x>.cuda` attribute access anywhere in their code
        tree = ast.parse(inspect.getsource(fn))
        assert not [n for n in ast.walk(tree) if isinstance(n, ast.Attribute) and n.attr == "cuda"], fn.__name__


def test_single_rank_runs_in_process():
    p = _run({}, "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "8")
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 1 and out["weights_broadcast"] is None


def test_failing_rank_fails_the_run_with_its_exit_code():
    p = _run({"PCV_BENCH_STUB_FAIL_RANK": "1"}, "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8")
    assert p.returncode == 7, (p.returncode, p.stderr[-1000:])
    assert p.stdout.strip() == ""                                # no result line for a failed run
    assert "rank 1 fails on request" in p.stderr


def test_under_a_launcher_it_is_one_of_the_ranks():
    """WORLD_SIZE set (what torch.distributed.run does): no children, this process IS rank 0 of a 1-rank gloo group."""
    p = _run({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(29900 + os.getpid() % 90),
              "PCV_BENCH_FORCE_DIST": "1"}, "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "8")
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 1 and out["weights_broadcast"]["messages"] >= 1


def test_kernel_class_of_names():
    sys.path.insert(0, ROOT)
    import bench
    f = bench.kernel_class_of
    assert f("void d3q_kernel<1, 4, 2, 2, 7, 2, false>(D3Params)") == "dense3x3"
    assert f("void d3q_kernel<1, 8, 1, 2, 7, 2, true>(D3Params)") == "dense1x1_kheavy"
    assert f("void d3w_kernel<1, 4, 2, 4, 7, 2, 3>(D3Params)") == "dense3x3" and f("void d3c_kernel<1>(D3Params)") == "dense3x3"
    assert f("void p1r_kernel<1, 32, 512>(D3Params)") == "dense1x1_kheavy" and f("void d3k_kernel<1>(D3Params)") == "dense3x3"
    assert f("void d1i_kernel<1, 1024>(D3Params)") == "dense1x1_kheavy"
    assert f("void d3i_kernel<1, 256>(D3Params)") == "dense3x3" and f("pack_d3i_kernel(unsigned int __vector(4) const*, ...)") is None
    assert f("void igemm_conv_kernel<1, 1, 4, 4, 2, 2, false, 9>(IgemmParams)") == "dense3x3"
    assert f("void igemm_conv_kernel<1, 1, 4, 4, 4, 1, false, 1>(IgemmParams)") == "dense1x1"
    assert f("void wpair1x1_kernel<1, 256, 1024, false>(WPairParams)") == "pair1x1"
    assert f("void pair1x1_kernel<1, 2, true, false>(PairParams)") == "pair1x1"
    assert f("void gconv3x3r_kernel<1, 2, 5>(GConvRParams)") == "grouped3x3"
    assert f("void dwconv_kernel<2, 3, 1, true, 8>(DwParams)") == "depthwise"
    assert f("void mbw_kernel<2, 1, 4, 2, 16, 1, 0>(MbParams)") == "fused_unit"
    assert f("void stem_conv_kernel<1, true, true>(StemParams)") == "stem"
    assert f("void head_gemm_f32_kernel<16>(HeadParams)") is None and f("void pack_gconv_kernel<1>(float const*, void*, int, int)") is None


def test_runner_with_foreign_packed_state_refuses_to_repack():
    """after parallel.broadcast_packed_state the receivers' packed weights no longer derive from their own fp32
    parameters; a key change (dtype, channel pitch, padding parity, touched parameter) must raise, not re-pack silently."""
    sys.path.insert(0, ROOT)
    import torch.nn as nn
    from pytorchcv_amd import engine, parallel

    class Block(nn.Module):
        def __init__(self):
            super(Block, self).__init__()
            self.conv = nn.Conv2d(16, 16, 1, bias=False)
            self.bn = nn.BatchNorm2d(16)
            self._pcv_runner = engine.ConvRunner(self.conv, self.bn)
            self._pcv_bnact = engine.BnActRunner(self.bn)

    blk = Block().eval()
    r, b = blk._pcv_runner, blk._pcv_bnact
    cpu = torch.device("cpu")
    s_bf = engine._ShapeOnly(1, 4, 4, 16, torch.bfloat16)
    x = engine._PrepHandle(s_bf, cpu)                              # dtype / cpitch / device: all prepare() reads of its input
    d = r.desc(s_bf, 0, 0, False)
    r._key = r._state_key(x.dtype, x.cpitch, d)                   # as if packed for this configuration
    r.packed, r.scale, r.shift = torch.zeros(1), torch.zeros(16), torch.zeros(16)
    b._key, b.scale, b.shift = b._state_key(), torch.zeros(16), torch.zeros(16)
    for q in parallel._runners(blk):
        q.adopt_foreign_state()                                   # what broadcast_packed_state does on ranks != src
    r.prepare(x, d)                                               # same key: the received state is used as is
    b.prepare(x)
    s_16 = engine._ShapeOnly(1, 4, 4, 16, torch.float16)
    with pytest.raises(RuntimeError, match="broadcast_packed_state"):
        r.prepare(engine._PrepHandle(s_16, cpu), r.desc(s_16, 0, 0, False))      # another dtype: would re-pack from the local fp32 weights
    with torch.no_grad():
        blk.bn.weight.mul_(2.0)                                   # a touched parameter
    with pytest.raises(RuntimeError, match="broadcast_packed_state"):
        b.prepare(x)
    parallel.mark_local_state_authoritative(blk)                  # (or broadcast_module_state): local tensors are the truth again
    assert r._key is None and not r._foreign and not b._foreign
