#!/usr/bin/env python
"""
bench.py - images/sec of the MI355X conv-net inference hot path (BASELINE.json metric), one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload resnet50_bs256]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (child processes, before anything in this
process touches a GPU) and relays rank 0's JSON line; under torch.distributed.run it is one of the ranks.

A step = one forward of the workload's batch through `get_model(name)` (fp32 NCHW input resident in HBM -> NHWC 16 bit ->
fused HIP kernels -> fp32 logits) followed, for N > 1, by the all-gather of the logits. Weak scaling: every rank runs the
full per-GPU batch. Rank 0 prints ONE JSON line (contract in the task statement) with
  roofline      the north-star kernel class of the workload (ResNet-50: dense 3x3 vs the 2.5 PFLOP/s dense bf16 MFMA peak;
                MobileNetV2: stand-alone depthwise vs the 8 TB/s HBM peak; ResNeXt-101: grouped 3x3) timed live with HIP events
                around every eager launch, beside the committed rocprofv3 all-launch average and PMC traffic, plus `classes`:
                every kernel class of the forward with us/step, achieved rate and fraction of its roof;
  cpu_baseline  the oracle's torch-fp32 CPU forward on the host cores;
  other_configs (default invocation only) BASELINE configs 3 and 4 - MobileNetV2 bs512, ResNeXt-101 bs256 - timed the same way
                over 10 steps each, so that they carry a driver-timed number too.
"""

import os
import sys
import json
import time
import socket
import argparse
import subprocess

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WORKLOADS = {
    # name: (model, per-GPU batch, north-star kernel class for `roofline`)
    "resnet50_bs256": ("resnet50", 256, "dense3x3"),
    "mobilenetv2_w1_bs512": ("mobilenetv2_w1", 512, "depthwise"),
    "resnext101_32x4d_bs256": ("resnext101_32x4d", 256, "grouped3x3"),
    "resnet18_bs256": ("resnet18", 256, "dense3x3"),
    "mobilenetv3_large_w1_bs512": ("mobilenetv3_large_w1", 512, "depthwise"),
    "efficientnet_b0_bs256": ("efficientnet_b0", 256, "depthwise"),
    "vgg16_bs128": ("vgg16", 128, "dense3x3"),
    "seresnet50_bs256": ("seresnet50", 256, "dense3x3"),
    "seresnext50_32x4d_bs256": ("seresnext50_32x4d", 256, "grouped3x3"),
}
OTHER_CONFIGS = ("mobilenetv2_w1_bs512", "resnext101_32x4d_bs256")      # BASELINE.json configs[2], configs[3]
MFMA_PEAK_TFLOPS = 2500.0     # dense bf16/fp16, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
HBM_PEAK_GBS = 8000.0         # HBM3E spec peak, MI355X_MICROARCH.md "HBM3E peak BW"
TERM_GRACE_S = 10.0                       # launcher: seconds between SIGTERM and SIGKILL for ranks that do not exit
OTHER_STEPS, OTHER_WARMUP = 30, 5          # timed / warm-up steps of each `other_configs` workload (r4: 10 steps = 17 ms was too short a window)
RIDGE_FLOP_PER_BYTE = MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)     # 312.5: a class above it is MFMA-bound, below it HBM-bound

# Kernel classes. Live timing classifies a launch by what the host asked for; the rocprofv3 / PMC summaries classify a dispatch
# by its kernel name (`kernel_class_of`, also used by tests/tools/*). The two agree except where the library routes a shape to
# another kernel family than the rule below assumes (d3q's 1x1 mode is chosen by tile count as well as by channels).


def kernel_class_of(name: str):
    """Kernel class of a rocprofv3 kernel name, None for kernels outside the convolution path (pools, head, layout, torch)."""
    head = name.split("(")[0].rstrip()
    if "pack_" in name:
        return None
    if "d3w_kernel" in name or "d3c_kernel" in name or "d3k_kernel" in name or "d3i_kernel" in name:
        return "dense3x3"
    if "p1r_kernel" in name or "d1i_kernel" in name:
        return "dense1x1_kheavy"
    if "d3q_kernel" in name:
        return "dense1x1_kheavy" if head.endswith("true>") else "dense3x3"
    if "igemm_conv_kernel" in name:
        if head.endswith(", 9>"):
            return "dense3x3"
        return "dense1x1" if head.endswith(", 1>") else "conv_other"
    if "wpair1x1_kernel" in name or "pair1x1_kernel" in name:
        return "pair1x1"
    if "gconv3x3" in name and "pack_" not in name:
        return "grouped3x3"
    if ("dwconv_kernel" in name or "dwconv5_kernel" in name) and "pack_" not in name:
        return "depthwise"
    if "mbr_kernel" in name or "mbw_kernel" in name or "mbconv_kernel" in name:
        return "fused_unit"
    if "stem_conv_kernel" in name:
        return "stem"
    return None


def calib_for(model):
    p = os.path.join(ROOT, "tests", "golden", "calib_{}.json".format(model))
    if not os.path.exists(p):
        return None
    with open(p) as f:
        return {k: tuple(v) for k, v in json.load(f).items()}


class LaunchTimer(object):
    """Brackets every convolution-path launch of an eager forward with events on the launch stream (torch's current stream,
    which is the stream handed to the C ABI) and accumulates algorithmic FLOPs / bytes per launch, per kernel class."""
    def __init__(self):
        self.records = []      # (start_event, end_event, flops, bytes, class, tag)
        self._saved = {}
        self._depth = 0

    @staticmethod
    def classify(runner, d, M):
        if runner.depthwise:
            return "depthwise"
        if d.groups > 1:
            return "grouped3x3" if d.kh == 3 else "conv_other"
        if d.kh == 3 and d.kw == 3 and d.x_cpitch != 4:
            return "dense3x3"
        if d.kh == 1 and d.kw == 1:
            return "dense1x1_kheavy" if (d.Cin >= 256 and d.Cout >= 128 and d.out_dtype == d.dtype) else "dense1x1"
        return "stem" if d.x_cpitch == 4 else "conv_other"

    def _timed(self, fn, account):
        """Run `fn()` between two events unless a timed launch is already open (a fused entry point that falls through to another
        timed one); `account(result)` -> (flops, bytes, class, tag) or None when nothing was launched."""
        if self._depth:
            return fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self._depth += 1
        try:
            s.record()
            y = fn()
            e.record()
        finally:
            self._depth -= 1
        if y is not None:
            rec = account(y)
            if rec is not None:
                self.records.append((s, e) + rec)
        return y

    def __enter__(self):
        from pytorchcv_amd import engine
        timer, CR = self, engine.ConvRunner
        self._saved = dict(_launch=CR._launch, run_pair=CR.run_pair, run_pair_idconv=CR.run_pair_idconv, run_maxpool=CR.run_maxpool,
                           _stem_from_nchw=CR._stem_from_nchw, mbconv_fused=engine.mbconv_fused)
        S = self._saved

        def conv_cost(x, d, yN, yH, yW, residual):
            es = 2 if d.dtype != 0 else 4
            cin_g = d.Cin // d.groups
            px_out = yN * yH * yW
            flops = 2.0 * px_out * d.Cout * cin_g * d.kh * d.kw
            nbytes = (x.N * x.H * x.W * d.Cin + px_out * d.Cout * (2 if residual else 1)) * es + d.Cout * cin_g * d.kh * d.kw * es
            return flops, nbytes

        def launch(runner, x, d, residual, out=None, gate=None):
            if out is not None:
                return S["_launch"](runner, x, d, residual, out, gate)

            def account(y):
                f, b = conv_cost(x, d, y.N, y.H, y.W, residual is not None)
                return f, b, timer.classify(runner, d, y.N * y.H * y.W), "{}x{}x{}->{} k{} s{}".format(x.H, x.W, d.Cin, d.Cout, d.kh, d.stride_h)
            return timer._timed(lambda: S["_launch"](runner, x, d, residual, None, gate), account)

        def run_pair(runner, x, residual, act, post_act, nxt, nxt_act, gate=None):
            def account(ys):
                y1, y2 = ys
                px = x.N * x.H * x.W
                cm, c1 = x.cpitch, y1.cpitch
                f = 2.0 * px * (cm * c1 + c1 * y2.cpitch)
                b = (px * (cm + 2 * c1 + y2.cpitch) + cm * c1 + c1 * y2.cpitch) * 2
                return f, b, "pair1x1", "{}x{}x{}->{}->{}".format(x.H, x.W, cm, c1, y2.cpitch)
            return timer._timed(lambda: S["run_pair"](runner, x, residual, act, post_act, nxt, nxt_act, gate), account)

        def run_pair_idconv(runner, x, x0, idr, act, post_act, nxt, nxt_act):
            def account(ys):
                y1, y2 = ys
                px = x.N * x.H * x.W
                cm, c1 = x.cpitch, y1.cpitch
                f = 2.0 * px * (cm * c1 + x0.cpitch * c1 + c1 * y2.cpitch)
                b = (px * (cm + x0.cpitch + c1 + y2.cpitch) + cm * c1 + x0.cpitch * c1 + c1 * y2.cpitch) * 2
                return f, b, "pair1x1", "{}x{}x{}->{}->{} +idconv".format(x.H, x.W, cm, c1, y2.cpitch)
            return timer._timed(lambda: S["run_pair_idconv"](runner, x, x0, idr, act, post_act, nxt, nxt_act), account)

        def stem_cost(runner, x, y):
            c = runner.conv
            kh, kw = c.kernel_size
            ho, wo = (x.H + 2 * c.padding[0] - kh) // c.stride[0] + 1, (x.W + 2 * c.padding[1] - kw) // c.stride[1] + 1
            f = 2.0 * x.N * ho * wo * y.cpitch * c.in_channels * kh * kw
            in_bytes = x.N * x.H * x.W * (c.in_channels * 4 if isinstance(x, engine.LazyNCHW) and not x.materialized else 8)
            return f, in_bytes + y.N * y.H * y.W * y.cpitch * 2 + y.cpitch * c.in_channels * kh * kw * 2

        def run_maxpool(runner, x, act, k, s, p, ceil_mode=False):
            def account(y):
                f, b = stem_cost(runner, x, y)
                return f, b, "stem", "{}x{} k{} +pool".format(x.H, x.W, runner.conv.kernel_size[0])
            return timer._timed(lambda: S["run_maxpool"](runner, x, act, k, s, p, ceil_mode), account)

        def stem_from_nchw(runner, x, act, pool):
            def account(y):
                f, b = stem_cost(runner, x, y)
                return f, b, "stem", "{}x{} k{} nchw{}".format(x.H, x.W, runner.conv.kernel_size[0], " +pool" if pool else "")
            return timer._timed(lambda: S["_stem_from_nchw"](runner, x, act, pool), account)

        def mbconv(exp, exp_act, dw, dw_act, proj, proj_act, x, residual, post_act):
            def account(y):
                cmid, cout = dw.conv.out_channels, proj.conv.out_channels
                px_in, px_out = x.N * x.H * x.W, y.N * y.H * y.W
                f = 2.0 * ((px_in * x.C * cmid if exp is not None else 0) + px_out * cmid * 9 + px_out * cmid * cout)
                b = (px_in * x.C + px_out * cout + (x.C * cmid if exp is not None else 0) + 9 * cmid + cmid * cout) * 2
                return f, b, "fused_unit", "{}x{}x{}->{}->{} s{}".format(x.H, x.W, x.C, cmid, cout, dw.conv.stride[0])
            return timer._timed(lambda: S["mbconv_fused"](exp, exp_act, dw, dw_act, proj, proj_act, x, residual, post_act), account)

        CR._launch, CR.run_pair, CR.run_pair_idconv, CR.run_maxpool, CR._stem_from_nchw = launch, run_pair, run_pair_idconv, run_maxpool, stem_from_nchw
        engine.mbconv_fused = mbconv
        return self

    def __exit__(self, *a):
        from pytorchcv_amd import engine
        CR, S = engine.ConvRunner, self._saved
        CR._launch, CR.run_pair, CR.run_pair_idconv, CR.run_maxpool, CR._stem_from_nchw = (
            S["_launch"], S["run_pair"], S["run_pair_idconv"], S["run_maxpool"], S["_stem_from_nchw"])
        engine.mbconv_fused = S["mbconv_fused"]

    def summary(self, passes):
        """{class: dict(launches_per_step, us_per_step, avg_launch_us, tflops, gbs, gflop_per_launch, mb_per_launch, per_shape)}"""
        torch.cuda.synchronize()
        out = {}
        for s, e, f, b, klass, tag in self.records:
            t = s.elapsed_time(e)
            c = out.setdefault(klass, dict(n=0, ms=0.0, flops=0.0, bytes=0.0, shapes={}))
            c["n"] += 1
            c["ms"] += t
            c["flops"] += f
            c["bytes"] += b
            a = c["shapes"].setdefault(tag, [0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += t
            a[2] += f
            a[3] += b
        res = {}
        for klass, c in out.items():
            sec = c["ms"] * 1e-3
            res[klass] = dict(
                launches_per_step=c["n"] // passes, us_per_step=1e3 * c["ms"] / passes, avg_launch_us=1e3 * c["ms"] / c["n"],
                tflops=c["flops"] / sec / 1e12, gbs=c["bytes"] / sec / 1e9, gflop_per_launch=c["flops"] / c["n"] / 1e9,
                mb_per_launch=c["bytes"] / c["n"] / 1e6,
                per_shape={k: dict(launches=v[0] // passes, avg_us=round(1e3 * v[1] / v[0], 2), tflops=round(v[2] / (v[1] * 1e-3) / 1e12, 2),
                                   gbs=round(v[3] / (v[1] * 1e-3) / 1e9, 1)) for k, v in c["shapes"].items()})
        return res


def _cpu_rate(fn, images_per_call, budget_s, min_calls=2, max_calls=64):
    calls, elapsed = 0, 0.0
    t_start = time.time()
    while calls < min_calls or (time.time() - t_start < budget_s and calls < max_calls):
        t1 = time.time()
        fn()
        elapsed += time.time() - t1
        calls += 1
    return images_per_call * calls / elapsed, calls


def usable_cores():
    """Host cores this process may actually use: the scheduler affinity, cut down to the cgroup CPU quota where one is set (a
    one-GPU box of the pool is a 16-CPU share of a 256-thread host; 256 OpenMP threads on that share ran 40x slower than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                      # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                quota = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    cap = int(os.environ.get("PCV_BENCH_CPU_THREADS", "0"))
    return min(n, cap) if cap > 0 else n


def cpu_baseline(model, sd_cpu, budget_s=12.0):
    """The oracle's CPU forward (torch eager fp32 = the ATen path the reference runs) on the GPU box's host cores, bounded
    samples: the workload's model on ALL host cores (batch 32) and on one thread (batch 4), plus BASELINE
    config 1 - resnet18, batch 1, fp32 - on all cores."""
    from oracle import refnet
    from pytorchcv_amd.synth import synth_input, synth_state_dict
    from pytorchcv_amd.model_provider import get_model
    cores = usable_cores()
    t0 = time.time()
    x = synth_input(32, seed=11)
    torch.set_num_threads(cores)
    refnet.forward(model, sd_cpu, x[:2])                      # warm-up (thread pool, oneDNN primitives)
    rate_all, n_all = _cpu_rate(lambda: refnet.forward(model, sd_cpu, x), 32, budget_s)
    sd18 = synth_state_dict(get_model("resnet18").state_dict(), seed=1234, calib=calib_for("resnet18"))
    refnet.forward("resnet18", sd18, x[:1])
    rate_c1, n_c1 = _cpu_rate(lambda: refnet.forward("resnet18", sd18, x[:1]), 1, 3.0, min_calls=5, max_calls=200)
    torch.set_num_threads(1)
    rate_one, n_one = _cpu_rate(lambda: refnet.forward(model, sd_cpu, x[:4]), 4, 6.0, min_calls=1, max_calls=8)
    torch.set_num_threads(cores)
    return dict(value=round(rate_all, 2), unit="images/sec", cores=cores, kind="port", host_logical_cpus=os.cpu_count(),
                one_thread_value=round(rate_one, 3),
                config1_resnet18_bs1_fp32=dict(value=round(rate_c1, 2), unit="images/sec", cores=cores, ms_per_image=round(1e3 / rate_c1, 2)),
                sample="{} at 224x224, fp32, torch {} eager CPU ops via oracle/refnet.py: {} forward(s) of batch 32 on {} threads; "
                       "{} forward(s) of batch 4 on 1 thread; resnet18 batch 1 x {} on {} threads ({:.1f} s in all)".format(
                           model, torch.__version__, n_all, cores, n_one, n_c1, cores, time.time() - t0))


def committed_profile(workload, dtype):
    """{class: dict(calls, avg_us)} over ALL launches of the committed rocprofv3 --kernel-trace --stats summary of this workload
    (profiles/r<NN>_<workload>_<dtype>_kernel_stats.csv, the newest round present: collected with PCV_BENCH_PROFILE=1, where bench.py
    runs nothing but full-batch forwards - every dispatch in the table is a full-batch launch), and the file; ({}, None) when there is none."""
    import csv
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_{}_{}_kernel_stats.csv".format(workload, dtype))))
    if not found:
        return {}, None
    path = found[-1]
    acc = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            k = kernel_class_of(row["Name"])
            if k is None:
                continue
            a = acc.setdefault(k, [0, 0])
            a[0] += int(row["Calls"])
            a[1] += int(row["TotalDurationNs"])
    return {k: dict(calls=v[0], avg_us=round(v[1] / v[0] / 1e3, 2)) for k, v in acc.items() if v[0]}, os.path.relpath(path, ROOT)


def committed_traffic(workload, dtype):
    """{class: dict(traffic_mb_per_launch, ...)} from profiles/pmc_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    passes over this command; cannot be read inside this process) for this workload and dtype, and its note."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            e = json.load(f).get(workload)
    except (OSError, ValueError):
        return {}, None
    if not e or e.get("dtype") != dtype or "classes" not in e:
        return {}, None
    return e["classes"], e.get("note")


# ---------------------------------------------------------------------------------------------------------------------------------
class Env(object):
    """Where this rank runs: device, process group, synchronisation. `stub`: the launcher / collective plumbing rehearsed on CPU
    with gloo and a stand-in forward (tests/test_bench_launcher.py) - never a measurement, the JSON line says so."""
    def __init__(self, stub):
        self.stub = stub
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # PCV_BENCH_REHEARSE=1 (dev, a ONE-GPU box): every rank runs the real product path on cuda:0 and the collectives go through gloo -
        # the N > 1 code path (packed-state broadcast of the real arenas, replica check, MAX-reduced timing) without a second GPU.
        # Never a measurement: the ranks share one GPU, and the JSON line says so.
        self.rehearse = not stub and os.environ.get("PCV_BENCH_REHEARSE") == "1"
        if self.rehearse:
            self.local_rank = 0
        self.dev = torch.device("cpu") if stub else torch.device("cuda", self.local_rank)
        self.use_dist = self.world > 1 or os.environ.get("PCV_BENCH_FORCE_DIST") == "1"   # (forced at world size 1: rehearses the RCCL calls)
        if self.use_dist:
            # the process group comes up first: nothing of ours touches the GPU before RCCL has bound this rank to its device
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            if stub or self.rehearse:
                dist.init_process_group(backend="gloo")
            else:
                dist.init_process_group(backend="nccl", device_id=self.dev)
        if not stub:
            if not torch.cuda.is_available():
                raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback for the product path)")
            torch.cuda.set_device(self.local_rank)

    def sync(self):
        if not self.stub:
            torch.cuda.synchronize()

    def fail(self, code, msg):
        print("bench.py: " + msg, file=sys.stderr, flush=True)
        if self.use_dist:
            dist.destroy_process_group()
        sys.exit(code)


class _StubNet(torch.nn.Module):
    """Stand-in forward for the launcher self-test: [B,3,H,W] -> 8x8 average pool -> Linear -> [B,1000]. Plain torch on CPU; has
    nothing to do with the product path (which has no CPU form) and is never timed for a result."""
    def __init__(self, seed):
        super(_StubNet, self).__init__()
        g = torch.Generator().manual_seed(seed)
        self.fc = torch.nn.Linear(3 * 8 * 8, 1000)
        with torch.no_grad():
            self.fc.weight.copy_(torch.randn(self.fc.weight.shape, generator=g) * 0.05)
            self.fc.bias.copy_(torch.randn(self.fc.bias.shape, generator=g) * 0.05)

    def forward(self, x):
        return self.fc(torch.nn.functional.adaptive_avg_pool2d(x, 8).flatten(1))


def run_workload(env, workload, args, steps, warmup, profile=False):
    """One workload, timed exactly as the contract says: W untimed steps, K timed ones between barrier + synchronize on both sides,
    MAX over ranks. Returns (result dict, context for the roofline / CPU passes)."""
    from pytorchcv_amd.parallel import ShardedInference, broadcast_packed_state, replicas_agree
    from pytorchcv_amd.synth import synth_state_dict, synth_input
    model, batch, klass = WORKLOADS[workload]
    if args.batch > 0:
        batch = args.batch
    if args.scaling == "strong":                       # BASELINE config 5 literally: a fixed global batch split over the ranks
        if args.global_batch % env.world != 0:
            env.fail(2, "--global-batch {} does not split over {} rank(s)".format(args.global_batch, env.world))
        batch = args.global_batch // env.world
    rank, dev = env.rank, env.dev
    sd_cpu = None
    if env.stub:
        net, dtype, ovf0 = _StubNet(seed=1234 if rank == 0 else 77 + rank).eval(), "fp32", 0
    else:
        import pytorchcv_amd
        from pytorchcv_amd import engine
        from pytorchcv_amd.model_provider import get_model
        net = get_model(model).eval()
        if rank == 0:
            sd_cpu = synth_state_dict(net.state_dict(), seed=1234, calib=calib_for(model))
            net.load_state_dict(sd_cpu, strict=True)                  # ranks != 0 keep their random init: the broadcast must deliver
        net = pytorchcv_amd.set_compute_dtype(net.to(dev), args.dtype)
        dtype = engine.compute_dtype_of(net)                          # what "auto" resolved to for this family: reported in the JSON line
        ovf0 = engine.fp16_overflow_count(dev) if dtype == "fp16" else 0

    # synthetic N(0,1)-like images: 8 distinct seeded images per rank tiled to the batch (performance is data independent)
    hw = 224 if not env.stub else 32
    base = synth_input(8, 3, hw, hw, seed=rank).to(dev)
    common = synth_input(8, 3, hw, hw, seed=4242).to(dev)            # the same on every rank: the replica check below
    x = base.repeat((batch + 7) // 8, 1, 1, 1)[:batch].contiguous()
    with torch.no_grad():
        net(x if profile else base)                  # builds and packs every layer (rank 0: the real weights)
    bcast = None
    if env.use_dist:
        # RCCL broadcast of rank 0's PACKED inference state over xGMI (16-bit arenas + fp32 scale/shift: half the fp32 state;
        # the receiving ranks do not re-pack)
        env.sync()
        dist.barrier()
        tb = time.perf_counter()
        bcast = broadcast_packed_state(net, src=0)
        env.sync()
        dist.barrier()
        bcast_s = time.perf_counter() - tb
        with torch.no_grad():
            if not replicas_agree(net(common)):      # a tensor the broadcast missed = plausible but different logits on ranks != 0
                env.fail(5, "replicas disagree on a common input after the weight broadcast")
    y_ref8 = None
    if not profile:
        with torch.no_grad():
            y_ref8 = net(base).clone()               # the 8 distinct images on their own: what every row of the timed batch must equal
    use_graph = args.graph != 0 and not env.stub and not profile
    fwd = net
    pipelined = False
    if use_graph:
        from pytorchcv_amd.graph import capture, capture_best, PipelinedNet
        try:
            # ~60 kernel launches replayed by one hipGraphLaunch; the inputs are static buffers resident in HBM. --inflight 0 (default):
            # the launcher that replays fastest HERE - one graph of two batch lanes, or two full-batch graphs in flight on alternating
            # streams (consecutive steps overlap: the single-round 7x7 tail of step n runs under the head of step n+1)
            if args.inflight == 0 and args.lanes == 0 and env.use_dist:
                # multi-rank runs do not time launchers per rank (ranks could keep different ones and the MAX-reduced figure would
                # be a mix): every rank replays the launcher that wins on 8 of the 9 workloads at N = 1
                fwd = PipelinedNet(net, x, depth=2, lanes=1, own_input=True)
            elif args.inflight == 0 and args.lanes == 0:
                fwd = capture_best(net, x, own_input=True)
            elif args.inflight >= 2:
                fwd = PipelinedNet(net, x, depth=args.inflight, lanes=max(1, args.lanes), own_input=True)
            else:
                fwd = capture(net, x, own_input=True, lanes=args.lanes if args.lanes > 0 else None)
            pipelined = isinstance(fwd, PipelinedNet)
        except Exception as e:                       # noqa: BLE001 - a run that cannot capture is not the benchmarked configuration
            env.fail(4, "hipGraph capture failed ({}); rerun with --graph 0 for eager launches".format(e))
    runner = ShardedInference(fwd)
    last = []                                        # the results of the last `depth` steps: one per slot in flight

    def step():
        if pipelined:                                # every slot owns a resident copy of the batch; the all-gather rides on the slot's stream
            y = fwd(None, then=(runner.gather_all if env.use_dist else None))
            last.append(y)
            del last[:-fwd.depth]
            return y
        y = runner.run_local(x)
        return runner.gather_all(y) if env.use_dist else y

    for _ in range(warmup):
        step()
    if env.use_dist:
        dist.barrier()
    env.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        y = step()
    env.sync()
    if env.use_dist:
        dist.barrier()
    env.sync()
    elapsed = time.perf_counter() - t0
    if env.use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if dtype == "fp16" and engine.fp16_overflow_count(dev) != ovf0:
        env.fail(3, "the fp16 range guard counted an overflow during the run")
    if y_ref8 is not None:
        # the timed forward computes what the small-batch forward computes: every row of the full batch (multi-round tile
        # schedules, graph lanes) equals the 8-image eager result bit for bit; a mismatch fails the run
        want = y_ref8.repeat((batch + 7) // 8, 1)[:batch]
        for yy in (last if pipelined else [y]):      # every step in flight at the end of the timed region
            y_local = yy[rank * batch:(rank + 1) * batch] if env.use_dist else yy
            same = torch.equal(y_local, want) if not env.stub else bool(torch.allclose(y_local, want, atol=1e-5))
            if not bool(torch.isfinite(y_local).all()) or not same:
                env.fail(3, "{} of {} rows of the timed batch differ from the 8-image forward".format(
                    int((y_local != want).any(1).sum()), batch))
    res = dict(value=round(env.world * batch * steps / elapsed, 1), ms_per_step=round(1e3 * elapsed / steps, 3), steps=steps, warmup=warmup,
               dtype=dtype, per_gpu_batch=batch,
               launch=(("{} hipGraphs in flight on alternating streams (consecutive steps overlap), {} batch lane(s) each".format(fwd.depth, fwd.lanes)
                        if pipelined else "hipGraph replay, {} batch lane(s) as parallel branches".format(fwd.lanes)) if use_graph else "eager"),
               launchers_timed=getattr(fwd, "tuning", None),
               launcher_choice=("fixed for multi-rank runs (the same on every rank)" if env.use_dist and args.inflight == 0 and args.lanes == 0
                                else ("timed at capture on this GPU" if args.inflight == 0 and args.lanes == 0 else "flags")),
               weights_broadcast=(dict(messages=bcast[0], bytes=bcast[1], seconds=round(bcast_s, 4), replicas_agree=True,
                                       what="packed inference state (RCCL broadcast from rank 0), then every rank's logits of a common "
                                            "input compared (all-reduce MIN / MAX)") if bcast else None))
    return res, dict(net=net, x=x, sd_cpu=sd_cpu, model=model, klass=klass, dtype=dtype, batch=batch)


def forward_latency(ctx, samples=24):
    """SURVEY 8d's latency figure: HIP events around ONE forward of the resident batch on ONE stream (a single-lane hipGraph replay:
    nothing of another step overlaps it), `samples` replays one after the other, each synchronised before the next starts."""
    from pytorchcv_amd.graph import capture
    g = capture(ctx["net"], ctx["x"], own_input=True, lanes=1)
    ms = []
    for i in range(samples + 3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g(None)
        e1.record()
        e1.synchronize()
        if i >= 3:
            ms.append(e0.elapsed_time(e1))
    del g
    ms.sort()
    med = ms[len(ms) // 2] if len(ms) % 2 else 0.5 * (ms[len(ms) // 2 - 1] + ms[len(ms) // 2])
    return dict(latency_ms_median=round(med, 4), latency_ms_min=round(ms[0], 4), latency_ms_max=round(ms[-1], 4), samples=len(ms),
                images_per_sec_at_median_latency=round(ctx["batch"] / med * 1e3, 1),
                how="HIP events around one single-lane hipGraph replay of the whole forward on one stream, one at a time (SURVEY 8d)")


def compact_roofline(roof):
    """The north-star class of a workload, as it goes into `other_configs`."""
    if roof is None:
        return None
    out = {k: roof.get(k) for k in ("kernel_class", "bound", "achieved", "peak", "unit", "frac", "avg_launch_us", "launches_per_step",
                                    "traffic", "rocprof_avg_launch_us", "rocprof_frac")}
    # every class of the step beside it (MobileNetV2's fused units hold 13 of its 17 depthwise layers and most of its time)
    out["classes"] = [{k: e.get(k) for k in ("kernel_class", "launches_per_step", "us_per_step", "share_of_conv_time", "bound", "frac",
                                             "frac_of_hbm_peak", "frac_of_mfma_peak", "rocprof_frac", "traffic_mb_per_launch")}
                      for e in roof.get("classes", [])]
    return out


def roofline_of(env, workload, ctx, passes):
    """Live per-launch timing of every kernel class of the workload's forward (eager: the same kernel mix `value` replays from its
    graph), the north-star class on top."""
    net, x, klass, dtype = ctx["net"], ctx["x"], ctx["klass"], ctx["dtype"]
    with torch.no_grad():
        net(x)                                       # untimed: first eager full-batch forward (code load, allocator)
    torch.cuda.synchronize()
    with LaunchTimer() as lt:
        for _ in range(passes):
            with torch.no_grad():
                net(x)
    live = lt.summary(passes)
    prof, prof_file = committed_profile(workload, dtype)
    pmc, pmc_note = committed_traffic(workload, dtype) if ctx["batch"] == WORKLOADS[workload][1] else ({}, None)
    total_us = sum(c["us_per_step"] for c in live.values()) or 1.0

    def entry(k, c, detail):
        intensity = c["gflop_per_launch"] * 1e9 / (c["mb_per_launch"] * 1e6)
        # the roof that bounds a class is read off its arithmetic intensity against the ridge (312.5 FLOP/B), not assumed per class
        # (the K-heavy 1x1 layers sit at 136 FLOP/B - HBM-bound); both fractions are printed
        bound = "mfma" if intensity >= RIDGE_FLOP_PER_BYTE else "hbm"
        e = dict(kernel_class=k, bound=bound, launches_per_step=c["launches_per_step"], us_per_step=round(c["us_per_step"], 1),
                 share_of_conv_time=round(c["us_per_step"] / total_us, 3), avg_launch_us=round(c["avg_launch_us"], 2),
                 achieved=round(c["tflops"], 1) if bound == "mfma" else round(c["gbs"], 1), unit="TFLOP/s" if bound == "mfma" else "GB/s",
                 frac=round(c["tflops"] / MFMA_PEAK_TFLOPS if bound == "mfma" else c["gbs"] / HBM_PEAK_GBS, 4),
                 frac_of_mfma_peak=round(c["tflops"] / MFMA_PEAK_TFLOPS, 4), frac_of_hbm_peak=round(c["gbs"] / HBM_PEAK_GBS, 4),
                 tflops=round(c["tflops"], 1), gbs=round(c["gbs"], 1), flop_per_byte=round(intensity, 1),
                 algorithmic_gflop_per_launch=round(c["gflop_per_launch"], 3), algorithmic_mb_per_launch=round(c["mb_per_launch"], 3),
                 rocprof_avg_launch_us=prof.get(k, {}).get("avg_us"), rocprof_calls=prof.get(k, {}).get("calls"),
                 traffic_mb_per_launch=pmc.get(k, {}).get("traffic_mb_per_launch"))
        if e["rocprof_avg_launch_us"]:
            rate = (c["gflop_per_launch"] * 1e9 / 1e12 if bound == "mfma" else c["mb_per_launch"] * 1e6 / 1e9) / (e["rocprof_avg_launch_us"] * 1e-6)
            e["rocprof_achieved"] = round(rate, 1)
            e["rocprof_frac"] = round(rate / (MFMA_PEAK_TFLOPS if bound == "mfma" else HBM_PEAK_GBS), 4)
        if detail:
            e["per_shape"] = c["per_shape"]
        return e

    classes = [entry(k, c, k == klass) for k, c in sorted(live.items(), key=lambda kv: -kv[1]["us_per_step"])]
    head = next((e for e in classes if e["kernel_class"] == klass), None)
    if head is None:
        return None
    roof = dict(bound=head["bound"], achieved=head["achieved"], peak=MFMA_PEAK_TFLOPS if head["bound"] == "mfma" else HBM_PEAK_GBS,
                unit=head["unit"], frac=head["frac"], traffic=head["traffic_mb_per_launch"],
                traffic_unit="MB per launch (HBM read + write)" if head["traffic_mb_per_launch"] is not None else None,
                traffic_source=("profiles/pmc_traffic.json: " + pmc_note) if head["traffic_mb_per_launch"] is not None and pmc_note else None,
                kernel_class=klass, launches_per_step=head["launches_per_step"], avg_launch_us=head["avg_launch_us"],
                avg_launch_us_source="live: HIP events around every eager launch of the class on the launch stream (this run, {} full-batch "
                                     "forwards)".format(passes),
                rocprof_avg_launch_us=head["rocprof_avg_launch_us"], rocprof_frac=head.get("rocprof_frac"),
                rocprof_source=("ALL {} launches of the class in the committed rocprofv3 --kernel-trace --stats summary of this command "
                                "(PCV_BENCH_PROFILE=1: full-batch forwards only, one lane, eager): {}".format(head["rocprof_calls"], prof_file))
                if head["rocprof_avg_launch_us"] else None,
                algorithmic_per_launch=head["algorithmic_gflop_per_launch"] if head["bound"] == "mfma" else head["algorithmic_mb_per_launch"],
                algorithmic_unit="GFLOP" if head["bound"] == "mfma" else "MB",
                algorithmic_mb_per_launch=head["algorithmic_mb_per_launch"], per_shape=head.get("per_shape"),
                classes=[{k: v for k, v in e.items() if k != "per_shape"} for e in classes],
                classes_note="every convolution-path kernel class of one forward, by time: live HIP-event us/step, algorithmic rate and its "
                             "fraction of the class's roof (mfma: 2.5 PFLOP/s; hbm: 8 TB/s), committed rocprofv3 all-launch average and PMC "
                             "traffic beside it where collected")
    return roof


def visible_gpus():
    """How many GPUs the ranks will see, WITHOUT touching the HIP / HSA runtime in this process (the launcher parent must stay
    GPU-free: torch.cuda.device_count() can fall back to hipGetDeviceCount). An explicit visibility list wins; else the KFD topology
    (a node with SIMDs is a GPU); None when neither says anything (the ranks then fail by themselves)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([t for t in v.split(",") if t.strip() != ""])
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(root):
            with open(os.path.join(root, node, "properties")) as f:
                for ln in f:
                    if ln.startswith("simd_count") and int(ln.split()[1]) > 0:
                        n += 1
        return n
    except (OSError, ValueError):
        return None


def launch_ranks(n, argv, stub):
    """`--gpus n` without a launcher: start the n ranks as child processes (one per GPU, rendezvous on 127.0.0.1) BEFORE this process
    touches a GPU - it never does (devices are counted from the environment / sysfs, `visible_gpus`) - relay rank 0's JSON line on stdout (everything else the ranks print goes to stderr) and return
    the worst child exit code. A rank that fails takes the others down (they would wait in a collective)."""
    if not stub and os.environ.get("PCV_BENCH_REHEARSE") != "1":
        have = visible_gpus()                                         # from the environment / sysfs: no HIP or HSA call in this process
        if have is not None and have < n:
            print("bench.py: --gpus {} but {} GPU(s) visible".format(n, have), file=sys.stderr)
            return 2
    limit = float(os.environ.get("PCV_BENCH_TIMEOUT_S", "1500"))     # a rank stuck in RCCL init or a collective must not hang the caller
    t_start = time.time()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    import tempfile
    procs, killed = [], set()
    out0 = tempfile.TemporaryFile(mode="w+")                          # rank 0's stdout (a pipe would have to be drained while polling)
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")             # dmabuf IPC: RCCL across processes needs it on this image
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out0 if r == 0 else sys.stderr))
    worst, line = 0, None
    term_at = None                                                    # when SIGTERM went out: SIGKILL after a grace period, then stop waiting
    try:
        pending = set(range(n))
        while pending:
            if term_at is None and time.time() - t_start > limit:
                print("bench.py: ranks {} still running after {:.0f} s (PCV_BENCH_TIMEOUT_S): terminating them".format(sorted(pending), limit),
                      file=sys.stderr)
                for o in pending:
                    killed.add(o)
                    procs[o].terminate()
                worst = worst or 124
                term_at = time.time()
            if term_at is not None and time.time() - term_at > TERM_GRACE_S:
                # a rank that ignores SIGTERM (stuck in a driver call, a library handler): kill it and stop waiting - the launcher
                # always returns
                print("bench.py: ranks {} ignored SIGTERM for {:.0f} s: killing them".format(sorted(pending), TERM_GRACE_S), file=sys.stderr)
                for o in pending:
                    procs[o].kill()
                break
            for r in sorted(pending):
                rc = procs[r].poll()
                if rc is None:
                    continue
                pending.discard(r)
                if rc != 0 and r not in killed:
                    worst = worst or rc                               # the first rank that failed by itself
                    for o in pending:                                 # exact children only; their exit codes are ours, not theirs
                        killed.add(o)
                        procs[o].terminate()
                    term_at = term_at or time.time()
            time.sleep(0.05)
        out0.seek(0)
        for ln in out0.read().splitlines():
            if ln.startswith('{"metric"'):
                line = ln
            elif ln.strip():
                print(ln, file=sys.stderr)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        out0.close()
    if worst == 0 and line is None:
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
        worst = 1
    if line is not None and worst == 0:
        print(line, flush=True)
    return worst


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="resnet50_bs256", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="auto", choices=["auto", "bf16", "fp16", "fp32"],
                    help="storage / MFMA type; auto = the family's 16-bit mode (bf16; fp16 for MobileNetV2 / V3 / EfficientNet, whose "
                         "bf16 logits miss the north-star 1e-2 - pytorchcv_amd.engine.compute_dtype_of); the JSON line states what ran")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of BASELINE configs 3 and 4 behind the headline")
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU batch (parity/debug only)")
    ap.add_argument("--graph", type=int, default=-1, help="replay the forward from a captured hipGraph (1), eager launches (0), "
                                                          "default: graph")
    ap.add_argument("--inflight", type=int, default=0, help="captured forwards in flight on alternating streams: 1 = one graph (steps "
                                                           "strictly one after the other), 2 = consecutive steps overlap, 0 = with --lanes 0: "
                                                           "time both launchers at capture and keep the faster one")
    ap.add_argument("--lanes", type=int, default=0, help="independent batch slices captured as parallel graph branches, so that "
                                                        "one slice's tile-schedule tails are filled by the other's kernels "
                                                        "(0: pytorchcv_amd.graph.auto_lanes, i.e. 2 from batch 64 up)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: the workload's batch per GPU (default); strong: --global-batch images split over the ranks (BASELINE config 5: 2048)")
    ap.add_argument("--global-batch", type=int, default=2048)
    args = ap.parse_args(argv)
    stub = os.environ.get("PCV_BENCH_STUB") == "1"
    # PCV_BENCH_PROFILE=1 (the rocprofv3 / PMC collection runs): nothing but full-batch eager forwards on one lane - no 8-image
    # forwards, no graph, no roofline / CPU passes - so that EVERY dispatch in the profiler's tables is a full-batch launch
    profile = os.environ.get("PCV_BENCH_PROFILE") == "1"

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args.gpus, argv, stub)                    # nothing above touched a GPU

    env = Env(stub)
    if env.world != args.gpus and env.rank == 0:
        print("bench.py: --gpus {} but the launcher started {} rank(s); reporting n_gpus = {}".format(args.gpus, env.world, env.world),
              file=sys.stderr)
    if stub and os.environ.get("PCV_BENCH_STUB_FAIL_RANK") == str(env.rank):
        env.fail(7, "rank {} fails on request (launcher self-test)".format(env.rank))
    if stub and os.environ.get("PCV_BENCH_STUB_HANG_RANK") == str(env.rank):
        time.sleep(3600)                               # (launcher self-test: a rank that never returns)
    tune_overrides = [kv for kv in os.environ.get("PCV_BENCH_TUNE", "").split(",") if kv]
    if not stub:
        for kv in tune_overrides:                    # dev only: "key=value,..." -> pcv_set_tuning; reported in the JSON line
            from pytorchcv_amd import _lib
            k, v = kv.split("=")
            _lib.check(_lib.lib().pcv_set_tuning(_lib.ctx_for(env.local_rank), k.encode(), int(v)), _lib.ctx_for(env.local_rank))

    res, ctx = run_workload(env, args.workload, args, args.steps, args.warmup, profile=profile)
    rank0_extras = env.rank == 0 and not stub and not profile
    roof = roofline_of(env, args.workload, ctx, max(3, min(args.steps, 10))) if rank0_extras else None
    lat = forward_latency(ctx) if (rank0_extras and args.graph != 0) else None
    cpu = cpu_baseline(ctx["model"], ctx["sd_cpu"]) if (rank0_extras and env.world == 1 and not args.no_cpu_baseline) else None
    others = None
    if rank0_extras and env.world == 1 and not args.no_other_configs and args.workload == "resnet50_bs256" and args.batch == 0:
        del ctx
        torch.cuda.empty_cache()
        others = {}
        for w in OTHER_CONFIGS:
            r, c = run_workload(env, w, args, OTHER_STEPS, OTHER_WARMUP)
            others[w] = dict(value=r["value"], unit="images/sec", ms_per_step=r["ms_per_step"], steps=OTHER_STEPS, warmup=OTHER_WARMUP, dtype=r["dtype"],
                             launch=r["launch"], rows_checked="every row of the timed batch equals the 8-image eager forward bit for bit",
                             roofline=compact_roofline(roofline_of(env, w, c, 3)), latency=forward_latency(c, samples=20))
            del r, c
            torch.cuda.empty_cache()

    if env.rank == 0:
        model, _, _ = WORKLOADS[args.workload]
        out = {
            "metric": "images/sec @224x224 ({} bs={}/GPU)".format(model, res["per_gpu_batch"]),
            "value_is": "steady-state throughput: images of the K timed steps / their wall time ({}); the one-forward latency figure of "
                        "SURVEY 8d is `latency`".format(res["launch"]),
            "value": res["value"],
            "unit": "images/sec",
            "n_gpus": env.world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"],
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": res["dtype"],
            "data": "synthetic (seeded N(0,1) images, seeded calibrated random-init weights of the named architecture)",
            "config": {"workload": args.workload, "per_gpu_batch": res["per_gpu_batch"], "global_batch": env.world * res["per_gpu_batch"],
                       "input": "fp32 NCHW 224x224 resident in HBM", "parallelism": "batch-sharded replicas x{}".format(env.world),
                       "launch": res["launch"], "launcher_choice": res.get("launcher_choice"),
                       "launchers_timed_at_capture_img_per_s": res.get("launchers_timed"),
                       "tuning_overrides": tune_overrides or None},
            "roofline": roof,
            "latency": lat,
            "cpu_baseline": cpu,
            "weights_broadcast": res["weights_broadcast"],
            "other_configs": others,
        }
        if stub:
            out["stub"] = "launcher / collective self-test on CPU with gloo and a stand-in forward: NOT a measurement"
        if env.rehearse:
            out["rehearsal"] = "PCV_BENCH_REHEARSE=1: {} ranks share cuda:0, collectives through gloo: NOT a measurement".format(env.world)
        if profile:
            out["profile_mode"] = "PCV_BENCH_PROFILE=1: full-batch eager forwards only (for rocprofv3); value is not the benchmark figure"
        print(json.dumps(out), flush=True)
    if env.use_dist:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
