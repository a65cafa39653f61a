#!/usr/bin/env python
"""
bench.py - images/sec of the MI355X conv-net inference hot path (BASELINE.json metric), one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload resnet50_bs256]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one forward of the workload's batch through `get_model(name)` (fp32 NCHW input resident in HBM -> NHWC bf16 ->
fused HIP kernels -> fp32 logits) followed, for N > 1, by the all-gather of the logits. Weak scaling: every rank runs the
full per-GPU batch. Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (3x3 dense convolutions
of ResNet-50: algorithmic FLOPs / HIP-event time per launch vs the 2.5 PFLOP/s dense bf16 MFMA peak; for MobileNetV2 the
depthwise kernels vs the 8 TB/s HBM peak) and `cpu_baseline` (the oracle's torch-fp32 CPU forward on the host cores).
"""

import os
import sys
import json
import time
import argparse

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WORKLOADS = {
    # name: (model, per-GPU batch, kernel class for the roofline, bound)
    "resnet50_bs256": ("resnet50", 256, "dense3x3", "mfma"),
    "mobilenetv2_w1_bs512": ("mobilenetv2_w1", 512, "fused_unit", "hbm"),      # the fused inverted-residual units: 40 % of the step
    "resnext101_32x4d_bs256": ("resnext101_32x4d", 256, "grouped3x3", "hbm"),
    "resnet18_bs256": ("resnet18", 256, "dense3x3", "mfma"),
    "mobilenetv3_large_w1_bs512": ("mobilenetv3_large_w1", 512, "depthwise", "hbm"),
    "efficientnet_b0_bs256": ("efficientnet_b0", 256, "depthwise", "hbm"),
    "vgg16_bs128": ("vgg16", 128, "dense3x3", "mfma"),
    "seresnet50_bs256": ("seresnet50", 256, "dense3x3", "mfma"),
    "seresnext50_32x4d_bs256": ("seresnext50_32x4d", 256, "grouped3x3", "hbm"),
}
MFMA_PEAK_TFLOPS = 2500.0     # dense bf16/fp16, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
HBM_PEAK_GBS = 8000.0         # HBM3E spec peak, MI355X_MICROARCH.md "HBM3E peak BW"


def calib_for(model):
    p = os.path.join(ROOT, "tests", "golden", "calib_{}.json".format(model))
    if not os.path.exists(p):
        return None
    with open(p) as f:
        return {k: tuple(v) for k, v in json.load(f).items()}


class LaunchTimer(object):
    """Brackets every ConvRunner launch of one kernel class with events on the launch stream (torch's current stream,
    which is the stream handed to the C ABI) and accumulates algorithmic FLOPs / bytes per launch."""
    def __init__(self, klass):
        self.klass = klass
        self.records = []      # (start_event, end_event, flops, bytes, tag)
        self._orig = None

    @staticmethod
    def classify(runner, d):
        if runner.depthwise:
            return "depthwise"
        if d.groups > 1:
            return "grouped3x3" if d.kh == 3 else "grouped"
        if d.kh == 3 and d.kw == 3 and d.x_cpitch != 4:
            return "dense3x3"
        if d.kh == 1 and d.kw == 1:
            return "dense1x1"
        return "stem"

    def __enter__(self):
        from pytorchcv_amd import engine
        timer = self
        self._orig = engine.ConvRunner._launch

        def timed(runner, x, d, residual, out=None, gate=None):
            if out is not None or timer.classify(runner, d) != timer.klass:
                return timer._orig(runner, x, d, residual, out, gate)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            y = timer._orig(runner, x, d, residual, None, gate)
            e.record()
            es = x.t.element_size()
            cin_g = d.Cin // d.groups
            flops = 2.0 * y.N * y.H * y.W * d.Cout * cin_g * d.kh * d.kw
            nbytes = (x.N * x.H * x.W * d.Cin + y.N * y.H * y.W * d.Cout * (2 if residual is not None else 1)) * es \
                + d.Cout * cin_g * d.kh * d.kw * es
            timer.records.append((s, e, flops, nbytes, "{}x{}x{}->{} k{} s{}".format(x.H, x.W, d.Cin, d.Cout, d.kh, d.stride_h)))
            return y

        engine.ConvRunner._launch = timed
        # "fused_unit": the one-launch inverted-residual units (pcv_mbconv_fused). Algorithmic bytes = the unit's input and output
        # once (the skip tensor IS the input) + the three weight sets; FLOPs = expand + depthwise + project.
        self._orig_mb = engine.mbconv_fused

        def timed_mb(exp, exp_act, dw, dw_act, proj, proj_act, x, residual, post_act):
            if timer.klass != "fused_unit":
                return timer._orig_mb(exp, exp_act, dw, dw_act, proj, proj_act, x, residual, post_act)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            y = timer._orig_mb(exp, exp_act, dw, dw_act, proj, proj_act, x, residual, post_act)
            e.record()
            if y is None:
                return None
            es = x.t.element_size()
            cmid, cout = dw.conv.out_channels, proj.conv.out_channels
            px_in, px_out = x.N * x.H * x.W, y.N * y.H * y.W
            flops = 2.0 * ((px_in * x.C * cmid if exp is not None else 0) + px_out * cmid * 9 + px_out * cmid * cout)
            nbytes = (px_in * x.C + px_out * cout + (x.C * cmid if exp is not None else 0) + 9 * cmid + cmid * cout) * es
            timer.records.append((s, e, flops, nbytes, "{}x{}x{}->{}->{} s{}".format(x.H, x.W, x.C, cmid, cout, dw.conv.stride[0])))
            return y

        engine.mbconv_fused = timed_mb
        return self

    def __exit__(self, *a):
        from pytorchcv_amd import engine
        engine.ConvRunner._launch = self._orig
        engine.mbconv_fused = self._orig_mb

    def summary(self):
        torch.cuda.synchronize()
        n = len(self.records)
        if n == 0:
            return None
        ms = [s.elapsed_time(e) for s, e, _, _, _ in self.records]
        total_ms = sum(ms)
        flops = sum(r[2] for r in self.records)
        nbytes = sum(r[3] for r in self.records)
        per_shape = {}
        for (s, e, f, b, tag), t in zip(self.records, ms):
            a = per_shape.setdefault(tag, [0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += t
            a[2] += f
            a[3] += b
        return dict(launches=n, avg_ms=total_ms / n, tflops=flops / (total_ms * 1e-3) / 1e12,
                    gbs=nbytes / (total_ms * 1e-3) / 1e9, flops_per_launch=flops / n, bytes_per_launch=nbytes / n,
                    per_shape={k: dict(launches=v[0], avg_us=1e3 * v[1] / v[0], tflops=v[2] / (v[1] * 1e-3) / 1e12,
                                       gbs=v[3] / (v[1] * 1e-3) / 1e9) for k, v in per_shape.items()})


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


def rocprof_class_us(workload, klass):
    """Average launch duration of the roofline kernel class from the COMMITTED rocprofv3 --kernel-trace --stats summary of this
    command (profiles/, single batch lane), beside the live HIP-event figure: (us, file) or (None, None)."""
    import csv
    import glob
    names = {"dense3x3": ("d3q_kernel", "false, 9>"), "depthwise": ("dwconv_kernel", "dwconv5_kernel"),
             "grouped3x3": ("gconv3x3_kernel", "gconv3x3r_kernel"), "fused_unit": ("mbw_kernel", "mbconv_kernel")}.get(klass)
    if not names:
        return None, None
    # preferred: one full-batch forward in dispatch order (tests/tools/trace_summary.py of the same rocprofv3 run) - the --stats
    # table also counts the 8-image warm-up forward's launches
    traces = sorted(glob.glob(os.path.join(ROOT, "profiles", "r02_{}_*per_launch*.txt".format(workload))))
    if traces:
        tot = calls = 0
        with open(traces[-1]) as f:
            for line in f:
                if any(n in line for n in names) and ", true>(D3Params)" not in line and line.rstrip().endswith("us"):
                    tot += float(line.split()[-2])
                    calls += 1
        if calls:
            return round(tot / calls, 2), os.path.relpath(traces[-1], ROOT)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r02_{}_*kernel_stats*.csv".format(workload))))
    if not files:
        return None, None
    tot = calls = 0
    with open(files[-1]) as f:
        for row in csv.DictReader(f):
            if any(n in row["Name"] for n in names) and ", true>(D3Params)" not in row["Name"]:      # (d3q's 1x1 mode is another class)
                tot += int(row["TotalDurationNs"])
                calls += 1 * int(row["Calls"])
    return (round(tot / calls / 1e3, 2), os.path.relpath(files[-1], ROOT)) if calls else (None, None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="resnet50_bs256", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="auto", choices=["auto", "bf16", "fp16", "fp32"],
                    help="storage / MFMA type; auto = the family's 16-bit mode (bf16; fp16 for MobileNetV2 / V3 / EfficientNet, whose "
                         "bf16 logits miss the north-star 1e-2 - pytorchcv_amd.engine.compute_dtype_of); the JSON line states what ran")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU batch (parity/debug only)")
    ap.add_argument("--graph", type=int, default=-1, help="replay the forward from a captured hipGraph (1), eager launches (0), "
                                                          "default: graph")
    ap.add_argument("--lanes", type=int, default=0, help="independent batch slices captured as parallel graph branches, so that "
                                                        "one slice's tile-schedule tails are filled by the other's kernels "
                                                        "(0: pytorchcv_amd.graph.auto_lanes, i.e. 2 from batch 64 up)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus {} needs `python -m torch.distributed.run --nproc-per-node {}`".format(args.gpus, args.gpus))
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("PCV_BENCH_FORCE_DIST") == "1"    # (forcing it at world size 1 rehearses the RCCL calls)
    if use_dist:
        # the process group comes up first: nothing of ours touches the GPU before RCCL has bound this rank to its device
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=dev)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)

    import pytorchcv_amd
    from pytorchcv_amd.model_provider import get_model
    from pytorchcv_amd.synth import synth_state_dict, synth_input
    from pytorchcv_amd.parallel import ShardedInference, broadcast_module_state, broadcast_packed_state

    for kv in filter(None, os.environ.get("PCV_BENCH_TUNE", "").split(",")):       # dev only: "key=value,..." -> pcv_set_tuning
        from pytorchcv_amd import _lib
        k, v = kv.split("=")
        _lib.check(_lib.lib().pcv_set_tuning(_lib.ctx_for(local_rank), k.encode(), int(v)), _lib.ctx_for(local_rank))
    model, batch, klass, bound = WORKLOADS[args.workload]
    if args.batch > 0:
        batch = args.batch
    net = get_model(model).eval()
    sd_cpu = None
    if rank == 0:
        sd_cpu = synth_state_dict(net.state_dict(), seed=1234, calib=calib_for(model))
        net.load_state_dict(sd_cpu, strict=True)
    net = pytorchcv_amd.set_compute_dtype(net.to(dev), args.dtype)
    from pytorchcv_amd import engine
    dtype = engine.compute_dtype_of(net)               # what "auto" resolved to for this family: reported in the JSON line
    ovf0 = engine.fp16_overflow_count(dev) if dtype == "fp16" else 0

    # synthetic N(0,1)-like images: 8 distinct seeded images tiled to the batch (performance is data independent)
    base = synth_input(8, seed=rank).to(dev)
    x = base.repeat((batch + 7) // 8, 1, 1, 1)[:batch].contiguous()
    with torch.no_grad():
        net(base)                                    # builds and packs every layer (rank 0: the real weights)
    bcast = None
    if use_dist:
        # RCCL broadcast of rank 0's PACKED inference state over xGMI (bf16 arenas + fp32 scale/shift: half the fp32 state;
        # the receiving ranks do not re-pack)
        bcast = broadcast_packed_state(net, src=0)
    with torch.no_grad():
        y_ref8 = net(base).clone()                   # the 8 distinct images on their own: what every row of the timed batch must equal
    use_graph = args.graph != 0
    fwd = net
    if use_graph:
        from pytorchcv_amd.graph import capture
        try:
            fwd = capture(net, x, own_input=True, lanes=args.lanes if args.lanes > 0 else None)    # ~60 kernel launches replayed by one hipGraphLaunch; x is the static input
        except Exception as e:                       # noqa: BLE001 - a run that cannot capture is not the benchmarked configuration
            print("bench.py: hipGraph capture failed ({}); rerun with --graph 0 for eager launches".format(e), file=sys.stderr)
            if use_dist:
                dist.destroy_process_group()
            sys.exit(4)
    runner = ShardedInference(fwd)

    def step():
        y = runner.run_local(x)
        return runner.gather_all(y) if use_dist else y

    for _ in range(args.warmup):
        step()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # the timed forward computes what the small-batch forward computes: every row of the full batch (multi-round tile
    # schedules, graph lanes) equals the 8-image eager result bit for bit; a mismatch fails the run
    y_local = y[rank * batch:(rank + 1) * batch] if use_dist else y
    want = y_ref8.repeat((batch + 7) // 8, 1)[:batch]
    if dtype == "fp16" and engine.fp16_overflow_count(dev) != ovf0:
        print("bench.py: the fp16 range guard counted an overflow during the run", file=sys.stderr)
        if use_dist:
            dist.destroy_process_group()
        sys.exit(3)
    if not bool(torch.isfinite(y_local).all()) or not torch.equal(y_local, want):
        bad = int((y_local != want).any(1).sum())
        print("bench.py: {} of {} rows of the timed batch differ from the 8-image forward".format(bad, batch), file=sys.stderr)
        if use_dist:
            dist.destroy_process_group()
        sys.exit(3)

    # HBM traffic of the class from the PMC counters: cannot be read inside this process (rocprofv3 --pmc is its own run);
    # the per-launch figure of the committed passes over this same command travels in profiles/pmc_traffic.json.
    pmc = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pmc = json.load(f).get(args.workload)
        if pmc is not None and (pmc.get("dtype") != dtype or args.batch > 0 or pmc.get("kernel_class") != klass):
            pmc = None
    except (OSError, ValueError):
        pmc = None

    # per-launch HIP-event timing of the roofline kernel class (separate pass, not part of `value`)
    roof = None
    if rank == 0:
        with torch.no_grad():                        # (the same kernel mix `value` runs: fused units stay fused)
            net(x)                                   # untimed: first use of the per-layer kernels this pass takes (code load, packing)
        torch.cuda.synchronize()
        with LaunchTimer(klass) as lt:
            for _ in range(max(3, min(args.steps, 10))):
                with torch.no_grad():
                    net(x)                           # eager: events bracket every launch of the class
        s = lt.summary()
        if s is not None:
            if bound == "mfma":
                roof = dict(bound="mfma", achieved=round(s["tflops"], 2), peak=MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                            frac=round(s["tflops"] / MFMA_PEAK_TFLOPS, 4), traffic=None)
            else:
                roof = dict(bound="hbm", achieved=round(s["gbs"], 1), peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=round(s["gbs"] / HBM_PEAK_GBS, 4), traffic=None)
            rp_us, rp_file = rocprof_class_us(args.workload, klass)
            roof.update(kernel_class=klass, launches_per_step=s["launches"] // max(3, min(args.steps, 10)),
                        avg_launch_us=round(1e3 * s["avg_ms"], 2),
                        avg_launch_us_source="live: HIP events around every eager launch of the class on the launch stream (this run)",
                        rocprof_avg_launch_us=rp_us,
                        rocprof_source=("committed rocprofv3 --kernel-trace of this command with --lanes 1 (one full-batch forward): " + rp_file) if rp_file else None,
                        algorithmic_per_launch=(round(s["flops_per_launch"] / 1e9, 3) if bound == "mfma"
                                                else round(s["bytes_per_launch"] / 1e6, 3)),
                        algorithmic_unit="GFLOP" if bound == "mfma" else "MB",
                        algorithmic_mb_per_launch=round(s["bytes_per_launch"] / 1e6, 3),
                        per_shape={k: {kk: round(vv, 2) for kk, vv in v.items()} for k, v in s["per_shape"].items()})
            if pmc is not None:
                roof.update(traffic=pmc["traffic_mb_per_launch"], traffic_unit="MB per launch (HBM read + write)",
                            traffic_source="profiles/pmc_traffic.json: " + pmc["note"])

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(model, sd_cpu)

    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        out = {
            "metric": "images/sec @224x224 ({} bs={}/GPU)".format(model, batch),
            "value": round(world * batch * args.steps / elapsed, 1),
            "unit": "images/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": dtype,
            "data": "synthetic (seeded N(0,1) images, seeded calibrated random-init weights of the named architecture)",
            "config": {"workload": args.workload, "per_gpu_batch": batch, "global_batch": world * batch,
                       "input": "fp32 NCHW 224x224 resident in HBM", "parallelism": "batch-sharded replicas x{}".format(world),
                       "launch": ("hipGraph replay, {} batch lane(s) as parallel branches".format(fwd.lanes) if use_graph else "eager")},
            "roofline": roof,
            "cpu_baseline": cpu,
            "weights_broadcast": (dict(messages=bcast[0], bytes=bcast[1], what="packed inference state (RCCL broadcast from rank 0)")
                                  if bcast else None),
        }
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
