import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tests/golden")
import torch, util, pytorchcv_amd
from pytorchcv_amd import engine
from pytorchcv_amd.models.mobilenetv2 import LinearBottleneck
from pytorchcv_amd.models.common.conv import mbconv_chain
from pytorchcv_amd.models.common.activ import create_activation_layer
dev = torch.device("cuda", 0)
for (N, H, W, Cin, Cout, act, dtype) in [(5, 14, 14, 32, 32, "hswish", "fp16"), (5, 14, 14, 32, 32, "hswish", "bf16"), (5, 14, 14, 32, 24, "hswish", "fp16"), (2, 28, 28, 32, 32, "hswish", "fp16"), (5, 14, 14, 32, 32, "relu", "fp16")]:
    unit = LinearBottleneck(in_channels=Cin, out_channels=Cout, stride=1, expansion=False, remove_exp_conv=False, activation=(lambda: create_activation_layer(act))).eval()
    unit.load_state_dict(util.synth_state_dict(unit.state_dict(), seed=31))
    unit = pytorchcv_amd.set_compute_dtype(unit.to(dev), dtype)
    tdt = {"bf16": torch.bfloat16, "fp16": torch.float16}[dtype]
    x = util.synth_input(N, Cin, H, W, seed=8).to(tdt)
    a = engine.NHWC(x.permute(0, 2, 3, 1).contiguous().to(dev), N, H, W, Cin)
    res = a if unit.residual else None
    with torch.no_grad():
        outs = {}
        for name, tune in (("xl", dict(mbr=1, mbr_xl=1)), ("nx", dict(mbr=1, mbr_xl=0)), ("mbw", dict(mbr=0))):
            with util.tuning(**tune):
                outs[name] = mbconv_chain(unit.conv1, unit.conv2, unit.conv3, a, residual=res).t.float().cpu()
    d = (outs["nx"] - outs["mbw"]).abs()
    bad = d > 0.02
    print(N, H, Cin, Cout, act, dtype, "xl-mbw max", float((outs["xl"] - outs["mbw"]).abs().max()), "nx-mbw max", float(d.max()), "bad", int(bad.sum()), "of", d.numel())
    if bad.any():
        idx = bad.nonzero()
        print("  images", sorted(set(idx[:, 0].tolist())), "rows", sorted(set(idx[:, 1].tolist())), "cols", sorted(set(idx[:, 2].tolist())), "ch", sorted(set(idx[:, 3].tolist()))[:40])
        i = idx[0].tolist()
        print("  first", i, float(outs["nx"][tuple(i)]), float(outs["mbw"][tuple(i)]), "x there", float(x[i[0], i[3], i[1], i[2]]) if Cin == Cout else None)
    if bad.any() and Cin == Cout:
        xn = x.float().permute(0, 2, 3, 1)      # NHWC
        D = outs["nx"] - outs["mbw"]
        for shift in (7, -7, 1, -1, 14):
            n_ok = 0; n_all = 0
            for i in idx[:200].tolist():
                n_, h_, w_, c_ = i
                h2 = h_ + shift
                if 0 <= h2 < H:
                    n_all += 1
                    if abs(float(D[n_, h_, w_, c_]) - (float(xn[n_, h2, w_, c_]) - float(xn[n_, h_, w_, c_]))) < 0.01: n_ok += 1
            print("   residual taken from row h%+d explains %d of %d" % (shift, n_ok, n_all))
        # other image?
        for dn in (1,):
            n_ok = n_all = 0
            for i in idx[:200].tolist():
                n_, h_, w_, c_ = i
                if n_ + dn < N:
                    n_all += 1
                    if abs(float(D[n_, h_, w_, c_]) - (float(xn[n_ + dn, h_, w_, c_]) - float(xn[n_, h_, w_, c_]))) < 0.01: n_ok += 1
            print("   residual from image n+1 same pixel explains %d of %d" % (n_ok, n_all))
        print("   residual missing explains", sum(1 for i in idx[:200].tolist() if abs(float(D[tuple(i)]) + float(xn[tuple(i)])) < 0.01), "of", min(200, len(idx)))
