"""Dev tool (CPU): where does the distance between the bf16 forward and the fp32 reference forward come from?
The oracle's 16-bit mode rounds (w) the MFMA weight operands, (a) every activation tensor stored to HBM - of which (s) the
residual stream written by a unit's last convolution - and (p) the pooled features; this script switches those rounding points
off one at a time on the golden fixtures (4 images, calibrated synthetic weights) and prints max|logit - fp32 golden|.
Usage: python tests/tools/bf16_drift.py [model ...]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import util
from oracle import refnet


class Abl(refnet.Quant):
    def __init__(self, dtype, w=True, a=True, s=True, p=True, c=True, dw=True, w2=False):
        super().__init__(dtype)
        self.w, self.a, self.s, self.p, self.c, self.dw, self.w2 = w, a, s, p, c, dw, w2

    def rc(self, t):                                                    # the classifier (product: fp32)
        return self.r(t) if self.c else t

    def rw(self, t):
        if t.dim() == 4 and t.shape[1] == 1:                            # depthwise weights (VALU kernel: fp32 costs nothing)
            return self.r(t) if self.dw else t
        if self.w2:                                                     # two-term split: W ~ bf16(W) + bf16(W - bf16(W))
            hi = self.r(t)
            return hi + self.r(t - hi)
        return self.r(t) if self.w else t

    def ro(self, y, is_unit_output=False):
        if is_unit_output:
            return self.r(y) if (self.a and self.s) else y
        return self.r(y) if self.a else y

    def rp(self, f):
        return self.r(f) if self.p else f


VARIANTS = [
    ("everything rounded (the round-1 pipeline)", dict()),
    ("fp32 head (= the GPU pipeline now)", dict(p=False, c=False)),
    ("weights NOT rounded", dict(w=False, c=False)),
    ("activations NOT rounded", dict(a=False, p=False)),
    ("pooled features + classifier weights fp32", dict(p=False, c=False)),
    ("residual stream fp32 (unit outputs unrounded)", dict(s=False)),
    ("residual stream + pool/classifier fp32", dict(s=False, p=False, c=False)),
    ("pool/classifier fp32 + depthwise weights fp32", dict(p=False, c=False, dw=False)),
    ("pool/classifier fp32 + dw fp32 + dense weights as hi+lo", dict(p=False, c=False, dw=False, w2=True)),
    ("dense weights as hi+lo only", dict(w2=True)),
]
models = sys.argv[1:] or ["resnet18", "resnet50", "mobilenetv2_w1", "resnext101_32x4d"]
torch.set_num_threads(8)
for name in models:
    logits, ids = util.model_golden(name)
    sd = util.model_state(name)
    x = util.images(ids)
    print(name, "(|logits| max {:.2f}, std {:.2f})".format(float(logits.abs().max()), float(logits.std())))
    for dt in ("bf16", "fp16"):
        for label, kw in (VARIANTS if dt == "bf16" else VARIANTS[:1]):
            real = refnet.Quant
            refnet.Quant = lambda d, _kw=kw, _dt=dt: Abl(_dt if d is not None else None, **_kw)     # forward() builds its own Quant
            try:
                y = refnet.forward(name, sd, x, quant=dt)
            finally:
                refnet.Quant = real
            d = (y - logits).abs()
            print("   {:5s} {:48s} max {:.2e}  rms {:.2e}  top-1 equal: {}".format(dt, label, float(d.max()), float(d.pow(2).mean().sqrt()),
                                                                        bool(torch.equal(y.argmax(1), logits.argmax(1)))))
