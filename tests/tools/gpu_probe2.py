import sys, time, faulthandler
faulthandler.dump_traceback_later(100, exit=True)
t0 = time.time()
import torch
print('torch imported %.1fs' % (time.time() - t0), flush=True)
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/tests/golden')
import util, pytorchcv_amd
from pytorchcv_amd import engine, _lib
dev = torch.device('cuda', 0)
print(torch.cuda.get_device_name(0), flush=True)
x = torch.randn(2, 32, 10, 10, device=dev)
torch.cuda.synchronize(); print('alloc ok', flush=True)
ctx = _lib.ctx_for(0); print('ctx ok', flush=True)
h = engine.from_nchw(x, 'fp32', stem=False); torch.cuda.synchronize(); print('from_nchw ok', h.t.shape, flush=True)
back = engine.to_nchw(h); torch.cuda.synchronize(); print('roundtrip err', float((back - x).abs().max()), flush=True)
case = util.BLOCK_CASES[0]
sd, xc = util.block_state_and_input(case)
blk = util.build_block(case); blk.load_state_dict(sd); blk = pytorchcv_amd.set_compute_dtype(blk.to(dev), 'fp32')
h = engine.from_nchw(xc.to(dev), 'fp32', stem=False)
r = engine.ConvRunner(blk.conv, blk.bn)
d = r.desc(h, 1, 0, False)
r.prepare(h, d); torch.cuda.synchronize(); print('prepare ok', r.packed.shape, flush=True)
y = r._launch(h, d, None); torch.cuda.synchronize(); print('launch ok', flush=True)
g = util.block_golden(case)
print('err', float((engine.to_nchw(y).cpu() - g).abs().max()), flush=True)
