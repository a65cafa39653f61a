import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/tests/golden')
import util, pytorchcv_amd
from oracle import refblocks
dev = torch.device('cuda', 0)
print(torch.cuda.get_device_name(0))
for case in util.BLOCK_CASES:
    for dt in ('fp32', 'bf16'):
        try:
            sd, x = util.block_state_and_input(case)
            blk = util.build_block(case); blk.load_state_dict(sd); blk = pytorchcv_amd.set_compute_dtype(blk.to(dev), dt)
            with torch.no_grad(): y = blk(x.to(dev))
            torch.cuda.synchronize()
            g = util.block_golden(case)
            print('%-28s %-5s err vs golden %.3e  (absmax %.2f)' % (case['name'], dt, float((y.cpu()-g).abs().max()), float(g.abs().max())), flush=True)
        except Exception as e:
            print('%-28s %-5s EXC %r' % (case['name'], dt, e), flush=True)
