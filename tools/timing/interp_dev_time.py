"""mzk_fast_interpolate_batch_dev alone (16 M128 registers of 2^14 - 3 points, trace already in HBM) and the upload of the trace alone:
    python tools/timing/interp_dev_time.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, orc, myzkp_amd as mz
mz.init(0)
fid, p = orc.M128, orc.MOD[orc.M128]
lgt, R = 14, 16
cycles = (1 << lgt) - 3
om = orc.root_of(fid, lgt)
dom, acc = [], 1
for _ in range(cycles):
    dom.append(acc); acc = acc * om % p
dom = orc.to_limbs(dom, 2)
trace = np.stack([orc.synth_vector(fid, 100 + r, cycles) for r in range(R)])
d_tr = torch.from_numpy(np.ascontiguousarray(trace).view(np.int64).reshape(-1)).cuda()
d_out = torch.zeros(R * cycles * 2, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3): mz.fast_interpolate_batch_dev(fid, dom, d_tr.data_ptr(), R, om, 1 << lgt, d_out.data_ptr(), st)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): mz.fast_interpolate_batch_dev(fid, dom, d_tr.data_ptr(), R, om, 1 << lgt, d_out.data_ptr(), st)
torch.cuda.synchronize(); print("dev call, trace already in HBM: %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
t0 = time.perf_counter()
for _ in range(20):
    x = torch.from_numpy(np.ascontiguousarray(trace).view(np.int64).reshape(-1)).cuda(); torch.cuda.synchronize()
print("upload of the trace alone: %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
