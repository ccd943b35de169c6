import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import myzkp_amd as mz
mz.init(0); L = mz.lib()
dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
fid = int(sys.argv[1]); lg = int(sys.argv[2]); nl = 4 if fid == 0 else 2
n = 1 << lg
vin = torch.empty(n*nl, dtype=torch.int64, device=dev); vout = torch.empty(n*nl, dtype=torch.int64, device=dev)
L.mzk_synth_field_dev(fid, ctypes.c_uint64(5), ctypes.c_size_t(n), ctypes.c_void_p(vin.data_ptr()), st)
root = mz.to_limbs([mz.root_of_unity(fid, lg)], nl)
for _ in range(4):
    L.mzk_ntt_dev(fid, root.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(vin.data_ptr()), ctypes.c_void_p(vout.data_ptr()), ctypes.c_size_t(n), 0, st)
torch.cuda.synchronize()
