"""Kernel timeline of ONE Merkle commit of 2^16 M128 elements for rocprofv3 --kernel-trace (summary: prints the kernel
sequence of the last commit):  rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/timing/merkle_trace.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import myzkp_amd as mz
mz.init(0); L = mz.lib()
dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = 1 << lg
d = torch.zeros(n * 2, dtype=torch.int64, device=dev)
assert L.mzk_synth_field_dev(1, ctypes.c_uint64(5), ctypes.c_size_t(n), ctypes.c_void_p(d.data_ptr()), st) == 0
root = (ctypes.c_uint8 * 48)(); ln = ctypes.c_size_t()
for _ in range(10):
    assert L.mzk_merkle_commit_field_dev(1, ctypes.c_void_p(d.data_ptr()), ctypes.c_size_t(n), root, ctypes.c_size_t(48), ctypes.byref(ln), st) == 0
torch.cuda.synchronize()
