"""Where a pageable host buffer's transfer time goes on this box: (a) one thread's memcpy pageable -> pinned, (b) pinned -> device
DMA, (c) the runtime's own pageable -> device path, (d) the same three for the way back; 32 MiB each."""
import time, numpy as np, torch
n = 32 << 20
page = np.random.randint(0, 255, n, dtype=np.uint8)
pin = torch.empty(n, dtype=torch.uint8, pin_memory=True)
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
pin_np = pin.numpy()
tpage = torch.from_numpy(page)


def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return n / ((time.perf_counter() - t0) / reps) / 1e9


print("memcpy pageable -> pinned, one thread : %6.1f GB/s" % t(lambda: np.copyto(pin_np, page)))
print("memcpy pinned -> pageable, one thread : %6.1f GB/s" % t(lambda: np.copyto(page, pin_np)))
print("DMA pinned -> device                  : %6.1f GB/s" % t(lambda: dev.copy_(pin, non_blocking=True)))
print("DMA device -> pinned                  : %6.1f GB/s" % t(lambda: pin.copy_(dev, non_blocking=True)))
print("runtime pageable -> device            : %6.1f GB/s" % t(lambda: dev.copy_(tpage)))
print("runtime device -> pageable            : %6.1f GB/s" % t(lambda: tpage.copy_(dev)))
torch.set_num_threads(8)
a = torch.empty(n, dtype=torch.uint8); 
print("torch copy pageable -> pinned, 8 thr  : %6.1f GB/s" % t(lambda: pin.copy_(tpage)))
