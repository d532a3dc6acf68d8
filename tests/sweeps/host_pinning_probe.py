import time, torch, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tomahawk_amd as T
lib = T.load_library()
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
def t(f):
    t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); return time.perf_counter() - t0, r
for mb in (64, 256, 1024, 2048):
    n = mb << 20
    p = ctypes.c_void_p()
    dt, _ = t(lambda: lib.twk_hip_host_alloc(ctypes.c_size_t(n), ctypes.byref(p)))
    print(f"twk_hip_host_alloc {mb} MB: {dt*1e3:.1f} ms ({mb/1024/dt:.1f} GB/s)")
    dt2, _ = t(lambda: lib.twk_hip_host_free(p))
    print(f"  free: {dt2*1e3:.1f} ms")
# parallel pinning from several threads
import threading
def alloc_one(res, i, n):
    p = ctypes.c_void_p(); lib.twk_hip_host_alloc(ctypes.c_size_t(n), ctypes.byref(p)); res[i] = p
for nt in (4, 8):
    res = [None] * nt
    t0 = time.perf_counter()
    th = [threading.Thread(target=alloc_one, args=(res, i, 256 << 20)) for i in range(nt)]
    [x.start() for x in th]; [x.join() for x in th]
    dt = time.perf_counter() - t0
    print(f"{nt} threads x 256 MB pinned allocs in parallel: {dt*1e3:.1f} ms ({nt*0.25/dt:.1f} GB/s)")
    for p in res: lib.twk_hip_host_free(p)
# copy rates
d = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
hp = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
hg = torch.empty(1 << 30, dtype=torch.uint8); hg.fill_(1)
for name, h in (("pinned", hp), ("pageable", hg)):
    for rep in range(3):
        dt, _ = t(lambda: d.copy_(h, non_blocking=True))
    print(f"H2D 1 GiB from {name}: {dt*1e3:.1f} ms ({1.0737/dt:.1f} GB/s)")
# hipHostRegister of existing memory
hip = ctypes.CDLL("libamdhip64.so")
buf = torch.empty(1 << 30, dtype=torch.uint8); buf.fill_(2)
dt, rc = t(lambda: hip.hipHostRegister(ctypes.c_void_p(buf.data_ptr()), ctypes.c_size_t(1 << 30), 0))
print(f"hipHostRegister 1 GiB of touched memory: rc={rc} {dt*1e3:.1f} ms")
dt, _ = t(lambda: d.copy_(buf, non_blocking=True)); dt, _ = t(lambda: d.copy_(buf, non_blocking=True))
print(f"H2D 1 GiB from registered: {dt*1e3:.1f} ms ({1.0737/dt:.1f} GB/s)")
hip.hipHostUnregister(ctypes.c_void_p(buf.data_ptr()))
print("cpus", os.cpu_count())
