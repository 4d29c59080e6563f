"""Write / read / copy stream rates of this box (scripts/dbg/fill.hip): the ceiling the fused STFT epilogue (3.93 GB of 16-byte stores per launch) is priced against."""
import ctypes, os, sys
import torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libfill.so"))
lib.fill_bench.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
region = 294912
nbytes = region * 13312                     # the fused launch of the bench: 52 chunks x 256 frames
a = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
b = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
def run(mode, grid, name):
    us = ctypes.c_float()
    rc = lib.fill_bench(a.data_ptr(), b.data_ptr(), nbytes, mode, grid, 20, ctypes.byref(us))
    assert rc == 0, rc
    moved = nbytes * (2 if mode == 5 else 1)
    print(f"{name:44s} grid {grid:6d}: {us.value:8.1f} us  {moved / us.value / 1e6:6.2f} TB/s", flush=True)
for g in (1024, 2048, 4096, 8192, 16384):
    run(0, g, "fill, grid-stride 16 B/lane")
run(1, 0, "fill, 128-thread WG per 288 KiB region (126)")
run(2, 0, "fill, 128-thread WG per 288 KiB region (128)")
run(3, 0, "fill, 256-thread WG per 288 KiB region")
for g in (2048, 8192):
    run(4, g, "read, grid-stride 16 B/lane")
for g in (2048, 8192):
    run(5, g, "copy, grid-stride 16 B/lane (read + write)")
for g in (512, 2048, 106496):
    run(6, g, "fill, m0 epilogue shape (64 B + 32 B per pixel)")
    run(7, g, "fill, m0 tile rows as linear 16 B/lane stores")
import time
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): a.zero_()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
print(f"torch zero_: {dt * 1e6:.1f} us {nbytes / dt / 1e12:.2f} TB/s")
