import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import mvtrim_amd as m
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
lib = m.load_library()
s = m.MotionScanner(m.ScanParams.from_config(1920, 1080), 0)
frame = 32640 * 40
buf = torch.empty(15837 * frame, dtype=torch.uint8, device=dev); buf.zero_()
nbytes = buf.numel()
def rate(idle, chunk=frame):
    st = torch.cuda.current_stream(dev).cuda_stream
    for _ in range(2):
        m._abi.check(lib.mtgpu_debug_read_ceiling_shape(s._ctx, buf.data_ptr(), nbytes, 2, chunk, idle, st))
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in evs:
        a.record(); m._abi.check(lib.mtgpu_debug_read_ceiling_shape(s._ctx, buf.data_ptr(), nbytes, 2, chunk, idle, st)); b.record()
    torch.cuda.synchronize()
    return nbytes / (np.mean([a.elapsed_time(b) for a, b in evs]) * 1e-3) / 1e9
print("slow state:", " | ".join(f"{i}: {rate(i):.0f}" for i in (0, 30)))
s2 = torch.cuda.Stream(device=dev)
with torch.cuda.stream(s2):
    x = torch.zeros(1 << 20, device=dev) + 1
s2.synchronize()
for rnd in range(2):
    print("fast state:", " | ".join(f"{i}: {rate(i):.0f}" for i in (0, 120, 60, 30, 20, 16, 15, 14, 12, 10, 8, 6, 4, 3, 2)), flush=True)
