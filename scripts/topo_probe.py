import os, glob, torch
print("pci bus id", torch.cuda.get_device_properties(0).pci_bus_id if hasattr(torch.cuda.get_device_properties(0), "pci_bus_id") else None)
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
buf = ctypes.create_string_buffer(64)
print("hipDeviceGetPCIBusId rc", hip.hipDeviceGetPCIBusId(buf, 64, 0), buf.value)
bdf = buf.value.decode().lower()
for f in ("local_cpulist", "numa_node", "class", "vendor"):
    p = f"/sys/bus/pci/devices/{bdf}/{f}"
    print(f, open(p).read().strip() if os.path.exists(p) else "missing")
gpus = []
for d in sorted(glob.glob("/sys/bus/pci/devices/*")):
    try:
        cls = open(d + "/class").read().strip(); ven = open(d + "/vendor").read().strip()
    except OSError:
        continue
    if ven == "0x1002" and cls.startswith(("0x0302", "0x0380", "0x0300", "0x1200")):
        gpus.append((os.path.basename(d), cls, open(d + "/numa_node").read().strip(), open(d + "/local_cpulist").read().strip()))
print("amd gpus visible in sysfs:", len(gpus))
for g in gpus: print(" ", g)
print("allowed cpus", len(os.sched_getaffinity(0)))
print(open("/proc/self/status").read().split("Cpus_allowed_list:")[1].split("\n")[0].strip())
print("numa nodes", sorted(glob.glob("/sys/devices/system/node/node*/cpulist")), [open(p).read().strip() for p in sorted(glob.glob("/sys/devices/system/node/node*/cpulist"))])
