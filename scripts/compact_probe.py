import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
dev = torch.device("cuda", 0)
for wl, frames in (("1080p_dense8x8", 4096), ("4k_dense8x8", 1024), ("4k_fine", 1024)):
    w = bench.build_workload(wl, "code_defaults", frames, 30, 1000, dev)
    k40 = bench.time_scan_only(w, 20)
    f40 = w["d_flags"].cpu().numpy()
    k8, f8 = bench.time_compact(w, 20)
    assert np.array_equal(f40, f8)
    print(wl, "aos40 %.4f ms %.0f GB/s | compact %.4f ms -> %.0f GB/s of compact bytes, %.2fx frames/s" % (
        k40, w["alg_bytes"] / k40 / 1e6, k8, (8 * w["n_records"] + 9 * frames) / k8 / 1e6, k40 / k8), flush=True)
    w["scanner"].close(); del w; torch.cuda.empty_cache()
