#!/usr/bin/env python3
"""Round-4 probe for the run-aggregated vote path (scan_kernels.hip: vote / bump_n): scan-kernel rate by HIP events
on typical and on vote-heavy ("pan": every record above the threshold) input, for the plans whose counters are
packed (thermometer fields) or banded (spill queue), plus the headline and the 1-bit tile as controls.
Alternate builds with scripts/ab_libs_r04.sh.  Prints one line per case."""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
CASES = [  # workload, params, frames, pan, compact
    ("4k_fine_dense4", "shipped_env", 256, True, False),     # 2 bands, 4-bit fields, 4 records per block
    ("4k_fine", "shipped_env", 1024, True, False),           # the 0.69 case of rounds 2-3: 2 bands, one record per block
    ("4k_fine_dense4", "shipped_env", 1024, False, False),   # the same plan on typical input (bench leg)
    ("4k_fine", "code_defaults", 1024, True, False),         # single 2-bit tile (VECTORS_NEEDED 2), vote-heavy
    ("4k_fine", "code_defaults", 1024, False, False),        # ... typical (bench leg)
    ("1080p_dense8x8", "code_defaults", 4096, False, False), # control: 32-bit adds, code path unchanged
    ("1080p_dense8x8", "code_defaults", 4096, True, False),
    ("4k_fine_dense4", "shipped_env", 256, True, True),      # compact records, banded, vote-heavy
]
only = os.environ.get("ONLY")
for i, (wl, pn, frames, pan, compact) in enumerate(CASES):
    if only and str(i) not in only.split(","):
        continue
    os.environ["AB_PAN"] = "1" if pan else "0"
    w = bench.build_workload(wl, pn, frames, 30, 1000, dev)
    if compact:
        ms, flags = bench.time_compact(w, 10)
        nbytes = 8 * w["n_records"] + 9 * frames
    else:
        ms = bench.time_scan_only(w, 10)
        flags = w["d_flags"].cpu().numpy()
        nbytes = w["alg_bytes"]
    plan = w["scanner"].plan
    print(f"case {i} {wl}:{pn}:{frames}{':pan' if pan else ''}{':compact' if compact else ''} bands {plan['bands']} fb {plan['counter_bits']} "
          f"{ms:.4f} ms {nbytes / ms / 1e9:.3f} TB/s frac {nbytes / ms / 1e9 / 8.0:.3f} motion {int(flags.sum())}/{frames}", flush=True)
    w["scanner"].close()
    del w
    torch.cuda.empty_cache()
