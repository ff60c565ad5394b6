"""bench.py contract checks that need no GPU: it refuses to run without one (no CPU fallback
can ever produce a number), and its declared defaults are the BASELINE.json configuration."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True)
    assert out.returncode != 0 and "needs a GPU" in (out.stderr + out.stdout)
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]      # no JSON line, no number


def test_bench_defaults_match_baseline_config():
    sys.path.insert(0, ROOT)
    import bench
    old = sys.argv
    sys.argv = ["bench.py"]
    try:
        a = bench.parse()
    finally:
        sys.argv = old
    assert (a.gpus, a.workload, a.params) == (1, "1080p_dense8x8", "code_defaults")
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "1080p" in base["metric"] and "120" in base["configs"][1]            # the 120x68 1080p grid config
    spec, (w, h, kw) = bench.make_spec(a.workload, seed=1)
    assert (w, h, spec.records_per_frame) == (1920, 1080, 32640)                 # dense8x8: 1 305 600 B per frame
