"""CPU-tier test of the C++ host layer (csrc/host/mtgpu_host.hpp): builds tests/cpp/host_layer_test.cpp
with g++ against the C ABI and checks Config / TaskQueue / ResultCollector / JobQueue semantics, the
.mtmv reader (against what mvfile.py wrote) and the backward-seek rule.  No GPU needed."""
import os
import subprocess


import mvtrim_amd as m
from mvtrim_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.dirname(m.LIB_PATH)


def test_cpp_host_layer(tmp_path):
    exe = str(tmp_path / "host_layer_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(PKG, "csrc", "host"), os.path.join(ROOT, "tests", "cpp", "host_layer_test.cpp"),
                           "-o", exe, "-L" + PKG, "-lmtgpu", "-lpthread", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    spec = synth.StreamSpec(width=320, height=240, block=16, sub=1, fps=30.0, gop=15, seed=2)
    n = 70
    frames = [synth.gen_frame(spec, i) for i in range(n)]          # I-frames (None) at 0, 15, 30, 45, 60
    ticks = [spec.pts_ticks(i) for i in range(n)]
    path = str(tmp_path / "s.mtmv")
    m.mvfile.write_mtmv(path, 320, 240, 1, spec.tb_den, spec.fps, n / spec.fps, ticks, frames)
    out = subprocess.run([exe, path], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout + out.stderr
    kv = dict(ln.split(" ", 1) for ln in out.stdout.splitlines() if " " in ln and not ln.startswith("seek"))
    nrec = sum(0 if f is None else len(f) for f in frames)
    assert (int(kv["width"]), int(kv["height"]), int(kv["frames"]), int(kv["records"])) == (320, 240, n, nrec)
    assert float(kv["fps"]) == 30.0 and int(kv["time_base_den"]) == 90000
    assert abs(float(kv["duration"]) - n / 30.0) < 1e-6
    assert (int(kv["iterated"]), int(kv["with_sd"]), int(kv["bytes"])) == (n, n - 5, 40 * nrec)
    assert int(kv["first_pts"]) == 0
    seeks = [ln.split() for ln in out.stdout.splitlines() if ln.startswith("seek")]
    # target tick = int64(seconds / time_base); lands on the last keyframe with pts <= target
    want = {0.0: 0, 0.2: 0, 1.0: 30, 1.49: 30, 1.5: 45, 2.26: 60, 100.0: 60}
    assert {float(s[1]): int(s[4]) for s in seeks} == {k: ticks[v] for k, v in want.items()}
    assert all(s[6] == "1" for s in seeks)                          # every landing frame is a keyframe
    if "skipped" not in kv["init_error"]:
        assert "no CPU fallback" in kv["init_error"]


def _build_recipe(tmp_path):
    exe = str(tmp_path / "integration_recipe")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "integration_recipe.cpp"), "-o", exe, "-L" + PKG, "-lmtgpu",
                           "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_integration_recipe_compiles_against_the_header(tmp_path):
    """INTEGRATION.md section 1 (the pipe at the check_frame call site), member for member in
    tests/cpp/integration_recipe.cpp: it must compile and link against include/mtgpu.h as written, and every
    entry point the text tells a maintainer to call must really appear in it."""
    _build_recipe(tmp_path)
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    src = open(os.path.join(ROOT, "tests", "cpp", "integration_recipe.cpp")).read()
    for call in ("mtgpu_pipe_create(", "mtgpu_pipe_acquire(", "mtgpu_batch_add_frame(", "mtgpu_pipe_submit(", "mtgpu_pipe_collect(",
                 "mtgpu_pipe_release(", "mtgpu_pipe_destroy(", "MT_ERR_CAPACITY", "MT_ERR_BUSY", "in_flight_"):
        assert call in text and call in src, call


import pytest  # noqa: E402


@pytest.mark.gpu
def test_integration_recipe_gives_the_oracles_answer(tmp_path):
    """... and run on the GPU it must find exactly the frames the oracle finds: 90 frames, a 2 x 1-cell object moving
    in frames 30..59, every 15th frame without side data (45 is one of them) — through 64-record batches, i.e. many
    rounds of capacity / submit / back-pressure / drain."""
    import numpy as np
    import oracle_binding as ob
    exe = _build_recipe(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    p = ob.params_from_config(1920, 1080)
    frames, sd = [], []
    for f in range(90):
        mv = np.zeros(4 if 30 <= f < 60 else 0, dtype=m.MV_DTYPE)
        for k in range(len(mv)):
            mv["dst_x"][k], mv["dst_y"][k] = 16 * (40 + k // 2) + 8, 16 * 30 + 8
            mv["src_x"][k], mv["src_y"][k] = mv["dst_x"][k] - 6, mv["dst_y"][k]
        frames.append(mv)
        sd.append(f % 15 != 0)
    want = [f / 30.0 for f in range(90) if sd[f] and ob.check_frame(p, frames[f], True)]
    assert len(want) == 28                                         # frames 30..59 without 30 and 45
    assert out.stdout.split() == ["motion", str(len(want)), "first", f"{want[0]:.6f}", "last", f"{want[-1]:.6f}"]
