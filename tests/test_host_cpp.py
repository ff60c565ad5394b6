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
