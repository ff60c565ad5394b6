"""The caller-sized buffers of include/mtgpu.h, from plain C, sized exactly as the header says, each with a canary
behind it (tests/c/abi_buffer_canaries.c).  Replaces nothing of the reference: it fences the drop-in boundary that
stands where src/pipeline.cpp:302-356 and src/motion_scanner.cpp:376 stood."""
import os
import re
import subprocess

import pytest

import mvtrim_amd as m

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    pkg = os.path.dirname(m.LIB_PATH)
    exe = str(tmp_path / "abi_buffer_canaries")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "abi_buffer_canaries.c"),
                           "-o", exe, "-L" + pkg, "-lmtgpu", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + pkg,
                           "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_header_states_the_merge_workspace_the_kernel_uses():
    """CPU tier: the three places that name the size of d_ts agree (header, launch interface, Python host)."""
    hdr = open(os.path.join(ROOT, "include", "mtgpu.h")).read()
    assert re.search(r"d_ts\s+workspace, 2 \* n_frames doubles", hdr)
    internal = open(os.path.join(ROOT, "motion-estimated-video-trimmer_amd", "csrc", "merge_kernels.h")).read()
    assert "2 * n_frames_total doubles" in internal
    py = open(os.path.join(ROOT, "motion-estimated-video-trimmer_amd", "scanner.py")).read()
    assert "2 * F" in py or "2 * n_frames" in py


def test_canary_program_builds(tmp_path):
    """CPU tier: the plain-C program compiles and links against the ABI (it cannot run without a device)."""
    exe = _build(tmp_path)
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_every_caller_sized_buffer_is_respected(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all caller-sized buffers respected" in out.stdout
    # the canaries do catch an under-stated size: with d_ts sized as round 4's header had it, the program fails on d_ts
    bad = subprocess.run([exe, "round4-header"], capture_output=True, text=True)
    assert bad.returncode == 1 and "d_ts (n_frames doubles)" in bad.stderr and "was overrun" in bad.stderr, bad.stdout + bad.stderr
