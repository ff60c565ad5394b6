import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """Build the oracle (gcc) and, if missing, the product library (hipcc cross-compiles
    without a GPU).  Both normally exist already (built by __graft_entry__.build())."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "libmt_oracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "all"])
    lib = os.path.join(ROOT, "motion-estimated-video-trimmer_amd", "libmtgpu.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "motion-estimated-video-trimmer_amd", "csrc")])
    yield


@pytest.fixture(scope="session")
def gpu_scanner_factory():
    """Creates MotionScanner contexts on device 0; fails loudly (no skip) if the HIP
    library or the device is unavailable — GPU tests must never pass on a fallback."""
    import mvtrim_amd as m
    made = []

    def make(params, force_fb=None, force_block=None):
        """force_fb / force_block select a counter form / workgroup size through the
        library's MTGPU_FORCE_FB / MTGPU_FORCE_BLOCK experiment knobs (read at create time)."""
        old = {k: os.environ.get(k) for k in ("MTGPU_FORCE_FB", "MTGPU_FORCE_BLOCK")}
        try:
            for k, v in (("MTGPU_FORCE_FB", force_fb), ("MTGPU_FORCE_BLOCK", force_block)):
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = str(v)
            s = m.MotionScanner(params, device=0)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        made.append(s)
        return s

    yield make
    for s in made:
        s.close()


def experiments_build():
    """True when the loaded libmtgpu is the A/B build (csrc/Makefile `experiments`, MTGPU_LIBRARY): only there are
    the exp_int() knobs of csrc/knobs.h (MTGPU_ALIGN, MTGPU_PREFETCH, MTGPU_PIPE_EAGER, MTGPU_PIPE_STREAMS, ...)
    read at all.  The default build ignores them, so the knob-off halves of a few tests run only on that build."""
    import mvtrim_amd as m
    return b"+experiments" in m.load_library().mtgpu_version()
