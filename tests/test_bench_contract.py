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
    assert not bench.needs_launch(a, {})                                        # N = 1: this process is the rank
    assert (a.gpus, a.workload, a.params) == (1, "1080p_dense8x8", "code_defaults")
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "1080p" in base["metric"] and "120" in base["configs"][1]            # the 120x68 1080p grid config
    spec, (w, h, kw) = bench.make_spec(a.workload, seed=1)
    assert (w, h, spec.records_per_frame) == (1920, 1080, 32640)                 # dense8x8: 1 305 600 B per frame


def test_bench_gpus_n_launches_n_ranks_itself(tmp_path):
    """`python bench.py --gpus N` (how the driver invokes it) must start N ranks on its own: the
    parent builds one environment per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* on 127.0.0.1),
    forwards rank 0's stdout, and fails if any rank fails.  Under a launcher (WORLD_SIZE already
    set by torchrun) it is one rank and launches nothing."""
    sys.path.insert(0, ROOT)
    import bench
    argv = ["--gpus", "4", "--steps", "5", "--warmup", "1"]
    a = bench.parse(argv)
    assert bench.needs_launch(a, {}) and bench.needs_launch(a, {"PATH": "/bin"})
    assert not bench.needs_launch(a, {"WORLD_SIZE": "4", "RANK": "2"})           # torchrun / the driver's launcher
    assert not bench.needs_launch(bench.parse(["--gpus", "1"]), {})
    envs = bench.rank_environments(4, {"PATH": "/bin"}, 29517)
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"] == [e["LOCAL_RANK"] for e in envs]
    assert all(e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29517"
               and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and e["PATH"] == "/bin" for e in envs)

    # the launcher itself, with a stand-in child that only reports its environment
    started = []

    class FakeProc:
        def __init__(self, cmd, env, stdout, stderr=None):
            started.append((cmd, env, stdout, getattr(stderr, "name", None)))
            self.rank = int(env["RANK"])
            self.code = None
            self.terminated = False

        def poll(self):
            if self.code is None:
                self.code = fail_rank_code if self.rank == fail_rank else (-15 if self.terminated else 0)
            return self.code

        def terminate(self):
            self.terminated = True

    fail_rank, fail_rank_code = -1, 0
    assert bench.launch_ranks(a, argv, environ={"PATH": "/bin"}, popen=FakeProc) == 0
    assert len(started) == 4
    assert all(c[0][1].endswith("bench.py") and c[0][2:] == argv for c in started)
    assert started[0][2] is None and all(c[2] == subprocess.DEVNULL for c in started[1:])   # rank 0's stdout is ours
    # rank 0 keeps our stderr; every other rank's stderr is kept in its own file
    assert started[0][3] is None
    assert [os.path.basename(c[3]) for c in started[1:]] == [f"bench_rank{r}.err.stderr.log" for r in (1, 2, 3)]
    assert len({c[1]["MASTER_PORT"] for c in started}) == 1
    started.clear()
    fail_rank, fail_rank_code = 2, 7
    assert bench.launch_ranks(a, argv, environ={}, popen=FakeProc) == 7

    # end to end with real child processes: a tiny script in place of bench.py's rank body
    probe = tmp_path / "probe.py"
    probe.write_text("import os, sys\n"
                     "r = int(os.environ['RANK'])\n"
                     "print('rank', r, 'of', os.environ['WORLD_SIZE'], flush=True)\n"
                     "sys.exit(3 if os.environ.get('FAIL_RANK') == str(r) else 0)\n")
    real_popen = subprocess.Popen

    def probe_popen(cmd, env, stdout, stderr=None):
        return real_popen([sys.executable, str(probe)], env=env, stdout=subprocess.PIPE if stdout is None else stdout,
                          stderr=stderr)
    assert bench.launch_ranks(bench.parse(["--gpus", "3"]), ["--gpus", "3"], environ=dict(os.environ), popen=probe_popen) == 0
    assert bench.launch_ranks(bench.parse(["--gpus", "3"]), ["--gpus", "3"],
                              environ=dict(os.environ, FAIL_RANK="1"), popen=probe_popen) == 3


def test_bench_helpers_traffic_quota_and_rank_logs():
    """CPU-tier pieces of the bench line: `traffic` is replayed from the committed PMC summary and labelled as such
    for EVERY leg the default run reports; the cgroup CPU quota is a positive number or None; rank logs are named
    per rank."""
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse([])
    t, src = bench.replayed_traffic(a.workload, a.params, a.frames)
    assert t and 1.0 <= t / (40 * 15837 * 32640 + 9 * 16384) < 1.001        # 16 384 frames: 15 837 P-frames x 32 640 records
    assert src.startswith("replayed, not measured in this run") and "pmc_traffic.json" in src and "round 3" in src
    for (wl, pn, frames, _steps) in bench.OTHER_WORKLOADS:
        t, src = bench.replayed_traffic(wl, pn, frames)
        assert t and "replayed" in src, (wl, pn, frames)
    assert bench.replayed_traffic("1080p_dense8x8", "code_defaults", 12345) == (None, None)
    q = bench.cpu_quota()
    assert q is None or q > 0
    assert os.path.basename(bench.rank_log_path(3)) == "bench_rank3.err"
    model, total, usable = bench.host_cpu_info()
    assert total >= usable >= 1 and isinstance(model, str)
