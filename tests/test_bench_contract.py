"""bench.py contract checks that need no GPU: it refuses to run without one (no CPU fallback
can ever produce a number), and its declared defaults are the BASELINE.json configuration."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gpurun_out_state():
    """(path, size, mtime) of every file under gpurun_out/ — the evidence the GPU runs brought back, which no CPU test
    may create, delete or append to."""
    d = os.path.join(ROOT, "gpurun_out")
    out = []
    for base, _dirs, files in os.walk(d):
        for f in files:
            st = os.stat(os.path.join(base, f))
            out.append((os.path.join(base, f), st.st_size, st.st_mtime_ns))
    return sorted(out)


def test_bench_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-pmc"],
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
    logs, reports = str(tmp_path / "logs"), []
    os.makedirs(logs)
    before = gpurun_out_state()
    assert bench.launch_ranks(a, argv, environ={"PATH": "/bin"}, popen=FakeProc, log_dir=logs, report=reports.append) == 0
    assert len(started) == 4
    assert all(c[0][1].endswith("bench.py") and c[0][2:] == argv for c in started)
    assert started[0][2] is None and all(c[2] == subprocess.DEVNULL for c in started[1:])   # rank 0's stdout is ours
    # rank 0 keeps our stderr; every other rank's stderr is kept in its own file
    assert started[0][3] is None
    assert [c[3] for c in started[1:]] == [os.path.join(logs, f"bench_rank{r}.err.stderr.log") for r in (1, 2, 3)]
    assert all(c[1]["MTGPU_BENCH_LOG_DIR"] == logs for c in started)             # the ranks log where the launcher looks
    assert len({c[1]["MASTER_PORT"] for c in started}) == 1
    started.clear()
    fail_rank, fail_rank_code = 2, 7
    assert bench.launch_ranks(a, argv, environ={}, popen=FakeProc, log_dir=logs, report=reports.append) == 7
    assert len(reports) == 1 and "rank 2 exited with code 7" in reports[0]

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
    assert bench.launch_ranks(bench.parse(["--gpus", "3"]), ["--gpus", "3"], environ=dict(os.environ), popen=probe_popen,
                              log_dir=logs, report=reports.append) == 0
    assert bench.launch_ranks(bench.parse(["--gpus", "3"]), ["--gpus", "3"],
                              environ=dict(os.environ, FAIL_RANK="1"), popen=probe_popen, log_dir=logs, report=reports.append) == 3
    # the default log directory is an environment setting away, and none of the above touched the evidence directory
    assert bench.rank_log_dir({"MTGPU_BENCH_LOG_DIR": "/x/y"}) == "/x/y"
    assert gpurun_out_state() == before


def test_bench_helpers_traffic_quota_and_rank_logs():
    """CPU-tier pieces of the bench line: `traffic` is replayed from the committed PMC summary and labelled as such
    for EVERY leg the default run reports; the cgroup CPU quota is a positive number or None; rank logs are named
    per rank."""
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse([])
    t, src = bench.replayed_traffic(a.workload, a.params, a.frames)
    assert t and 1.0 <= t / (40 * 15837 * 32640 + 9 * 16384) < 1.001        # 16 384 frames: 15 837 P-frames x 32 640 records
    assert src.startswith("replayed, not measured in this run") and "pmc_traffic.json" in src and "round 6" in src
    for (wl, pn, frames, _steps) in bench.OTHER_WORKLOADS:
        t, src = bench.replayed_traffic(wl, pn, frames)
        assert t and "replayed" in src, (wl, pn, frames)
    assert bench.replayed_traffic("1080p_dense8x8", "code_defaults", 12345) == (None, None)
    q = bench.cpu_quota()
    assert q is None or q > 0
    assert os.path.basename(bench.rank_log_path(3)) == "bench_rank3.err"
    model, total, usable = bench.host_cpu_info()
    assert total >= usable >= 1 and isinstance(model, str)


def test_launcher_watchdog_ends_a_hung_run_and_keeps_the_evidence(tmp_path, monkeypatch):
    """First contact with N > 1 ranks happens on the driver's box: a rank that stops making progress (a
    collective that never completes) must not leave `python bench.py --gpus N` waiting for the driver's kill.
    A live rank whose log / stderr stay silent past --rank-timeout gets every rank terminated (killed if it
    ignores that), exit code 124, and a verdict line; a rank that keeps writing is never taken for hung."""
    sys.path.insert(0, ROOT)
    import bench
    before = gpurun_out_state()
    now = [0.0]
    reports = []

    class Proc:
        """rank 1 hangs forever (and ignores SIGTERM); the others would run for 1000 s, writing as they go"""
        procs = []

        def __init__(self, cmd, env, stdout, stderr=None):
            self.rank = int(env["RANK"])
            self.terminated = self.killed = False
            Proc.procs.append(self)

        def poll(self):
            if self.killed:
                return -9
            if self.terminated and self.rank != 1:
                return -15
            return None

        def terminate(self):
            self.terminated = True

        def kill(self):
            self.killed = True

    def evidence(rank):                    # every rank but 1 logs something new every 10 s of fake time; rank 1 wrote
        return (5, 1.0) if rank == 1 else (int(now[0] // 10), now[0] // 10)      # its "start" line and then nothing

    def sleep(dt):
        now[0] += 1.0                      # fake time: one second per poll

    a = bench.parse(["--gpus", "4", "--rank-timeout", "30"])
    rc = bench.launch_ranks(a, ["--gpus", "4"], environ={}, popen=Proc, clock=lambda: now[0], sleep=sleep,
                            evidence=evidence, report=reports.append, log_dir=str(tmp_path))
    assert rc == 124
    assert 30 < now[0] < 60                                                    # ended soon after the limit, not at 1000 s
    assert all(p.terminated for p in Proc.procs) and Proc.procs[1].killed      # SIGTERM for all, SIGKILL for the deaf one
    assert not any(p.killed for p in Proc.procs if p.rank != 1)
    assert len(reports) == 1 and "[1]" in reports[0] and "silent" in reports[0]

    # a rank that never wrote a byte (still importing torch on a cold box) gets twice the limit
    Proc.procs.clear()
    now[0] = 0.0
    reports.clear()
    rc = bench.launch_ranks(a, ["--gpus", "4"], environ={}, popen=Proc, clock=lambda: now[0], sleep=sleep,
                            evidence=lambda r: (0, 0.0) if r == 1 else (int(now[0] // 10), 0.0), report=reports.append,
                            log_dir=str(tmp_path))
    assert rc == 124 and 60 < now[0] < 90 and len(reports) == 1
    reports[:] = reports[:1]

    # a slow but living run is left alone; --rank-timeout 0 switches the watchdog off
    Proc.procs.clear()
    now[0] = 0.0

    class Slow(Proc):
        def poll(self):
            return 0 if now[0] > 200 else None
    a = bench.parse(["--gpus", "2", "--rank-timeout", "30"])
    assert bench.launch_ranks(a, ["--gpus", "2"], environ={}, popen=Slow, clock=lambda: now[0], sleep=sleep,
                              evidence=lambda r: (int(now[0] // 10), 0.0), report=reports.append, log_dir=str(tmp_path)) == 0
    assert len(reports) == 1
    now[0] = 0.0
    a = bench.parse(["--gpus", "2", "--rank-timeout", "0"])
    assert bench.launch_ranks(a, ["--gpus", "2"], environ={}, popen=Slow, clock=lambda: now[0], sleep=sleep,
                              evidence=lambda r: (0, 0.0), report=reports.append, log_dir=str(tmp_path)) == 0

    # real processes: rank 1 sleeps "forever", the watchdog (1 s) ends the run; logs stay
    probe = tmp_path / "probe.py"
    probe.write_text("import os, sys, time\n"
                     "r = int(os.environ['RANK'])\n"
                     "time.sleep(600 if r == 1 else 0.1)\n")
    real_popen = subprocess.Popen

    def probe_popen(cmd, env, stdout, stderr=None):
        return real_popen([sys.executable, str(probe)], env=env, stdout=subprocess.DEVNULL, stderr=stderr)
    import time as _t
    t0 = _t.monotonic()
    a = bench.parse(["--gpus", "2", "--rank-timeout", "1"])
    assert bench.launch_ranks(a, ["--gpus", "2"], environ=dict(os.environ), popen=probe_popen, report=reports.append,
                              log_dir=str(tmp_path)) == 124
    assert _t.monotonic() - t0 < 20
    assert os.path.exists(str(tmp_path / "bench_rank1.err.stderr.log"))
    assert gpurun_out_state() == before


def test_rank_under_torchrun_sets_the_ipc_mode_before_the_gpu_runtime_starts():
    """Under `python -m torch.distributed.run ... bench.py --gpus N` (the driver's N > 1 command) no launcher of ours
    builds the rank's environment: the rank itself must set HSA_ENABLE_IPC_MODE_LEGACY=0 (RCCL on this pool) before
    torch — hence HIP — is imported; a value the caller exported is left alone; a single rank sets nothing."""
    import inspect
    sys.path.insert(0, ROOT)
    import bench
    assert bench.prepare_rank_environment({}, True) == {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    assert bench.prepare_rank_environment({"HSA_ENABLE_IPC_MODE_LEGACY": "1"}, True) == {"HSA_ENABLE_IPC_MODE_LEGACY": "1"}
    assert bench.prepare_rank_environment({}, False) == {}
    src = inspect.getsource(bench._run_rank)
    assert 0 < src.index("prepare_rank_environment(os.environ, multi)") < src.index("import torch")


def test_bench_line_config_is_machine_readable_when_truncated():
    """The driver's record keeps only the head of long strings: `config.workload` must show the parameter set within
    its first 100 characters, and the details live in keys of their own."""
    import inspect
    sys.path.insert(0, ROOT)
    import bench
    src = inspect.getsource(bench._run_rank)
    a = bench.parse([])
    wl = f"{a.workload} params={a.params} grid=120x68 frames/GPU/step={a.frames}"
    assert len(wl) < 100 and "params=code_defaults" in wl
    assert 'f"{a.workload} params={a.params} grid=' in src and '"distinct_frames_tiled": a.distinct' in src
    assert '"vs_cpu_baseline_same_quota"' in src


def test_traffic_is_measured_by_child_runs_or_the_fallback_says_why(tmp_path):
    """roofline.traffic at N = 1: two rocprofv3 child runs (FETCH_SIZE, WRITE_SIZE in separate passes, the interpreter
    directly after `--`), parsed from rocprofv3's counter CSVs with the guide's gfx950 correction; when rocprofv3 is
    missing or a child fails the caller gets (None, reason) and replays the committed figure."""
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse([])
    assert not a.no_pmc and not a.pmc_child
    # the parser: per scan launch, the scan kernel's counter plus the planning kernels' of the same call
    d = tmp_path / "FETCH_SIZE" / "host" / "123"
    d.mkdir(parents=True)
    rows = ["Correlation_Id,Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value"]
    for i in range(3):
        rows.append(f'{i},{3 * i},"void mtgpu::plan_scatter_kernel(...)",FETCH_SIZE,100')
        rows.append(f'{i},{3 * i + 1},"void mtgpu::scan_frames_kernel<512, 4, 32, 0, 0, 40, false>(...)",FETCH_SIZE,10000000')
        rows.append(f'{i},{3 * i + 1},"void mtgpu::scan_frames_kernel<512, 4, 32, 0, 0, 40, false>(...)",GRBM_COUNT,5')
        rows.append(f'{i},{3 * i + 2},"void at::native::vectorized_elementwise_kernel<...>",FETCH_SIZE,777')
    (d / "123_counter_collection.csv").write_text("\n".join(rows) + "\n")
    assert bench.parse_pmc_dir(str(tmp_path / "FETCH_SIZE"), "FETCH_SIZE") == (10000100.0, 3)
    assert bench.pmc_bytes(10000100.0, 50.0) == 10000100.0 * 2048.0 + 50.0 * 1024.0
    import pytest
    with pytest.raises(RuntimeError):
        bench.parse_pmc_dir(str(tmp_path / "FETCH_SIZE"), "WRITE_SIZE")           # no dispatch carries that counter
    # fallback 1: no rocprofv3 on the box (and none at the default ROCm path: patched out)
    real_exists = os.path.exists
    try:
        os.path.exists = lambda p_: False if p_ == "/opt/rocm/bin/rocprofv3" else real_exists(p_)
        t, why, detail = bench.measure_traffic(a, [], which=lambda name: None)
    finally:
        os.path.exists = real_exists
    assert t is None and detail is None and "not found" in why
    # fallback 2: a child fails; the command is rocprofv3 ... -- <python> bench.py --pmc-child (no hop in between)
    seen = []

    class R:
        returncode, stderr = 1, "boom"

    def run(cmd, **kw):
        seen.append(cmd)
        return R()
    t, why, detail = bench.measure_traffic(a, [], which=lambda name: "/bin/rocprofv3", run=run, tmp_root=str(tmp_path))
    assert t is None and "exited 1" in why and "boom" in why
    cmd = seen[0]
    assert cmd[0] == "/bin/rocprofv3" and cmd[1:4] == ["--kernel-trace", "--pmc", "FETCH_SIZE"]
    assert not any(flag in cmd for flag in ("-s", "--sys-trace", "-r", "--runtime-trace", "--hip-trace", "--hsa-trace"))
    sep = cmd.index("--")
    assert cmd[sep + 1] == sys.executable and cmd[sep + 2].endswith("bench.py") and cmd[sep + 3] == "--pmc-child"
    # success path with a stand-in child that writes the CSVs rocprofv3 would
    def run_ok(cmd, **kw):
        counter = cmd[cmd.index("--pmc") + 1]
        out = os.path.join(cmd[cmd.index("-d") + 1], "h", "1")
        os.makedirs(out)
        with open(os.path.join(out, "1_counter_collection.csv"), "w") as fh:
            fh.write("Kernel_Name,Counter_Name,Counter_Value\n")
            for _ in range(4):
                fh.write(f'"mtgpu::scan_frames_kernel<...>",{counter},{1000 if counter == "FETCH_SIZE" else 10}\n')

        class Ok:
            returncode, stderr = 0, ""
        return Ok()
    t, src, detail = bench.measure_traffic(a, [], which=lambda name: "/bin/rocprofv3", run=run_ok, tmp_root=str(tmp_path))
    assert t == 1000 * 2048.0 + 10 * 1024.0 and src.startswith("measured in this run") and detail["launches_averaged"] == [4, 4]
