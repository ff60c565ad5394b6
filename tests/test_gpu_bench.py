"""bench.py on the GPU box (`-m gpu`): the one-line JSON contract at N = 1, and the N > 1 control flow — rank
processes started by bench.py itself, barrier + max-over-ranks timing, the exchange of segment lists, the `ranks`
list — rehearsed with two gloo ranks on the one device this box has (RCCL refuses two ranks on one device; real
multi-device RCCL runs only on the driver's 8-GPU node)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QUICK = ["--steps", "4", "--warmup", "2", "--frames", "600", "--cpu-seconds", "0", "--no-others", "--no-host"]


def _run(args, log_dir=None):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if log_dir is not None:
        env["MTGPU_BENCH_LOG_DIR"] = str(log_dir)              # rank logs of the test go to the test's directory
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                       env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]                    # exactly ONE line on stdout
    return json.loads(lines[0])


def _check_line(d, world, frames):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["metric"].startswith("MV-scan frames/sec") and d["unit"] == "frames/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == world and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.05 < r["frac"] < 1.0
    assert r["algorithmic_bytes_per_launch"] == d["config"]["bytes_per_step_per_gpu"]
    # value = frames of ALL ranks / max-over-ranks time: consistent with ms_per_step
    assert abs(d["value"] - frames * world / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    # the kernel cannot be slower than the whole step it is part of
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.05
    assert 0 < d["motion_frames_in_batch"] < frames
    assert len(d["ranks"]) == world and [x["rank"] for x in d["ranks"]] == list(range(world))
    assert all(x["frames_scanned"] == frames * d["steps"] and x["kernel_ms"] > 0 for x in d["ranks"])


def test_bench_line_single_gpu():
    d = _run(QUICK)
    _check_line(d, 1, 600)
    assert d["distinct_devices"] == 1 and d["ranks"][0]["pci_bus_id"]
    assert d["config"]["step"] == "scan + stream-merge kernels"
    # traffic is MEASURED in the run (two rocprofv3 --pmc child runs of bench.py, before the parent touches the GPU):
    # a read-once scan moves its algorithmic bytes and little else (600 frames: the batch is small, allow 3 %)
    r = d["roofline"]
    assert r["traffic_source"].startswith("measured in this run"), r["traffic_source"]
    assert 0.97 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.03 and r["traffic_detail"]["launches_averaged"] == [4, 4]
    # the scan kernel and the planning kernels ahead of it are timed separately, by the library's own events
    assert 0 < r["plan_ms"] < r["kernel_ms"] and abs(r["call_ms"] - r["plan_ms"] - r["kernel_ms"]) < 1e-9
    d2 = _run(QUICK + ["--no-pmc"])
    assert d2["roofline"]["traffic"] is None or "replayed" in d2["roofline"]["traffic_source"]


def test_bench_two_ranks_rehearsal_gloo_same_device(tmp_path):
    d = _run(["--gpus", "2", "--backend", "gloo", "--same-device"] + QUICK, log_dir=tmp_path)
    _check_line(d, 2, 600)
    assert d["config"]["streams_total"] == 2 * d["config"]["streams_per_gpu"]
    assert "all_gather" in d["config"]["step"]
    assert d["distinct_devices"] == 1                              # --same-device: both ranks report the one bus id
    assert d["cpu_baseline"] is None and d["other_workloads"] is None      # N > 1: the headline only
    logs = [os.path.join(str(tmp_path), f"bench_rank{r}.err") for r in (0, 1)]
    for r, p in enumerate(logs):                                   # every rank left its own evidence
        recs = [json.loads(ln) for ln in open(p)]
        # one line per stage (the launcher's watchdog reads their arrival as the rank's sign of life)
        assert [x["stage"] for x in recs] == ["start", "process_group_ready", "workload_resident", "warm", "timed",
                                              "line_printed" if r == 0 else "waiting_for_rank_0", "done"]
        timed = [x for x in recs if x["stage"] == "timed"][0]
        assert timed["rank"] == r and recs[0]["rank"] == r and timed["frames_scanned"] > 0


def test_bench_rccl_branch_with_one_rank(tmp_path):
    """The `nccl` (= RCCL) branch of the distributed path — process group bound to the device, asynchronous
    all_gather_into_tensor of the packed segment lists issued from the merge stream, barriers, MAX all_reduce of the
    wall time, all_gather_object of the rank identities — executed for real, with the one rank a 1-GPU box allows
    (`--force-dist`).  Peers are the only thing missing; the driver's 8-GPU node supplies those."""
    d = _run(["--force-dist"] + QUICK, log_dir=tmp_path)
    _check_line(d, 1, 600)
    assert "RCCL all_gather" in d["config"]["step"]
    assert d["distinct_devices"] == 1 and d["ranks"][0]["pci_bus_id"]


def test_bench_under_torchrun_exactly_as_the_driver_launches_it(tmp_path):
    """For N > 1 the driver does not call `python bench.py --gpus N` (bench.py's own launcher) but
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
    --gpus N --steps K --warmup W`: WORLD_SIZE is already set, so bench.py is ONE rank and starts nothing.  The same
    command line here with two gloo ranks on the one device (RCCL refuses two ranks on one device): one JSON line on
    rank 0's stdout, `n_gpus` 2, both ranks' stage logs complete."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["MTGPU_BENCH_LOG_DIR"] = str(tmp_path)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
           "--same-device"] + QUICK
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    _check_line(d, 2, 600)
    assert "all_gather" in d["config"]["step"]
    logs = [os.path.join(str(tmp_path), f"bench_rank{k}.err") for k in (0, 1)]
    for k, p in enumerate(logs):
        stages = [json.loads(ln)["stage"] for ln in open(p)]
        assert stages[0] == "start" and stages[-1] == "done" and "timed" in stages, (k, stages)
