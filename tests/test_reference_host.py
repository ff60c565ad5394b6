"""CPU tier: the host half of the scan path against answers of THE REFERENCE ITSELF.

tests/golden/reference_host_vectors.json was recorded by running the reference's own config.hpp, types.hpp,
src/task_queue.cpp and src/ffmpeg_queue.cpp (compiled where they lie under /root/reference by `make -C oracle
ref`, driven by oracle/ref_host_probe.cpp; generator tests/golden/make_reference_host_vectors.py).  These files
are the part of SURVEY.md §8's path that builds here without FFmpeg: env parsing and defaults of every scan /
merge parameter (row a2's inputs), TaskQueue / ResultCollector (a7), FFmpegQueue / FFmpegJob (a10) and the
TimeSegment layout the merge writes (a9/a10).  Replayed against

  * mvtrim_amd.config              (Python mirror; same libc strto* calls as std::sto*)
  * csrc/host/mtgpu_host.hpp       (C++ host layer, through tests/cpp/host_probe.cpp)
  * include/mt_types.h             (mt_segment == TimeSegment)

and, when oracle/_ref/ref_host_probe is present (build container, or the GPU box where the built binary
travels), live against it on fresh random inputs.
"""
import json
import os
import random
import subprocess

import numpy as np
import pytest

import mvtrim_amd as m

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.dirname(m.LIB_PATH)
VEC = json.load(open(os.path.join(HERE, "golden", "reference_host_vectors.json")))
REF_PROBE = os.path.join(ROOT, "oracle", "_ref", "ref_host_probe")

GETTERS = {
    "mv_threshold_sq": m.config.mv_threshold_sq, "block_size": m.config.block_size,
    "block_shift": m.config.block_shift, "vectors_needed": m.config.vectors_needed,
    "clusters_needed": m.config.clusters_needed, "vertical_mask": m.config.vertical_mask,
    "max_gap_sec": m.config.max_gap_sec, "padding_sec": m.config.padding_sec,
    "chunk_duration_sec": m.config.chunk_duration_sec, "target_fps": m.config.target_fps,
    "min_savings_pct": m.config.min_savings_pct, "parallel_streams": m.config.parallel_streams,
    "threads_per_stream": m.config.threads_per_stream,
}
CONFIG_VARS = [k.upper() for k in GETTERS]


@pytest.fixture(scope="module")
def host_probe(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("probe") / "host_probe")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(PKG, "csrc", "host"), os.path.join(HERE, "cpp", "host_probe.cpp"),
                           "-o", exe, "-L" + PKG, "-lmtgpu", "-lpthread", "-Wl,-rpath," + PKG,
                           "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def run(exe, args, env=None, stdin=None):
    e = {"PATH": os.environ.get("PATH", ""), "LD_LIBRARY_PATH": os.environ.get("LD_LIBRARY_PATH", "")}
    e.update(env or {})
    out = subprocess.run([exe] + args, env=e, input=stdin, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    return out.stdout.splitlines()


def python_config_answers(env):
    """The probe's `config` output, produced by mvtrim_amd.config under `env`."""
    import struct
    saved = {k: os.environ.pop(k, None) for k in CONFIG_VARS}
    os.environ.update(env)
    m.config.forget()                           # a fresh process, as far as the getters can tell
    try:
        return [_python_get(name) for name in GETTERS]
    finally:
        for k in CONFIG_VARS:
            os.environ.pop(k, None)
            if saved[k] is not None:
                os.environ[k] = saved[k]
        m.config.forget()


def _python_get(name):
    import struct
    try:
        v = GETTERS[name]()
    except ValueError:
        return f"{name} error invalid_argument"
    except OverflowError:
        return f"{name} error out_of_range"
    if name == "vertical_mask":
        return f"{name} f32 {struct.unpack('<I', struct.pack('<f', v))[0]:#010x}"
    if isinstance(v, float):
        return f"{name} f64 {struct.unpack('<Q', struct.pack('<d', v))[0]:#018x}"
    return f"{name} int {v}"


def python_memo_answers(script):
    """The probe's `memo` output, produced by mvtrim_amd.config: one process, environment edited between gets."""
    saved = {k: os.environ.pop(k, None) for k in CONFIG_VARS}
    m.config.forget()
    lines = []
    try:
        for ln in script:
            op, _, rest = ln.partition(" ")
            if op == "get":
                lines.append(_python_get(rest))
            elif op == "set":
                k, _, v = rest.partition(" ")
                os.environ[k] = v
                lines.append("set ok")
            elif op == "unset":
                os.environ.pop(rest, None)
                lines.append("unset ok")
    finally:
        for k in CONFIG_VARS:
            os.environ.pop(k, None)
            if saved[k] is not None:
                os.environ[k] = saved[k]
        m.config.forget()
    return lines


def bits_only(lines):
    """Drop the decimal rendering of float answers: '<name> f64 <dec> 0x<bits>' -> '<name> f64 0x<bits>'."""
    out = []
    for ln in lines:
        p = ln.split()
        out.append(" ".join([p[0], p[1], p[3]]) if p[1] in ("f64", "f32") else ln)
    return out


def test_fixture_is_reference_output():
    assert "src/task_queue.cpp" in VEC["source"] and len(VEC["config"]) >= 80 and len(VEC["queue"]) >= 10
    assert VEC["race"] == ["race values 6000 once 2000 of 2000 triples_intact 1"]


@pytest.mark.parametrize("case", VEC["config"], ids=[c["name"] for c in VEC["config"]])
def test_python_config_matches_reference(case):
    assert python_config_answers(case["env"]) == bits_only(case["answers"])


def test_cpp_host_config_matches_reference(host_probe):
    for case in VEC["config"]:
        assert run(host_probe, ["config"], case["env"]) == case["answers"], case["name"]


@pytest.mark.parametrize("case", VEC["memo"], ids=[c["name"] for c in VEC["memo"]])
def test_config_memoisation_matches_reference(host_probe, case):
    """get / setenv / get again inside ONE process: the reference's getters are function-local statics
    (config.hpp:56-59) — the first successful parse sticks, a throwing parse is retried.  Answers recorded from
    the reference's own config.hpp; replayed against the C++ host layer and the Python mirror."""
    script = "\n".join(case["script"]) + "\n"
    assert run(host_probe, ["memo"], case["env"], stdin=script) == case["answers"]
    if not case["env"]:
        assert python_memo_answers(case["script"]) == bits_only_mixed(case["answers"])


def bits_only_mixed(lines):
    return [ln if ln.split()[1] in ("ok", "int", "error") else bits_only([ln])[0] for ln in lines]


def test_shipped_env_file_values():
    """config/motion_trim.env as the reference reads it == the SHIPPED_ENV set bench and tests use."""
    case = next(c for c in VEC["config"] if c["name"] == "shipped_env_file")
    got = {ln.split()[0]: ln.split()[2] for ln in case["answers"]}
    want = m.config.SHIPPED_ENV
    assert float(got["mv_threshold_sq"]) == want["mv_threshold_sq"]
    for k in ("block_size", "block_shift", "vectors_needed", "clusters_needed"):
        assert int(got[k]) == want[k]
    import numpy as np
    assert np.float32(got["vertical_mask"]) == np.float32(want["vertical_mask"])
    dflt = next(c for c in VEC["config"] if c["name"] == "code_defaults")
    got = {ln.split()[0]: ln.split()[2] for ln in dflt["answers"]}
    want = m.config.CODE_DEFAULTS
    assert float(got["mv_threshold_sq"]) == want["mv_threshold_sq"]
    for k in ("block_size", "block_shift", "vectors_needed", "clusters_needed"):
        assert int(got[k]) == want[k]
    assert np.float32(got["vertical_mask"]) == np.float32(want["vertical_mask"])


def test_params_from_config_takes_reference_values():
    """mtgpu_params_from_config over the reference's parsed values: u8 VECTORS_NEEDED, float32 mask."""
    checked = 0
    for case in VEC["config"]:
        ans = {ln.split()[0]: ln.split() for ln in case["answers"]}
        if any(v[1] == "error" for v in ans.values()):
            continue
        thr, mask = float(ans["mv_threshold_sq"][2]), float(ans["vertical_mask"][2])
        bs, sh = int(ans["block_size"][2]), int(ans["block_shift"][2])
        vn, cn = int(ans["vectors_needed"][2]), int(ans["clusters_needed"][2])
        try:
            p = m.ScanParams.from_config(1920, 1080, thr, bs, sh, vn, cn, mask)
        except (m.MtgpuError, ValueError):
            continue            # rejected parameter sets are covered by tests/test_host_logic.py
        assert p.vectors_needed == vn and p.clusters_needed == cn and p.block_shift == sh
        assert p.mv_threshold_sq == thr or thr != thr
        checked += 1
    assert checked >= 20


def test_layout_matches_reference(host_probe):
    ref = dict((ln.split()[0], ln.split()[1:]) for ln in VEC["layout"])
    ours = dict((ln.split()[0], ln.split()[1:]) for ln in run(host_probe, ["layout"]))
    assert ref["TimeSegment"] == "size 16 align 16 start 0 end 8".split()
    # same size and offsets; mt_segment asks for LESS alignment (8), so a TimeSegment array is always a valid
    # mt_segment array — the direction INTEGRATION.md casts in (include/mt_types.h)
    assert ours["TimeSegment"] == "size 16 align 8 start 0 end 8".split()
    assert ours["ScanTask"] == ref["ScanTask"][-6:]                       # same field order and offsets
    assert m.SEGMENT_DTYPE.itemsize == 16
    assert (m.SEGMENT_DTYPE.fields["start"][1], m.SEGMENT_DTYPE.fields["end"][1]) == (0, 8)


@pytest.mark.parametrize("case", VEC["queue"], ids=[c["name"] for c in VEC["queue"]])
def test_cpp_host_queues_match_reference(host_probe, case):
    assert run(host_probe, ["queue"], stdin="\n".join(case["script"]) + "\n") == case["answers"]


def test_cpp_host_race_matches_reference(host_probe):
    assert run(host_probe, ["race", "2000", "8"]) == VEC["race"]


def _random_env(rng):
    pool = ["3", "4.75", "-2", " 9", "0x1F", "1e1", "12abc", "0.0625", "255", "256", ".5", "1e-3", "7."]
    return {v: rng.choice(pool) for v in rng.sample(CONFIG_VARS, rng.randrange(1, len(CONFIG_VARS)))}


@pytest.mark.skipif(not os.path.exists(REF_PROBE), reason="oracle/_ref/ref_host_probe not built (make -C oracle ref)")
def test_live_against_reference_binary(host_probe):
    rng = random.Random(7)
    for _ in range(40):
        env = _random_env(rng)
        ref = run(REF_PROBE, ["config"], env)
        assert run(host_probe, ["config"], env) == ref, env
        assert python_config_answers(env) == bits_only(ref), env
    for _ in range(10):
        ops, tid = [], 0
        for _ in range(200):
            c = rng.random()
            if c < 0.35:
                ops.append(f"tpush {rng.uniform(0, 9e4)!r} {rng.uniform(0, 9e4)!r} {tid}")
                tid += 1
            elif c < 0.6:
                ops.append("tpop")
            elif c < 0.75:
                n = rng.randrange(0, 6)
                ops.append(("radd %d %s" % (n, " ".join(repr(rng.uniform(-1, 1e5)) for _ in range(n)))).strip())
            elif c < 0.8:
                ops.append("rextract")
            elif c < 0.9:
                n = rng.randrange(0, 4)
                ops.append(("jpush %d %d %s" % (rng.randrange(64), n, " ".join(repr(rng.uniform(0, 1e4))
                                                                             for _ in range(2 * n)))).strip())
            else:
                ops.append(rng.choice(["jpop", "jdone", "jempty", "tfinish", "jfinish"]))
        ops += ["tfinish", "tpop", "jfinish", "jpop", "rextract"]
        script = "\n".join(ops) + "\n"
        assert run(host_probe, ["queue"], stdin=script) == run(REF_PROBE, ["queue"], stdin=script)
    assert run(REF_PROBE, ["layout"]) == VEC["layout"]
    pool = ["3", "4.75", "abc", "", " 9", "0x1F", "1e400", "300", "-1", "12abc", "99999999999"]
    for _ in range(20):
        ops = []
        for _ in range(40):
            c = rng.random()
            name = rng.choice(list(GETTERS))
            if c < 0.5:
                ops.append(f"get {name}")
            elif c < 0.9:
                ops.append(f"set {name.upper()} {rng.choice(pool)}")
            else:
                ops.append(f"unset {name.upper()}")
        script = "\n".join(ops) + "\n"
        ref = run(REF_PROBE, ["memo"], stdin=script)
        assert run(host_probe, ["memo"], stdin=script) == ref
        assert python_memo_answers(ops) == bits_only_mixed(ref)


def test_cpp_host_queues_under_thread_sanitizer(tmp_path):
    """TaskQueue / ResultCollector / JobQueue of the C++ host layer under -fsanitize=thread: 8 workers drain
    2000 tasks (the reference's worker loop shape, pipeline.cpp:216-223) plus one scripted session; any data
    race makes the sanitizer exit non-zero."""
    exe = str(tmp_path / "host_probe_tsan")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(PKG, "csrc", "host"), os.path.join(HERE, "cpp", "host_probe.cpp"),
                           "-o", exe, "-L" + PKG, "-lmtgpu", "-lpthread", "-Wl,-rpath," + PKG,
                           "-Wl,-rpath,/opt/rocm/lib"])
    env = {"TSAN_OPTIONS": "halt_on_error=1 exitcode=66"}
    for _ in range(3):
        assert run(exe, ["race", "2000", "8"], env) == VEC["race"]
    case = VEC["queue"][-1]
    assert run(exe, ["queue"], env, stdin="\n".join(case["script"]) + "\n") == case["answers"]


def test_stream_to_gpu_assignment(host_probe):
    """process_batch hands stream s the device base s * threads_per_stream; worker i then takes
    (base + i) % n_devices.  BASELINE config 4 (64 streams, 8 GPUs): with one worker per stream every GPU
    serves 8 streams; with 2 workers the pairs (0,1) (2,3) ... rotate; on one GPU everything is device 0."""
    rows = [ln.split(":")[1].split() for ln in run(host_probe, ["devices", "64", "1", "8"])]
    flat = [int(r[0]) for r in rows]
    assert flat == [s % 8 for s in range(64)] and all(flat.count(d) == 8 for d in range(8))
    rows = [[int(x) for x in ln.split(":")[1].split()] for ln in run(host_probe, ["devices", "8", "2", "8"])]
    assert rows == [[(2 * s) % 8, (2 * s + 1) % 8] for s in range(8)]
    rows = [[int(x) for x in ln.split(":")[1].split()] for ln in run(host_probe, ["devices", "3", "4", "1"])]
    assert rows == [[0, 0, 0, 0]] * 3
    rows = [[int(x) for x in ln.split(":")[1].split()] for ln in run(host_probe, ["devices", "2", "3", "4"])]
    assert rows == [[0, 1, 2], [3, 0, 1]]


def test_concat_list_text(host_probe):
    """The text stage right behind the path (src/ffmpeg_executor.cpp:38-50, hand-derived: the reference's executor
    cannot run here — no ffmpeg): three lines per segment, `end <= start` skipped, two decimals of the exact binary
    value ("%.2f" == fmt {:.2f}: 0.125 -> 0.12, 0.375 -> 0.38, 2.675 -> 2.67 because 2.675 is 2.67499999...).
    C++ host layer and Python mirror must print the same bytes."""
    segs = [(0.0, 1.5), (7.540000000000001, 14.46), (3.0, 3.0), (5.0, 4.0), (0.125, 0.375), (2.675, 1e3 + 0.005),
            (59.533333333333339, 61.466666666666669)]
    want = ("file '/v/a b.mp4'\ninpoint 0.00\noutpoint 1.50\n"
            "file '/v/a b.mp4'\ninpoint 7.54\noutpoint 14.46\n"
            "file '/v/a b.mp4'\ninpoint 0.12\noutpoint 0.38\n"
            "file '/v/a b.mp4'\ninpoint 2.67\noutpoint 1000.00\n"
            "file '/v/a b.mp4'\ninpoint 59.53\noutpoint 61.47\n")
    assert m.concat_list(segs, "/v/a b.mp4") == want
    stdin = "".join(f"{a!r} {b!r}\n" for a, b in segs)
    out = subprocess.run([host_probe, "concat", "/v/a b.mp4"], input=stdin, capture_output=True, text=True, check=True).stdout
    assert out == want
    assert m.concat_list([], "/x") == "" and m.concat_list(np.zeros((0, 2)), "/x") == ""
    # fmt's {:.2f} prints EVERY digit of a huge value and never follows the C locale: neither may the C++ layer
    # (it once formatted into char[64] with snprintf: 1e300 lost its tail and its newline; a decimal-comma
    # locale would have printed "0,12")
    big = [(1e300, 1.7976931348623157e308), (0.125, 1e60)]
    want_big = m.concat_list(big, "/b")
    assert want_big.count("\n") == 6 and ("inpoint " + "%.2f" % 1e300 + "\n") in want_big and len(want_big) > 600
    stdin = "".join(f"{a!r} {b!r}\n" for a, b in big)
    for loc in ("C", "de_DE.UTF-8", "fr_FR.UTF-8"):
        out = subprocess.run([host_probe, "concat", "/b", "--setlocale"], input=stdin, capture_output=True, text=True,
                             check=True, env=dict(os.environ, LC_ALL=loc)).stdout
        assert out == want_big, loc


def _ref_has_sizing():
    if not os.path.exists(REF_PROBE):
        return False
    try:
        return subprocess.run([REF_PROBE, "sizing"], capture_output=True, text=True).stdout.strip() not in ("", "unavailable")
    except OSError:
        return False


@pytest.mark.skipif(not _ref_has_sizing(), reason="oracle/_ref/ref_host_probe was built without the reference's system.cpp")
def test_what_the_reference_would_choose_on_this_machine_vs_the_cpu_budget(host_probe):
    """Row a11 (stream fan-out).  How many streams the REFERENCE starts on this machine is asked of the reference's
    own object code (src/system.cpp compiled where it lies into oracle/_ref/ref_host_probe: detect_cpu_limit(),
    calculate_parallel_streams(), get_available_cpus(), src/system.cpp:107-197) — the product holds no restatement of
    it.  The host layer sizes from cpu_budget() instead (the cgroup quota when there is one), which can only be
    tighter: detect_cpu_limit() takes the LARGER of the quota and the cpuset's CPU count (:155-161)."""
    budget = int(run(host_probe, ["cpulimit"])[0])
    for ps in (None, "0", "3", "64"):
        env = {} if ps is None else {"PARALLEL_STREAMS": ps}
        ref = run(REF_PROBE, ["sizing"], env)
        assert len(ref) == 3 and ref[0].startswith("detect_cpu_limit ") and ref[1].startswith("calculate_parallel_streams ")
        limit, streams = int(ref[0].split()[1]), int(ref[1].split()[1])
        assert 1 <= budget <= max(limit, len(os.sched_getaffinity(0)))
        assert streams == (max(1, limit) if ps in (None, "0") else max(1, min(int(ps), limit)))   # its own documented rule
    for ps in ("abc",):                                        # std::stoi throws in the reference; here: a configuration error
        assert "throws" in run(REF_PROBE, ["sizing"], {"PARALLEL_STREAMS": ps})[1]


def test_batch_sizing_hand_cases(host_probe):
    """default_batch_sizing (host layer; its own rule, not src/batch_processor.cpp:81-95 — a worker here decodes and
    copies out, the scan runs on a GPU): streams = PARALLEL_STREAMS, or one per CPU of the budget and at least one per
    device, never more than there are videos; threads = THREADS_PER_STREAM, or two workers per CPU of the budget
    spread over the streams, at least one each.  (n_videos, n_devices, budget, configured streams, configured threads)."""
    cases = [((64, 1, 16, 0, 0), (16, 2)), ((64, 8, 16, 0, 0), (16, 2)), ((8, 1, 16, 0, 0), (8, 4)), ((3, 8, 16, 0, 0), (3, 11)),
             ((64, 1, 16, 64, 0), (64, 1)), ((64, 1, 16, 64, 1), (64, 1)), ((8, 1, 16, 4, 2), (4, 2)), ((2, 1, 16, 4, 0), (2, 16)),
             ((64, 8, 4, 0, 0), (8, 1)), ((1, 1, 1, 0, 0), (1, 2)), ((0, 0, 0, -3, -1), (1, 2)), ((5, 2, 3, 0, 0), (3, 2))]
    for args, want in cases:
        assert tuple(int(x) for x in run(host_probe, ["batchsizing"] + [str(a) for a in args])[0].split()) == want, args


def test_cpu_list_parser(host_probe):
    """parse_cpu_list (host layer): sysfs cpulists and MTGPU_CPU_WINDOW — numbers and ranges, order kept, duplicates
    dropped; anything else is refused (the caller then does not pin)."""
    def parse(t):
        return run(host_probe, ["cpulist", t])[0]
    assert parse("0-3") == "0,1,2,3" and parse("64-66,192-193") == "64,65,66,192,193"
    assert parse("0,2,4") == "0,2,4" and parse(" 5 , 7-8\n") == "5,7,8" and parse("3,1-3") == "3,1,2"
    assert parse("7") == "7" and run(host_probe, ["cpulist", ""]) in ([], [""])
    for bad in ("a", "1-", "-3", "4-2", "1,,2", "1,", "1;2", "0x10", "1-2-3"):
        assert parse(bad) == "invalid", bad


def test_cpu_window_of_another_device_asked_from_a_pinned_worker(host_probe):
    """cpu_window_for_device computes from the PROCESS's CPUs: a worker already pinned to device 0's window that is
    the first to ask for device 1's must not get an empty window (round 4 used the calling thread's mask: on a
    multi-GPU node every device but the first of a stream thread's came back empty and stayed so)."""
    n = len(os.sched_getaffinity(0))
    if n < 3:
        pytest.skip("needs three CPUs")
    out = run(host_probe, ["windowpinned"], {"MTGPU_CPU_WINDOW": "2"})[0].split()
    got = dict(zip(out[0::2], (int(x) for x in out[1::2])))
    assert got == {"allowed": n, "window0": 2, "window1": 2}


def test_cpu_gate_and_cpu_budget(host_probe):
    """The host layer's CPU budget for the gate: cpu_budget() = the cgroup QUOTA (v2 cpu.max, v1 cfs quota, rounded
    up), else the CPUs the process may run on — deliberately not the reference's detect_cpu_limit(), which takes the
    larger of quota and cpuset size (see test_what_the_reference_would_choose_...) — and the CpuGate never lets more workers fill
    batches at once than it has tokens (0 tokens = no gate)."""
    def want_limit():
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                return -(-int(q) // int(per))
        except OSError:
            pass
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                return -(-q // per)
        except (OSError, ValueError):
            pass
        return len(os.sched_getaffinity(0))
    assert int(run(host_probe, ["cpulimit"])[0]) == want_limit() >= 1
    peak, passes, waited, tokens = (int(x) for x in run(host_probe, ["gate", "3", "12", "20"])[0].split()[1::2])
    assert peak <= 3 and passes == 240 and waited == 1 and tokens == 3
    peak, passes, waited, tokens = (int(x) for x in run(host_probe, ["gate", "0", "8", "10"])[0].split()[1::2])
    assert passes == 80 and waited == 0 and tokens == 0 and 1 <= peak <= 8
    peak, passes, waited, tokens = (int(x) for x in run(host_probe, ["gate", "16", "4", "10"])[0].split()[1::2])
    assert peak <= 4 and passes == 40 and waited == 0


def test_cpu_window_next_to_the_device(host_probe):
    """pick_cpu_window (host layer): the workers of a device are confined to a window of CPUs next to it, sized from
    the CPU budget; the GPUs of a NUMA node get disjoint windows — cores first, then the SMT siblings of the same
    cores.  Hand cases on the GPU box's layout (node 1: "64-127,192-255", four GPUs) and on flat lists."""
    def win(local, idx, n, want):
        out = run(host_probe, ["window", local, str(idx), str(n), str(want)])[0]
        return [int(x) for x in out.split(",")] if out else []
    node1 = "64-127,192-255"
    assert win(node1, 0, 4, 16) == list(range(64, 80))
    assert win(node1, 3, 4, 16) == list(range(112, 128))
    assert win(node1, 3, 4, 24) == list(range(112, 128)) + list(range(64, 72))         # 24 CORES: on into the next part, wrapping
    assert win(node1, 1, 4, 24) == list(range(80, 104))
    assert win(node1, 0, 1, 80) == list(range(64, 128)) + list(range(192, 208))        # more than the node's cores: siblings
    assert all(len(set(win(node1, g, 4, 24))) == 24 and set(win(node1, g, 4, 24)) <= set(range(64, 128)) for g in range(4))
    assert win("0-7", 0, 1, 6) == [0, 1, 2, 3, 4, 5] and win("0-7", 1, 2, 6) == [4, 5, 6, 7, 0, 1]
    assert win("0,2,4,6,8,10,12,14", 1, 2, 3) == [8, 10, 12]                            # compose pins cpuset 0,2,...,14 in the reference
    assert win("0-7", 5, 2, 2) == [4, 5] and win("0-7", -1, 2, 2) == [0, 1]             # index clamped
    assert win("0-7", 0, 1, 0) == []
