"""Generates tests/golden/reference_host_vectors.json by RUNNING THE REFERENCE ITSELF.

oracle/_ref/ref_host_probe (recipe: `make -C oracle ref`) is oracle/ref_host_probe.cpp compiled together
with the reference's own src/task_queue.cpp, src/ffmpeg_queue.cpp, include/motion_trim/config.hpp and
types.hpp from where they lie under /root/reference — the std-only part of the scan path.  This script
feeds it environment sets and queue scripts and records what the reference answered:

  config cases   every Config getter under an environment: defaults, the shipped config/motion_trim.env
                 (its KEY=VALUE lines, read as data), and parse edge cases of std::stoi / stod / stof
                 (prefix parse, hex, whitespace, range errors, the uint8 cast of VECTORS_NEEDED)
  memo cases     get / setenv / get again inside one process: the getters are function-local statics
                 (config.hpp:56-59), so the first successful parse sticks and a throwing one is retried
  layout         sizeof / alignof / offsets of TimeSegment and ScanTask
  queue cases    scripted push / pop / finish on TaskQueue, ResultCollector::add / extract order,
                 FFmpegQueue push / pop / finish / is_done / empty

Run in the build container (the reference tree does not exist on the GPU box):
    make -C oracle ref && python tests/golden/make_reference_host_vectors.py
The fixture is data (inputs + the reference's printed answers); tests/test_reference_host.py replays it
against config.py and the C++ host layer (tests/cpp/host_probe.cpp over csrc/host/mtgpu_host.hpp).
"""
import json
import os
import random
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
PROBE = os.path.join(ROOT, "oracle", "_ref", "ref_host_probe")
ENV_FILE = "/root/reference/config/motion_trim.env"

CONFIG_VARS = ["MV_THRESHOLD_SQ", "BLOCK_SIZE", "BLOCK_SHIFT", "VECTORS_NEEDED", "CLUSTERS_NEEDED", "VERTICAL_MASK",
               "MAX_GAP_SEC", "PADDING_SEC", "CHUNK_DURATION_SEC", "TARGET_FPS", "MIN_SAVINGS_PCT",
               "PARALLEL_STREAMS", "THREADS_PER_STREAM"]

INT_STRINGS = ["7", " 7", "\t7", "+7", "-3", "007", "7abc", "7 ", "0x10", "1_000", "3.9", "-0", "", " ", "abc", "-",
               "2147483647", "2147483648", "-2147483648", "-2147483649", "99999999999999999999", "1e3"]
U8_STRINGS = ["0", "1", "4", "255", "256", "300", "-1", "-255", "511", "65536"]
F64_STRINGS = ["4", "4.5", "-4.5", "1e2", "1E2", ".5", "5.", "0x10", "0x1p-2", "inf", "-inf", "infinity", "nan",
               "nan(0x1)", "1e400", "-1e400", "1e-400", "4.9e-324", "2.2250738585072014e-308", "1,5", " 2.5x",
               "2.5 ", "", "abc", "+", "0.1", "0.30000000000000004", "1e", "1e+", "--1"]
F32_STRINGS = ["0.05", "0.1", "0.3", "0", "-0.0", "0.5", "0.49", "0.25", "1e39", "-1e39", "1e-50", "1e-40", "16777217",
               "1.000000178813934326171874999999", "1.00000017881393432617187500", "0.1abc", "abc", "",
               "inf", "nan", "0x1p-4", " 0.2", "3.4028235e38", "3.4028236e38"]


def run(args, env=None, stdin=None):
    e = {"PATH": os.environ.get("PATH", "")}
    e.update(env or {})
    out = subprocess.run([PROBE] + args, env=e, input=stdin, capture_output=True, text=True, check=True)
    return out.stdout.splitlines()


def config_case(name, env):
    return {"name": name, "env": env, "answers": run(["config"], env)}


def shipped_env():
    env = {}
    for ln in open(ENV_FILE):
        ln = ln.strip()
        if ln and not ln.startswith("#") and "=" in ln:
            k, v = ln.split("=", 1)
            if k in CONFIG_VARS:
                env[k] = v
    return env


def queue_scripts():
    rng = random.Random(20261004)
    scripts = []
    scripts.append(("fifo_then_finish", ["tpush 0 30 0", "tpush 30 60 1", "tpush 60 61.5 2", "tpop", "tpop", "tpop",
                                          "tpop", "tfinish", "tpop", "tpop"]))
    scripts.append(("finish_with_backlog", ["tpush 0 30 0", "tpush 30 60 1", "tfinish", "tpop", "tpush 60 90 2",
                                             "tpop", "tpop", "tpop"]))
    scripts.append(("collector_order", ["radd 3 1.5 0.25 9", "radd 0", "radd 2 0.25 -1", "rextract", "rextract",
                                         "radd 1 7", "rextract"]))
    scripts.append(("jobs", ["jempty", "jdone", "jpush 3 2 0 1.5 4 9.25", "jpush 1 0", "jempty", "jdone", "jpop",
                             "jfinish", "jdone", "jpop", "jdone", "jempty", "jpop", "jpush 7 1 2 3", "jdone", "jpop",
                             "jpop"]))
    for k in range(6):
        ops, tid = [], 0
        for _ in range(60):
            c = rng.random()
            if c < 0.30:
                s = round(rng.uniform(0, 3600), 3)
                ops.append(f"tpush {s} {s + rng.choice([30, 60, 0.5])} {tid}")
                tid += 1
            elif c < 0.55:
                ops.append("tpop")
            elif c < 0.60:
                ops.append("tfinish")
            elif c < 0.72:
                n = rng.randrange(0, 5)
                ops.append(("radd %d %s" % (n, " ".join(repr(round(rng.uniform(0, 100), 6)) for _ in range(n)))).strip())
            elif c < 0.77:
                ops.append("rextract")
            elif c < 0.87:
                n = rng.randrange(0, 4)
                segs = []
                t = 0.0
                for _ in range(n):
                    a = t + rng.uniform(0, 10)
                    b = a + rng.uniform(0.1, 20)
                    segs += [repr(a), repr(b)]
                    t = b
                ops.append(("jpush %d %d %s" % (rng.randrange(0, 64), n, " ".join(segs))).strip())
            elif c < 0.96:
                ops.append(rng.choice(["jpop", "jdone", "jempty"]))
            else:
                ops.append("jfinish")
        ops += ["tfinish", "tpop", "tpop", "jfinish", "jpop", "jpop", "rextract"]
        scripts.append((f"random_{k}", ops))
    return scripts


def memo_scripts():
    rng = random.Random(20261005)
    getters = [v.lower() for v in CONFIG_VARS]
    scripts = [
        ("first_parse_sticks", {}, ["get target_fps", "set TARGET_FPS 10", "get target_fps", "get chunk_duration_sec",
                                    "set CHUNK_DURATION_SEC 4", "get chunk_duration_sec", "set MAX_GAP_SEC 2",
                                    "get max_gap_sec", "set MAX_GAP_SEC 3", "get max_gap_sec", "unset MAX_GAP_SEC",
                                    "get max_gap_sec"]),
        ("value_at_first_call_not_at_start", {"VECTORS_NEEDED": "4"},
         ["set VECTORS_NEEDED 1", "get vectors_needed", "set VECTORS_NEEDED 2", "get vectors_needed",
          "unset VECTORS_NEEDED", "get vectors_needed"]),
        ("throwing_parse_is_retried", {}, ["set VECTORS_NEEDED abc", "get vectors_needed", "get vectors_needed",
                                           "set VECTORS_NEEDED 300", "get vectors_needed", "set VECTORS_NEEDED 3",
                                           "get vectors_needed", "set MV_THRESHOLD_SQ 1e400", "get mv_threshold_sq",
                                           "set MV_THRESHOLD_SQ 0x10", "get mv_threshold_sq", "set MV_THRESHOLD_SQ 4",
                                           "get mv_threshold_sq", "set VERTICAL_MASK", "get vertical_mask",
                                           "unset VERTICAL_MASK", "get vertical_mask", "set VERTICAL_MASK 0.2",
                                           "get vertical_mask"]),
        ("default_sticks_too", {}, ["get block_size", "get block_shift", "set BLOCK_SIZE 4", "set BLOCK_SHIFT 2",
                                    "get block_size", "get block_shift", "get clusters_needed",
                                    "set CLUSTERS_NEEDED 0", "get clusters_needed"]),
    ]
    pool = ["3", "4.75", "abc", "", " 9", "0x1F", "1e400", "300", "-1", "12abc", "99999999999", "0.0625"]
    for k in range(6):
        ops = []
        for _ in range(50):
            c = rng.random()
            g = rng.choice(getters)
            if c < 0.5:
                ops.append(f"get {g}")
            elif c < 0.9:
                ops.append(f"set {g.upper()} {rng.choice(pool)}")
            else:
                ops.append(f"unset {g.upper()}")
        scripts.append((f"random_{k}", {}, ops))
    return scripts


def main():
    if not os.path.exists(PROBE):
        raise SystemExit("build it first: make -C oracle ref")
    cases = [config_case("code_defaults", {}), config_case("shipped_env_file", shipped_env())]
    for s in INT_STRINGS:
        cases.append(config_case(f"int:{s!r}", {v: s for v in ["BLOCK_SIZE", "BLOCK_SHIFT", "CLUSTERS_NEEDED",
                                                                  "VECTORS_NEEDED", "PARALLEL_STREAMS",
                                                                  "THREADS_PER_STREAM"]}))
    for s in U8_STRINGS:
        cases.append(config_case(f"u8:{s!r}", {"VECTORS_NEEDED": s}))
    for s in F64_STRINGS:
        cases.append(config_case(f"f64:{s!r}", {v: s for v in ["MV_THRESHOLD_SQ", "MAX_GAP_SEC", "PADDING_SEC",
                                                                  "CHUNK_DURATION_SEC", "TARGET_FPS",
                                                                  "MIN_SAVINGS_PCT"]}))
    for s in F32_STRINGS:
        cases.append(config_case(f"f32:{s!r}", {"VERTICAL_MASK": s}))
    queues = [{"name": n, "script": ops, "answers": run(["queue"], stdin="\n".join(ops) + "\n")}
              for n, ops in queue_scripts()]
    memos = [{"name": n, "env": env, "script": ops, "answers": run(["memo"], env, stdin="\n".join(ops) + "\n")}
             for n, env, ops in memo_scripts()]
    doc = {
        "source": "the reference's own config.hpp / types.hpp / src/task_queue.cpp / src/ffmpeg_queue.cpp compiled "
                  "and run in the build container through oracle/ref_host_probe.cpp (make -C oracle ref); "
                  "generator: tests/golden/make_reference_host_vectors.py",
        "answer_format": "config: '<getter> f64|f32|int <value> [0x<bits>]' or '<getter> error <std exception>'; "
                         "queue / memo: one line per script op",
        "layout": run(["layout"]),
        "race": run(["race", "2000", "8"]),
        "config": cases,
        "memo": memos,
        "queue": queues,
    }
    path = os.path.join(HERE, "reference_host_vectors.json")
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)
    print(path, len(cases), "config cases,", len(queues), "queue scripts")


if __name__ == "__main__":
    main()
