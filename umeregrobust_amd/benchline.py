"""The ONE line bench.py prints, and the launcher behind `python bench.py --gpus N`.

The driver keeps the last 8 KB of stdout and parses the final line as JSON.  Round 4's line carried every note, stage table and
counter block of the run (22 KB) and did not survive that; since round 5 the full result goes to a file (`bench_detail.json`) and
the line is a bounded extract of it: `compact()` keeps the keys of the bench contract (metric / value / ... / config / roofline /
cpu_baseline), shortens every kernel's roofline to the same few numeric keys, drops prose, and asserts the size.  Torch-free and
numpy-free: tests/test_host_logic.py builds a line from a canned result on the CPU.
"""
import json
import os
import socket
import sys

LINE_LIMIT = 4096          # bytes; the driver's stdout tail is 8 KB

ROOFLINE_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "in_situ_avg_launch_ms")
REQUIRED_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline", "cpu_baseline")


def _roof(r):
    if not r:
        return None
    out = {k: r.get(k) for k in ROOFLINE_KEYS}
    for k in ("counters_match_library", "peak_source", "traffic_frac"):
        if k in r:
            out[k] = r[k]
    return out


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[:n - 3] + "..."


def compact(result, detail_path=None):
    """The bounded extract of bench.py's full result dict (see the module docstring) -> dict."""
    cfg = result.get("config") or {}
    hard = cfg.get("named_path_on_hard_pairs") or {}
    rag = cfg.get("named_path_on_ragged_pairs") or {}
    e2e = dict(cfg.get("end_to_end_pairs_per_s") or {})
    e2e.pop("what", None)
    world = result.get("world") or {}
    out = {k: result.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                      "vs_baseline", "dtype", "data")}
    out["config"] = {
        "workload": _short(cfg.get("workload", ""), 260),
        "pairs_per_step_per_gpu": cfg.get("pairs_per_step_per_gpu"),
        "ms_per_pair": cfg.get("ms_per_pair"),
        "sharding": cfg.get("sharding"),
        "value_is": _short(cfg.get("value_is_short") or cfg.get("value_is", ""), 200),
        "end_to_end_pairs_per_s": e2e,
        "named_path_on_hard_pairs": {"pairs_per_s": hard.get("pairs_per_s")} if hard else None,
        "named_path_on_ragged_pairs": {k: rag.get(k) for k in ("pairs_per_s", "ratio_to_value", "graphs_captured_during_the_leg")} if rag else None,
        "stream_plan_check": (cfg.get("stream_plan_check") or {}).get("ratio_shipped_over_creation_order"),
        "world": {"ranks": world.get("ranks"), "backend": world.get("backend"), "launched_by": world.get("launched_by")},
    }
    out["roofline"] = _roof(result.get("roofline"))
    out["rooflines"] = {k: _roof(v) for k, v in (result.get("rooflines") or {}).items()}
    cb = result.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind") if k in cb}
        out["cpu_baseline"]["sample"] = _short(cb.get("sample", ""), 160)
        for k in ("rr_pairs", "rr_pairs_with_a_different_gate_outcome"):
            if k in cb:
                out["cpu_baseline"][k] = cb[k]
    else:
        out["cpu_baseline"] = None      # (N > 1, or --no-cpu-baseline: rank 0 at N = 1 only, by the bench contract)
    # recall: the discriminating numbers (hard pairs: the reference's port and this library on replayed draws, the KT-size hard pairs);
    # a run without those legs (N > 1: no CPU leg) falls back to the end-to-end leg's own pairs, labelled as what they are
    e = result.get("end_to_end") or {}
    if result.get("recall"):
        out["recall"] = result["recall"]
    elif e:
        out["recall"] = {"pairs_are": "exact rigid copies (cannot fail)",
                         **{k: e.get(k) for k in ("rr_1.5deg_0.6m", "rr_1.5deg_0.3m", "rr_1deg_0.1m", "mRRE_deg", "mRTE_m")}}
    if "counters_match_library" in result:
        out["counters_match_library"] = result["counters_match_library"]
    out["detail"] = detail_path
    return out


def line(result, detail_path=None):
    """compact(result) as ONE JSON line of at most LINE_LIMIT bytes (strictly parseable: no NaN / Infinity)."""
    c = compact(result, detail_path)
    s = json.dumps(c, allow_nan=False, separators=(", ", ": "))
    if len(s.encode()) > LINE_LIMIT:          # never reached with the keys above; if a future key overflows, prose goes first
        c["config"]["workload"] = _short(c["config"]["workload"], 80)
        c["config"]["value_is"] = _short(c["config"]["value_is"], 60)
        if c.get("cpu_baseline"):
            c["cpu_baseline"]["sample"] = _short(c["cpu_baseline"]["sample"], 40)
        s = json.dumps(c, allow_nan=False, separators=(", ", ": "))
    if len(s.encode()) > LINE_LIMIT:
        raise ValueError(f"bench line is {len(s.encode())} bytes (> {LINE_LIMIT}): move the new keys to bench_detail.json")
    return s


def safe_line(result, detail_path=None):
    """line(), and if that raises (a key that does not serialise, an oversized extract): the contract's keys alone -- a timed run must
    not end without its line, and rank 0 must reach the barrier the other ranks wait in."""
    try:
        return line(result, detail_path)
    except Exception as e:   # noqa: BLE001
        print(f"[bench] full line failed ({e!r}); printing the required keys only", file=sys.stderr)
        c = {k: result.get(k) for k in REQUIRED_KEYS}
        cfg = result.get("config") or {}
        c["config"] = {"workload": _short(cfg.get("workload", ""), 200)}
        c["roofline"] = _roof(result.get("roofline"))
        cb = result.get("cpu_baseline")
        c["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind") if k in cb} if cb else None
        c["detail"] = detail_path
        return json.dumps(sanitize(c), allow_nan=False, separators=(", ", ": "), default=str)


def sanitize(o):
    """NaN / +-Infinity -> None, numpy scalars -> Python numbers (json.dumps(allow_nan=False) must accept the result)."""
    if isinstance(o, dict):
        return {str(k): sanitize(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [sanitize(v) for v in o]
    if isinstance(o, bool) or o is None or isinstance(o, (str, int)):
        return o
    if isinstance(o, float):
        return o if o == o and o not in (float("inf"), float("-inf")) else None
    if hasattr(o, "item"):
        return sanitize(o.item())
    return str(o)


def write_detail(result, path):
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        json.dump(result, f, indent=1, allow_nan=False)
        f.write("\n")
    return path


# ---- `python bench.py --gpus N` as typed --------------------------------------------------------------------------------------------

def _arg_value(argv, name, default=None):
    for i, v in enumerate(argv):
        if v == name and i + 1 < len(argv):
            return argv[i + 1]
        if v.startswith(name + "="):
            return v.split("=", 1)[1]
    return default


def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch_command(script, argv, environ):
    """The command a plain `python <script> --gpus N ...` (N > 1, no WORLD_SIZE in the environment) re-executes itself as: one
    process per GPU under torch.distributed.run on this node, rendezvous on 127.0.0.1 at a free port -- or None when the process
    is already a rank of a launched job (WORLD_SIZE set: the driver's torch.distributed.run form) or N <= 1."""
    try:
        n = int(_arg_value(argv, "--gpus", "1"))
    except ValueError:
        return None
    if n <= 1 or "WORLD_SIZE" in environ:
        return None
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(free_port()), script] + list(argv)


def maybe_self_launch(script, argv=None, environ=None):
    """Replaces this process by the launcher when self_launch_command() says so (does not return then)."""
    argv = sys.argv[1:] if argv is None else argv
    environ = os.environ if environ is None else environ
    cmd = self_launch_command(script, argv, environ)
    if cmd is None:
        return
    env = dict(environ)
    env["UMEREG_BENCH_SELF_LAUNCHED"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print(f"[bench] --gpus {_arg_value(argv, '--gpus')} without a launcher: starting " + " ".join(cmd[1:8]) + " ...", file=sys.stderr, flush=True)
    sys.stdout.flush()
    os.execve(cmd[0], cmd, env)
