"""HIP streams that really run side by side.

The loops that keep several pairs in flight (evaluate.evaluate_pairs: two; RegistrationPipeline: `depth`) put consecutive pairs on
different HIP streams.  The HIP runtime multiplexes ALL streams of a process onto a few hardware queues (four on ROCm 7) by a rule of
its own: measured on MI355X it is neither creation order modulo four nor anything a caller can read back, it changes with what the
process created before, and two streams on ONE queue run strictly one after the other.  Consequences measured (tools/
exp_eval_pairs_matrix.py, exp_stream_classes.py; profiles/r06/stream_classes.txt): evaluate_pairs 440 pairs/s on two streams of
different queues, 310 on two streams of one queue, 380 when one of them shares the queue of the process's NULL stream -- and which
of these a run got depended on how many streams the process happened to create earlier (BENCH r05 446 vs 375 on the same code).

So the streams are chosen by MEASUREMENT, once per device and process: umereg_streams_run_side_by_side (a spin kernel on one stream,
a one-thread kernel on the other) sorts candidate streams of torch's pool into classes (= hardware queues); `concurrent_streams`
hands out one stream per class, the null stream's class last.  ~30 ms, results do not depend on it -- only which kernels overlap.
"""
import ctypes

import torch

from . import _lib

_classes = {}          # device index -> list of classes, each a list of torch.cuda.Stream; class 0 = the null stream's
_report = {}           # device index -> dict for logs (bench_detail.json)
SPIN_MS = 0.25
MAX_CANDIDATES = 24


def run_side_by_side(a, b, spin_ms=SPIN_MS):
    """True if work on stream `b` proceeds while stream `a` is busy (different hardware queues).  Synchronises both."""
    lib = _lib.load()
    out = ctypes.c_int(0)
    ms = ctypes.c_float(0.0)
    with torch.cuda.device(a.device):
        rc = lib.umereg_streams_run_side_by_side(a.cuda_stream, b.cuda_stream, float(spin_ms), ctypes.byref(out), ctypes.byref(ms))
    _lib.check(rc, "umereg_streams_run_side_by_side")
    return bool(out.value)


def stream_classes(device, want=3, per_class=1, refresh=False):
    """Candidate streams of torch's pool sorted into classes that run side by side (hardware queues), measured.  Class 0 is the null
    stream's (a stream in it is serialised with everything the process puts on the default stream).  Stops once `want` classes
    beside the null stream's hold `per_class` streams each, or MAX_CANDIDATES streams were tried."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()

    def enough(cl):
        return len(cl) - 1 >= want and all(len(c) >= per_class for c in cl[1:want + 1])
    if idx in _classes and not refresh and (enough(_classes[idx]) or _report[idx]["candidates_tried"] >= MAX_CANDIDATES):
        return _classes[idx]
    with torch.cuda.device(idx):
        null = torch.cuda.default_stream(idx)
        classes = _classes.get(idx) if not refresh else None
        tried = _report[idx]["candidates_tried"] if classes else 0
        if not classes:
            run_side_by_side(null, null)                              # (warm-up: module load, first launch)
            classes = [[null]]
        seen = {s_.cuda_stream for c in classes for s_ in c}
        while tried < MAX_CANDIDATES and not enough(classes):
            s = torch.cuda.Stream(idx)
            tried += 1
            if s.cuda_stream in seen:                                 # (torch's pool wrapped around)
                continue
            seen.add(s.cuda_stream)
            for c in classes:
                if not run_side_by_side(c[0], s):
                    c.append(s)
                    break
            else:
                classes.append([s])
    _classes[idx] = classes
    _report[idx] = {"hardware_queues_seen": len(classes), "classes_beside_null": len(classes) - 1, "candidates_tried": tried,
                    "streams_sharing_the_null_streams_queue": len(classes[0]) - 1, "class_sizes": [len(c) for c in classes]}
    return classes


def concurrent_streams(device, n):
    """n DISTINCT torch streams for work that is meant to overlap: dealt round-robin over the measured classes that do not hold the
    null stream (up to three), so that consecutive entries always run side by side; entries i and i + (number of classes) share a
    queue -- like the runtime would have dealt them, but never the null stream's.  Cached per device: the same streams every time
    (the native workspaces are per stream)."""
    k = max(1, min(n, 3))
    classes = stream_classes(device, want=k, per_class=(n + k - 1) // k)
    order = [c for c in classes[1:k + 1]]
    if not order:                                                      # (one hardware queue only: nothing to choose from)
        order = [classes[0][1:]] if len(classes[0]) > 1 else []
    out = []
    for i in range(n):
        c = order[i % len(order)] if order else []
        j = i // max(len(order), 1)
        out.append(c[j] if j < len(c) else torch.cuda.Stream(device))
    return out


def report(device):
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    return dict(_report.get(idx) or {})
