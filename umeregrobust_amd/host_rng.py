"""Host-side RNG step of the path: the tau-weighted match sub-sample of reference evaluate.py:238,
`np.random.choice(n, size, replace=False, p=prob)`.

`choice_noreplace` returns exactly what numpy's legacy RandomState.choice returns (same indices, same
order, same RNG state afterwards): the uniforms still come from the caller's RandomState / the global
np.random stream, only the per-round cumsum / searchsorted / unique bookkeeping runs natively
(umereg_host_choice_round) instead of ~10 small numpy calls per round.
"""
import threading

import numpy as np

from . import _lib


def _is_legacy(rng):
    return rng is np.random or isinstance(rng, np.random.RandomState)


_addr_cache = {}


def _mt19937_of(rng):
    """The MT19937 BitGenerator behind a legacy RandomState / the np.random module, or None."""
    if rng is np.random:
        rng = getattr(np.random.mtrand, "_rand", None)
    bitgen = getattr(rng, "_bit_generator", None)
    if isinstance(bitgen, np.random.MT19937) and hasattr(bitgen, "ctypes") and hasattr(bitgen, "lock"):
        return bitgen
    return None


def _state_address(bitgen):
    k = id(bitgen)
    ent = _addr_cache.get(k)
    if ent is None or ent[0] is not bitgen:
        if len(_addr_cache) > 64:
            _addr_cache.clear()
        ent = _addr_cache[k] = (bitgen, int(bitgen.ctypes.state_address))
    return ent[1]


_ERRORS = {1: "probabilities contain NaN or are not non-negative", 2: "probabilities do not sum to 1",
           3: "Fewer non-zero entries in p than size"}
_tls = threading.local()     # scratch buffers are per host thread: the native calls run without the GIL


def _thread_scratch(name):
    d = getattr(_tls, name, None)
    if d is None:
        d = {}
        setattr(_tls, name, d)
    return d


def choice_noreplace(rng, n, size, p):
    """Bit-identical replacement for rng.choice(n, size, replace=False, p=p) (legacy numpy RNGs: same
    indices, same order, same generator state afterwards).  Any other generator type is forwarded to its
    own .choice().  One native call (umereg_host_choice_mt19937) on the generator's MT19937 state."""
    if not _is_legacy(rng):
        return rng.choice(n, size, replace=False, p=p)
    lib = _lib.load()
    p = np.asarray(p)
    if p.dtype != np.float32:
        p = p.astype(np.float64, copy=False)
    p = np.ascontiguousarray(p).ravel()
    if p.shape[0] != n:
        raise ValueError("'a' and 'p' must have same size")
    if size > n:
        raise ValueError("Cannot take a larger sample than population when 'replace=False'")
    if size <= 0:
        return rng.choice(n, size, replace=False, p=p)
    _scratch = _thread_scratch("choice")
    sc = _scratch.get((n, size))
    if sc is None:
        work = np.empty(2 * n + size, dtype=np.float64)
        seen = np.empty(n, dtype=np.uint8)
        pos = np.zeros(1, dtype=np.int32)
        sc = _scratch[(n, size)] = (work, seen, pos, work.ctypes.data, seen.ctypes.data, pos.ctypes.data)
        if len(_scratch) > 16:
            _scratch.pop(next(iter(_scratch)))
    work, seen, pos, work_p, seen_p, pos_p = sc
    found = np.empty(size, dtype=np.int64)
    is_f32 = int(p.dtype == np.float32)
    # numpy's legacy RandomState wraps an MT19937 bit generator whose state struct {uint32 key[624]; int pos;}
    # is exposed through the documented BitGenerator.ctypes interface: advance it in place, under its lock
    bitgen = _mt19937_of(rng)
    if bitgen is not None:
        addr = _state_address(bitgen)
        with bitgen.lock:
            rc = lib.umereg_host_choice_mt19937(addr, addr + 624 * 4, p.ctypes.data, is_f32, n, size,
                                                found.ctypes.data, work_p, seen_p, None)
    else:
        st = rng.get_state()
        if st[0] != "MT19937":
            return rng.choice(n, size, replace=False, p=p)
        key = st[1]
        pos[0] = st[2]
        rc = lib.umereg_host_choice_mt19937(key.ctypes.data, pos_p, p.ctypes.data, is_f32, n, size,
                                            found.ctypes.data, work_p, seen_p, None)
        if rc == 0:
            rng.set_state((st[0], key, int(pos[0]), st[3], st[4]))
    if rc != 0:
        raise ValueError(_ERRORS.get(rc, "umereg_host_choice_mt19937 failed"))
    return found


def choice_uniform_noreplace(rng, n, size):
    """Bit-identical replacement for rng.choice(n, size, replace=False) WITHOUT p (the keypoint draws of reference
    evaluate.py:199-200 and the correlation sub-sampling of :280, :284): numpy's legacy RandomState computes
    permutation(n)[:size]; one native call does the same shuffle on the generator's MT19937 state (same indices, same
    order, same state afterwards; ~4x faster).  Other generators are forwarded to their own .choice()."""
    n, size = int(n), int(size)
    bitgen = _mt19937_of(rng) if _is_legacy(rng) else None
    if bitgen is None or size > n or n <= 0 or n > 0xffffffff:
        return rng.choice(n, size, replace=False)
    lib = _lib.load()
    _perm_scratch = _thread_scratch("perm")
    perm = _perm_scratch.get(n)
    if perm is None:
        if len(_perm_scratch) > 16:
            _perm_scratch.clear()
        perm = _perm_scratch[n] = np.empty(n, dtype=np.int64)
    out = np.empty(size, dtype=np.int64)
    addr = _state_address(bitgen)
    with bitgen.lock:
        rc = lib.umereg_host_permutation_mt19937(addr, addr + 624 * 4, n, size, perm.ctypes.data, out.ctypes.data)
    if rc != 0:
        raise ValueError("umereg_host_permutation_mt19937 failed")
    return out


class RecordingRNG:
    """Wraps a generator and keeps every `choice` result in order: the host draws of one evaluation-loop iteration (keypoints x 2,
    weighted match draw, correlation sub-samples x 2) can then be replayed into another implementation of the loop."""

    def __init__(self, rng):
        self.rng, self.log = rng, []

    def choice(self, *args, **kw):
        out = self.rng.choice(*args, **kw)
        self.log.append(np.array(out, copy=True))
        return out


class ReplayRNG:
    """Hands out recorded draws in order (see RecordingRNG); checks that each fits the call it answers."""

    def __init__(self, log):
        self.log = list(log)

    def choice(self, n, size=None, replace=True, p=None):
        if not self.log:
            raise ValueError("ReplayRNG: no recorded draw left")
        out = self.log.pop(0)
        if out.shape[0] != size or (out.size and int(out.max()) >= n):
            raise ValueError(f"ReplayRNG: recorded draw of {out.shape[0]} indices (max {int(out.max()) if out.size else -1}) "
                             f"does not fit choice({n}, {size})")
        return out
