"""Host-side RNG step of the path: the tau-weighted match sub-sample of reference evaluate.py:238,
`np.random.choice(n, size, replace=False, p=prob)`.

`choice_noreplace` returns exactly what numpy's legacy RandomState.choice returns (same indices, same
order, same RNG state afterwards): the uniforms still come from the caller's RandomState / the global
np.random stream, only the per-round cumsum / searchsorted / unique bookkeeping runs natively
(umereg_host_choice_round) instead of ~10 small numpy calls per round.
"""
import numpy as np

from . import _lib


def _is_legacy(rng):
    return rng is np.random or isinstance(rng, np.random.RandomState)


def choice_noreplace(rng, n, size, p):
    """Bit-identical replacement for rng.choice(n, size, replace=False, p=p) (legacy numpy RNGs).
    Any other generator type is forwarded to its own .choice()."""
    if not _is_legacy(rng):
        return rng.choice(n, size, replace=False, p=p)
    lib = _lib.load()
    p64 = np.array(p, dtype=np.float64, copy=True, order="C").ravel()
    if p64.shape[0] != n:
        raise ValueError("'a' and 'p' must have same size")
    if size > n:
        raise ValueError("Cannot take a larger sample than population when 'replace=False'")
    chk = np.empty(3, dtype=np.float64)
    lib.umereg_host_choice_check(p64.ctypes.data, n, chk.ctypes.data)
    if chk[2] != 0.0:
        raise ValueError("probabilities contain NaN or are not non-negative")
    pd = np.asarray(p).dtype
    atol = np.sqrt(np.finfo(np.float64).eps)
    if np.issubdtype(pd, np.floating):
        atol = max(atol, np.sqrt(np.finfo(pd).eps))          # numpy relaxes the tolerance for float32 input
    if abs(chk[0] - 1.0) > atol:
        raise ValueError("probabilities do not sum to 1")
    if chk[1] < size:
        raise ValueError("Fewer non-zero entries in p than size")
    found = np.empty(size, dtype=np.int64)
    cdf = np.empty(n, dtype=np.float64)
    seen = np.zeros(n, dtype=np.uint8)
    n_uniq = 0
    while n_uniq < size:
        x = rng.rand(size - n_uniq)
        n_new = lib.umereg_host_choice_round(p64.ctypes.data, n, x.ctypes.data, x.shape[0], found.ctypes.data, n_uniq,
                                             cdf.ctypes.data, seen.ctypes.data)
        if n_new < 0:
            raise RuntimeError("umereg_host_choice_round failed")
        n_uniq += n_new
    return found
