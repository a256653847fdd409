"""
Seed plumbing.  The reference draws from the global legacy numpy.random state with no seed
arguments anywhere; the device uses counter-based Philox streams keyed by
(seed, ray stream id, event).  This module owns the process-wide seed and hands a fresh 64-bit
seed to every source / trace / optics call that was not given one, so that repeated calls draw
independent streams like successive numpy.random calls do, and `seed(s)` makes a whole script
reproducible.
"""
import os
import threading

_MASK = (1 << 64) - 1
_lock = threading.Lock()
_state = {'seed': int.from_bytes(os.urandom(8), 'little'), 'calls': 0, 'rays': 0}


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _MASK
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
    return z ^ (z >> 31)


def seed(s):
    """Make every following draw of this process reproducible."""
    with _lock:
        _state['seed'] = int(s) & _MASK
        _state['calls'] = 0
        _state['rays'] = 0


def next_seed():
    """A fresh stream seed for one call."""
    with _lock:
        _state['calls'] += 1
        return _splitmix64((_state['seed'] + 0x632BE59BD9B4E019 * _state['calls']) & _MASK)


def current_seed():
    return _state['seed']
