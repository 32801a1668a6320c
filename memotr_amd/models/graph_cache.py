"""The one cache behind every hipGraph capture of the model (decoder loop, batched encode, forward-only inference
graphs): keyed entries in least-recently-used order, a capture on first use, and a guard that stops capturing when
keys never recur.

Why a guard: multi-scale / random-crop training gives almost every clip its own pyramid.  A capture costs two warm-up
runs plus the capture itself (~3x an eager pass, forward and backward) and a private memory pool, so capturing per clip
would cost more than the graphs save.  After ``MISS_LIMIT`` consecutive NEW keys without one replay in between the
cache stops capturing; replays of what is already captured continue, and after ``RETRY_AFTER`` eager calls it re-arms
(the input sizes may have settled).  ``MEMOTR_REQUIRE_GRAPHS=1`` (set by bench.py) never gives up: there a capture
failure is an error, not a silent eager fallback.
"""
from __future__ import annotations

import os
from collections import OrderedDict

MISS_LIMIT = 12
RETRY_AFTER = 200


PROBE_EVERY = 32         # polls of one cache between probes of its records that sit at a level without windows


def selector_signature(owner, root) -> int:
    """The kernel-selection signature of the deformable-attention modules under ``root`` (csrc/msda_select.h): a capture
    bakes their kernel choice in, replayed launches keep counting the points that leave their windows, and a cache keyed
    on this replays -- or captures -- the graph of the levels in force.  Only the modules the graphs of ``owner`` hold
    are hashed (a level move or probe of some other module's record used to change every cache's key: advisor, round 5),
    and ``owner`` counts its own polls: every ``PROBE_EVERY``-th one announces the probe one level down."""
    from .. import _lib
    from ..modules.ms_deform_attn import MSDeformAttn
    mods = owner.__dict__.get("_sel_modules")
    if mods is None:
        mods = owner.__dict__["_sel_modules"] = [m for m in root.modules() if isinstance(m, MSDeformAttn)]
    sites = [m.__dict__["_msda_site"] for m in mods if m.__dict__.get("_msda_site") is not None]
    owner.__dict__["_sel_polls"] = polls = owner.__dict__.get("_sel_polls", 0) + 1
    return _lib.selector_poll_sites(sites, probe=polls % PROBE_EVERY == 0)


def require_graphs() -> bool:
    return os.environ.get("MEMOTR_REQUIRE_GRAPHS", "0") == "1"


class GraphCache:
    """``lookup(key, capture)`` -> the entry stored under ``key`` or None (the caller then runs eagerly).

    ``capture()`` builds the entry on a miss; it returns None when the capture failed (and is expected to have set
    ``failed`` and counted itself in ``captures`` otherwise).  Counters: ``captures`` graphs built, ``replays`` calls
    served from a graph (counted by the owner), ``eager`` eligible calls that ran eagerly."""

    def __init__(self, what: str, max_graphs: int, grow_cap: int = 1):
        """``grow_cap`` > 1: when a key that was EVICTED is asked for again the limit doubles, up to ``grow_cap`` times
        the initial one -- an LRU cache one entry short of its working set misses on every lookup, and a capture costs
        ~0.1 s; the consecutive-miss guard below does not see that pattern (its misses are interleaved with hits)."""
        self.what = what
        self.max_graphs = max_graphs
        self._max_cap = max_graphs * max(1, int(grow_cap))
        self._evicted: "OrderedDict[tuple, bool]" = OrderedDict()
        self.regrown = 0
        self.slots: "OrderedDict[tuple, object]" = OrderedDict()
        self.failed = False
        self.captures = 0
        self.replays = 0
        self.eager = 0
        self._misses = 0         # consecutive new keys
        self._paused_at = None   # eager count when the guard tripped

    def lookup(self, key, capture):
        entry = self.slots.get(key)
        if entry is not None:
            self._misses = 0
            self.slots.move_to_end(key)
            return entry
        if self.failed:
            return None
        if self._paused_at is not None:
            if self.eager - self._paused_at < RETRY_AFTER:
                self.eager += 1
                return None
            self._paused_at, self._misses = None, 0
        if key in self._evicted and self.max_graphs < self._max_cap:
            self.max_graphs = min(2 * self.max_graphs, self._max_cap)      # the working set did not fit: make room
            self.regrown += 1
        self._misses += 1
        if self._misses > MISS_LIMIT and not require_graphs():
            self._paused_at = self.eager
            self.eager += 1
            return None
        entry = capture()
        if entry is None:
            self.eager += 1
            return None
        self.slots[key] = entry
        self._evicted.pop(key, None)
        while len(self.slots) > self.max_graphs:
            gone, _ = self.slots.popitem(last=False)
            self._evicted[gone] = True
            while len(self._evicted) > 8 * self._max_cap:
                self._evicted.popitem(last=False)
        return entry

    def capture_failed(self, exc: Exception):
        """Shared failure policy of the capture sites: an error under MEMOTR_REQUIRE_GRAPHS=1, else a warning, and the
        cache is marked failed (eager from here on)."""
        if require_graphs():
            raise RuntimeError(f"{self.what} graph capture failed and MEMOTR_REQUIRE_GRAPHS=1: "
                               f"{type(exc).__name__}: {exc}") from exc
        import warnings
        warnings.warn(f"{self.what} graph capture failed ({type(exc).__name__}: {exc}); running eager")
        self.failed = True
        return None
