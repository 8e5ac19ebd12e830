"""HIP-graph capture of a framework forward, safe beside other threads that use the GPU.

The services capture their fallback forwards lazily, per (batch, width) bucket, on first use - and a /query request runs its
token classifier in a worker thread while the main thread embeds and searches (services/multi_diagnosis_service.py). In the
default (global) capture-error mode ANY thread's hipMalloc / synchronising copy while a capture is open invalidates that
capture. Two guards:
  * captures run in thread-local error mode: only the capturing thread's own calls can invalidate it;
  * ONE process-wide lock serialises the captures themselves (two open captures would share the allocator's private pool).
A failed capture costs the caller one eager forward and is retried at the bucket's next use; only a bucket that failed
MAX_FAILURES times stays eager - one unlucky capture never turns the graphs off for the life of the service.
"""
from __future__ import annotations

import threading

LOCK = threading.Lock()
MAX_FAILURES = 3


def capture(torch, build):
    """build() runs the forward on static inputs and returns its static outputs. -> (graph, outputs); raises what the capture raises."""
    with LOCK:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):   # warm-up outside the capture: lazy initialisations, the allocator's pools
                build()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"), torch.no_grad():
            out = build()
        return g, out


class Buckets:
    """the captured graphs of one service by bucket key; a key maps to an entry, or to False once it has failed MAX_FAILURES times"""

    def __init__(self):
        self.entries = {}
        self.failures = {}

    def get(self, key):
        return self.entries.get(key)

    def put(self, key, entry):
        self.entries[key] = entry

    def failed(self, key) -> bool:
        """count a failed capture of `key`; True when the bucket is given up (stays eager from now on)"""
        n = self.failures[key] = self.failures.get(key, 0) + 1
        if n >= MAX_FAILURES:
            self.entries[key] = False
            return True
        return False

    def __bool__(self):
        return any(v is not False for v in self.entries.values())

    def __len__(self):
        return sum(1 for v in self.entries.values() if v is not False)
