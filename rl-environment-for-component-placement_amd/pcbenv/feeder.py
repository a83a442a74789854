"""InstanceFeeder: keeps the instance queue of a `BatchedPlacementEnv` fed with FRESH instances.

SURVEY.md §7 step 5: instances are action-independent, so the host generates them ahead of time.  A background
thread advances every environment's reference RNG stream with the native generator (`pcbenv_instgen_next_batch`
releases the GIL) and keeps up to `prefetch` packed batches ready; `refill()` -- called from the thread that steps
the environment, e.g. once per PPO iteration -- asks the device for the smallest queue cursor, and overwrites every
slot whose episode all environments have already consumed with the next batch.  The copy is enqueued on the
caller's stream, i.e. ordered after the kernels already launched and before the next ones: no kernel ever sees a
half-written slot.  Episode k of environment i is always instance k of its stream, as long as the environments
do not drift apart by `queue_depth` episodes or more between two `refill()` calls (then an old slot is replayed).
"""
from __future__ import annotations

import queue
import threading
from typing import Optional

import numpy as np


class InstanceFeeder:
    def __init__(self, env, prefetch: int = 4):
        if env._native is None:
            raise RuntimeError("call env.generate_instances(native=True) first (the feeder continues those streams)")
        self.env = env
        self.Q = env.queue_depth
        self.next_episode = self.Q          # episodes 0..Q-1 are in the queue already
        self._ready: "queue.Queue[np.ndarray]" = queue.Queue(maxsize=max(1, prefetch))
        self._stop = threading.Event()
        self._error: Optional[BaseException] = None
        self._thread = threading.Thread(target=self._produce, name="pcbenv-feeder", daemon=True)
        self._thread.start()

    def _produce(self):
        try:
            while not self._stop.is_set():
                batch = self.env._native.next_packed()   # C++ threads, GIL released
                while not self._stop.is_set():
                    try:
                        self._ready.put(batch, timeout=0.05)
                        break
                    except queue.Full:
                        continue
        except BaseException as exc:  # surfaced by refill()
            self._error = exc

    def refill(self, block: bool = False) -> int:
        """Refill every fully consumed slot for which a generated batch is ready.  Returns the number of slots
        written.  `block=True` waits for the generator instead of skipping."""
        if self._error is not None:
            raise RuntimeError("instance generator thread failed") from self._error
        lo, _hi = self.env.queue_cursors()
        n = 0
        # slot s = episode % Q may take episode `next_episode` once every env has consumed episode next_episode - Q
        while self.next_episode - self.Q < lo:
            try:
                batch = self._ready.get(block=block, timeout=5.0 if block else None)
            except queue.Empty:
                break
            self.env.load_packed(batch, slot=self.next_episode % self.Q)
            self.next_episode += 1
            n += 1
        return n

    def close(self):
        self._stop.set()
        self._thread.join(timeout=2.0)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
