"""sl.JobQueue (reference src/job_queue.cpp:30-176, python/src/py_job_queue.cpp:18-48).

The reference runs `hardware_concurrency()/2` CPU worker threads, each settling one queued scene
with PhysX.  Here the queue is the batch dimension of the settle kernel: scenes added since the
last retrieval are settled TOGETHER in one slhip_settle launch when the first of them is
retrieved; `retrieve_scene()` returns scenes in submission order like the reference
(job_queue.cpp:75-82)."""
import collections

from ._context import require_context


class JobQueue:
    def __init__(self, num_threads=-1):
        require_context()
        import os

        # kept for API compatibility: the degree of parallelism is the GPU batch, not threads
        self._num_threads = int(num_threads) if num_threads and num_threads > 0 else max(1, (os.cpu_count() or 2) // 2)
        self._pending = collections.deque()
        self._done = collections.deque()

    @property
    def num_threads(self):
        return self._num_threads

    def add_scene(self, scene):
        scene.load_physics()              # job_queue.cpp:58 (hull decomposition happens on the caller's thread)
        self._pending.append(scene)

    def stop(self):
        """job_queue.cpp:84-93: joins the workers; nothing to join here."""

    def retrieve_scene(self):
        if not self._done:
            if not self._pending:
                # std::logic_error -> RuntimeError through pybind11 (job_queue.cpp:70-71)
                raise RuntimeError("PhysicsSim::retrieveScene(): No scenes in work queue. You need to add scenes first!")
            from . import physics

            batch = list(self._pending)
            self._pending.clear()
            physics.settle_batch(batch)
            self._done.extend(batch)
        return self._done.popleft()
