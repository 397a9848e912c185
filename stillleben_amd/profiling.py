"""Nestable wall-clock timer with the interface of the reference's stillleben.profiling.Timer
(python/stillleben/profiling.py:9-52): a ContextDecorator gated by ``Timer.enabled`` that
prints a tree when the outermost timer exits."""
import time
from contextlib import ContextDecorator


class Timer(ContextDecorator):
    enabled = False
    _stack = []
    _records = []

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if Timer.enabled:
            Timer._stack.append((self.name, time.time(), len(Timer._records)))
            Timer._records.append(None)
        return self

    def __exit__(self, *exc):
        if not Timer.enabled or not Timer._stack:
            return False
        name, t0, slot = Timer._stack.pop()
        Timer._records[slot] = (len(Timer._stack), name, time.time() - t0)
        if not Timer._stack:
            for depth, n, dt in Timer._records:
                print("%s%s: %.3f ms" % ("  " * depth, n, dt * 1e3))
            Timer._records = []
        return False
