"""placeholder"""


class JobQueue:
    def __init__(self, *a, **k):
        raise NotImplementedError
