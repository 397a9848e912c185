"""placeholder"""


class ManipulationSim:
    def __init__(self, *a, **k):
        raise NotImplementedError
