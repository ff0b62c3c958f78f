"""No-op stand-in for tensorboardX (absent in the build container); test tooling only."""


class SummaryWriter:
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        def noop(*a, **k):
            return None
        return noop
