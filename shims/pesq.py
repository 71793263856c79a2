"""pesq stand-in (metrics are out of scope): importable, not callable."""


def _unavailable(*a, **k):
    raise RuntimeError("pesq is not installed in this image; validation metrics are outside the hot path")


pesq = stoi = _unavailable
