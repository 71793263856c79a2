"""librosa.display stand-in (trainer/trainer.py:2)."""


def waveplot(*a, **k):
    raise RuntimeError("librosa is not installed in this image")


specshow = waveplot
