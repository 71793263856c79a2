"""json5 stand-in: the reference's shipped configs (config/train/train.json) are plain JSON."""
from json import dump, dumps, load, loads  # noqa: F401
