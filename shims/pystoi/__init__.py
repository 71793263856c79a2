"""pystoi stand-in (util/utils.py:8 `from pystoi.stoi import stoi`): importable, not callable."""
