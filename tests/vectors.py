"""Loader for the golden fixtures in tests/golden (see tests/golden/make_fixtures.py)."""
import json
import lzma
import os

_HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_cache = {}


def _pool():
    if "pool" not in _cache:
        raw = lzma.decompress(open(os.path.join(_HERE, "pool.bin.xz"), "rb").read())
        idx = json.load(open(os.path.join(_HERE, "pool_index.json")))
        _cache["pool"] = [raw[o:o + n] for o, n in idx]
    return _cache["pool"]


def _resolve(v):
    if isinstance(v, dict):
        if "$b" in v and len(v) == 1:
            return _pool()[v["$b"]]
        return {k: _resolve(x) for k, x in v.items()}
    if isinstance(v, list):
        return [_resolve(x) for x in v]
    return v


def load(family):
    """Return {case_name: {"input": ..., "output": ...}} with byte strings materialised."""
    if "vectors" not in _cache:
        _cache["vectors"] = json.load(open(os.path.join(_HERE, "vectors.json")))
    return {k: _resolve(v) for k, v in _cache["vectors"][family].items()}


SRS_PATH = os.path.join(os.path.dirname(_HERE), "..", "rust-eth-kzg_amd", "data", "trusted_setup_4096.bin")
