"""Scratch-side helper: apply `DFW_CFG="field=value,field=value"` (fields of dfw_config, include/diffews_hip.h) through
diffews_amd._lib.configure().  The library itself reads nothing from the environment; sweep scripts that run one
configuration per child process pass it this way."""
import os


def apply_env_config():
    spec = os.environ.get("DFW_CFG", "")
    if not spec:
        return {}
    from diffews_amd import _lib
    kw = {}
    for item in spec.split(","):
        k, v = item.split("=")
        kw[k.strip()] = int(v)
    return _lib.configure(**kw)
