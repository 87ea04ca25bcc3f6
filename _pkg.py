"""Loads the product package (directory `vi-slam_amd`, not a valid Python identifier) as the
module `vi_slam_amd`.  Used by tests/, bench.py and __graft_entry__.py."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "vi-slam_amd")


def load_package():
    if "vi_slam_amd" in sys.modules:
        return sys.modules["vi_slam_amd"]
    spec = importlib.util.spec_from_file_location(
        "vi_slam_amd", os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["vi_slam_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def load_synth():
    load_package()
    import importlib
    return importlib.import_module("vi_slam_amd.synth")
