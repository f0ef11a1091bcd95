"""Loader for the `sdr-modem_amd/` package directory (hyphenated name) under the module name `sdr_modem_amd`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "sdr-modem_amd")


def load():
    if "sdr_modem_amd" in sys.modules:
        return sys.modules["sdr_modem_amd"]
    spec = importlib.util.spec_from_file_location("sdr_modem_amd", os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["sdr_modem_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
