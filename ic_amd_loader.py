"""Imports the `image-compression_amd/` package (its directory name is not a valid Python identifier)
under the module name `image_compression_amd`."""
import importlib.util
import os
import sys

_NAME = "image_compression_amd"


def load_package():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    root = os.path.dirname(os.path.abspath(__file__))
    pkg_dir = os.path.join(root, "image-compression_amd")
    spec = importlib.util.spec_from_file_location(_NAME, os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
