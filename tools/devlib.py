"""A/B tools load libmingnative_dev.so (the product code + the hooks of include/mingnative_dev.h; `make -C ming_univision_amd/csrc dev`)
instead of the shipped library.  Import this module BEFORE anything calls ming_univision_amd._lib.lib()."""
import os
import subprocess

from ming_univision_amd import _lib

_DEV = os.environ.get("MINGNATIVE_DEV_LIB") or os.path.join(os.path.dirname(_lib.LIB_PATH), "libmingnative_dev.so")     # (A/B builds: another dev library)
if not os.path.exists(_DEV):
    subprocess.run(["make", "-C", os.path.join(os.path.dirname(_lib.LIB_PATH), "csrc"), "-j8", "dev"], check=True)
assert _lib._lib is None, "tools.devlib must be imported before the library is loaded"
_lib.LIB_PATH = _DEV
