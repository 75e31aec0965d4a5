"""Run a script of this repository against an EXPERIMENT flavour of the library (profiles/tools/build_variant.sh):
    python3 profiles/tools/with_variant.py tuning bench.py --mode train ...      (PS_* knobs are only read by the -DPS_TUNING_ENV flavour)
The package's loader is pointed at point-unet_amd/csrc/build/variants/libps_<name>.so before the script starts; the default library
ignores every PS_* experiment variable (csrc/common.h, struct Tuning)."""
import os, runpy, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
name, script = sys.argv[1], sys.argv[2]
from point_unet_amd import _lib
_lib.LIB_PATH = os.path.join(root, "point-unet_amd", "csrc", "build", "variants", "libps_%s.so" % name)
assert os.path.exists(_lib.LIB_PATH), "build it first: sh profiles/tools/build_variant.sh %s \"-D...\" <file.hip ...>" % name
sys.argv = [script] + sys.argv[3:]
runpy.run_path(script, run_name="__main__")
