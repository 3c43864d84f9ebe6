"""float64 twin of oracle.py: the SAME C source (athena_oracle.c) compiled with `float` -> `double` and the libm calls
widened (oracle/Makefile: liboracle64.so), behind the SAME python front-end with float32 -> float64.

TEST INFRASTRUCTURE ONLY.  It is the yardstick of the anchored tolerance in tests/helpers.py: where a chain of fp32
roundings puts the fp32 oracle itself more than 1e-5 from exact arithmetic, a device result passes when it is no further
from this float64 evaluation of the reference's formulas than the fp32 oracle is, plus 1e-5
(|gpu - f64| <= |oracle - f64| + 1e-5 * scale).  Inputs are the same fp32 values, widened."""
import os
import subprocess
import sys
import types

_HERE = os.path.dirname(os.path.abspath(__file__))


def _load():
    so = os.path.join(_HERE, "liboracle64.so")
    src = os.path.join(_HERE, "athena_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle64.so"], stdout=subprocess.DEVNULL)
    with open(os.path.join(_HERE, "oracle.py")) as fh:
        text = fh.read()
    text = (text.replace("np.float32", "np.float64").replace("C.c_float", "C.c_double")
                .replace('"liboracle.so"', '"liboracle64.so"'))
    mod = types.ModuleType("oracle.oracle64_impl")
    mod.__file__ = os.path.join(_HERE, "oracle.py")
    exec(compile(text, mod.__file__, "exec"), mod.__dict__)
    return mod


_impl = _load()
sys.modules[__name__].__dict__.update({k: v for k, v in _impl.__dict__.items() if not k.startswith("__")})
