"""The oracle-side restatement of the three layers lives in oracle/layers.py (bench.py's parity block uses it too);
this name is what the tests have always imported."""
from oracle.layers import *  # noqa: F401,F403
from oracle.layers import double_precision, f64_lazy  # noqa: F401
