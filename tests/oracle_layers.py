"""The oracle-side restatement of the three layers lives in oracle/layers.py (bench.py's parity block uses it too); this name is
what the tests have always imported -- it IS that module (same globals: double_precision() switches it for every caller)."""
import sys

from oracle import layers as _layers

sys.modules[__name__] = _layers
