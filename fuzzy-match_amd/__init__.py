"""fuzzy-match_amd — MI355X-native fuzzy-match lookup engine.

The product is `libfmx.so` (hand-written HIP kernels behind the C ABI of include/fmx.h) and the C++
facade in `fuzzy-match_amd/include/fuzzy/`.  `fmx` is the ctypes binding used by tests and bench.py.
The directory name is not a Python identifier; import it through `fmx_import.load()` at the repo root
(registers it as `fuzzy_match_amd`).
"""
from . import fmx  # noqa: F401
from . import shard  # noqa: F401
