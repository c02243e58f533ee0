"""bcos_hip -- host-side plumbing for the MI355X-native B-cos kernels (libbcos_hip.so).

`lib`   ctypes binding of the C ABI (include/bcos_hip.h), build helper
`ops`   tensor-level wrappers (torch used for device memory / streams only)
"""
from .lib import BcosHipError, build, load  # noqa: F401
