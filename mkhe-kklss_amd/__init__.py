"""mkhe_kklss_amd -- MI355X-native multi-key RLWE key-switch engine (host-side mirror).

Product path: Python host code mirroring the reference's mkrlwe.KeySwitcher / mkckks.Evaluator
interface over the C ABI of `lib/libmkhe_hip.so` (include/mkhe.h).  There is NO CPU fallback:
importing `_abi` raises if the HIP library is missing.
"""
__version__ = "0.1.0"
