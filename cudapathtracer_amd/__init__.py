"""MI355X-native unidirectional path tracer (drop-in for the reference's
``launch_unidirectional`` / ``launch_naive_unidirectional`` path, deviceCode.cuh:8-12).

The compute path is the HIP library ``cudapathtracer_amd/csrc/libptamd.so`` (C ABI declared in
``include/pt_api.h``); importing :mod:`cudapathtracer_amd.api` fails loudly when it is missing.
"""
__version__ = "0.1.0"
