"""vslam_amd — MI355X-native front-end (extract / match / RANSAC / k-d tree) behind a C ABI.

The product is vslam_amd/libvslam_amd.so (HIP, gfx950) + include/vslam_amd.h + the C++ adapters in
include/vslam/.  This Python package is harness plumbing: a ctypes binding and synthetic inputs.
"""
from .capi import Context, VslamError, load_library, LIB_PATH, SYMBOLS  # noqa: F401
