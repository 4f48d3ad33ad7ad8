"""MI355X-native Huffman encode/decode engine behind the aws-c-compression C ABI.

The product is the shared library next to this file (C99 host layer + HIP kernels,
built by the Makefile here).  This Python module only locates and loads it; the
ctypes bindings the tests and the benchmark use live in tests/harness.py.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBRARY_PATH = os.path.join(_HERE, "libaws-c-compression-amd.so")

_REQUIRED = ("aws_huffman_encode", "aws_huffman_decode", "aws_huffman_get_encoded_length",
             "aws_huffman_amd_engine_new", "aws_huffman_amd_encode_plan_launch", "aws_huffman_amd_decode_plan_launch")


def library():
    """dlopen the HIP library; raises when it has not been built (there is no fallback)."""
    if not os.path.exists(LIBRARY_PATH):
        raise RuntimeError("%s is missing: run `make -C %s`" % (LIBRARY_PATH, _HERE))
    lib = ctypes.CDLL(LIBRARY_PATH)
    for name in _REQUIRED:
        getattr(lib, name)
    return lib
