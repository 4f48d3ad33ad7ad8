"""ctypes plumbing shared by the tests, bench.py and __graft_entry__.smoke().

Two libraries speak the same C ABI (include/aws/compression/huffman.h):

  * the ORACLE  (oracle/libhuffman_oracle.so, entry points oracle_huffman_*): the CPU
    restatement of the reference, test infrastructure only;
  * the PRODUCT (aws-c-compression_amd/libaws-c-compression-amd.so, entry points
    aws_huffman_*): the HIP implementation.

`Codec` wraps either one behind the same Python calls so that a parity test is
"run the same scenario on both and compare everything observable".
"""
import ctypes as C
import time
import json
import os

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
ORACLE_SO = os.path.join(REPO, "oracle", "libhuffman_oracle.so")
PRODUCT_SO = os.path.join(REPO, "aws-c-compression_amd", "libaws-c-compression-amd.so")

AWS_OP_SUCCESS = 0
AWS_OP_ERR = -1
AWS_ERROR_SHORT_BUFFER = 4
AWS_ERROR_UNSUPPORTED_OPERATION = 6
AWS_ERROR_INVALID_ARGUMENT = 34
AWS_ERROR_COMPRESSION_UNKNOWN_SYMBOL = 0x0C00


# ----------------------------------------------------------------------------- ABI structs
class HuffmanCode(C.Structure):  # 8 bytes
    _fields_ = [("pattern", C.c_uint32), ("num_bits", C.c_uint8)]


ENCODE_FN = C.CFUNCTYPE(HuffmanCode, C.c_uint8, C.c_void_p)
DECODE_FN = C.CFUNCTYPE(C.c_uint8, C.c_uint32, C.POINTER(C.c_uint8), C.c_void_p)


class SymbolCoder(C.Structure):  # 24 bytes
    _fields_ = [("encode", C.c_void_p), ("decode", C.c_void_p), ("userdata", C.c_void_p)]


class Encoder(C.Structure):  # 24 bytes
    _fields_ = [("coder", C.POINTER(SymbolCoder)), ("eos_padding", C.c_uint8), ("overflow_bits", HuffmanCode)]


class Decoder(C.Structure):  # 32 bytes
    _fields_ = [
        ("coder", C.POINTER(SymbolCoder)),
        ("allow_growth", C.c_bool),
        ("working_bits", C.c_uint64),
        ("num_bits", C.c_uint8),
    ]


class ByteCursor(C.Structure):  # 16 bytes
    _fields_ = [("len", C.c_size_t), ("ptr", C.c_void_p)]


class ByteBuf(C.Structure):  # 32 bytes
    _fields_ = [("len", C.c_size_t), ("buffer", C.c_void_p), ("capacity", C.c_size_t), ("allocator", C.c_void_p)]


def check_abi_layout():
    """Sizes/offsets of SURVEY.md section 8b."""
    assert C.sizeof(HuffmanCode) == 8 and HuffmanCode.num_bits.offset == 4
    assert C.sizeof(SymbolCoder) == 24
    assert C.sizeof(Encoder) == 24 and Encoder.eos_padding.offset == 8 and Encoder.overflow_bits.offset == 12
    assert C.sizeof(Decoder) == 32 and Decoder.allow_growth.offset == 8
    assert Decoder.working_bits.offset == 16 and Decoder.num_bits.offset == 24
    assert C.sizeof(ByteCursor) == 16 and C.sizeof(ByteBuf) == 32


# ----------------------------------------------------------------------------- fixtures
def load_table():
    rows = json.load(open(os.path.join(GOLDEN, "test_coder_table.json")))["rows"]
    patterns = (C.c_uint32 * 256)(*[r["pattern"] for r in rows])
    lens = (C.c_uint8 * 256)(*[r["num_bits"] for r in rows])
    return patterns, lens


def load_json(name):
    return json.load(open(os.path.join(GOLDEN, name)))


def splitmix64_bytes(seed, n):
    """numpy twin of oracle_splitmix64_fill (SURVEY.md section 8c generator)."""
    draws = (n + 7) // 8
    with np.errstate(over="ignore"):
        idx = np.arange(1, draws + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z.astype("<u8").view(np.uint8)[:n].copy()


def printable_map(raw):
    """G16KP mapping of SURVEY.md section 8c: 32 + (b % 95)."""
    return (32 + (raw.astype(np.uint16) % 95)).astype(np.uint8)


# ----------------------------------------------------------------------------- libraries
_PROTOS = {
    "huffman_encoder_init": (None, [C.POINTER(Encoder), C.POINTER(SymbolCoder)]),
    "huffman_encoder_reset": (None, [C.POINTER(Encoder)]),
    "huffman_decoder_init": (None, [C.POINTER(Decoder), C.POINTER(SymbolCoder)]),
    "huffman_decoder_reset": (None, [C.POINTER(Decoder)]),
    "huffman_decoder_allow_growth": (None, [C.POINTER(Decoder), C.c_bool]),
    "huffman_get_encoded_length": (C.c_size_t, [C.POINTER(Encoder), ByteCursor]),
    "huffman_encode": (C.c_int, [C.POINTER(Encoder), C.POINTER(ByteCursor), C.POINTER(ByteBuf)]),
    "huffman_decode": (C.c_int, [C.POINTER(Decoder), C.POINTER(ByteCursor), C.POINTER(ByteBuf)]),
}


def _bind(lib, name, restype, argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = argtypes
    return fn


def load_oracle():
    if not os.path.exists(ORACLE_SO):
        raise RuntimeError("oracle not built: run `make -C oracle` (or __graft_entry__.build())")
    lib = C.CDLL(ORACLE_SO)
    for short, (res, args) in _PROTOS.items():
        _bind(lib, "oracle_" + short, res, args)
    _bind(lib, "oracle_last_error", C.c_int, [])
    _bind(lib, "oracle_reset_error", None, [])
    _bind(lib, "oracle_default_allocator", C.c_void_p, [])
    _bind(lib, "oracle_table_coder_new", C.POINTER(SymbolCoder), [C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)])
    _bind(lib, "oracle_table_coder_destroy", None, [C.POINTER(SymbolCoder)])
    _bind(lib, "oracle_huffman_test_transitive", C.c_int,
          [C.POINTER(SymbolCoder), C.c_char_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_char_p)])
    _bind(lib, "oracle_huffman_test_transitive_chunked", C.c_int,
          [C.POINTER(SymbolCoder), C.c_char_p, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_char_p)])
    _bind(lib, "oracle_splitmix64_fill", None, [C.c_void_p, C.c_size_t, C.c_uint64])
    return lib


# ---- extension API of include/aws/compression/huffman_amd.h
class AmdEncodeItem(C.Structure):
    _fields_ = [("in_offset", C.c_uint64), ("in_len", C.c_uint64), ("out_offset", C.c_uint64),
                ("out_capacity", C.c_uint64), ("overflow_in", HuffmanCode), ("eos_padding", C.c_uint8)]


class AmdEncodeResult(C.Structure):
    _fields_ = [("rc", C.c_int32), ("error", C.c_int32), ("consumed", C.c_uint64), ("produced", C.c_uint64),
                ("overflow_out", HuffmanCode)]


class AmdDecodeItem(C.Structure):
    _fields_ = [("in_offset", C.c_uint64), ("in_len", C.c_uint64), ("first_bit", C.c_uint32),
                ("out_offset", C.c_uint64), ("out_capacity", C.c_uint64)]


class AmdDecodeResult(C.Structure):
    _fields_ = [("rc", C.c_int32), ("error", C.c_int32), ("produced", C.c_uint64), ("bits_consumed", C.c_uint64)]


class StridedItems(C.Structure):
    """struct aws_huffman_amd_strided_items: a batch of equal items a stride apart."""
    _fields_ = [("count", C.c_uint64), ("in_offset", C.c_uint64), ("in_stride", C.c_uint64), ("in_len", C.c_uint64),
                ("out_offset", C.c_uint64), ("out_stride", C.c_uint64), ("out_capacity", C.c_uint64),
                ("first_bit", C.c_uint8), ("eos_padding", C.c_uint8)]


class PlanStats(C.Structure):
    """struct aws_huffman_amd_plan_stats: how a plan's items are taken (which kernels an item goes through)."""
    _fields_ = [(name, C.c_uint64) for name in (
        "items", "thread_limit", "by_thread", "by_wave", "by_workgroup", "by_blocks", "by_pieces", "pieces",
        "end_pieces_packed", "end_pieces_single", "empty", "end_pieces_folded")]

    def as_dict(self):
        return {name: int(getattr(self, name)) for name, _ in self._fields_}


# the library's testing hooks (huffman_amd.h): the ways BACK a launch carries, by the names the scenarios use
ENCODE_ROADS = {None: 0, "three-kernel": 1, "one-pass-fails": 2}
DECODE_ROADS = {None: 0, "long-way": 1, "wide-fails": 2, "wide-fn-fails": 4, "lean-sync": 8, "all-kernels": 16, "tails-apart": 32}


class encode_road:
    """with harness.encode_road(lib, "three-kernel"): engines MADE inside keep to count / scan / pack."""

    def __init__(self, lib, name):
        self.lib, self.flags = lib, ENCODE_ROADS[name]

    def __enter__(self):
        self.lib.aws_huffman_amd_testing_set_encode_road(self.flags)

    def __exit__(self, *exc):
        self.lib.aws_huffman_amd_testing_set_encode_road(0)


class decode_road:
    """with harness.decode_road(lib, "long-way"): decode LAUNCHES inside take that way back ("lean-sync": short
    end-of-stream chunks a workgroup each)."""

    def __init__(self, lib, name):
        self.lib, self.flags = lib, DECODE_ROADS[name]

    def __enter__(self):
        self.lib.aws_huffman_amd_testing_set_decode_road(self.flags)

    def __exit__(self, *exc):
        self.lib.aws_huffman_amd_testing_set_decode_road(0)


class items_per_byte:
    """with harness.items_per_byte(lib, encode=1): plans MADE inside give a thread to every item of a short-item class that
    holds at least that many items per byte of its longest item (0: the built-in rule)."""

    def __init__(self, lib, encode=0, decode=0):
        self.lib, self.encode, self.decode = lib, encode, decode

    def __enter__(self):
        self.lib.aws_huffman_amd_testing_set_items_per_byte(self.encode, self.decode)

    def __exit__(self, *exc):
        self.lib.aws_huffman_amd_testing_set_items_per_byte(0, 0)


class ShardIo(C.Structure):
    _fields_ = [("device_input", C.c_void_p), ("device_output", C.c_void_p)]


EXPORTED_SYMBOLS = [
    # include/aws/compression/huffman.h
    "aws_huffman_encoder_init", "aws_huffman_encoder_reset", "aws_huffman_decoder_init", "aws_huffman_decoder_reset",
    "aws_huffman_decoder_allow_growth", "aws_huffman_get_encoded_length", "aws_huffman_encode", "aws_huffman_decode",
    # include/aws/compression/compression.h
    "aws_compression_library_init", "aws_compression_library_clean_up",
    # include/aws/compression/huffman_amd.h
    "aws_huffman_amd_engine_new", "aws_huffman_amd_engine_destroy", "aws_huffman_amd_engine_max_code_bits",
    "aws_huffman_amd_engine_can_decode", "aws_huffman_amd_encode_plan_new", "aws_huffman_amd_encode_plan_destroy",
    "aws_huffman_amd_encode_plan_launch", "aws_huffman_amd_encode_plan_launch_staged",
    "aws_huffman_amd_encode_plan_results", "aws_huffman_amd_decode_plan_new", "aws_huffman_amd_decode_plan_destroy",
    "aws_huffman_amd_decode_plan_launch", "aws_huffman_amd_decode_plan_launch_staged",
    "aws_huffman_amd_decode_plan_results", "aws_huffman_amd_decode_plan_road", "aws_huffman_amd_encode_plan_road",
    "aws_huffman_amd_decode_plan_is_quiet",
    "aws_huffman_amd_testing_set_decode_piece_bytes",
    "aws_huffman_amd_testing_set_wide_min_bytes",
    "aws_huffman_amd_testing_set_encode_road", "aws_huffman_amd_testing_set_decode_road",
    "aws_huffman_amd_testing_set_items_per_byte", "aws_huffman_amd_encode_plan_stats", "aws_huffman_amd_decode_plan_stats",
    "aws_huffman_amd_encode_plan_reset_strided", "aws_huffman_amd_decode_plan_reset_strided",
    "aws_huffman_amd_encode_plan_reset_device_items", "aws_huffman_amd_decode_plan_reset_device_items",
    "aws_huffman_amd_engine_device", "aws_huffman_amd_current_device", "aws_huffman_amd_encode_plan_reset",
    "aws_huffman_amd_decode_plan_reset", "aws_huffman_amd_decode_plan_from_encode",
    "aws_huffman_amd_device_count", "aws_huffman_amd_device_alloc", "aws_huffman_amd_device_free",
    "aws_huffman_amd_copy_to_device", "aws_huffman_amd_copy_to_host", "aws_huffman_amd_device_fill",
    "aws_huffman_amd_device_fill_splitmix64", "aws_huffman_amd_engine_stream", "aws_huffman_amd_stream_synchronize",
    "aws_huffman_amd_event_new", "aws_huffman_amd_event_destroy", "aws_huffman_amd_event_record",
    "aws_huffman_amd_event_elapsed_ms", "aws_huffman_amd_table_coder_new", "aws_huffman_amd_table_coder_destroy",
    "aws_huffman_amd_table_coder_from_def", "aws_huffman_amd_encode_plan_encoded_lengths",
    "aws_huffman_amd_engine_encodes_in_one_pass",
    "aws_huffman_amd_shards_new", "aws_huffman_amd_shards_destroy", "aws_huffman_amd_shards_count",
    "aws_huffman_amd_shards_engine", "aws_huffman_amd_shards_encode", "aws_huffman_amd_shards_decode",
]
# include/aws/compression/private/huffman_testing.h (the reference's names: no aws_ prefix)
TESTING_SYMBOLS = ["huffman_test_transitive", "huffman_test_transitive_chunked"]


def load_product(path=None):
    """The HIP library.  Fails loudly when it is missing: there is no CPU fallback."""
    path = path or PRODUCT_SO
    if not os.path.exists(path):
        raise RuntimeError("HIP library not built: %s (run __graft_entry__.build())" % path)
    lib = C.CDLL(path)
    for short, (res, args) in _PROTOS.items():
        _bind(lib, "aws_" + short, res, args)
    _bind(lib, "aws_last_error", C.c_int, [])
    _bind(lib, "aws_reset_error", None, [])
    _bind(lib, "aws_default_allocator", C.c_void_p, [])
    V, P = C.c_void_p, C.POINTER
    _bind(lib, "aws_huffman_amd_table_coder_new", P(SymbolCoder), [P(C.c_uint32), P(C.c_uint8)])
    _bind(lib, "aws_huffman_amd_table_coder_destroy", None, [P(SymbolCoder)])
    _bind(lib, "aws_huffman_amd_table_coder_from_def", P(SymbolCoder), [C.c_char_p, C.c_size_t])
    _bind(lib, "aws_huffman_amd_encode_plan_encoded_lengths", C.c_int, [V, P(C.c_uint64), V])
    _bind(lib, "huffman_test_transitive", C.c_int, [P(SymbolCoder), C.c_char_p, C.c_size_t, C.c_size_t, P(C.c_char_p)])
    _bind(lib, "huffman_test_transitive_chunked", C.c_int,
          [P(SymbolCoder), C.c_char_p, C.c_size_t, C.c_size_t, C.c_size_t, P(C.c_char_p)])
    _bind(lib, "aws_huffman_amd_engine_new", C.c_int, [P(V), P(SymbolCoder), C.c_int])
    _bind(lib, "aws_huffman_amd_engine_destroy", None, [V])
    _bind(lib, "aws_huffman_amd_engine_max_code_bits", C.c_uint32, [V])
    _bind(lib, "aws_huffman_amd_engine_can_decode", C.c_bool, [V])
    _bind(lib, "aws_huffman_amd_engine_stream", V, [V])
    _bind(lib, "aws_huffman_amd_engine_encodes_in_one_pass", C.c_bool, [V])
    _bind(lib, "aws_huffman_amd_shards_new", C.c_int, [P(V), P(SymbolCoder), P(C.c_int), C.c_size_t])
    _bind(lib, "aws_huffman_amd_shards_destroy", None, [V])
    _bind(lib, "aws_huffman_amd_shards_count", C.c_size_t, [V])
    _bind(lib, "aws_huffman_amd_shards_engine", V, [V, C.c_size_t])
    _bind(lib, "aws_huffman_amd_shards_encode", C.c_int, [V, P(AmdEncodeItem), C.c_size_t, P(ShardIo), P(AmdEncodeResult)])
    _bind(lib, "aws_huffman_amd_shards_decode", C.c_int, [V, P(AmdDecodeItem), C.c_size_t, P(ShardIo), P(AmdDecodeResult)])
    _bind(lib, "aws_huffman_amd_encode_plan_new", C.c_int, [P(V), V, P(AmdEncodeItem), C.c_size_t])
    _bind(lib, "aws_huffman_amd_encode_plan_destroy", None, [V])
    _bind(lib, "aws_huffman_amd_encode_plan_launch", C.c_int, [V, V, V, C.c_bool, V])
    _bind(lib, "aws_huffman_amd_encode_plan_launch_staged", C.c_int, [V, V, V, C.c_bool, V, P(V)])
    _bind(lib, "aws_huffman_amd_decode_plan_launch_staged", C.c_int, [V, V, V, V, P(V)])
    _bind(lib, "aws_huffman_amd_encode_plan_results", C.c_int, [V, P(AmdEncodeResult), V])
    _bind(lib, "aws_huffman_amd_decode_plan_new", C.c_int, [P(V), V, P(AmdDecodeItem), C.c_size_t])
    _bind(lib, "aws_huffman_amd_decode_plan_destroy", None, [V])
    _bind(lib, "aws_huffman_amd_decode_plan_launch", C.c_int, [V, V, V, V])
    _bind(lib, "aws_huffman_amd_decode_plan_results", C.c_int, [V, P(AmdDecodeResult), V])
    _bind(lib, "aws_huffman_amd_decode_plan_road", C.c_int, [V, V, P(C.c_uint32), P(C.c_uint32)])
    _bind(lib, "aws_huffman_amd_encode_plan_road", C.c_int, [V, P(C.c_uint32)])
    if hasattr(lib, "aws_huffman_amd_decode_plan_is_quiet"):  # (profiles/tools/ab.sh loads older builds beside this one)
        _bind(lib, "aws_huffman_amd_decode_plan_is_quiet", C.c_bool, [V])
    _bind(lib, "aws_huffman_amd_testing_set_decode_piece_bytes", None, [C.c_size_t])
    _bind(lib, "aws_huffman_amd_testing_set_wide_min_bytes", None, [C.c_uint64])
    _bind(lib, "aws_huffman_amd_testing_set_encode_road", None, [C.c_uint32])
    _bind(lib, "aws_huffman_amd_testing_set_decode_road", None, [C.c_uint32])
    _bind(lib, "aws_huffman_amd_testing_set_items_per_byte", None, [C.c_uint64, C.c_uint64])
    _bind(lib, "aws_huffman_amd_encode_plan_reset_strided", C.c_int, [V, P(StridedItems), V])
    _bind(lib, "aws_huffman_amd_decode_plan_reset_strided", C.c_int, [V, P(StridedItems), V])
    _bind(lib, "aws_huffman_amd_encode_plan_reset_device_items", C.c_int, [V, V, C.c_size_t, V])
    _bind(lib, "aws_huffman_amd_decode_plan_reset_device_items", C.c_int, [V, V, C.c_size_t, V])
    _bind(lib, "aws_huffman_amd_encode_plan_stats", C.c_int, [V, P(PlanStats)])
    _bind(lib, "aws_huffman_amd_decode_plan_stats", C.c_int, [V, P(PlanStats)])
    _bind(lib, "aws_huffman_amd_device_count", C.c_int, [])
    _bind(lib, "aws_huffman_amd_engine_device", C.c_int, [V])
    _bind(lib, "aws_huffman_amd_current_device", C.c_int, [])
    _bind(lib, "aws_huffman_amd_device_alloc", V, [V, C.c_size_t])
    _bind(lib, "aws_huffman_amd_device_free", None, [V, V])
    _bind(lib, "aws_huffman_amd_copy_to_device", C.c_int, [V, V, V, C.c_size_t])
    _bind(lib, "aws_huffman_amd_copy_to_host", C.c_int, [V, V, V, C.c_size_t])
    _bind(lib, "aws_huffman_amd_device_fill", C.c_int, [V, V, C.c_int, C.c_size_t])
    _bind(lib, "aws_huffman_amd_device_fill_splitmix64", C.c_int, [V, V, C.c_size_t, C.c_uint64])
    _bind(lib, "aws_huffman_amd_stream_synchronize", C.c_int, [V, V])
    _bind(lib, "aws_huffman_amd_event_new", V, [V])
    _bind(lib, "aws_huffman_amd_event_destroy", None, [V, V])
    _bind(lib, "aws_huffman_amd_event_record", C.c_int, [V, V, V])
    _bind(lib, "aws_huffman_amd_event_elapsed_ms", C.c_int, [V, V, V, P(C.c_float)])
    return lib


class Engine:
    """Device-pointer / batched face of the product library (huffman_amd.h)."""

    def __init__(self, lib, coder, device=-1):
        self.lib = lib
        h = C.c_void_p()
        if lib.aws_huffman_amd_engine_new(C.byref(h), coder, device) != 0:
            raise RuntimeError("aws_huffman_amd_engine_new failed, error %d" % lib.aws_last_error())
        self.h = h
        self.stream = lib.aws_huffman_amd_engine_stream(h)

    def close(self):
        if self.h:
            self.lib.aws_huffman_amd_engine_destroy(self.h)
            self.h = None

    def alloc(self, n):
        p = self.lib.aws_huffman_amd_device_alloc(self.h, max(int(n), 1))
        if not p:
            raise MemoryError("device alloc of %d bytes failed" % n)
        return p

    def free(self, p):
        self.lib.aws_huffman_amd_device_free(self.h, p)

    def upload(self, dptr, arr, offset=0):
        arr = np.ascontiguousarray(arr, dtype=np.uint8)
        assert self.lib.aws_huffman_amd_copy_to_device(self.h, dptr + offset, arr.ctypes.data, arr.size) == 0

    def download(self, dptr, n, offset=0):
        out = np.empty(int(n), dtype=np.uint8)
        assert self.lib.aws_huffman_amd_copy_to_host(self.h, out.ctypes.data, dptr + offset, out.size) == 0
        return out

    def fill(self, dptr, byte, n):
        assert self.lib.aws_huffman_amd_device_fill(self.h, dptr, byte, int(n)) == 0

    def fill_splitmix64(self, dptr, n, seed):
        assert self.lib.aws_huffman_amd_device_fill_splitmix64(self.h, dptr, int(n), seed) == 0

    def sync(self):
        assert self.lib.aws_huffman_amd_stream_synchronize(self.h, None) == 0

    # items: list of dicts / tuples
    def encode_plan(self, items):
        arr = (AmdEncodeItem * max(len(items), 1))()
        for i, it in enumerate(items):
            arr[i].in_offset, arr[i].in_len = it["in_offset"], it["in_len"]
            arr[i].out_offset, arr[i].out_capacity = it["out_offset"], it["out_capacity"]
            ov = it.get("overflow_in", (0, 0))
            arr[i].overflow_in.pattern, arr[i].overflow_in.num_bits = ov
            arr[i].eos_padding = it.get("eos_padding", 0xFF)
        plan = C.c_void_p()
        t0 = time.perf_counter()
        rc = self.lib.aws_huffman_amd_encode_plan_new(C.byref(plan), self.h, arr, len(items))
        self.last_plan_ms = (time.perf_counter() - t0) * 1e3  # what making the plan cost (the C call alone)
        if rc != 0:
            raise RuntimeError("encode_plan_new failed, error %d" % self.lib.aws_last_error())
        return plan

    def _encode_item_array(self, items):
        arr = (AmdEncodeItem * max(len(items), 1))()
        for i, it in enumerate(items):
            arr[i].in_offset, arr[i].in_len = it["in_offset"], it["in_len"]
            arr[i].out_offset, arr[i].out_capacity = it["out_offset"], it["out_capacity"]
            ov = it.get("overflow_in", (0, 0))
            arr[i].overflow_in.pattern, arr[i].overflow_in.num_bits = ov
            arr[i].eos_padding = it.get("eos_padding", 0xFF)
        return arr

    def empty_encode_plan(self):
        plan = C.c_void_p()
        assert self.lib.aws_huffman_amd_encode_plan_new(C.byref(plan), self.h, None, 0) == 0
        return plan

    def empty_decode_plan(self):
        plan = C.c_void_p()
        assert self.lib.aws_huffman_amd_decode_plan_new(C.byref(plan), self.h, None, 0) == 0
        return plan

    def encode_plan_from_device_items(self, items, plan=None):
        """The items' records uploaded, the plan made from them ON THE DEVICE (no host loop over the items).  Returns
        (plan, device array): the array is the caller's to free."""
        arr = self._encode_item_array(items)
        d_items = self.alloc(C.sizeof(arr))
        self.upload(d_items, np.frombuffer(arr, dtype=np.uint8))
        plan = plan or self.empty_encode_plan()
        t0 = time.perf_counter()
        rc = self.lib.aws_huffman_amd_encode_plan_reset_device_items(plan, d_items, len(items), None)
        self.last_plan_ms = (time.perf_counter() - t0) * 1e3
        if rc != 0:
            raise RuntimeError("encode_plan_reset_device_items failed, error %d" % self.lib.aws_last_error())
        return plan, d_items

    def decode_plan_from_device_items(self, items, plan=None):
        arr = self._decode_item_array(items)
        d_items = self.alloc(C.sizeof(arr))
        self.upload(d_items, np.frombuffer(arr, dtype=np.uint8))
        plan = plan or self.empty_decode_plan()
        t0 = time.perf_counter()
        rc = self.lib.aws_huffman_amd_decode_plan_reset_device_items(plan, d_items, len(items), None)
        self.last_plan_ms = (time.perf_counter() - t0) * 1e3
        if rc != 0:
            raise RuntimeError("decode_plan_reset_device_items failed, error %d" % self.lib.aws_last_error())
        return plan, d_items

    def plan_strided(self, encode, plan=None, **fields):
        """A plan of `count` equal items a stride apart, made on the device from the description alone."""
        desc = StridedItems(**fields)
        plan = plan or (self.empty_encode_plan() if encode else self.empty_decode_plan())
        fn = self.lib.aws_huffman_amd_encode_plan_reset_strided if encode else self.lib.aws_huffman_amd_decode_plan_reset_strided
        t0 = time.perf_counter()
        rc = fn(plan, C.byref(desc), None)
        self.last_plan_ms = (time.perf_counter() - t0) * 1e3
        if rc != 0:
            raise RuntimeError("plan_reset_strided failed, error %d" % self.lib.aws_last_error())
        return plan

    def encode_launch(self, plan, d_in, d_out, length_only=False, events=None):
        assert self.lib.aws_huffman_amd_encode_plan_launch_staged(plan, d_in, d_out, length_only, None, events) == 0

    def new_events(self, n):
        arr = (C.c_void_p * n)()
        for i in range(n):
            arr[i] = self.lib.aws_huffman_amd_event_new(self.h)
            assert arr[i]
        return arr

    def record(self, event):
        assert self.lib.aws_huffman_amd_event_record(self.h, event, None) == 0

    def elapsed_ms(self, start, stop):
        ms = C.c_float()
        assert self.lib.aws_huffman_amd_event_elapsed_ms(self.h, start, stop, C.byref(ms)) == 0
        return ms.value

    def encode_results(self, plan, n):
        res = (AmdEncodeResult * max(n, 1))()
        assert self.lib.aws_huffman_amd_encode_plan_results(plan, res, None) == 0
        return [(r.rc, r.error, r.consumed, r.produced, r.overflow_out.num_bits,
                 r.overflow_out.pattern if r.overflow_out.num_bits else 0) for r in res[:n]]

    def encode_road(self, plan):
        """Of the last launch whose results were fetched.  0: count / scan / pack, 1: one pass, 2: one pass gave up, done over."""
        road = C.c_uint32(99)
        assert self.lib.aws_huffman_amd_encode_plan_road(plan, C.byref(road)) == 0
        return road.value

    def encoded_lengths(self, plan, n):
        out = (C.c_uint64 * max(n, 1))()
        assert self.lib.aws_huffman_amd_encode_plan_encoded_lengths(plan, out, None) == 0
        return list(out[:n])

    def _decode_item_array(self, items):
        arr = (AmdDecodeItem * max(len(items), 1))()
        for i, it in enumerate(items):
            arr[i].in_offset, arr[i].in_len = it["in_offset"], it["in_len"]
            arr[i].first_bit = it.get("first_bit", 0)
            arr[i].out_offset, arr[i].out_capacity = it["out_offset"], it["out_capacity"]
        return arr

    def decode_plan(self, items):
        arr = self._decode_item_array(items)
        plan = C.c_void_p()
        t0 = time.perf_counter()
        rc = self.lib.aws_huffman_amd_decode_plan_new(C.byref(plan), self.h, arr, len(items))
        self.last_plan_ms = (time.perf_counter() - t0) * 1e3
        if rc != 0:
            raise RuntimeError("decode_plan_new failed, error %d" % self.lib.aws_last_error())
        return plan

    def decode_plan_from_encode(self, plan, encode_plan):
        """plan reset to decode what encode_plan's last launch produced; the lengths stay on the device.  False: the batch is
        not one of short items (AWS_ERROR_UNSUPPORTED_OPERATION, the plan is as it was)."""
        self.lib.aws_huffman_amd_decode_plan_from_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        rc = self.lib.aws_huffman_amd_decode_plan_from_encode(plan, encode_plan, None)
        if rc != 0:
            assert self.lib.aws_last_error() == AWS_ERROR_UNSUPPORTED_OPERATION, self.lib.aws_last_error()
            return False
        return True

    def decode_launch(self, plan, d_in, d_out, events=None):
        assert self.lib.aws_huffman_amd_decode_plan_launch_staged(plan, d_in, d_out, None, events) == 0

    def decode_road(self, plan):
        """0: sync + scan + emit, the one road since round 5 (the query is kept for callers that link it)."""
        road, detail = C.c_uint32(99), (C.c_uint32 * 2)()
        assert self.lib.aws_huffman_amd_decode_plan_road(plan, None, C.byref(road), detail) == 0
        return road.value

    def decode_plan_is_quiet(self, plan):
        """The plan's last fetched launch listed no chunk: its next launch goes without the kernels for listed chunks."""
        return bool(self.lib.aws_huffman_amd_decode_plan_is_quiet(plan))

    def encode_stats(self, plan):
        st = PlanStats()
        assert self.lib.aws_huffman_amd_encode_plan_stats(plan, C.byref(st)) == 0
        return st.as_dict()

    def decode_stats(self, plan):
        st = PlanStats()
        assert self.lib.aws_huffman_amd_decode_plan_stats(plan, C.byref(st)) == 0
        return st.as_dict()

    def decode_results(self, plan, n):
        res = (AmdDecodeResult * max(n, 1))()
        assert self.lib.aws_huffman_amd_decode_plan_results(plan, res, None) == 0
        return [(r.rc, r.error, r.produced, r.bits_consumed) for r in res[:n]]


class Shards:
    """huffman_amd.h "several GPUs": item i on shard i mod G, one engine and one host thread per shard."""

    def __init__(self, lib, coder, devices):
        self.lib, self.G = lib, len(devices)
        h = C.c_void_p()
        devs = (C.c_int * self.G)(*devices)
        if lib.aws_huffman_amd_shards_new(C.byref(h), coder, devs, self.G) != 0:
            raise RuntimeError("aws_huffman_amd_shards_new failed, error %d" % lib.aws_last_error())
        self.h = h
        assert lib.aws_huffman_amd_shards_count(h) == self.G
        self.engines = []
        for g in range(self.G):
            e = Engine.__new__(Engine)
            e.lib, e.h = lib, C.c_void_p(lib.aws_huffman_amd_shards_engine(h, g))
            e.stream = lib.aws_huffman_amd_engine_stream(e.h)
            self.engines.append(e)

    def close(self):
        if self.h:
            self.lib.aws_huffman_amd_shards_destroy(self.h)
            self.h = None

    def _io(self, bases):
        io = (ShardIo * self.G)()
        for g, (d_in, d_out) in enumerate(bases):
            io[g].device_input, io[g].device_output = d_in, d_out
        return io

    def encode(self, items, bases):
        arr = (AmdEncodeItem * max(len(items), 1))()
        for i, it in enumerate(items):
            arr[i].in_offset, arr[i].in_len = it["in_offset"], it["in_len"]
            arr[i].out_offset, arr[i].out_capacity = it["out_offset"], it["out_capacity"]
            ov = it.get("overflow_in", (0, 0))
            arr[i].overflow_in.pattern, arr[i].overflow_in.num_bits = ov
            arr[i].eos_padding = it.get("eos_padding", 0xFF)
        res = (AmdEncodeResult * max(len(items), 1))()
        rc = self.lib.aws_huffman_amd_shards_encode(self.h, arr, len(items), self._io(bases), res)
        assert rc == 0, "shards_encode failed, error %d" % self.lib.aws_last_error()
        return [(r.rc, r.error, r.consumed, r.produced, r.overflow_out.num_bits,
                 r.overflow_out.pattern if r.overflow_out.num_bits else 0) for r in res[:len(items)]]

    def decode(self, items, bases):
        arr = (AmdDecodeItem * max(len(items), 1))()
        for i, it in enumerate(items):
            arr[i].in_offset, arr[i].in_len = it["in_offset"], it["in_len"]
            arr[i].first_bit = it.get("first_bit", 0)
            arr[i].out_offset, arr[i].out_capacity = it["out_offset"], it["out_capacity"]
        res = (AmdDecodeResult * max(len(items), 1))()
        rc = self.lib.aws_huffman_amd_shards_decode(self.h, arr, len(items), self._io(bases), res)
        assert rc == 0, "shards_decode failed, error %d" % self.lib.aws_last_error()
        return [(r.rc, r.error, r.produced, r.bits_consumed) for r in res[:len(items)]]


class CallResult:
    """Everything observable after one encode/decode call."""

    def __init__(self, rc, err, consumed, produced, state):
        self.rc, self.err, self.consumed, self.produced, self.state = rc, err, consumed, produced, state

    def key(self):
        return (self.rc, self.err, self.consumed, self.produced, self.state)

    def __repr__(self):
        return "CallResult(rc=%d err=%d consumed=%d produced=%d state=%s)" % self.key()


class Codec:
    """One of the two libraries behind a uniform Python face."""

    def __init__(self, lib, prefix):
        self.lib, self.prefix = lib, prefix
        g = lambda n: getattr(lib, prefix + n)
        self.encoder_init = g("huffman_encoder_init")
        self.encoder_reset = g("huffman_encoder_reset")
        self.decoder_init = g("huffman_decoder_init")
        self.decoder_reset = g("huffman_decoder_reset")
        self.decoder_allow_growth = g("huffman_decoder_allow_growth")
        self._get_len = g("huffman_get_encoded_length")
        self._encode = g("huffman_encode")
        self._decode = g("huffman_decode")
        self.last_error = getattr(lib, "oracle_last_error" if prefix == "oracle_" else "aws_last_error")
        self.reset_error = getattr(lib, "oracle_reset_error" if prefix == "oracle_" else "aws_reset_error")

    # -- state objects
    def new_encoder(self, coder, eos_padding=None):
        e = Encoder()
        self.encoder_init(C.byref(e), coder)
        if eos_padding is not None:
            e.eos_padding = eos_padding
        return e

    def new_decoder(self, coder):
        d = Decoder()
        self.decoder_init(C.byref(d), coder)
        return d

    def encoded_length(self, enc, data):
        """aws_huffman_get_encoded_length; an empty `data` is handed over as the cursor {0, NULL}."""
        arr = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8)) if not isinstance(data, np.ndarray) else data
        cur = ByteCursor(arr.size, arr.ctypes.data if arr.size else None)
        return self._get_len(C.byref(enc), cur)

    # -- one call; `src`/`dst` are numpy uint8 arrays owned by the caller
    def encode_call(self, enc, src, src_off, dst, dst_len, dst_cap, null_when_empty=False):
        """aws_huffman_encode on src[src_off:] into dst with len=dst_len, capacity=dst_cap.
        null_when_empty: an empty cursor is handed over as {0, NULL} (valid for the reference, source/huffman.c:161-167)."""
        cur = ByteCursor(src.size - src_off, src.ctypes.data + src_off if src.size else None)
        if null_when_empty and cur.len == 0:
            cur = ByteCursor(0, None)
            src = src[:0]
            src_off = 0
        buf = ByteBuf(dst_len, dst.ctypes.data, dst_cap, None)
        self.reset_error()
        rc = self._encode(C.byref(enc), C.byref(cur), C.byref(buf))
        err = self.last_error() if rc != 0 else 0
        consumed = (src.size - src_off) - cur.len
        ptr_ok = (cur.ptr or 0) == ((src.ctypes.data + src_off + consumed) if src.size else (cur.ptr or 0))
        assert ptr_ok, "cursor pointer and length disagree"
        nb = enc.overflow_bits.num_bits
        # the pattern is unspecified once num_bits is 0
        state = (nb, enc.overflow_bits.pattern if nb else 0)
        return CallResult(rc, err, consumed, buf.len - dst_len, state)

    def decode_call(self, dec, src, src_off, src_end, dst, dst_len, dst_cap, null_when_empty=False):
        """aws_huffman_decode on src[src_off:src_end] into dst with len=dst_len, capacity=dst_cap."""
        n = src_end - src_off
        cur = ByteCursor(n, src.ctypes.data + src_off if src.size else None)
        if null_when_empty and n == 0:
            cur = ByteCursor(0, None)
        buf = ByteBuf(dst_len, dst.ctypes.data, dst_cap, None)
        self.reset_error()
        rc = self._decode(C.byref(dec), C.byref(cur), C.byref(buf))
        err = self.last_error() if rc != 0 else 0
        state = (dec.num_bits, dec.working_bits)
        assert not (null_when_empty and n == 0) or not cur.ptr, "an empty NULL cursor came back with a pointer"
        return CallResult(rc, err, n - cur.len, buf.len - dst_len, state)

    # -- conveniences
    def encode_all(self, coder, data, eos_padding=None, slack=64):
        """One-shot encode into a buffer that is certainly large enough."""
        src = np.ascontiguousarray(data, dtype=np.uint8)
        cap = src.size * 4 + slack
        dst = np.zeros(cap, dtype=np.uint8)
        enc = self.new_encoder(coder, eos_padding)
        r = self.encode_call(enc, src, 0, dst, 0, cap)
        assert r.rc == 0 and r.consumed == src.size, r
        return dst[: r.produced].copy()

    def decode_all(self, coder, data, out_cap):
        src = np.ascontiguousarray(data, dtype=np.uint8)
        dst = np.zeros(max(out_cap, 1), dtype=np.uint8)
        dec = self.new_decoder(coder)
        r = self.decode_call(dec, src, 0, src.size, dst, 0, out_cap)
        return r, dst[: r.produced].copy()


def oracle_codec():
    return Codec(load_oracle(), "oracle_")


def product_codec():
    return Codec(load_product(), "aws_")
