"""Pins the CPU oracle to the reference (CPU only, no GPU).

Mirrors the reference's own unit tests (tests/huffman_test.c, cited per test) on
the oracle, then checks the oracle against the decision tree of the reference's
generated coder and against the records SURVEY.md section 8c holds from a run of
the real reference.  The GPU parity tests compare the HIP path with this oracle.
"""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

import harness
from harness import AWS_ERROR_COMPRESSION_UNKNOWN_SYMBOL as UNKNOWN_SYMBOL
from harness import AWS_ERROR_SHORT_BUFFER as SHORT_BUFFER

VEC = harness.load_json("reference_vectors.json")
TREE = harness.load_json("test_coder_decode_tree.json")
TABLE = harness.load_json("test_coder_table.json")["rows"]
PROBE = harness.load_json("survey_probe_records.json")

K1_PLAIN = np.frombuffer(bytes.fromhex(VEC["K1_url"]["plain"]), dtype=np.uint8)
K1_ENC = np.frombuffer(bytes.fromhex(VEC["K1_url"]["encoded"]), dtype=np.uint8)
K2_PLAIN = np.frombuffer(bytes.fromhex(VEC["K2_all_codes"]["plain"]), dtype=np.uint8)
K2_ENC = np.frombuffer(bytes.fromhex(VEC["K2_all_codes"]["encoded"]), dtype=np.uint8)
STEPS = VEC["step_sizes"]


def call_encode_cb(coder, sym):
    fn = harness.ENCODE_FN(coder.contents.encode)
    code = fn(sym, coder.contents.userdata)
    return code.pattern, code.num_bits


def call_decode_cb(coder, bits):
    fn = harness.DECODE_FN(coder.contents.decode)
    sym = C.c_uint8(0xEE)
    n = fn(bits, C.byref(sym), coder.contents.userdata)
    return n, sym.value


def test_abi_layout():
    harness.check_abi_layout()


# ---- tests/huffman_test.c:42-60 huffman_symbol_encoder
def test_symbol_encoder(oracle_coder):
    for row in TABLE:
        assert call_encode_cb(oracle_coder, row["symbol"]) == (row["pattern"], row["num_bits"])


# ---- tests/huffman_test.c:199-220 huffman_symbol_decoder
def test_symbol_decoder(oracle_coder):
    for row in TABLE:
        bits = (row["pattern"] << (32 - row["num_bits"])) & 0xFFFFFFFF
        assert call_decode_cb(oracle_coder, bits) == (row["num_bits"], row["symbol"])


# ---- decision tree of tests/test_huffman_static.c:276-2381
def test_decode_tree_leaves_and_dead_ends(oracle_coder):
    rng = np.random.default_rng(7)
    assert len(TREE["leaves"]) == 256 and len(TREE["dead_ends"]) == 9
    for leaf in TREE["leaves"]:
        p = leaf["prefix"]
        base = int(p, 2) << (32 - len(p))
        for fill in (0, (1 << (32 - len(p))) - 1, int(rng.integers(0, 1 << (32 - len(p))))):
            assert call_decode_cb(oracle_coder, base | fill) == (leaf["num_bits"], leaf["symbol"])
    for p in TREE["dead_ends"]:
        base = int(p, 2) << (32 - len(p))
        for fill in (0, (1 << (32 - len(p))) - 1):
            n, sym = call_decode_cb(oracle_coder, base | fill)
            assert n == 0 and sym == 0xEE  # *symbol untouched on a miss (huffman.h:41-42)


def test_decode_tree_as_10bit_lut(oracle_coder):
    """779 valid 10-bit windows (320/144/40/40/26/209 by length), 245 invalid."""
    by_len = {}
    for w in range(1024):
        n, _ = call_decode_cb(oracle_coder, w << 22)
        by_len[n] = by_len.get(n, 0) + 1
    assert by_len == {0: 245, 5: 320, 6: 144, 7: 40, 8: 40, 9: 26, 10: 209}


# ---- tests/huffman_test.c:62-115 huffman_encoder, huffman_encoder_all_code_points
@pytest.mark.parametrize("plain,enc", [(K1_PLAIN, K1_ENC), (K2_PLAIN, K2_ENC)], ids=["K1", "K2"])
def test_encoder_known_answer(oracle, oracle_coder, plain, enc):
    e = oracle.new_encoder(oracle_coder)
    assert oracle.encoded_length(e, plain) == enc.size
    dst = np.zeros(enc.size + 1, dtype=np.uint8)
    r = oracle.encode_call(e, plain, 0, dst, 0, enc.size)
    assert (r.rc, r.consumed, r.produced) == (0, plain.size, enc.size)
    assert dst[enc.size] == 0  # byte past the buffer untouched (huffman_test.c:83,111)
    assert bytes(dst[: enc.size]) == bytes(enc)


# ---- tests/huffman_test.c:117-165 huffman_encoder_partial_output
@pytest.mark.parametrize("step", STEPS)
def test_encoder_partial_output(oracle, oracle_coder, step):
    e = oracle.new_encoder(oracle_coder)
    oracle.encoder_reset(e)
    dst = np.zeros(K2_ENC.size, dtype=np.uint8)
    cap = length = off = 0
    while length < K2_ENC.size:
        cap = min(cap + step, K2_ENC.size)
        r = oracle.encode_call(e, K2_PLAIN, off, dst, length, cap)
        assert r.produced > 0
        length += r.produced
        off += r.consumed
        assert bytes(dst[:length]) == bytes(K2_ENC[:length])
        if length == K2_ENC.size:
            assert r.rc == 0
        else:
            assert r.rc == -1 and r.err == SHORT_BUFFER
    assert off == K2_PLAIN.size


# ---- tests/huffman_test.c:167-197 huffman_encoder_exact_output
def test_encoder_exact_output(oracle, oracle_coder):
    e = oracle.new_encoder(oracle_coder)
    for case in VEC["K3_exact_fit"]:
        plain = np.frombuffer(bytes.fromhex(case["plain"]), dtype=np.uint8)
        want = bytes.fromhex(case["encoded"])
        dst = np.zeros(2, dtype=np.uint8)
        r = oracle.encode_call(e, plain, 0, dst, 0, len(want))
        assert r.rc == 0 and r.produced == len(want) and bytes(dst[: len(want)]) == want


# ---- tests/huffman_test.c:222-273 huffman_decoder, huffman_decoder_all_code_points
@pytest.mark.parametrize("plain,enc", [(K1_PLAIN, K1_ENC), (K2_PLAIN, K2_ENC)], ids=["K1", "K2"])
def test_decoder_known_answer(oracle, oracle_coder, plain, enc):
    d = oracle.new_decoder(oracle_coder)
    dst = np.zeros(plain.size + 1, dtype=np.uint8)
    r = oracle.decode_call(d, enc, 0, enc.size, dst, 0, plain.size)
    assert (r.rc, r.consumed, r.produced) == (0, enc.size, plain.size)
    assert dst[plain.size] == 0 and bytes(dst[: plain.size]) == bytes(plain)


# ---- tests/huffman_test.c:275-314 huffman_decoder_partial_input
@pytest.mark.parametrize("step", STEPS)
def test_decoder_partial_input(oracle, oracle_coder, step):
    d = oracle.new_decoder(oracle_coder)
    oracle.decoder_reset(d)
    dst = np.zeros(150, dtype=np.uint8)
    off = length = 0
    while length < K2_PLAIN.size:
        chunk = min(step, K2_ENC.size - off)
        r = oracle.decode_call(d, K2_ENC, off, off + chunk, dst, length, K2_PLAIN.size)
        assert r.consumed == chunk  # every chunk is swallowed whole (huffman_test.c:301)
        off += chunk
        length += r.produced
        assert bytes(dst[:length]) == bytes(K2_PLAIN[:length])
        if length == K2_PLAIN.size:
            assert r.rc == 0
    assert length == K2_PLAIN.size


# ---- tests/huffman_test.c:316-363 huffman_decoder_partial_output
@pytest.mark.parametrize("step", STEPS)
def test_decoder_partial_output(oracle, oracle_coder, step):
    d = oracle.new_decoder(oracle_coder)
    dst = np.zeros(150, dtype=np.uint8)
    off = length = cap = 0
    while length < K2_PLAIN.size:
        cap = min(cap + step, K2_PLAIN.size)
        r = oracle.decode_call(d, K2_ENC, off, K2_ENC.size, dst, length, cap)
        assert r.produced > 0
        off += r.consumed
        length += r.produced
        assert bytes(dst[:length]) == bytes(K2_PLAIN[:length])
        if length == K2_PLAIN.size:
            assert r.rc == 0
        else:
            assert r.rc == -1 and r.err == SHORT_BUFFER


# ---- tests/huffman_test.c:365-385 huffman_decoder_allow_growth
def test_decoder_allow_growth(oracle, oracle_coder):
    lib = oracle.lib
    d = oracle.new_decoder(oracle_coder)
    oracle.decoder_allow_growth(d, True)
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.free.argtypes = [C.c_void_p]
    buf = harness.ByteBuf(0, libc.malloc(1), 1, lib.oracle_default_allocator())
    cur = harness.ByteCursor(K1_ENC.size, K1_ENC.ctypes.data)
    assert lib.oracle_huffman_decode(C.byref(d), C.byref(cur), C.byref(buf)) == 0
    assert cur.len == 0 and buf.len == K1_PLAIN.size
    assert C.string_at(buf.buffer, buf.len) == bytes(K1_PLAIN)
    assert buf.capacity == 16  # 1 -> 2 -> 4 -> 8 -> 16 by doubling (huffman.c:260-264)
    libc.free(buf.buffer)


# ---- tests/huffman_test.c:387-446 huffman_transitive*, via the restated helpers of source/huffman_testing.c
def test_transitive_helpers(oracle, oracle_coder):
    lib = oracle.lib
    msg = C.c_char_p()
    k4 = VEC["K4_even_bytes"]
    cases = [(bytes(K1_PLAIN), K1_ENC.size), (bytes.fromhex(k4["plain"]), k4["encoded_len"]), (bytes(K2_PLAIN), K2_ENC.size)]
    for plain, enc_len in cases:
        assert lib.oracle_huffman_test_transitive(oracle_coder, plain, len(plain), enc_len, C.byref(msg)) == 0, msg.value
    for step in STEPS:
        rc = lib.oracle_huffman_test_transitive_chunked(
            oracle_coder, bytes(K2_PLAIN), K2_PLAIN.size, K2_ENC.size, step, C.byref(msg))
        assert rc == 0, (step, msg.value)
    # the helpers do report a wrong expected size
    assert lib.oracle_huffman_test_transitive(oracle_coder, bytes(K1_PLAIN), 15, 13, C.byref(msg)) == -1
    assert msg.value == b"encoded length is incorrect"


# ---- digests from the real reference
def stream_input(rec):
    raw = harness.splitmix64_bytes(rec["seed"], rec["len"])
    return harness.printable_map(raw) if rec["map"] == "printable" else raw


@pytest.mark.parametrize("name", ["G4K", "G16K", "G16KP", "G1M"])
def test_survey_stream_digests(oracle, oracle_coder, name):
    rec = PROBE["streams"][name]
    plain = stream_input(rec)
    assert hashlib.sha256(plain.tobytes()).hexdigest() == rec["sha256_input"]
    enc = oracle.encode_all(oracle_coder, plain)
    assert enc.size == rec["encoded_len"]
    assert hashlib.sha256(enc.tobytes()).hexdigest() == rec["sha256_encoded"]
    if "encoded_prefix" in rec:
        assert enc[:8].tobytes().hex() == rec["encoded_prefix"] and enc[-4:].tobytes().hex() == rec["encoded_suffix"]
    e = oracle.new_encoder(oracle_coder)
    assert oracle.encoded_length(e, plain) == rec["encoded_len"]
    r, back = oracle.decode_all(oracle_coder, enc, plain.size)
    assert (r.rc, r.consumed, r.produced) == (0, enc.size, plain.size)
    assert r.state[0] == rec["decoder_tail_num_bits"]
    assert np.array_equal(back, plain)


def test_splitmix_twins_agree(oracle):
    a = harness.splitmix64_bytes(12345, 1003)
    b = np.zeros(1003, dtype=np.uint8)
    oracle.lib.oracle_splitmix64_fill(b.ctypes.data, b.size, 12345)
    assert np.array_equal(a, b)
    assert a[:8].tobytes().hex() != "00" * 8
    assert harness.splitmix64_bytes(1, 8).tobytes().hex() == PROBE["streams"]["G4K"]["input_prefix"]


@pytest.mark.slow
@pytest.mark.skipif(os.environ.get("HUFFMAN_SLOW") != "1", reason="1 GiB through the scalar oracle takes about 2 minutes")
def test_survey_stream_digest_G1G(oracle, oracle_coder):
    rec = PROBE["streams"]["G1G"]
    plain = stream_input(rec)
    assert hashlib.sha256(plain.tobytes()).hexdigest() == rec["sha256_input"]
    src = plain
    dst = np.zeros(rec["encoded_len"] + 16, dtype=np.uint8)
    e = oracle.new_encoder(oracle_coder)
    r = oracle.encode_call(e, src, 0, dst, 0, dst.size)
    assert (r.rc, r.consumed, r.produced) == (0, src.size, rec["encoded_len"])
    assert hashlib.sha256(dst[: r.produced].tobytes()).hexdigest() == rec["sha256_encoded"]


def test_survey_partial_encode_records(oracle, oracle_coder):
    plain = stream_input(PROBE["streams"]["G16K"])
    for rec in PROBE["G16K_partial_encode"]:
        e = oracle.new_encoder(oracle_coder)
        dst = np.zeros(20000, dtype=np.uint8)
        r = oracle.encode_call(e, plain, 0, dst, 0, rec["cap"])
        assert (r.rc, r.consumed, r.produced) == (rec["rc"], rec["consumed"], rec["out_len"]), rec
        if rec["rc"]:
            assert r.err == SHORT_BUFFER
            assert r.state == (rec["overflow_num_bits"], rec["overflow_pattern"])
        else:
            assert r.state[0] == 0
        # resume into a roomy buffer: the total is the one-shot length every time
        r2 = oracle.encode_call(e, plain, r.consumed, dst, r.produced, dst.size)
        assert r2.rc == 0 and r.produced + r2.produced == 19438
        assert hashlib.sha256(dst[:19438].tobytes()).hexdigest() == PROBE["streams"]["G16K"]["sha256_encoded"]


def test_survey_K1_step1_encode_trace(oracle, oracle_coder):
    trace = PROBE["K1_encode_step1_trace"]
    e = oracle.new_encoder(oracle_coder)
    dst = np.zeros(12, dtype=np.uint8)
    off = length = 0
    for cap in range(1, 13):
        r = oracle.encode_call(e, K1_PLAIN, off, dst, length, cap)
        off += r.consumed
        length += r.produced
        assert off == trace["consumed"][cap - 1] and r.state[0] == trace["overflow_num_bits"][cap - 1], cap
        assert (r.rc == 0) == (cap == 12)
    assert bytes(dst) == bytes(K1_ENC)


def test_survey_K1_decode_partial_output_trace(oracle, oracle_coder):
    t = PROBE["K1_decode_partial_output"]
    d = oracle.new_decoder(oracle_coder)
    dst = np.zeros(16, dtype=np.uint8)
    off = length = 0
    for cap, want_off, want_bits in zip(t["caps"], t["input_consumed"], t["num_bits"]):
        r = oracle.decode_call(d, K1_ENC, off, K1_ENC.size, dst, length, cap)
        off += r.consumed
        length += r.produced
        assert off == want_off and r.state[0] == want_bits, cap
    assert length == 15 and "%016x" % d.working_bits == t["final_working_bits"]


def test_survey_raw_bytes_decode_records(oracle, oracle_coder):
    for rec in PROBE["raw_bytes_as_stream_decode"]:
        if "input" in rec:
            data = stream_input(PROBE["streams"][rec["input"]])
        else:
            data = np.frombuffer(bytes.fromhex(rec["input_hex"]), dtype=np.uint8)
        d = oracle.new_decoder(oracle_coder)
        dst = np.zeros(rec["out_cap"], dtype=np.uint8)
        r = oracle.decode_call(d, data, 0, data.size, dst, 0, rec["out_cap"])
        assert (r.rc, r.err) == (rec["rc"], rec["error"]), rec
        assert dst[: r.produced].tobytes().hex() == rec["symbols"]
        assert r.consumed == rec["input_pulled"]
        if "num_bits" in rec:
            assert r.state[0] == rec["num_bits"]
        if "working_bits" in rec:
            assert "%016x" % r.state[1] == rec["working_bits"]


def test_survey_eos_padding_uses_low_bits(oracle, oracle_coder):
    for rec in PROBE["eos_padding_probe"]:
        plain = np.frombuffer(bytes.fromhex(rec["plain"]), dtype=np.uint8)
        enc = oracle.encode_all(oracle_coder, plain, eos_padding=rec["eos_padding"])
        assert enc.tobytes().hex() == rec["encoded"]


# ---- fuzz properties of tests/fuzz/*.c as seeded tests
def test_fuzz_decode_arbitrary_bytes_is_safe(oracle, oracle_coder):
    rng = np.random.default_rng(101)
    for _ in range(300):
        n = int(rng.integers(1, 200))
        data = rng.integers(0, 256, n, dtype=np.uint8)
        d = oracle.new_decoder(oracle_coder)
        dst = np.zeros(2 * n + 1, dtype=np.uint8)
        r = oracle.decode_call(d, data, 0, n, dst, 0, 2 * n)
        assert r.rc in (0, -1) and r.produced <= 2 * n and dst[2 * n] == 0
        if r.rc:
            assert r.err in (UNKNOWN_SYMBOL, SHORT_BUFFER)


def test_fuzz_transitive_and_chunked(oracle, oracle_coder):
    rng = np.random.default_rng(202)
    msg = C.c_char_p()
    for i in range(120):
        n = int(rng.integers(1, 400))
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert oracle.lib.oracle_huffman_test_transitive(oracle_coder, data, n, 0, C.byref(msg)) == 0, msg.value
        step = STEPS[i % len(STEPS)]
        assert oracle.lib.oracle_huffman_test_transitive_chunked(oracle_coder, data, n, 0, step, C.byref(msg)) == 0, msg.value


def test_unknown_symbol_and_empty_input(oracle):
    """Coder with holes: symbol 7 has no code (huffman.c:62-64); empty input succeeds and writes nothing."""
    patterns, lens = harness.load_table()
    lens[7] = 0
    coder = oracle.lib.oracle_table_coder_new(patterns, lens)
    e = oracle.new_encoder(coder)
    src = np.array([97, 98, 7, 99], dtype=np.uint8)
    dst = np.zeros(16, dtype=np.uint8)
    r = oracle.encode_call(e, src, 0, dst, 0, 16)
    assert (r.rc, r.err, r.consumed) == (-1, UNKNOWN_SYMBOL, 3)
    assert r.produced == 1  # 'a' 5 bits + 'b' 6 bits: one whole byte out, the partial byte is lost
    e2 = oracle.new_encoder(coder)
    r = oracle.encode_call(e2, np.zeros(0, dtype=np.uint8), 0, dst, 0, 16)
    assert (r.rc, r.consumed, r.produced) == (0, 0, 0)
    oracle.lib.oracle_table_coder_destroy(coder)


def test_config5_pins_are_anchored_to_the_survey():
    """tests/golden/config5_stream_pins.json (the oracle over the 1 GiB streams of seeds 5 .. 12, make_config5_pins.py):
    its seed 5 must be the survey's record of the REAL reference, and every other stream is the same generator's with
    another seed -- same length in, about the same length out."""
    pins = harness.load_json("config5_stream_pins.json")
    ref = PROBE["streams"]["G1G"]
    five = pins["streams"]["5"]
    assert pins["len"] == ref["len"] == 1 << 30
    assert (five["encoded_len"], five["sha256_input"], five["sha256_encoded"]) == (ref["encoded_len"], ref["sha256_input"], ref["sha256_encoded"])
    assert sorted(int(k) for k in pins["streams"]) == list(range(5, 13))
    for seed, rec in pins["streams"].items():
        assert rec["seed"] == int(seed) and abs(rec["encoded_len"] - ref["encoded_len"]) < 100_000
        assert len(rec["sha256_encoded"]) == 64 and len(rec["sha256_input"]) == 64
    assert len({r["sha256_encoded"] for r in pins["streams"].values()}) == 8
