"""Parity scenarios: the product library against the CPU oracle, call by call.

Every scenario runs the same calls on both libraries and compares everything a
caller can observe: return code, raised error, how far the cursor moved, how
many bytes were appended, the streaming state left in the encoder/decoder
struct, the bytes written AND the bytes that must stay untouched.

The scenarios are library-agnostic: tests/test_emulated_kernels.py runs them on
the CPU-emulated build of the kernels (logic + UBSan, no GPU), and
tests/test_gpu_parity.py runs them through the hipcc build on an MI355X.
"""
import ctypes as C
import hashlib
import os

import numpy as np

import harness
from harness import AWS_ERROR_COMPRESSION_UNKNOWN_SYMBOL as UNKNOWN_SYMBOL
from harness import AWS_ERROR_SHORT_BUFFER as SHORT_BUFFER

SENTINEL = 0x5A
VEC = harness.load_json("reference_vectors.json")
PROBE = harness.load_json("survey_probe_records.json")
K1_PLAIN = np.frombuffer(bytes.fromhex(VEC["K1_url"]["plain"]), dtype=np.uint8)
K1_ENC = np.frombuffer(bytes.fromhex(VEC["K1_url"]["encoded"]), dtype=np.uint8)
K2_PLAIN = np.frombuffer(bytes.fromhex(VEC["K2_all_codes"]["plain"]), dtype=np.uint8)
K2_ENC = np.frombuffer(bytes.fromhex(VEC["K2_all_codes"]["encoded"]), dtype=np.uint8)
STEPS = VEC["step_sizes"]


class World:
    """Both libraries, each with its own instance of the same coder."""

    def __init__(self, oracle, product):
        self.oracle, self.product = oracle, product
        patterns, lens = harness.load_table()
        self.table = (patterns, lens)
        self.ocoder = oracle.lib.oracle_table_coder_new(patterns, lens)
        self.pcoder = product.lib.aws_huffman_amd_table_coder_new(patterns, lens)
        assert self.ocoder and self.pcoder
        # the same table with holes: symbols 7 and 200 have no code
        holes = (C.c_uint8 * 256)(*lens)
        holes[7] = 0
        holes[200] = 0
        self.ocoder_holes = oracle.lib.oracle_table_coder_new(patterns, holes)
        self.pcoder_holes = product.lib.aws_huffman_amd_table_coder_new(patterns, holes)
        assert self.ocoder_holes and self.pcoder_holes


# ----------------------------------------------------------------------------- input families
def inputs(rng, n, kind):
    if kind == "uniform":
        return rng.integers(0, 256, n, dtype=np.uint8)
    if kind == "printable":
        return harness.printable_map(rng.integers(0, 256, n, dtype=np.uint8))
    if kind == "short":  # only the ten 5-bit symbols: densest possible output per bit
        alphabet = np.frombuffer(b" aeinorst", dtype=np.uint8)
        return alphabet[rng.integers(0, alphabet.size, n)]
    if kind == "long":  # only 10-bit symbols: no self-synchronisation between bit offsets
        return rng.integers(128, 256, n, dtype=np.uint8)
    if kind == "constant":
        return np.full(n, 101, dtype=np.uint8)
    raise ValueError(kind)


KINDS = ["uniform", "printable", "short", "long", "constant"]


# ----------------------------------------------------------------------------- paired calls
def paired_encode(w, eo, ep, src, off, dst_o, dst_p, length, cap, coder_pair=None, null_when_empty=False):
    ro = w.oracle.encode_call(eo, src, off, dst_o, length, cap, null_when_empty)
    rp = w.product.encode_call(ep, src, off, dst_p, length, cap, null_when_empty)
    assert rp.key() == ro.key(), "encode call differs: product %r oracle %r (off=%d len=%d cap=%d)" % (rp, ro, off, length, cap)
    assert np.array_equal(dst_p, dst_o), "encode output bytes differ (off=%d len=%d cap=%d)" % (off, length, cap)
    return ro


def paired_decode(w, do, dp, src, off, end, dst_o, dst_p, length, cap, null_when_empty=False):
    ro = w.oracle.decode_call(do, src, off, end, dst_o, length, cap, null_when_empty)
    rp = w.product.decode_call(dp, src, off, end, dst_p, length, cap, null_when_empty)
    assert rp.key() == ro.key(), "decode call differs: product %r oracle %r (off=%d end=%d len=%d cap=%d)" % (
        rp, ro, off, end, length, cap)
    assert np.array_equal(dst_p, dst_o), "decode output bytes differ (off=%d end=%d len=%d cap=%d)" % (off, end, length, cap)
    return ro


def oracle_encode(w, data, coder=None, eos=None):
    return w.oracle.encode_all(coder or w.ocoder, data, eos_padding=eos)


# ----------------------------------------------------------------------------- scenario: the reference's unit tests on the product
def reference_unit_tests(codec, coder):
    """tests/huffman_test.c:62-385 on `codec` (known answers, no oracle involved)."""
    # huffman_encoder / _all_code_points
    for plain, enc in ((K1_PLAIN, K1_ENC), (K2_PLAIN, K2_ENC)):
        e = codec.new_encoder(coder)
        assert codec.encoded_length(e, plain) == enc.size
        dst = np.zeros(enc.size + 1, dtype=np.uint8)
        r = codec.encode_call(e, plain, 0, dst, 0, enc.size)
        assert (r.rc, r.consumed, r.produced) == (0, plain.size, enc.size)
        assert dst[enc.size] == 0 and bytes(dst[: enc.size]) == bytes(enc)
        d = codec.new_decoder(coder)
        out = np.zeros(plain.size + 1, dtype=np.uint8)
        r = codec.decode_call(d, enc, 0, enc.size, out, 0, plain.size)
        assert (r.rc, r.consumed, r.produced) == (0, enc.size, plain.size)
        assert out[plain.size] == 0 and bytes(out[: plain.size]) == bytes(plain)
    # huffman_encoder_partial_output
    for step in STEPS:
        e = codec.new_encoder(coder)
        dst = np.zeros(K2_ENC.size, dtype=np.uint8)
        cap = length = off = 0
        while length < K2_ENC.size:
            cap = min(cap + step, K2_ENC.size)
            r = codec.encode_call(e, K2_PLAIN, off, dst, length, cap)
            assert r.produced > 0
            length += r.produced
            off += r.consumed
            assert bytes(dst[:length]) == bytes(K2_ENC[:length])
            assert (r.rc == 0) if length == K2_ENC.size else (r.rc == -1 and r.err == SHORT_BUFFER)
    # huffman_encoder_exact_output
    e = codec.new_encoder(coder)
    for case in VEC["K3_exact_fit"]:
        plain = np.frombuffer(bytes.fromhex(case["plain"]), dtype=np.uint8)
        want = bytes.fromhex(case["encoded"])
        dst = np.zeros(2, dtype=np.uint8)
        r = codec.encode_call(e, plain, 0, dst, 0, len(want))
        assert r.rc == 0 and bytes(dst[: len(want)]) == want
    # huffman_decoder_partial_input / _partial_output
    for step in STEPS:
        d = codec.new_decoder(coder)
        dst = np.zeros(150, dtype=np.uint8)
        off = length = 0
        while length < K2_PLAIN.size:
            chunk = min(step, K2_ENC.size - off)
            r = codec.decode_call(d, K2_ENC, off, off + chunk, dst, length, K2_PLAIN.size)
            assert r.consumed == chunk
            off += chunk
            length += r.produced
            assert bytes(dst[:length]) == bytes(K2_PLAIN[:length])
        d = codec.new_decoder(coder)
        dst = np.zeros(150, dtype=np.uint8)
        off = length = cap = 0
        while length < K2_PLAIN.size:
            cap = min(cap + step, K2_PLAIN.size)
            r = codec.decode_call(d, K2_ENC, off, K2_ENC.size, dst, length, cap)
            assert r.produced > 0
            off += r.consumed
            length += r.produced
            assert bytes(dst[:length]) == bytes(K2_PLAIN[:length])
            assert (r.rc == 0) if length == K2_PLAIN.size else (r.rc == -1 and r.err == SHORT_BUFFER)
    # huffman_decoder_allow_growth
    alloc_name = "oracle_default_allocator" if codec.prefix == "oracle_" else "aws_default_allocator"
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.free.argtypes = [C.c_void_p]
    d = codec.new_decoder(coder)
    codec.decoder_allow_growth(d, True)
    buf = harness.ByteBuf(0, libc.malloc(1), 1, getattr(codec.lib, alloc_name)())
    cur = harness.ByteCursor(K1_ENC.size, K1_ENC.ctypes.data)
    assert codec._decode(C.byref(d), C.byref(cur), C.byref(buf)) == 0
    assert cur.len == 0 and buf.len == K1_PLAIN.size and buf.capacity == 16
    assert C.string_at(buf.buffer, buf.len) == bytes(K1_PLAIN)
    libc.free(buf.buffer)


# ----------------------------------------------------------------------------- scenario: one-shot calls
def one_shot_roundtrips(w, sizes, seed=11):
    rng = np.random.default_rng(seed)
    for n in sizes:
        for kind in KINDS:
            data = inputs(rng, n, kind)
            cap = n * 2 + 64
            do, dp = np.full(cap, SENTINEL, np.uint8), np.full(cap, SENTINEL, np.uint8)
            eo, ep = w.oracle.new_encoder(w.ocoder), w.product.new_encoder(w.pcoder)
            assert w.product.encoded_length(ep, data) == w.oracle.encoded_length(eo, data)
            r = paired_encode(w, eo, ep, data, 0, do, dp, 0, cap)
            enc = do[: r.produced].copy()
            # decode: exact capacity, one byte short, and roomy
            for out_cap in (n, max(n - 1, 0), n + 7):
                oo, op = np.full(n + 8, SENTINEL, np.uint8), np.full(n + 8, SENTINEL, np.uint8)
                ddo, ddp = w.oracle.new_decoder(w.ocoder), w.product.new_decoder(w.pcoder)
                paired_decode(w, ddo, ddp, enc, 0, enc.size, oo, op, 0, out_cap)


# ----------------------------------------------------------------------------- scenario: streaming encode (output in pieces)
def streaming_encode(w, sizes, seed=12, coder_pair=None):
    rng = np.random.default_rng(seed)
    oc, pc = coder_pair or (w.ocoder, w.pcoder)
    for n in sizes:
        for kind in ("uniform", "printable", "short"):
            data = inputs(rng, n, kind)
            total = n * 2 + 64
            do, dp = np.full(total, SENTINEL, np.uint8), np.full(total, SENTINEL, np.uint8)
            eo, ep = w.oracle.new_encoder(oc, eos_padding=0xA7), w.product.new_encoder(pc, eos_padding=0xA7)
            off = length = cap = 0
            for _ in range(10000):
                grow = int(rng.choice([0, 1, 1, 2, 3, 7, 64, 1000, 5000, 20000]))
                cap = min(cap + grow, total)
                r = paired_encode(w, eo, ep, data, off, do, dp, length, cap)
                off += r.consumed
                length += r.produced
                if r.rc == 0:
                    break
                assert r.err == SHORT_BUFFER
            else:
                raise AssertionError("streaming encode did not finish")
            assert off == n


# ----------------------------------------------------------------------------- scenario: the reference's coder-testing helpers, exported by the product
def transitive_helpers(w):
    """include/aws/compression/private/huffman_testing.h on the product: tests/huffman_test.c:387-446 as the
    reference runs them, plus a coder with HPACK's code lengths."""
    import ctypes as C

    lib = w.product.lib
    msg = C.c_char_p()
    k4 = VEC["K4_even_bytes"]
    cases = [(bytes(K1_PLAIN), K1_ENC.size), (bytes.fromhex(k4["plain"]), k4["encoded_len"]), (bytes(K2_PLAIN), K2_ENC.size)]
    for plain, enc_len in cases:
        assert lib.huffman_test_transitive(w.pcoder, plain, len(plain), enc_len, C.byref(msg)) == 0, msg.value
    for step in STEPS:
        rc = lib.huffman_test_transitive_chunked(w.pcoder, bytes(K2_PLAIN), K2_PLAIN.size, K2_ENC.size, step, C.byref(msg))
        assert rc == 0, (step, msg.value)
    assert lib.huffman_test_transitive(w.pcoder, bytes(K1_PLAIN), 15, 13, C.byref(msg)) == -1
    assert msg.value == b"encoded length is incorrect"
    assert lib.huffman_test_transitive_chunked(w.pcoder, bytes(K1_PLAIN), 15, 11, 3, C.byref(msg)) == -1
    assert msg.value == b"encoded length is incorrect"
    # what a downstream user does with its own table: here HPACK's code lengths, printable text
    _, hp, _ = profile_coders(w, "hpack_lengths")
    text = bytes(inputs(np.random.default_rng(5), 700, "printable"))
    assert lib.huffman_test_transitive(hp, text, len(text), 0, C.byref(msg)) == 0, msg.value
    assert lib.huffman_test_transitive_chunked(hp, text[:90], 90, 0, 7, C.byref(msg)) == 0, msg.value


# ----------------------------------------------------------------------------- scenario: threads sharing a coder
def shared_coder_threads(w, n_threads=4, calls=60, seed=47):
    """Distinct encoders/decoders on different threads with the SAME coder (the reference's generated coders are
    function-static, so every connection of a process shares one): the calls meet inside one engine."""
    import threading

    rng = np.random.default_rng(seed)
    jobs = []
    for t in range(n_threads):
        mine = []
        for _ in range(calls):
            data = inputs(rng, int(rng.choice([1, 15, 100, 700, 5000, 40000])), KINDS[t % 3])
            mine.append((data, oracle_encode(w, data)))
        jobs.append(mine)
    errors = []

    def run(mine):
        try:
            for data, want in mine:
                got = w.product.encode_all(w.pcoder, data)
                assert np.array_equal(got, want), "encode differs under threads"
                r, back = w.product.decode_all(w.pcoder, want, data.size)
                assert r.rc == 0 and np.array_equal(back, data), "decode differs under threads"
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=run, args=(mine,)) for mine in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[0]


# ----------------------------------------------------------------------------- scenario: streaming decode (input and output in pieces)
def streaming_decode(w, sizes, seed=13):
    rng = np.random.default_rng(seed)
    for n in sizes:
        for kind in ("uniform", "printable", "long"):
            data = inputs(rng, n, kind)
            enc = oracle_encode(w, data)
            oo, op = np.full(n + 8, SENTINEL, np.uint8), np.full(n + 8, SENTINEL, np.uint8)
            ddo, ddp = w.oracle.new_decoder(w.ocoder), w.product.new_decoder(w.pcoder)
            fed = length = cap = 0
            pending_lo = 0  # start of the not-yet-consumed part of what was fed
            for _ in range(20000):
                # offer a few more input bytes and a little more output room
                fed = min(fed + int(rng.choice([0, 1, 2, 3, 5, 8, 100, 3000, 40000])), enc.size)
                cap = min(cap + int(rng.choice([0, 1, 2, 9, 200, 5000, 50000])), n)
                r = paired_decode(w, ddo, ddp, enc, pending_lo, fed, oo, op, length, cap)
                pending_lo += r.consumed
                length += r.produced
                if length == n and pending_lo == enc.size:
                    break
                if r.rc != 0:
                    assert r.err == SHORT_BUFFER
            else:
                raise AssertionError("streaming decode did not finish")
            assert np.array_equal(oo[:n], data)


# ----------------------------------------------------------------------------- scenario: symbols without a code
def unknown_symbols(w, seed=14, big=16384 * 2 + 100):
    rng = np.random.default_rng(seed)
    oc, pc = w.ocoder_holes, w.pcoder_holes
    cases = []
    for n, bad_at in ((40, [0]), (40, [39]), (300, [150, 151]), (big, [5]), (big, [16383]), (big, [16384]),
                      (big, [16385]), (big, [big - 1]), (big, [16380, 20000])):
        data = inputs(rng, n, "printable")
        data[data == 7] = 8
        data[data == 200] = 201
        for k, b in enumerate(bad_at):
            data[b] = 7 if k % 2 == 0 else 200
        cases.append(data)
    for data in cases:
        n = data.size
        full = w.oracle.encoded_length(w.oracle.new_encoder(oc), data)
        first_bad = int(np.flatnonzero((data == 7) | (data == 200))[0])
        bits_before = w.oracle.encoded_length(w.oracle.new_encoder(oc), data[:first_bad])
        for cap in sorted({0, 1, max(bits_before - 1, 0), bits_before, bits_before + 1, bits_before + 2, full, full + 50,
                           n * 2}):
            do, dp = np.full(n * 2 + 64, SENTINEL, np.uint8), np.full(n * 2 + 64, SENTINEL, np.uint8)
            eo, ep = w.oracle.new_encoder(oc), w.product.new_encoder(pc)
            r = paired_encode(w, eo, ep, data, 0, do, dp, 0, min(cap, do.size))
            if r.rc == -1 and r.err == SHORT_BUFFER:
                # resume into a roomy buffer: now the bad symbol must surface identically
                paired_encode(w, eo, ep, data, r.consumed, do, dp, r.produced, do.size)
        assert w.product.encoded_length(w.product.new_encoder(pc), data) == full


# ----------------------------------------------------------------------------- scenario: decode of arbitrary bytes (tests/fuzz/decode.c) + pinned error records
def garbage_decode(w, seed=15, rounds=120, big=70000):
    rng = np.random.default_rng(seed)
    streams = [rng.integers(0, 256, int(rng.integers(1, 300)), dtype=np.uint8) for _ in range(rounds)]
    streams += [np.full(k, v, np.uint8) for k in (1, 3, 4, 8, 200) for v in (0x00, 0xFF, 0x55)]
    # valid streams with damage in the middle, across a chunk boundary
    good = oracle_encode(w, inputs(rng, big, "uniform"))
    for at in (0, 1000, 32767, 32768, 32769, good.size - 3):
        bad = good.copy()
        bad[at] ^= 0xFF
        bad[min(at + 1, bad.size - 1)] = 0x00
        streams.append(bad)
    streams.append(good[: good.size - 1])
    # cut around a chunk boundary: the last whole lane, the careful lane and the next chunk's first bytes
    for cut in (32768 - 140, 32768 - 9, 32768 - 3, 32768, 32768 + 3, 32768 + 5, 32768 + 7, 32768 + 8, 32768 + 135, 32768 + 137):
        streams.append(good[:cut])
    for data in streams:
        n = data.size
        for out_cap in (2 * n, 3):
            oo, op = np.full(2 * n + 8, SENTINEL, np.uint8), np.full(2 * n + 8, SENTINEL, np.uint8)
            ddo, ddp = w.oracle.new_decoder(w.ocoder), w.product.new_decoder(w.pcoder)
            paired_decode(w, ddo, ddp, data, 0, n, oo, op, 0, out_cap)
    for rec in PROBE["raw_bytes_as_stream_decode"]:
        if "input" in rec:
            s = PROBE["streams"][rec["input"]]
            data = harness.splitmix64_bytes(s["seed"], s["len"])
        else:
            data = np.frombuffer(bytes.fromhex(rec["input_hex"]), dtype=np.uint8)
        d = w.product.new_decoder(w.pcoder)
        dst = np.zeros(rec["out_cap"], dtype=np.uint8)
        r = w.product.decode_call(d, data, 0, data.size, dst, 0, rec["out_cap"])
        assert (r.rc, r.err, r.consumed) == (rec["rc"], rec["error"], rec["input_pulled"]), (rec, r)
        assert dst[: r.produced].tobytes().hex() == rec["symbols"]


# ----------------------------------------------------------------------------- scenario: host-pointer decode calls of one workgroup's size (dec_block: up to 32 KiB encoded, 8 KiB a turn)
def block_decode_calls(w, seed=73, wants=None, kinds=None):
    rng = np.random.default_rng(seed)

    def both(coders, stream, caps, first=None):
        for cap in caps:
            oo, op = np.full(cap + 8, SENTINEL, np.uint8), np.full(cap + 8, SENTINEL, np.uint8)
            ddo, ddp = w.oracle.new_decoder(coders[0]), w.product.new_decoder(coders[1])
            if first:  # a first call that leaves bits in the decoder's window: the next one starts inside a byte
                r = paired_decode(w, ddo, ddp, stream, 0, first, oo, op, 0, min(cap, 2))
                paired_decode(w, ddo, ddp, stream, r.consumed, stream.size, oo, op, r.produced, cap)
            else:
                paired_decode(w, ddo, ddp, stream, 0, stream.size, oo, op, 0, cap)

    plain = (w.ocoder, w.pcoder)
    # whole valid streams around the road's limits: 129 and 8192 encoded bytes, a lane's 8 bytes, a wave's 512
    for kind in kinds or KINDS:
        for want in wants or (129, 130, 136, 137, 511, 512, 513, 520, 1000, 4096, 8184, 8191, 8192, 8193, 8200, 12000, 16384,
                              16385, 32760, 32768, 32769):
            n = want  # symbols; trimmed until the stream has the wanted length
            data = inputs(rng, 2 * want, kind)
            lo, hi = 0, data.size
            while lo < hi:  # the most symbols whose stream is at most `want` bytes
                mid = (lo + hi + 1) // 2
                if (w.oracle.encoded_length(w.oracle.new_encoder(w.ocoder), data[:mid]) + 7) // 8 <= want:
                    lo = mid
                else:
                    hi = mid - 1
            n = lo
            stream = oracle_encode(w, data[:n])
            both(plain, stream, (n, n - 1, n // 2, n + 9, 0))
            both(plain, stream, (n,), first=3)
            both(plain, stream[: stream.size - 1], (n,))  # the last code cut off (or the padding missing)
    # any bytes at all; bytes that never fall into step (one value); no code at all for 30 bits of ones in the middle
    for size in (200, 777, 2048, 5000, 8192, 8200, 20000, 32768):
        both(plain, rng.integers(0, 256, size, dtype=np.uint8), (2 * size, 17))
        for v in (0x00, 0xFF, 0x55, 0x9C):
            both(plain, np.full(size, v, np.uint8), (2 * size, size // 3))
    good = oracle_encode(w, inputs(rng, 3000, "uniform"))
    for at in (0, 8, 63, 64, 1000, good.size - 9, good.size - 5, good.size - 4):
        bad = good.copy()
        bad[at : at + 4] = 0xFF
        both(plain, bad, (3000, 40))
    good = oracle_encode(w, inputs(rng, 20000, "uniform"))  # three turns of the workgroup
    for at in (8188, 8192, 8196, 16380, 16384, good.size - 6):
        bad = good.copy()
        bad[at : at + 4] = 0xFF
        both(plain, bad, (20000, 7000))
    both(plain, good, (20000, 6913, 6914, 13000), first=5)
    # a coder with holes: windows without a code stop the walk where the reference's does
    holes = (w.ocoder_holes, w.pcoder_holes)
    for size in (300, 3000, 8000):
        both(holes, rng.integers(0, 256, size, dtype=np.uint8), (2 * size, 5))


# ----------------------------------------------------------------------------- scenario: long streams with damage / short output (several scan runs)
def damaged_long_streams(w, seed=19, big=10_000_000):
    """Streams of several hundred chunks (more than one scan run of 256 chunks): damage inside a
    sub-chunk, at chunk and run boundaries, and output capacities that end in either run."""
    rng = np.random.default_rng(seed)
    plain = inputs(rng, big, "uniform")
    good = oracle_encode(w, plain)
    run_bytes = 256 * 32768
    assert good.size > run_bytes + 4 * 32768
    cases = []
    for at in (5 * 32768 + 777, run_bytes - 1, run_bytes, run_bytes + 32768 * 3 + 129, good.size - 40000):
        bad = good.copy()
        bad[at] ^= 0xFF
        bad[at + 1] = 0x00
        bad[at + 2] = 0x00
        cases.append((bad, 2 * big))
    for cap in (big, big - 1, 7_000_000, 8_000_001, 3):
        cases.append((good, cap))
    cases.append((good[: run_bytes + 5], 2 * big))
    for data, out_cap in cases:
        n = data.size
        oo, op = np.full(out_cap + 8, SENTINEL, np.uint8), np.full(out_cap + 8, SENTINEL, np.uint8)
        ddo, ddp = w.oracle.new_decoder(w.ocoder), w.product.new_decoder(w.pcoder)
        paired_decode(w, ddo, ddp, data, 0, n, oo, op, 0, out_cap)


# ----------------------------------------------------------------------------- scenario: other coders (every kernel variant, not only the reference's test table)
def canonical_code(lengths):
    """patterns[256], lens[256] of the canonical prefix code with these code lengths (Kraft sum <= 1)."""
    assert sum(2.0 ** -l for l in lengths if l) <= 1.0 + 1e-12
    order = sorted((l, s) for s, l in enumerate(lengths) if l)
    patterns = [0] * 256
    code, prev = 0, order[0][0]
    for l, s in order:
        code <<= l - prev
        patterns[s] = code
        code += 1
        prev = l
    return patterns, list(lengths)


CODER_PROFILES = {
    # name: [(how many symbols, code length), ...] in symbol order
    "len4to12": [(8, 4), (16, 6), (32, 8), (64, 10), (136, 12)],      # wave packer with 4-word octs; decode tables of 12 bits
    "len4to15": [(4, 4), (8, 5), (16, 7), (32, 9), (64, 12), (132, 15)],  # wave packer with 5-word octs; decode through linked tables
    "len1to16": [(1, 1), (1, 2), (2, 4), (4, 7), (8, 9), (240, 16)],      # codes shorter than 4 bits: the streaming packer
    "len2to12": [(2, 2), (4, 4), (8, 6), (16, 8), (226, 12)],            # short codes and decode: > 255 symbols a sub-chunk
    "len2to30": [(2, 2), (4, 5), (10, 8), (240, 30)],                    # long codes: the per-symbol packer
    "len8": [(256, 8)],                                                  # fixed length, decode table of 8 bits: dec_fixed
    "len9": [(256, 9)],                                                  # fixed length, half the windows without a code
    # the code lengths of RFC 7541 appendix B (HPACK, the reference's one production coder, in aws-c-http) less its
    # 30-bit EOS: an incomplete code, decode through linked tables
    "hpack_lengths": [(10, 5), (26, 6), (32, 7), (6, 8), (5, 10), (3, 11), (2, 12), (6, 13), (2, 14), (3, 15), (3, 19),
                      (8, 20), (13, 21), (26, 22), (29, 23), (12, 24), (4, 25), (15, 26), (19, 27), (29, 28), (3, 30)],
}


def profile_coders(w, name):
    """(oracle coder, product coder, code lengths) of one of CODER_PROFILES"""
    import ctypes as C

    lengths = [l for count, l in CODER_PROFILES[name] for _ in range(count)]
    assert len(lengths) == 256
    patterns, lens = canonical_code(lengths)
    pat_arr = (C.c_uint32 * 256)(*patterns)
    len_arr = (C.c_uint8 * 256)(*lens)
    oc = w.oracle.lib.oracle_table_coder_new(pat_arr, len_arr)
    pc = w.product.lib.aws_huffman_amd_table_coder_new(pat_arr, len_arr)
    assert oc and pc
    return oc, pc, lengths


def other_coders(w, n=60000, seed=23):
    rng = np.random.default_rng(seed)
    for name in CODER_PROFILES:
        oc, pc, lengths = profile_coders(w, name)
        prob = np.array([2.0 ** -l for l in lengths])
        prob /= prob.sum()
        for kind in ("matched", "uniform"):
            data = (rng.choice(256, size=n, p=prob) if kind == "matched" else rng.integers(0, 256, n)).astype(np.uint8)
            want = w.oracle.encode_all(oc, data, slack=64 + n)
            got = w.product.encode_all(pc, data, slack=64 + n)
            assert np.array_equal(got, want), "encode differs for coder %s on %s data" % (name, kind)
            # short output, then the rest: the capacity edge inside a whole segment
            eo, ep = w.oracle.new_encoder(oc), w.product.new_encoder(pc)
            do, dp = np.full(want.size + 8, SENTINEL, np.uint8), np.full(want.size + 8, SENTINEL, np.uint8)
            r = paired_encode(w, eo, ep, data, 0, do, dp, 0, want.size // 2)
            paired_encode(w, eo, ep, data, r.consumed, do, dp, r.produced, want.size)
            # (codes of more than 12 bits: one thread walks the whole item through linked tables)
            ddo, ddp = w.oracle.new_decoder(oc), w.product.new_decoder(pc)
            oo, op = np.full(n + 8, SENTINEL, np.uint8), np.full(n + 8, SENTINEL, np.uint8)
            paired_decode(w, ddo, ddp, want, 0, want.size, oo, op, 0, n)
            assert np.array_equal(op[:n], data)
            # damage: the same answer as the oracle, whatever it is
            bad = want.copy()
            bad[bad.size // 3] ^= 0x5A
            ddo, ddp = w.oracle.new_decoder(oc), w.product.new_decoder(pc)
            oo, op = np.full(2 * n + 8, SENTINEL, np.uint8), np.full(2 * n + 8, SENTINEL, np.uint8)
            paired_decode(w, ddo, ddp, bad, 0, bad.size, oo, op, 0, 2 * n)
            # output room for a third of the symbols, then the rest
            ddo, ddp = w.oracle.new_decoder(oc), w.product.new_decoder(pc)
            oo, op = np.full(n + 8, SENTINEL, np.uint8), np.full(n + 8, SENTINEL, np.uint8)
            r = paired_decode(w, ddo, ddp, want, 0, want.size, oo, op, 0, n // 3)
            paired_decode(w, ddo, ddp, want, r.consumed, want.size, oo, op, r.produced, n)
            assert np.array_equal(op[:n], data)
            # the stream cut short in a few places (long codes: around the lanes and blocks of dec_deep)
            for cut in (513, 640, 32768, 32769, 32768 + 127, want.size // 2, want.size - 1):
                if cut < want.size:
                    ddo, ddp = w.oracle.new_decoder(oc), w.product.new_decoder(pc)
                    oo, op = np.full(n + 8, SENTINEL, np.uint8), np.full(n + 8, SENTINEL, np.uint8)
                    paired_decode(w, ddo, ddp, want, 0, cut, oo, op, 0, n)


def long_streams_of_other_coders(w, names=("len4to12", "len2to12"), n=22_000_000, seed=29):
    """One long stream of a coder whose decode table has 12 bits (the 12-bit builds of dec_sync_one / dec_emit_fast, twelve
    entry states in dec_scan's tables), long enough for several scan runs: whole, damaged in its middle, cut, short of
    room -- every record and byte as the oracle has them."""
    rng = np.random.default_rng(seed)
    for name in names:
        oc, pcoder, lengths = profile_coders(w, name)
        prob = np.array([2.0 ** -l for l in lengths])
        data = rng.choice(256, size=n, p=prob / prob.sum()).astype(np.uint8)
        enc = w.oracle.encode_all(oc, data, slack=64 + n)
        damaged = enc.copy()
        damaged[enc.size // 2 + 777:enc.size // 2 + 781] ^= 0xA5
        eng = harness.Engine(w.product.lib, pcoder)
        streams = [(enc, 0, n), (damaged, 0, n), (enc[: enc.size // 3 + 5], 0, n), (enc, 0, n // 2 + 3)]
        decode_items_like_the_oracle(w, eng, oc, streams, rng, "long stream of %s" % name, kinds=2)
        eng.close()


# ----------------------------------------------------------------------------- scenario: items sharded over several engines (one per GPU)
def sharded_items(w, devices=(0, 0, 0), n_items=23, seed=71, item_len=16384):
    """huffman_amd.h "several GPUs": item i runs on shard i mod G (the split of BASELINE configs[3]),
    every shard with its own engine, stream, host thread and device buffers; the records come back in item order
    and every item must equal the oracle's result for that item alone -- whole buffers, capacity-limited ones
    (SHORT_BUFFER with the reference's record), ragged ones, an empty one.  Then everything is decoded back the
    same way.  (The same device may be listed several times: that is how one GPU, or the emulator, runs this.)"""
    rng = np.random.default_rng(seed)
    G = len(devices)
    here = w.product.lib.aws_huffman_amd_current_device()  # (no call below may leave this thread on another device)
    sh = harness.Shards(w.product.lib, w.pcoder, list(devices))
    assert [w.product.lib.aws_huffman_amd_engine_device(sh.engines[g].h) for g in range(G)] == list(devices)
    assert w.product.lib.aws_huffman_amd_current_device() == here
    lens = [item_len, item_len, 0, 1, item_len + 17, 3 * item_len + 5, 15][:n_items]
    lens += [int(rng.integers(1, 2 * item_len)) for _ in range(n_items - len(lens))]
    blobs = [inputs(rng, n, KINDS[i % 3]) for i, n in enumerate(lens)]
    full = [oracle_encode(w, b) for b in blobs]
    caps = [[f.size + 9, f.size, max(f.size - 1, 0), f.size // 2][i % 4] for i, f in enumerate(full)]
    # per shard: its items' inputs back to back, outputs with guard gaps
    in_off, out_off = [0] * n_items, [0] * n_items
    in_pos, out_pos = [3] * G, [5] * G
    for i in range(n_items):
        g = i % G
        in_off[i], out_off[i] = in_pos[g], out_pos[g]
        in_pos[g] += blobs[i].size + int(rng.integers(0, 5))
        out_pos[g] += caps[i] + 24
    bases = []
    for g in range(G):
        eng = sh.engines[g]
        host_in = np.zeros(in_pos[g] + 64, np.uint8)
        for i in range(g, n_items, G):
            host_in[in_off[i]:in_off[i] + blobs[i].size] = blobs[i]
        d_in, d_out = eng.alloc(host_in.size), eng.alloc(out_pos[g] + 64)
        eng.upload(d_in, host_in)
        eng.fill(d_out, SENTINEL, out_pos[g] + 64)
        bases.append((d_in, d_out))
    items = [dict(in_offset=in_off[i], in_len=blobs[i].size, out_offset=out_off[i], out_capacity=caps[i]) for i in range(n_items)]
    res = sh.encode(items, bases)
    assert sh.encode(items, bases) == res  # (the same items again: the shards' plans are kept, nothing is built or uploaded)
    assert w.product.lib.aws_huffman_amd_current_device() == here
    outs = [sh.engines[g].download(bases[g][1], out_pos[g] + 64) for g in range(G)]
    for i in range(n_items):
        eo = w.oracle.new_encoder(w.ocoder)
        want = np.full(caps[i] + 8, SENTINEL, np.uint8)
        r = w.oracle.encode_call(eo, blobs[i], 0, want, 0, caps[i])
        assert res[i] == (r.rc, r.err, r.consumed, r.produced, r.state[0], r.state[1]), (i, res[i], r)
        got = outs[i % G][out_off[i]:out_off[i] + caps[i] + 8]
        assert np.array_equal(got[:r.produced], want[:r.produced]), "item %d bytes" % i
        assert np.all(got[r.produced:caps[i] + 8] == SENTINEL), "item %d wrote past what the call produced" % i
    # decode the complete streams back, sharded the same way
    dbases, dn = [], [0] * G
    d_items = []
    for g in range(G):
        eng = sh.engines[g]
        pos_in, pos_out = 0, 0
        blob_in = []
        for i in range(g, n_items, G):
            d_items.append((i, dict(in_offset=pos_in, in_len=full[i].size, out_offset=pos_out, out_capacity=blobs[i].size)))
            blob_in.append(full[i])
            pos_in += full[i].size
            pos_out += blobs[i].size
        host = np.concatenate(blob_in + [np.zeros(64, np.uint8)])
        d_in, d_out = eng.alloc(host.size), eng.alloc(pos_out + 64)
        eng.upload(d_in, host)
        dbases.append((d_in, d_out))
        dn[g] = pos_out
    d_items.sort(key=lambda t: t[0])
    dres = sh.decode([it for _, it in d_items], dbases)
    assert sh.decode([it for _, it in d_items], dbases) == dres
    # other items through the kept plans: the first half of every stream only (a code may be cut: what matters is the oracle's record)
    halves = [dict(it, in_len=it["in_len"] // 2) for _, it in d_items]
    hres = sh.decode(halves, dbases)
    for (i, it), r in zip(d_items, hres):
        ro, _ = w.oracle.decode_all(w.ocoder, full[i][:it["in_len"] // 2], blobs[i].size)
        assert r[:3] == (ro.rc, ro.err, ro.produced), (i, r, ro)
    dres = sh.decode([it for _, it in d_items], dbases)
    assert w.product.lib.aws_huffman_amd_current_device() == here
    backs = [sh.engines[g].download(dbases[g][1], dn[g] + 64) for g in range(G)]
    for i, it in d_items:
        assert dres[i][0] == 0 and dres[i][2] == blobs[i].size, (i, dres[i])
        assert np.array_equal(backs[i % G][it["out_offset"]:it["out_offset"] + blobs[i].size], blobs[i]), "item %d round trip" % i
    for g in range(G):
        for ptr in bases[g] + dbases[g]:
            sh.engines[g].free(ptr)
    sh.close()


# ----------------------------------------------------------------------------- scenario: a coder destroyed and another one made (often at the same address)
def recreated_coders(w, rounds=6, n=3000, seed=61):
    """The product finds its device tables again through the coder's address.  An address says nothing about the
    table behind it: destroy a coder, make one with another table -- malloc likes to hand the same block out
    again -- and the next call must encode with the NEW table (checked against the oracle's coder of that table).
    Also: the library's clean-up in between, which drops every cached engine."""
    import ctypes as C

    rng = np.random.default_rng(seed)
    names = ["len4to12", "len8", "len4to15", "len2to12"]
    seen = set()
    previous = None
    for k in range(rounds):
        name = names[k % len(names)]
        lengths = [l for count, l in CODER_PROFILES[name] for _ in range(count)]
        patterns, lens = canonical_code(lengths)
        if k % 2:
            # a permutation of the same lengths: same callbacks, same sizes, different codes per symbol
            perm = rng.permutation(256)
            patterns, lens = [patterns[i] for i in perm], [lens[i] for i in perm]
        pat_arr, len_arr = (C.c_uint32 * 256)(*patterns), (C.c_uint8 * 256)(*lens)
        if previous is not None:
            w.product.lib.aws_huffman_amd_table_coder_destroy(previous)  # ... and at once a new one of the same size
        pc = w.product.lib.aws_huffman_amd_table_coder_new(pat_arr, len_arr)
        previous = pc
        oc = w.oracle.lib.oracle_table_coder_new(pat_arr, len_arr)
        seen.add(C.addressof(pc.contents))
        data = rng.integers(0, 256, n).astype(np.uint8)
        want = w.oracle.encode_all(oc, data, slack=64 + 3 * n)
        got = w.product.encode_all(pc, data, slack=64 + 3 * n)
        assert np.array_equal(got, want), "round %d (%s): encoded with a stale table" % (k, name)
        r, back = w.product.decode_all(pc, want, n)
        assert r.rc == 0 and np.array_equal(back, data), "round %d (%s): decoded with a stale table" % (k, name)
        if k == rounds // 2:
            w.product.lib.aws_compression_library_clean_up()
            got = w.product.encode_all(pc, data, slack=64 + 3 * n)  # engines are made again on demand
            assert np.array_equal(got, want)
        w.oracle.lib.oracle_table_coder_destroy(oc)
    w.product.lib.aws_huffman_amd_table_coder_destroy(previous)
    return len(seen)  # (how many distinct addresses the coders had: fewer than `rounds` means addresses came back)


# ----------------------------------------------------------------------------- scenario: streams cut at every kind of place (end-of-stream handling of the chunked decoder)
def cut_streams(w, seed=29, chunks=(1, 2, 5), step=7, span=140, n=200_000):
    """A valid stream cut short at byte positions all around chunk and sub-chunk boundaries: the decoder must
    stop where the oracle stops (END / incomplete code / padding-like garbage), with every output capacity."""
    rng = np.random.default_rng(seed)
    plain = inputs(rng, n, "uniform")
    good = oracle_encode(w, plain)
    cuts = set()
    for k in chunks:
        base = k * 32768
        if base + span >= good.size:
            continue
        cuts.update(range(base - span, base + span + 1, step))
        cuts.update((base - 129, base - 128, base - 127, base - 9, base - 8, base - 7, base - 1, base, base + 1,
                     base + 7, base + 8, base + 9, base + 127, base + 128, base + 129, base + 135, base + 136, base + 137))
    cuts.update((135, 136, 137, 263, 264, 265, 1000, 4095, 4096, 4097))
    for cut in sorted(c for c in cuts if 0 < c < good.size):
        data = good[:cut]
        for out_cap in (n, max(int(cut / 1.3), 1)):
            oo, op = np.full(n + 8, SENTINEL, np.uint8), np.full(n + 8, SENTINEL, np.uint8)
            ddo, ddp = w.oracle.new_decoder(w.ocoder), w.product.new_decoder(w.pcoder)
            paired_decode(w, ddo, ddp, data, 0, cut, oo, op, 0, out_cap)


# ----------------------------------------------------------------------------- scenario: streams of short codes (more symbols in a chunk than the emit stage holds)
def dense_symbols(w, n=200_000, seed=31):
    rng = np.random.default_rng(seed)
    lens = np.array([w.table[1][i] for i in range(256)])
    short = np.flatnonzero(lens == lens[lens > 0].min())  # the symbols with the shortest codes
    cases = [np.full(n, short[0], np.uint8), short[rng.integers(0, short.size, n)].astype(np.uint8)]
    mixed = short[rng.integers(0, short.size, n)].astype(np.uint8)
    mixed[rng.integers(0, n, n // 50)] = rng.integers(0, 256, n // 50)  # a few long codes in between
    cases.append(mixed)
    for data in cases:
        enc = oracle_encode(w, data)
        got = w.product.encode_all(w.pcoder, data)
        assert np.array_equal(got, enc)
        for out_cap in (n, n - 1, n // 3):
            oo, op = np.full(n + 8, SENTINEL, np.uint8), np.full(n + 8, SENTINEL, np.uint8)
            ddo, ddp = w.oracle.new_decoder(w.ocoder), w.product.new_decoder(w.pcoder)
            paired_decode(w, ddo, ddp, enc, 0, enc.size, oo, op, 0, out_cap)


def streams_out_of_step(w, n=260_000, seed=109, modes=(None, "long-way")):
    """Streams whose walks from different entry bits never become one: one symbol over and over, two symbols of one
    length taking turns, a short pattern repeated -- every chunk of them is one dec_sync_one gives up after a few rows;
    dec_sync_few / dec_sync_true take those inside the stream (a few walks a lane, then the true walk's records for the
    fast emit kernels), "long-way" sends them through dec_sync / dec_emit as before round 4.  Whole, cut, damaged in
    the middle (a true walk that stops inside a chunk), entered inside a byte, short of room, and behind a stretch of
    ordinary symbols (chunks of both kinds in one item)."""
    rng = np.random.default_rng(seed)
    lens = np.array([w.table[1][i] for i in range(256)])
    short = np.flatnonzero(lens == lens[lens > 0].min())
    mid = np.flatnonzero(lens == np.sort(np.unique(lens[lens > 0]))[2])
    longest = np.flatnonzero(lens == lens.max())
    pattern = np.array([short[0], short[1], short[0], short[2]], np.uint8)
    bases = [
        np.full(n, short[0], np.uint8),
        np.tile(np.array([short[1], short[2]], np.uint8), n // 2),
        np.tile(pattern, n // 4),
        np.full(n // 2, longest[0], np.uint8),
        np.tile(np.array([mid[0], mid[1 % mid.size]], np.uint8), n // 2),
        np.concatenate([inputs(rng, n // 3, "uniform"), np.full(n // 2, short[3 % short.size], np.uint8), inputs(rng, n // 4, "printable")]),
    ]
    eng = harness.Engine(w.product.lib, w.pcoder)
    for b, data in enumerate(bases):
        enc = oracle_encode(w, data)
        assert enc.size > 3 * 32768, enc.size
        damaged = enc.copy()
        at = enc.size // 2 + 1000
        damaged[at:at + 4] = 0xFF
        streams = [(enc, 0, data.size), (enc[: 2 * 32768 + 4000], 0, data.size), (damaged, 0, data.size),
                   (enc[5:], 3, data.size), (enc, 0, data.size // 3), (enc[: 3 * 32768], 0, data.size + 9)]
        decode_items_like_the_oracle(w, eng, w.ocoder, streams, rng, "out of step %d" % b, modes=modes, kinds=1)
    eng.close()


def few_ends_among_many_chunks(w, n=330_000, seed=157, engine=None, modes=(None, "tails-apart")):
    """Long items (nine chunks and more each): the FEW chunks such a plan's streams end in are workgroups of the big kernels'
    own grids, the stream's last symbols followed by the workgroup itself (dec_sync_one_mixed_kernel) -- "tails-apart": kernels
    of their own, as for a plan of many such chunks.  Ends of every kind: fewer than 8 bytes in the last chunk, 8 .. 135 (one
    thread's work), a lane or two more, nearly a whole chunk, exactly a whole chunk; cut inside a code, damaged in the last
    bytes, damaged in front of them, arbitrary bytes at the end, short of room, entered inside a byte."""
    rng = np.random.default_rng(seed)
    eng = engine or harness.Engine(w.product.lib, w.pcoder)
    data = inputs(rng, n, "uniform")
    enc = oracle_encode(w, data)
    chunk = 32768
    k = 9
    assert enc.size > (k + 1) * chunk + 6000, enc.size
    batches = []
    ends = [0, 1, 7, 8, 9, 100, 135, 136, 137, 263, 264, 300, 5000, chunk - 129, chunk - 1]
    streams = [(enc[:k * chunk + e].copy(), 0, n) for e in ends]
    streams.append((enc.copy(), 0, n))  # (whole: ends with its padding)
    batches.append(("cuts", streams, 1))
    streams = []
    for e in (50, 135, 200, 700, 9000):
        cut = enc[:k * chunk + e].copy()
        for back in (3, 40, 140, 400):  # (damage in the stream's last bytes, in the lane in front of them, further in front)
            if back < cut.size:
                bad = cut.copy()
                bad[-back:-back + 2 if back > 2 else None] = 0xFF
                streams.append((bad, 0, n))
        noise = cut.copy()
        noise[-min(e, 120):] = rng.integers(0, 256, min(e, 120), dtype=np.uint8)
        streams.append((noise, 0, n))
    batches.append(("damage", streams, 2))
    whole_syms = n
    streams = [(enc[:k * chunk + 200].copy(), 0, cap) for cap in (n // 2, 280_000, 290_000, 5)]
    streams += [(enc[5:k * chunk + 5 + e].copy(), 3, n) for e in (60, 135, 2000)]
    streams += [(enc.copy(), 0, whole_syms - 1), (enc.copy(), 0, whole_syms)]
    batches.append(("room and first bits", streams, 2))
    # (such a plan says so: three streams of ten chunks -- three ends among thirty chunks; one of two chunks: not a few among many)
    probe = eng.decode_plan([dict(in_offset=i * 11 * chunk, in_len=k * chunk + 200, out_offset=i * n, out_capacity=n) for i in range(3)])
    st = eng.decode_stats(probe)
    assert st["end_pieces_folded"] == 3 and st["end_pieces_single"] == 0 and st["end_pieces_packed"] == 0, st
    eng.lib.aws_huffman_amd_decode_plan_destroy(probe)
    probe = eng.decode_plan([dict(in_offset=0, in_len=chunk + 200, out_offset=0, out_capacity=n)])
    st = eng.decode_stats(probe)
    assert st["end_pieces_folded"] == 0 and st["end_pieces_single"] == 1, st
    eng.lib.aws_huffman_amd_decode_plan_destroy(probe)
    for label, streams, kinds in batches:
        decode_items_like_the_oracle(w, eng, w.ocoder, streams, rng, label, modes=modes, kinds=kinds)
    if engine is None:
        eng.close()


def quiet_plans(w, n=300_000, seed=151, engine=None):
    """A plan whose last fetched launch listed no chunk for any kernel but the regular ones is QUIET: its launches go without
    dec_sync_guess / _few / _true / dec_emit_big.  The same plan over OTHER bytes of the same lengths -- a stream whose walks
    never fall into step, chunks of more symbols than the emit stage holds, damage, arbitrary bytes -- sends what it lists the
    long way: the oracle's records and bytes all the same, the fetch says that chunks were listed, and the next launch has
    the kernels back (and "all-kernels" has them in every launch)."""
    rng = np.random.default_rng(seed)
    eng = engine or harness.Engine(w.product.lib, w.pcoder)
    lens = np.array([w.table[1][i] for i in range(256)])
    short = np.flatnonzero(lens == lens[lens > 0].min())
    plain = [oracle_encode(w, inputs(rng, m, "uniform")) for m in (n, n // 3, 40_000)]
    sizes = [e.size for e in plain]
    caps = [e.size * 8 // int(lens[lens > 0].min()) + 9 for e in plain]  # (room for the densest stream of that many bytes)

    def of_length(data, size):
        enc = oracle_encode(w, data)
        assert enc.size >= size
        return enc[:size].copy()

    def variant(kind):
        out = [e.copy() for e in plain]
        if kind == "one symbol":  # (every chunk of item 0 listed: its walks never fall into step)
            out[0] = of_length(np.full(sizes[0] * 8 // 5 + 8, short[0], np.uint8), sizes[0])
        elif kind == "short codes":  # (more symbols in a chunk than the emit stage holds)
            out[1] = of_length(short[rng.integers(0, short.size, sizes[1] * 8 // 5 + 8)].astype(np.uint8), sizes[1])
        elif kind == "damage":
            out[0][sizes[0] // 2:sizes[0] // 2 + 4] = 0xFF
        elif kind == "bytes":
            out[2] = rng.integers(0, 256, sizes[2], dtype=np.uint8)
        return out

    offs, pos = [], 5
    for e in plain:
        offs.append(pos)
        pos += e.size + int(rng.integers(0, 9))
    enc_total = pos + 64
    items, pos = [], 3
    for o, e, cap in zip(offs, plain, caps):
        items.append(dict(in_offset=o, in_len=e.size, first_bit=0, out_offset=pos, out_capacity=cap))
        pos += cap + int(rng.integers(1, 9))
    sym_total = pos + 64
    d_enc, d_sym = eng.alloc(enc_total), eng.alloc(sym_total)
    plan = eng.decode_plan(items)
    assert eng.decode_stats(plan)["by_pieces"] == len(items)

    def launch_and_check(streams, label):
        host = np.zeros(enc_total, np.uint8)
        want = np.full(sym_total, SENTINEL, np.uint8)
        keys = []
        for it, e in zip(items, streams):
            host[it["in_offset"]:it["in_offset"] + e.size] = e
            d = w.oracle.new_decoder(w.ocoder)
            dst = np.full(it["out_capacity"] + 1, SENTINEL, np.uint8)
            r = w.oracle.decode_call(d, e, 0, e.size, dst, 0, it["out_capacity"])
            keys.append((r.rc, r.err, r.produced, r.consumed * 8 - r.state[0]))
            want[it["out_offset"]:it["out_offset"] + it["out_capacity"]] = dst[:it["out_capacity"]]
        eng.upload(d_enc, host)
        eng.fill(d_sym, SENTINEL, sym_total)
        eng.decode_launch(plan, d_enc, d_sym)
        res = eng.decode_results(plan, len(items))
        assert res == keys, (label, res, keys)
        assert np.array_equal(eng.download(d_sym, sym_total), want), label

    assert not eng.decode_plan_is_quiet(plan)  # (nothing fetched yet)
    launch_and_check(plain, "plain, first launch")
    assert eng.decode_plan_is_quiet(plan)
    launch_and_check(plain, "plain, quiet")
    for kind in ("one symbol", "short codes", "damage", "bytes"):
        assert eng.decode_plan_is_quiet(plan), kind
        streams = variant(kind)
        launch_and_check(streams, kind + ", by a quiet plan")  # (what is listed goes the long way)
        assert not eng.decode_plan_is_quiet(plan), kind  # (chunks were listed)
        launch_and_check(streams, kind + ", with the kernels back")
        assert not eng.decode_plan_is_quiet(plan), kind
        launch_and_check(plain, "plain after " + kind)
        assert eng.decode_plan_is_quiet(plan), kind
        with harness.decode_road(eng.lib, "all-kernels"):
            launch_and_check(streams, kind + ", quiet plan, every kernel queued")
        launch_and_check(plain, "plain again after " + kind)
    # a reset makes the plan forget
    assert eng.decode_plan_is_quiet(plan)
    arr = eng._decode_item_array(items[:2])
    eng.lib.aws_huffman_amd_decode_plan_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    assert eng.lib.aws_huffman_amd_decode_plan_reset(plan, arr, 2) == 0
    assert not eng.decode_plan_is_quiet(plan)
    eng.lib.aws_huffman_amd_decode_plan_destroy(plan)
    eng.free(d_enc)
    eng.free(d_sym)
    if engine is None:
        eng.close()


def walks_that_never_meet(w, seed=137, engine=None, runs=(130, 260, 420, 900), modes=(None,)):
    """Ordinary streams with a stretch of ONE long-code symbol in them whose code, rotated, is a code of the same length
    again (the test coder's symbols 255, 254, 252 ...: seven of nine rotations): a lane whose sub-chunk lies inside the
    stretch starts its guessed walk on a wrong phase and stays on it to the sub-chunk's end -- the walk from its true entry
    never stands where the guessed one stood.  A handful of such lanes in a chunk (fewer than the wave that gives a chunk
    up as out of step): dec_sync_one keeps the true walk's own records for them, lets them leave as the true walk does
    and has the lane behind walk again.  Inside long streams and in the chunk a stream ends in; whole, cut inside the
    stretch, damaged, entered inside a byte, short of room."""
    rng = np.random.default_rng(seed)
    patterns, lens = w.table
    longest = max(int(x) for x in lens)

    def rotations_that_are_codes(sym):
        code = int(patterns[sym])
        n = 0
        for r in range(1, longest):
            rot = ((code << r) | (code >> (longest - r))) & ((1 << longest) - 1)
            n += any(int(lens[s]) == longest and int(patterns[s]) == rot for s in range(256))
        return n

    stuck = sorted((s for s in range(256) if int(lens[s]) == longest), key=rotations_that_are_codes, reverse=True)[:4]
    assert rotations_that_are_codes(stuck[0]) >= 3, "no symbol of this coder keeps a walk on a wrong phase"
    eng = engine or harness.Engine(w.product.lib, w.pcoder)
    for b, run in enumerate(runs):
        sym = stuck[b % len(stuck)]
        for total in (90_000, 21_000, 9_000):  # chunks inside a stream; a wide end-of-stream chunk; a narrow one
            front = int(rng.integers(300, total // 2))
            data = np.concatenate([inputs(rng, front, "uniform"), np.full(run, sym, np.uint8),
                                   inputs(rng, 700, "uniform"), np.full(run // 2 + 60, stuck[(b + 1) % len(stuck)], np.uint8),
                                   inputs(rng, max(total - front - run - 700, 100), "uniform")])
            enc = oracle_encode(w, data)
            inside = (front + run // 2) * enc.size // data.size  # (about the middle of the first stretch)
            damaged = enc.copy()
            damaged[inside:inside + 3] ^= 0x5A
            streams = [(enc, 0, data.size), (enc[:inside], 0, data.size), (damaged, 0, data.size), (enc[3:], 5, data.size),
                       (enc, 0, front + run // 3), (enc, 0, data.size + 7)]
            decode_items_like_the_oracle(w, eng, w.ocoder, streams, rng, "never meet %d/%d" % (run, total), modes=modes, kinds=1)
    if engine is None:
        eng.close()


# ----------------------------------------------------------------------------- scenario: empty cursors with a NULL pointer
def null_empty_cursors(w, seed=91):
    """aws_byte_cursor{0, NULL} is a valid cursor: the reference never touches `ptr` when `len` is 0
    (source/huffman.c:149-167 flush of pending overflow bits, :196-211 refill, :107-129 length query)."""
    rng = np.random.default_rng(seed)
    for n in (0, 1, 9, 300, 5000, 40000):
        data = inputs(rng, n, "uniform")
        total = 2 * n + 32
        # encode: run out of room (pending overflow bits), then flush with {0, NULL} calls of growing capacity
        for first_cap in (0, 1, n // 2 + 1, total):
            do, dp = np.full(total, SENTINEL, np.uint8), np.full(total, SENTINEL, np.uint8)
            eo, ep = w.oracle.new_encoder(w.ocoder, eos_padding=0x3C), w.product.new_encoder(w.pcoder, eos_padding=0x3C)
            off = length = 0
            cap = min(first_cap, total)
            for _ in range(10000):
                r = paired_encode(w, eo, ep, data, off, do, dp, length, cap, null_when_empty=True)
                off += r.consumed
                length += r.produced
                if r.rc == 0:
                    break
                assert r.err == SHORT_BUFFER
                cap = min(cap + int(rng.choice([0, 1, 2, 500, 50000])), total)
            assert off == n
            # and once more on the finished encoder: nothing pending, nothing to encode
            paired_encode(w, eo, ep, data, n, do, dp, length, total, null_when_empty=True)
        # the length query on {0, NULL}, fresh and with bits pending
        eo, ep = w.oracle.new_encoder(w.ocoder), w.product.new_encoder(w.pcoder)
        assert w.product.encoded_length(ep, b"") == w.oracle.encoded_length(eo, b"")
        if n:
            do, dp = np.full(total, SENTINEL, np.uint8), np.full(total, SENTINEL, np.uint8)
            paired_encode(w, eo, ep, data, 0, do, dp, 0, min(1, total))
            assert w.product.encoded_length(ep, b"") == w.oracle.encoded_length(eo, b"")
        # decode: feed everything with too little room (bits stay in the decoder), then {0, NULL} calls with more room
        enc = oracle_encode(w, data)
        for first_cap in (0, 1, n // 3 + 1):
            oo, op = np.full(n + 8, SENTINEL, np.uint8), np.full(n + 8, SENTINEL, np.uint8)
            ddo, ddp = w.oracle.new_decoder(w.ocoder), w.product.new_decoder(w.pcoder)
            lo = length = 0
            cap = min(first_cap, n)
            for _ in range(20000):
                r = paired_decode(w, ddo, ddp, enc, lo, enc.size, oo, op, length, cap, null_when_empty=True)
                lo += r.consumed
                length += r.produced
                if r.rc == 0 and length == n:
                    break
                if r.rc != 0:
                    assert r.err == SHORT_BUFFER
                cap = min(cap + int(rng.choice([0, 1, 3, 700, 50000])), n)
            else:
                raise AssertionError("decode with empty cursors did not finish")
            assert np.array_equal(oo[:n], data)
            paired_decode(w, ddo, ddp, enc, enc.size, enc.size, oo, op, length, n + 8, null_when_empty=True)


# ----------------------------------------------------------------------------- scenario: plans made on the device
def many_header_sized_items(w, n_items=6000, seed=97, engine=None):
    """A batch in which EVERY item is one thread's work (header-sized strings, 1..90 symbols, none empty) and that has
    at least PLAN_ON_DEVICE_MIN_ITEMS (4096) of them: the plan is then made on the device from the caller's records as
    they are (csrc/host/engine.c, hufk_*_plan_tiny_items).  Encode -- some items with carried overflow bits, some with
    too little room -- and decode of the results, every record and every byte against the oracle's call for that item."""
    rng = np.random.default_rng(seed)
    eng = engine or harness.Engine(w.product.lib, w.pcoder)
    lens = rng.integers(1, 91, n_items)
    plains = [inputs(rng, int(n), KINDS[i % 3]) for i, n in enumerate(lens)]
    in_offs = np.concatenate([[3], 3 + np.cumsum(lens[:-1] + rng.integers(0, 3, n_items - 1))]).astype(np.int64)
    blob = np.full(int(in_offs[-1] + lens[-1]) + 64, 0xC3, np.uint8)
    for pl, o in zip(plains, in_offs):
        blob[o:o + pl.size] = pl
    caps = [int(2 * n + 8) if i % 7 else int(rng.integers(0, n + 1)) for i, n in enumerate(lens)]  # every seventh: short
    out_offs = np.concatenate([[1], 1 + np.cumsum(np.array(caps[:-1]) + 5)]).astype(np.int64)
    out_total = int(out_offs[-1] + caps[-1]) + 64
    carried = []  # (pattern, bits) as an encoder that ran out of room leaves them: nothing above the bits
    for i in range(n_items):
        bits = int(rng.integers(1, 10)) if i % 5 == 0 else 0
        carried.append((int(rng.integers(0, 1 << bits)) if bits else 0, bits))
    d_in, d_out = eng.alloc(blob.size), eng.alloc(out_total)
    eng.upload(d_in, blob)
    eng.fill(d_out, SENTINEL, out_total)
    plan = eng.encode_plan([dict(in_offset=int(in_offs[i]), in_len=int(lens[i]), out_offset=int(out_offs[i]), out_capacity=caps[i],
                                 overflow_in=carried[i], eos_padding=0x5F) for i in range(n_items)])
    eng.encode_launch(plan, d_in, d_out)
    got = eng.encode_results(plan, n_items)
    back = eng.download(d_out, out_total)
    enc_streams = []
    for i in range(n_items):
        eo = w.oracle.new_encoder(w.ocoder, eos_padding=0x5F)
        eo.overflow_bits.pattern, eo.overflow_bits.num_bits = carried[i]
        dst = np.full(caps[i] + 4, SENTINEL, np.uint8)
        r = w.oracle.encode_call(eo, plains[i], 0, dst, 0, caps[i])
        assert got[i][:4] == (r.rc, r.err, r.consumed, r.produced) and (got[i][4], got[i][5] if got[i][4] else 0) == r.state, (i, got[i], r)
        assert np.array_equal(back[out_offs[i]:out_offs[i] + caps[i] + 4], dst), i
        enc_streams.append(dst[:r.produced].copy())
    eng.lib.aws_huffman_amd_encode_plan_destroy(plan)
    # decode what was produced (whole streams, cut ones where the room ran out, empty ones left out: every item non-empty)
    keep = [i for i in range(n_items) if enc_streams[i].size]
    assert len(keep) >= 4096
    sym_caps = [int(lens[i]) if i % 3 else int(rng.integers(0, lens[i] + 1)) for i in keep]
    sym_offs = np.concatenate([[2], 2 + np.cumsum(np.array(sym_caps[:-1]) + 3)]).astype(np.int64)
    sym_total = int(sym_offs[-1] + sym_caps[-1]) + 64
    d_back = eng.alloc(sym_total)
    eng.fill(d_back, SENTINEL, sym_total)
    dplan = eng.decode_plan([dict(in_offset=int(out_offs[i]), in_len=int(enc_streams[i].size), out_offset=int(sym_offs[k]),
                                  out_capacity=sym_caps[k]) for k, i in enumerate(keep)])
    eng.decode_launch(dplan, d_out, d_back)
    dres = eng.decode_results(dplan, len(keep))
    dback = eng.download(d_back, sym_total)
    for k, i in enumerate(keep):
        dd = w.oracle.new_decoder(w.ocoder)
        want = np.full(sym_caps[k] + 3, SENTINEL, np.uint8)
        r = w.oracle.decode_call(dd, enc_streams[i], 0, enc_streams[i].size, want, 0, sym_caps[k])
        assert dres[k][:3] == (r.rc, r.err, r.produced), (k, i, dres[k], r)
        assert np.array_equal(dback[sym_offs[k]:sym_offs[k] + sym_caps[k] + 3], want), (k, i)
    eng.lib.aws_huffman_amd_decode_plan_destroy(dplan)
    for ptr in (d_in, d_out, d_back):
        eng.free(ptr)
    if engine is None:
        eng.close()


def encode_then_decode_on_the_device(w, seed=119, engine=None, batches=((40, 50), (9000, 90), (30000, 300))):
    """aws_huffman_amd_decode_plan_from_encode: a batch of header-sized strings encoded, then decoded back by a plan made
    on the device from the encode launch's records -- the encoded lengths never come to the host.  Some items run out of
    room (they decode to the symbols that fit), some are empty, some symbols have no code (holes coder: checked apart);
    every decode record and every byte as the oracle's call on that item's encoded bytes gives them.  A few items and
    thousands of them (the classes of the thread-per-item rule); a batch with a long item is refused."""
    rng = np.random.default_rng(seed)
    eng = engine or harness.Engine(w.product.lib, w.pcoder)
    for n_items, longest in batches:  # (40 items a byte of the largest capacity from 128 bytes on: HUFD_DEC_TINY_PER_BYTE)
        lens = rng.integers(0, longest + 1, n_items)
        lens[rng.integers(0, n_items, 3)] = 0
        plains = [inputs(rng, int(n), KINDS[i % 3]) for i, n in enumerate(lens)]
        in_offs = np.concatenate([[5], 5 + np.cumsum(lens[:-1] + rng.integers(0, 4, n_items - 1))]).astype(np.int64)
        in_total = int(in_offs[-1] + lens[-1]) + 64
        blob = np.full(in_total, 0xC3, np.uint8)
        for pl, o in zip(plains, in_offs):
            blob[o:o + pl.size] = pl
        caps = [min(int(2 * n + 4), 2 * longest) if i % 9 else int(rng.integers(0, n + 1)) for i, n in enumerate(lens)]  # every ninth: short
        out_offs = np.concatenate([[1], 1 + np.cumsum(np.array(caps[:-1]) + 2)]).astype(np.int64)
        out_total = int(out_offs[-1] + caps[-1]) + 64
        d_in, d_enc, d_back = eng.alloc(in_total), eng.alloc(out_total), eng.alloc(in_total)
        eng.upload(d_in, blob)
        eng.fill(d_enc, SENTINEL, out_total)
        eng.fill(d_back, SENTINEL, in_total)
        eplan = eng.encode_plan([dict(in_offset=int(in_offs[i]), in_len=int(lens[i]), out_offset=int(out_offs[i]), out_capacity=caps[i])
                                 for i in range(n_items)])
        dplan = eng.decode_plan([])
        eng.encode_launch(eplan, d_in, d_enc)
        assert eng.decode_plan_from_encode(dplan, eplan)  # (behind the launch on the engine's stream: no results fetched)
        eng.decode_launch(dplan, d_enc, d_back)
        dres = eng.decode_results(dplan, n_items)
        eres = eng.encode_results(eplan, n_items)
        enc_bytes, back = eng.download(d_enc, out_total), eng.download(d_back, in_total)
        want = np.full(in_total, SENTINEL, np.uint8)
        for i in range(n_items):
            stream = enc_bytes[out_offs[i]:out_offs[i] + eres[i][3]]
            dd = w.oracle.new_decoder(w.ocoder)
            sym = np.full(int(lens[i]) + 1, SENTINEL, np.uint8)
            r = w.oracle.decode_call(dd, stream, 0, stream.size, sym, 0, int(lens[i]))
            assert dres[i][:3] == (r.rc, r.err, r.produced), (n_items, i, dres[i], r, eres[i])
            want[in_offs[i]:in_offs[i] + lens[i]] = sym[:lens[i]]
            if eres[i][0] == 0:
                assert r.produced == lens[i] and np.array_equal(sym[:lens[i]], plains[i]), i  # the round trip itself
        assert np.array_equal(back, want)
        # the plan is an ordinary one again after a reset from records
        eng.lib.aws_huffman_amd_decode_plan_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        eng.lib.aws_huffman_amd_encode_plan_destroy(eplan)
        eng.lib.aws_huffman_amd_decode_plan_destroy(dplan)
        for ptr in (d_in, d_enc, d_back):
            eng.free(ptr)
    # a batch that is not one of short items: the general plan, made on the device from the launch's records all the same
    # (plans_made_on_the_device has the whole of it)
    d_in = eng.alloc(70000)
    both = inputs(rng, 70000, "uniform")
    eng.upload(d_in, both)
    eplan = eng.encode_plan([dict(in_offset=0, in_len=10, out_offset=0, out_capacity=40),
                             dict(in_offset=100, in_len=30000, out_offset=100, out_capacity=60000)])
    dplan = C.c_void_p()  # (a plan of no items, from no array at all)
    assert eng.lib.aws_huffman_amd_decode_plan_new(C.byref(dplan), eng.h, None, 0) == 0
    spare = C.c_void_p()
    assert eng.lib.aws_huffman_amd_encode_plan_new(C.byref(spare), eng.h, None, 0) == 0
    eng.lib.aws_huffman_amd_encode_plan_destroy(spare)
    # (an encode plan that was never launched has no records to take lengths from: INVALID_ARGUMENT, not a refusal)
    eng.lib.aws_huffman_amd_decode_plan_from_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    assert eng.lib.aws_huffman_amd_decode_plan_from_encode(dplan, eplan, None) != 0
    assert eng.lib.aws_last_error() == harness.AWS_ERROR_INVALID_ARGUMENT, eng.lib.aws_last_error()
    d_enc = eng.alloc(70000)
    eng.encode_launch(eplan, d_in, d_enc)
    assert eng.decode_plan_from_encode(dplan, eplan)
    assert eng.decode_stats(dplan)["by_pieces"] == 1 and eng.decode_stats(dplan)["by_thread"] == 1
    d_back = eng.alloc(70000)
    eng.fill(d_back, SENTINEL, 70000)
    eng.decode_launch(dplan, d_enc, d_back)
    assert [r[:3] for r in eng.decode_results(dplan, 2)] == [(0, 0, 10), (0, 0, 30000)]
    back = eng.download(d_back, 70000)
    assert np.array_equal(back[:10], both[:10]) and np.array_equal(back[100:30100], both[100:30100])
    assert np.all(back[10:100] == SENTINEL) and np.all(back[30100:] == SENTINEL)
    eng.lib.aws_huffman_amd_encode_plan_destroy(eplan)
    eng.lib.aws_huffman_amd_decode_plan_destroy(dplan)
    eng.free(d_in)
    eng.free(d_enc)
    eng.free(d_back)
    if engine is None:
        eng.close()


# ----------------------------------------------------------------------------- scenario: padding byte values
def eos_padding_values(w):
    rng = np.random.default_rng(16)
    for eos in (0x00, 0xFF, 0x55, 0x06, 0x80, 0x7F):
        for n in (1, 2, 3, 17, 1000):
            data = inputs(rng, n, "printable")
            do, dp = np.full(2 * n + 16, SENTINEL, np.uint8), np.full(2 * n + 16, SENTINEL, np.uint8)
            eo, ep = w.oracle.new_encoder(w.ocoder, eos_padding=eos), w.product.new_encoder(w.pcoder, eos_padding=eos)
            paired_encode(w, eo, ep, data, 0, do, dp, 0, do.size)
    for rec in PROBE["eos_padding_probe"]:
        plain = np.frombuffer(bytes.fromhex(rec["plain"]), dtype=np.uint8)
        assert w.product.encode_all(w.pcoder, plain, eos_padding=rec["eos_padding"]).tobytes().hex() == rec["encoded"]


# ----------------------------------------------------------------------------- scenario: pinned records of SURVEY.md 8c on the product
def survey_records_on_product(w, names=("G4K", "G16K", "G16KP")):
    for name in names:
        rec = PROBE["streams"][name]
        raw = harness.splitmix64_bytes(rec["seed"], rec["len"])
        plain = harness.printable_map(raw) if rec["map"] == "printable" else raw
        enc = w.product.encode_all(w.pcoder, plain)
        assert enc.size == rec["encoded_len"]
        assert hashlib.sha256(enc.tobytes()).hexdigest() == rec["sha256_encoded"]
        r, back = w.product.decode_all(w.pcoder, enc, plain.size)
        assert (r.rc, r.consumed, r.produced, r.state[0]) == (0, enc.size, plain.size, rec["decoder_tail_num_bits"])
        assert np.array_equal(back, plain)
    plain = harness.splitmix64_bytes(2, 16384)
    for rec in PROBE["G16K_partial_encode"]:
        e = w.product.new_encoder(w.pcoder)
        dst = np.zeros(20000, dtype=np.uint8)
        r = w.product.encode_call(e, plain, 0, dst, 0, rec["cap"])
        assert (r.rc, r.consumed, r.produced) == (rec["rc"], rec["consumed"], rec["out_len"]), (rec, r)
        if rec["rc"]:
            assert r.err == SHORT_BUFFER and r.state == (rec["overflow_num_bits"], rec["overflow_pattern"])
    t = PROBE["K1_encode_step1_trace"]
    e = w.product.new_encoder(w.pcoder)
    dst = np.zeros(12, dtype=np.uint8)
    off = length = 0
    for cap in range(1, 13):
        r = w.product.encode_call(e, K1_PLAIN, off, dst, length, cap)
        off += r.consumed
        length += r.produced
        assert off == t["consumed"][cap - 1] and r.state[0] == t["overflow_num_bits"][cap - 1]
    t = PROBE["K1_decode_partial_output"]
    d = w.product.new_decoder(w.pcoder)
    dst = np.zeros(16, dtype=np.uint8)
    off = length = 0
    for cap, want_off, want_bits in zip(t["caps"], t["input_consumed"], t["num_bits"]):
        r = w.product.decode_call(d, K1_ENC, off, K1_ENC.size, dst, length, cap)
        off += r.consumed
        length += r.produced
        assert off == want_off and r.state[0] == want_bits
    assert "%016x" % d.working_bits == t["final_working_bits"]


# ----------------------------------------------------------------------------- scenario: foreign callbacks
def foreign_coder_callbacks(w):
    """The product tabulates whatever callbacks it is given: hand it the ORACLE's coder object."""
    rng = np.random.default_rng(17)
    data = inputs(rng, 5000, "uniform")
    want = oracle_encode(w, data)
    got = w.product.encode_all(w.ocoder, data)  # oracle callbacks, product engine
    assert np.array_equal(got, want)
    r, back = w.product.decode_all(w.ocoder, want, data.size)
    assert r.rc == 0 and np.array_equal(back, data)


# ----------------------------------------------------------------------------- scenario: device-pointer batched API
def batched_device_api(w, n_items=9, seed=18, item_len=16384, engine=None):
    """huffman_amd.h: every item must equal the oracle's result for that item alone."""
    rng = np.random.default_rng(seed)
    eng = engine or harness.Engine(w.product.lib, w.pcoder)
    lens = [0, 1, 15, 16, 17, item_len - 1, item_len, item_len + 1, 3 * item_len + 5][:n_items]
    lens += [int(rng.integers(1, 2 * item_len)) for _ in range(n_items - len(lens))]
    blobs = [inputs(rng, n, KINDS[i % 3]) for i, n in enumerate(lens)]
    # inputs packed back to back at odd offsets; outputs with guard gaps
    in_offs, pos = [], 3
    for b in blobs:
        in_offs.append(pos)
        pos += b.size + int(rng.integers(0, 5))
    in_total = pos + 64
    full = [oracle_encode(w, b) for b in blobs]
    caps = []
    for i, f in enumerate(full):
        caps.append([f.size + 9, f.size, max(f.size - 1, 0), f.size // 2, 0][i % 5])
    out_offs, pos = [], 5
    for c in caps:
        out_offs.append(pos)
        pos += c + 24
    out_total = pos + 64
    host_in = np.zeros(in_total, np.uint8)
    for b, o in zip(blobs, in_offs):
        host_in[o:o + b.size] = b
    d_in, d_out = eng.alloc(in_total), eng.alloc(out_total)
    eng.upload(d_in, host_in)
    eng.fill(d_out, SENTINEL, out_total)
    items = [dict(in_offset=o, in_len=b.size, out_offset=oo, out_capacity=c, eos_padding=0xFF)
             for b, o, oo, c in zip(blobs, in_offs, out_offs, caps)]
    plan = eng.encode_plan(items)
    eng.encode_launch(plan, d_in, d_out)
    res = eng.encode_results(plan, len(items))
    got = eng.download(d_out, out_total)
    want = np.full(out_total, SENTINEL, np.uint8)
    carried = []  # the oracle's encoders of the items that ran out of room, for the resume pass below
    for i, (b, oo, c) in enumerate(zip(blobs, out_offs, caps)):
        e = w.oracle.new_encoder(w.ocoder)
        dst = np.full(c + 1, SENTINEL, np.uint8)
        r = w.oracle.encode_call(e, b, 0, dst, 0, c)
        want[oo:oo + c] = dst[:c]
        assert res[i] == (r.rc, r.err, r.consumed, r.produced, r.state[0], r.state[1]), (i, res[i], r)
        if r.rc != 0 and r.err == SHORT_BUFFER:
            carried.append((i, e, r))
    assert np.array_equal(got, want), "batched encode wrote outside its items or wrote wrong bytes"
    # the items that ran out of room, taken up where they stopped: the rest of the input (at whatever address that
    # is) with the bits that did not fit carried in, each into an output of its own
    if carried:
        rest_caps = [full[i].size - r.produced + [5, 0, 1][k % 3] for k, (i, e, r) in enumerate(carried)]
        rest_offs, pos = [], 11
        for c in rest_caps:
            rest_offs.append(pos)
            pos += c + 24
        rest_total = pos + 64
        d_rest = eng.alloc(rest_total)
        eng.fill(d_rest, SENTINEL, rest_total)
        ritems = [dict(in_offset=in_offs[i] + r.consumed, in_len=blobs[i].size - r.consumed, out_offset=ro, out_capacity=rc_,
                       overflow_in=(r.state[1], r.state[0]), eos_padding=0xFF)
                  for (i, e, r), ro, rc_ in zip(carried, rest_offs, rest_caps)]
        rplan = eng.encode_plan(ritems)
        eng.encode_launch(rplan, d_in, d_rest)
        rres = eng.encode_results(rplan, len(ritems))
        got = eng.download(d_rest, rest_total)
        want = np.full(rest_total, SENTINEL, np.uint8)
        for k, ((i, e, r), ro, rc_) in enumerate(zip(carried, rest_offs, rest_caps)):
            dst = np.full(rc_ + 1, SENTINEL, np.uint8)
            r2 = w.oracle.encode_call(e, blobs[i], r.consumed, dst, 0, rc_)
            want[ro:ro + rc_] = dst[:rc_]
            assert rres[k] == (r2.rc, r2.err, r2.consumed, r2.produced, r2.state[0], r2.state[1]), (k, rres[k], r2)
        assert np.array_equal(got, want), "resumed batched encode wrote outside its items or wrote wrong bytes"
        eng.lib.aws_huffman_amd_encode_plan_destroy(rplan)
        eng.free(d_rest)
    # length-only launch: nothing written, totals reported
    eng.fill(d_out, SENTINEL, out_total)
    eng.encode_launch(plan, d_in, d_out, length_only=True)
    assert eng.encoded_lengths(plan, len(items)) == [f.size for f in full]  # capacity planning: whatever the item's own capacity
    assert np.all(eng.download(d_out, out_total) == SENTINEL)
    eng.lib.aws_huffman_amd_encode_plan_destroy(plan)

    # decode side: the full encodings packed at odd offsets, with a bit offset on some
    enc_offs, pos = [], 1
    for f in full:
        enc_offs.append(pos)
        pos += f.size + int(rng.integers(0, 4))
    enc_total = pos + 64
    host_enc = np.zeros(enc_total, np.uint8)
    for f, o in zip(full, enc_offs):
        host_enc[o:o + f.size] = f
    dcaps = [[b.size, b.size + 3, max(b.size - 1, 0), b.size // 3][i % 4] for i, b in enumerate(blobs)]
    sym_offs, pos = [], 7
    for c in dcaps:
        sym_offs.append(pos)
        pos += c + 24
    sym_total = pos + 64
    d_enc, d_sym = eng.alloc(enc_total), eng.alloc(sym_total)
    eng.upload(d_enc, host_enc)
    eng.fill(d_sym, SENTINEL, sym_total)
    ditems = [dict(in_offset=o, in_len=f.size, first_bit=0, out_offset=so, out_capacity=c)
              for f, o, so, c in zip(full, enc_offs, sym_offs, dcaps)]
    dplan = eng.decode_plan(ditems)
    eng.decode_launch(dplan, d_enc, d_sym)
    dres = eng.decode_results(dplan, len(ditems))
    got = eng.download(d_sym, sym_total)
    want = np.full(sym_total, SENTINEL, np.uint8)
    for i, (f, so, c) in enumerate(zip(full, sym_offs, dcaps)):
        d = w.oracle.new_decoder(w.ocoder)
        dst = np.full(c + 1, SENTINEL, np.uint8)
        r = w.oracle.decode_call(d, f, 0, f.size, dst, 0, c)
        want[so:so + c] = dst[:c]
        # bits consumed by the emitted symbols = bytes pulled * 8 - read-ahead left in the decoder
        assert dres[i] == (r.rc, r.err, r.produced, r.consumed * 8 - r.state[0]), (i, dres[i], r)
    assert np.array_equal(got, want), "batched decode wrote outside its items or wrote wrong bytes"
    eng.lib.aws_huffman_amd_decode_plan_destroy(dplan)
    for p in (d_in, d_out, d_enc, d_sym):
        eng.free(p)
    if engine is None:
        eng.close()


def tiny_encode_items(w, n_items=1500, seed=37, engine=None, holes=False, max_len=560, edge_lens=True, thread_limit=None,
                      wave_limit=None, more_lens=(), profile=None):
    """Items either side of HUFD_ENC_TINY_BYTES (512 symbols; one thread each below it), with every kind
    of stop: roomy, exact, one byte short, cut anywhere, no room at all, carried overflow bits that fit,
    fill the output exactly or do not fit, symbols without a code.  thread_limit: the longest item the plan must give a
    thread of its own (aws_huffman_amd_encode_plan_stats) -- the road is asserted, not assumed; wave_limit: the longest
    one that is a wave's work without segments (HUFD_ENC_SOLO_BYTES where the coder encodes in one pass, else 0);
    profile: one of CODER_PROFILES instead of the test coder (len4to15: the packing kernel's build for codes of 13-15 bits)."""
    rng = np.random.default_rng(seed)
    own = engine is None or holes or profile is not None
    code_lens = [int(w.table[1][b]) for b in range(256)]
    if profile is not None:
        oc, profile_pcoder, code_lens = profile_coders(w, profile)
        eng = harness.Engine(w.product.lib, profile_pcoder)
    else:
        eng = harness.Engine(w.product.lib, w.pcoder_holes if holes else w.pcoder) if own else engine
        oc = w.ocoder_holes if holes else w.ocoder
    longest_code = max(code_lens)
    lens = ([0, 0, 1, 2, 3, 511, 512, 513, 600] if edge_lens else [0, 0, 1, 2, 3, 127, 128, 129, max_len - 1]) + list(more_lens)
    lens += [int(rng.integers(0, max_len)) for _ in range(n_items - len(lens))]
    blobs = []
    for i, n in enumerate(lens):
        b = inputs(rng, n, KINDS[i % 4])
        if holes:
            b[b == 7] = 8
            b[b == 200] = 201
            if n and i % 3 == 0:
                b[int(rng.integers(0, n))] = 7 if i % 2 else 200
        blobs.append(b)
    in_offs, pos = [], 1
    for b in blobs:
        in_offs.append(pos)
        pos += b.size + int(rng.integers(0, 3))
    in_total = pos + 64
    host_in = np.zeros(in_total, np.uint8)
    for b, o in zip(blobs, in_offs):
        host_in[o:o + b.size] = b
    items, out_offs, pos = [], [], 3
    for i, b in enumerate(blobs):
        ov = (0, 0)
        if i % 4 == 1:
            nb = int(rng.integers(1, 33))
            ov = (int(rng.integers(0, 1 << nb)), nb)
        full = (ov[1] + longest_code * b.size + 7) // 8 + 2
        # (4 * size + 8: room for the worst case of any coder here -- the road enc_tiny takes without asking for bytes)
        cap = [full, int(rng.integers(0, full + 1)), max(b.size * 5 // 8, 0), 0, int(rng.integers(0, 6)), 4 * b.size + 8][int(rng.integers(0, 6))]
        if i % 7 == 3:  # the exact size, and one less
            e = w.oracle.new_encoder(oc)
            e.overflow_bits.pattern, e.overflow_bits.num_bits = ov
            dst = np.zeros(full + 8, np.uint8)
            r = w.oracle.encode_call(e, b, 0, dst, 0, full + 8)
            cap = max(r.produced - (i % 2), 0)
        out_offs.append(pos)
        items.append(dict(in_offset=in_offs[i], in_len=b.size, out_offset=pos, out_capacity=cap, overflow_in=ov,
                          eos_padding=[0xFF, 0x00, 0xA5][i % 3]))
        pos += cap + int(rng.integers(1, 9))
    out_total = pos + 64
    d_in, d_out = eng.alloc(in_total), eng.alloc(out_total)
    eng.upload(d_in, host_in)
    eng.fill(d_out, SENTINEL, out_total)
    plan = eng.encode_plan(items)
    stats = eng.encode_stats(plan)
    assert stats["items"] == len(items), stats
    assert stats["by_thread"] + stats["by_wave"] + stats["by_pieces"] + stats["empty"] == len(items), stats
    if wave_limit is not None:
        by_wave = sum(1 for it in items if stats["thread_limit"] < it["in_len"] <= wave_limit)
        assert stats["by_wave"] == by_wave and (by_wave > len(items) // 4 or not wave_limit), (stats, by_wave)
    if thread_limit is not None:
        by_thread = sum(1 for it in items if it["in_len"] <= thread_limit and (it["in_len"] or it["overflow_in"][1]))
        assert stats["thread_limit"] == thread_limit and stats["by_thread"] == by_thread, (stats, by_thread)
        assert max(it["in_len"] for it in items if it["in_len"] <= thread_limit) > thread_limit * 3 // 4  # (such items exist)
    eng.encode_launch(plan, d_in, d_out)
    res = eng.encode_results(plan, len(items))
    got = eng.download(d_out, out_total)
    want = np.full(out_total, SENTINEL, np.uint8)
    kinds = set()
    for i, (b, it) in enumerate(zip(blobs, items)):
        e = w.oracle.new_encoder(oc, eos_padding=it["eos_padding"])
        e.overflow_bits.pattern, e.overflow_bits.num_bits = it["overflow_in"]
        c = it["out_capacity"]
        dst = np.full(c + 1, SENTINEL, np.uint8)
        r = w.oracle.encode_call(e, b, 0, dst, 0, c)
        want[it["out_offset"]:it["out_offset"] + c] = dst[:c]
        assert res[i] == (r.rc, r.err, r.consumed, r.produced, r.state[0], r.state[1]), (i, it, res[i], r)
        kinds.add((r.rc, r.err))
    assert np.array_equal(got, want), "tiny items: wrong bytes, or bytes outside an item"
    assert len(kinds) >= (3 if holes else 2)
    eng.encode_launch(plan, d_in, d_out, length_only=True)
    lens_of = list(code_lens)
    if holes:
        lens_of[7] = lens_of[200] = 0
    lens_arr = np.asarray(lens_of, dtype=np.int64)
    want_lens = [(it["overflow_in"][1] + int(lens_arr[b].sum()) + 7) // 8 for b, it in zip(blobs, items)]
    assert eng.encoded_lengths(plan, len(items)) == want_lens
    assert np.array_equal(eng.download(d_out, out_total), want), "a length-only launch wrote something"
    eng.lib.aws_huffman_amd_encode_plan_destroy(plan)
    eng.free(d_in)
    eng.free(d_out)
    if own:
        eng.close()


def tiny_decode_items(w, n_items=1500, seed=41, engine=None, profile=None, max_len=460, thread_limit=None):
    """Streams either side of HUFD_DEC_TINY_BYTES (512 bytes; one thread each below it): whole encodings, cut
    ones, arbitrary bytes, starting inside their first byte, with room for all, some or none of their symbols.
    thread_limit: the longest item the plan must give a thread of its own (aws_huffman_amd_decode_plan_stats)."""
    rng = np.random.default_rng(seed)
    ocoder, pcoder = w.ocoder, w.pcoder
    if profile:
        ocoder, pcoder, _ = profile_coders(w, profile)
        engine = None
    eng = engine or harness.Engine(w.product.lib, pcoder)
    streams = []
    for i in range(n_items):
        kind = i % 5
        if kind == 4:
            enc = rng.integers(0, 256, int(rng.integers(1, max_len + 80)), dtype=np.uint8)
        else:
            n = [0, 1, 2, 409, 410, 420][i // 5] if i < 30 else int(rng.integers(0, max_len))
            enc = oracle_encode(w, inputs(rng, n, KINDS[i % 3]), coder=ocoder, eos=[None, 0x00, 0x5A][i % 3])
            if kind == 3 and enc.size:
                enc = enc[:int(rng.integers(0, enc.size + 1))]
        streams.append(enc)
    offs, pos = [], 1
    for e in streams:
        offs.append(pos)
        pos += e.size + int(rng.integers(0, 4))
    enc_total = pos + 64
    host_enc = np.zeros(enc_total, np.uint8)
    for e, o in zip(streams, offs):
        host_enc[o:o + e.size] = e
    items, expect, pos = [], [], 5
    for i, e in enumerate(streams):
        fb = int(rng.integers(0, 8)) if (i % 4 == 2 and e.size) else 0
        # the oracle's view: the rest of the first byte is what a previous call left in the decoder
        d = w.oracle.new_decoder(ocoder)
        start = 0
        if fb:
            d.working_bits = (int(e[0]) & (0xFF >> fb)) << (56 + fb)
            d.num_bits = 8 - fb
            start = 1
        probe = np.zeros(e.size * 2 + 8, np.uint8)
        r_all = w.oracle.decode_call(d, e, start, e.size, probe, 0, probe.size)
        n = r_all.produced
        cap = [n, n + 3, max(n - 1, 0), n // 3, 0][int(rng.integers(0, 5))]
        d = w.oracle.new_decoder(ocoder)
        if fb:
            d.working_bits = (int(e[0]) & (0xFF >> fb)) << (56 + fb)
            d.num_bits = 8 - fb
        dst = np.full(cap + 1, SENTINEL, np.uint8)
        r = w.oracle.decode_call(d, e, start, e.size, dst, 0, cap)
        bits = (8 - fb if fb else 0) + r.consumed * 8 - r.state[0]
        expect.append(((r.rc, r.err, r.produced, bits), dst[:cap].copy()))
        items.append(dict(in_offset=offs[i], in_len=e.size, first_bit=fb, out_offset=pos, out_capacity=cap))
        pos += cap + int(rng.integers(1, 9))
    sym_total = pos + 64
    d_enc, d_sym = eng.alloc(enc_total), eng.alloc(sym_total)
    eng.upload(d_enc, host_enc)
    eng.fill(d_sym, SENTINEL, sym_total)
    plan = eng.decode_plan(items)
    stats = eng.decode_stats(plan)
    assert stats["items"] == len(items), stats
    assert stats["by_thread"] + stats["by_wave"] + stats["by_workgroup"] + stats["by_blocks"] + stats["by_pieces"] + stats["empty"] == len(items), stats
    if thread_limit is not None:
        by_thread = sum(1 for it in items if 0 < it["in_len"] <= thread_limit)
        assert stats["thread_limit"] == thread_limit and stats["by_thread"] == by_thread, (stats, by_thread)
        assert max(it["in_len"] for it in items if it["in_len"] <= thread_limit) > thread_limit * 3 // 4
    eng.decode_launch(plan, d_enc, d_sym)
    res = eng.decode_results(plan, len(items))
    got = eng.download(d_sym, sym_total)
    want = np.full(sym_total, SENTINEL, np.uint8)
    kinds = set()
    for i, (it, (key, sym)) in enumerate(zip(items, expect)):
        want[it["out_offset"]:it["out_offset"] + it["out_capacity"]] = sym
        assert res[i] == key, (i, it, res[i], key)
        kinds.add(key[:2])
    assert np.array_equal(got, want), "tiny items: wrong symbols, or bytes outside an item"
    assert len(kinds) >= (2 if profile else 3)
    eng.lib.aws_huffman_amd_decode_plan_destroy(plan)
    eng.free(d_enc)
    eng.free(d_sym)
    if engine is None:
        eng.close()


def decode_items_like_the_oracle(w, eng, ocoder, streams, rng, label, modes=(None,), kinds=3):
    """streams: (encoded bytes, first bit, output capacity) each; one plan of them all, launched twice per mode, every
    record and every output byte (and the bytes between the outputs) as the oracle has them."""
    offs, pos = [], 3
    for e, _, _ in streams:
        offs.append(pos)
        pos += e.size + int(rng.integers(0, 9))
    host_enc = np.zeros(pos + 64, np.uint8)
    for (e, _, _), o in zip(streams, offs):
        host_enc[o:o + e.size] = e
    items, expect, pos = [], [], 7
    for (e, fb, cap), o in zip(streams, offs):
        d = w.oracle.new_decoder(ocoder)
        start = 0
        if fb:  # the rest of the first byte is what a previous call left in the decoder
            d.working_bits = (int(e[0]) & (0xFF >> fb)) << (56 + fb)
            d.num_bits = 8 - fb
            start = 1
        dst = np.full(cap + 1, SENTINEL, np.uint8)
        r = w.oracle.decode_call(d, e, start, e.size, dst, 0, cap)
        bits = (8 - fb if fb else 0) + r.consumed * 8 - r.state[0]
        expect.append(((r.rc, r.err, r.produced, bits), dst[:cap].copy()))
        items.append(dict(in_offset=o, in_len=e.size, first_bit=fb, out_offset=pos, out_capacity=cap))
        pos += cap + int(rng.integers(1, 9))
    sym_total = pos + 64
    d_enc, d_sym = eng.alloc(host_enc.size), eng.alloc(sym_total)
    eng.upload(d_enc, host_enc)
    want = np.full(sym_total, SENTINEL, np.uint8)
    for it, (key, sym) in zip(items, expect):
        want[it["out_offset"]:it["out_offset"] + it["out_capacity"]] = sym
    assert len({key[:2] for key, _ in expect}) >= kinds
    plan = eng.decode_plan(items)
    for mode in modes:
        with harness.decode_road(eng.lib, mode):
            for _ in range(2):  # (a second launch of the plan finds the scratch words of the first)
                eng.fill(d_sym, SENTINEL, sym_total)
                eng.decode_launch(plan, d_enc, d_sym)
                res = eng.decode_results(plan, len(items))
                for i, (it, (key, _)) in enumerate(zip(items, expect)):
                    assert res[i] == key, (label, mode, i, it, res[i], key)
                assert np.array_equal(eng.download(d_sym, sym_total), want), (label, mode)
    eng.lib.aws_huffman_amd_decode_plan_destroy(plan)
    eng.free(d_enc)
    eng.free(d_sym)


def mid_sized_items(w, n_items=150, seed=101, engine=None, modes=(None, "lean-sync"), longest=9000):
    """A batch of items between header size and a whole chunk (0.8 .. ~10 KiB of encoded bytes each): every one is a
    chunk its stream ENDS in, with a handful of whole lanes -- several of them share a workgroup (dec_sync_pack,
    dec_emit_pack), one slot each.  Whole streams, cut ones, damaged ones, arbitrary bytes, streams that start inside a
    byte, room for all, some or none of the symbols; a plan with a few longer items among them (the slot width follows
    the longest end-of-stream chunk) and, "lean-sync", the same through a workgroup per chunk."""
    rng = np.random.default_rng(seed)
    eng = engine or harness.Engine(w.product.lib, w.pcoder)
    for label, lens in (("2-3 KiB", rng.integers(1700, 2600, n_items)), ("0.7-9 KiB", rng.integers(700, longest, n_items)),
                        ("1 KiB and one long", np.concatenate([rng.integers(900, 1200, n_items - 1), [70_000]]))):
        streams = []
        for i, n in enumerate(lens):
            kind = i % 7
            if kind == 6:
                enc = rng.integers(0, 256, int(n), dtype=np.uint8)  # arbitrary bytes
            else:
                enc = oracle_encode(w, inputs(rng, int(n), KINDS[i % 3]), eos=[None, 0x00, 0x5A][i % 3])
                if kind == 3:
                    enc = enc[:int(rng.integers(min(600, enc.size), enc.size + 1))]  # cut
                if kind == 4:
                    enc = enc.copy()
                    at = int(rng.integers(0, max(enc.size - 4, 1)))
                    enc[at:at + 4] = 0xFF  # ten one bits: no code
            fb = int(rng.integers(0, 8)) if kind == 5 else 0
            cap = [int(n) + 8, int(n), int(rng.integers(0, n)), int(n) + 8][i % 4]
            streams.append((enc, fb, cap))
        decode_items_like_the_oracle(w, eng, w.ocoder, streams, rng, label, modes=modes, kinds=3)
    if engine is None:
        eng.close()


def streams_with_two_last_chunks(w, seed=127, engine=None, modes=(None, "lean-sync")):
    """Items of k * 32 KiB + 1..7 encoded bytes: the item's last chunk holds fewer than 8 bytes, so the chunk in front of
    it is listed as one a stream ends in as well -- two entries of the plan's end-of-stream list an item, one wide and
    one narrow.  Plans in which such items outnumber the others (the list is then full to its last slot: wide chunks are
    collected from its back and moved in behind the narrow ones), alone and mixed with short items."""
    rng = np.random.default_rng(seed)
    eng = engine or harness.Engine(w.product.lib, w.pcoder)
    long_enc = oracle_encode(w, inputs(rng, 70_000, "uniform"))
    assert long_enc.size > 2 * 32768 + 7

    def cut_to(n, damaged=False):
        e = long_enc[:n].copy()
        if damaged:
            e[n - 3:n] = 0xFF
        return (e, 0, n)  # (room for every symbol: at least 5 bits each)

    short = lambda n: (oracle_encode(w, inputs(rng, n, "uniform")), 0, n + 8)
    for label, streams in (
            ("two of 32772", [cut_to(32772), cut_to(32772)]),
            ("three of 32772", [cut_to(32772)] * 3),
            ("32773 + 65543 + 32769", [cut_to(32773), cut_to(65543), cut_to(32769, damaged=True)]),
            ("one of 32772", [cut_to(32772)]),
            ("three of 32780", [cut_to(32780)] * 3),
            ("mixed with short ones", [cut_to(32775), short(900), cut_to(65537), cut_to(32770), short(1500), cut_to(32771)]),
            ("more short ones than long", [short(700), cut_to(32772), short(800), short(2000), cut_to(65540), short(40)])):
        decode_items_like_the_oracle(w, eng, w.ocoder, streams, rng, label, modes=modes, kinds=1)
    if engine is None:
        eng.close()


def plans_made_on_the_device(w, seed=131, engine=None, big=2_200_000, n_small=300,
                             strided_batches=((70, 16384, 2 * 16384), (70, 16384, 16384), (5000, 700, 1400), (5000, 100, 90), (3, 2_000_000, 4_000_000))):
    """A plan made on the device -- from the caller's records lying in DEVICE memory, from a stride, or (decode) from what
    an encode launch left -- is the plan the host's loop makes: the same counts (aws_huffman_amd_*_plan_stats), and for
    the same buffers the same records, output bytes and guard bytes.  Items of every class in one plan: empty, one
    thread's work, one wave's, a chunk or a few, two end-of-stream chunks, more than 64 chunks / segments; carried
    overflow bits, short outputs, first-bit offsets, cut and damaged streams; then batches of equal buffers by stride
    (BASELINE configs[3]'s shape, some with too little room), and encode -> decode of such a batch without its lengths
    ever coming to the host."""
    rng = np.random.default_rng(seed)
    eng = engine or harness.Engine(w.product.lib, w.pcoder)
    lib = eng.lib

    # ---- encode: host records against the same records in device memory
    lens = [0, 0, 1, 100, 128, 129, 600, 1024, 1025, 5000, 16384, 16385, 40_000, big] + [int(x) for x in rng.integers(0, 3000, n_small)]
    blobs = [inputs(rng, n, KINDS[i % 4]) for i, n in enumerate(lens)]
    in_offs, pos = [], 5
    for b in blobs:
        in_offs.append(pos)
        pos += b.size + int(rng.integers(0, 4))
    host_in = np.zeros(pos + 64, np.uint8)
    for b, o in zip(blobs, in_offs):
        host_in[o:o + b.size] = b
    items, pos = [], 3
    for i, b in enumerate(blobs):
        ov = (int(rng.integers(0, 1 << 7)), 7) if i % 5 == 1 else (0, 0)
        full = (ov[1] + 10 * b.size + 7) // 8 + 2
        cap = [full, full, max(full // 2, 0), full, 0][i % 5] if b.size < 50_000 else full
        items.append(dict(in_offset=in_offs[i], in_len=b.size, out_offset=pos, out_capacity=cap, overflow_in=ov,
                          eos_padding=[0xFF, 0x00, 0xA5][i % 3]))
        pos += cap + int(rng.integers(1, 9))
    out_total = pos + 64
    d_in, d_a, d_b = eng.alloc(host_in.size), eng.alloc(out_total), eng.alloc(out_total)
    eng.upload(d_in, host_in)
    host_plan = eng.encode_plan(items)
    dev_plan, d_items = eng.encode_plan_from_device_items(items)
    assert eng.encode_stats(host_plan) == eng.encode_stats(dev_plan), (eng.encode_stats(host_plan), eng.encode_stats(dev_plan))
    # (segments: 16 385, 40 000 and `big` symbols at least -- 5 000 and 16 384 too in a plan of fewer than 256 items; a wave each: the rest above a thread's)
    assert eng.encode_stats(dev_plan)["by_pieces"] >= 3 and eng.encode_stats(dev_plan)["by_thread"] >= 3 and eng.encode_stats(dev_plan)["by_wave"] >= 3
    for d_out, plan in ((d_a, host_plan), (d_b, dev_plan)):
        eng.fill(d_out, SENTINEL, out_total)
        eng.encode_launch(plan, d_in, d_out)
    res_host, res_dev = eng.encode_results(host_plan, len(items)), eng.encode_results(dev_plan, len(items))
    assert res_host == res_dev, [(i, a, b) for i, (a, b) in enumerate(zip(res_host, res_dev)) if a != b][:3]
    assert len({r[:2] for r in res_host}) >= 2
    enc_all = eng.download(d_a, out_total)
    assert np.array_equal(enc_all, eng.download(d_b, out_total)), "encode: the device-made plan wrote other bytes"
    # (and the host-made plan is the oracle's: a few items of every class)
    for i in list(range(14)) + [20, 21]:
        e = w.oracle.new_encoder(w.ocoder, eos_padding=items[i]["eos_padding"])
        e.overflow_bits.pattern, e.overflow_bits.num_bits = items[i]["overflow_in"]
        c = items[i]["out_capacity"]
        dst = np.full(c + 1, SENTINEL, np.uint8)
        r = w.oracle.encode_call(e, blobs[i], 0, dst, 0, c)
        assert res_host[i] == (r.rc, r.err, r.consumed, r.produced, r.state[0], r.state[1]), (i, res_host[i], r)
        assert np.array_equal(enc_all[items[i]["out_offset"]:items[i]["out_offset"] + c], dst[:c]), i

    # ---- decode what was written (whole streams, cut ones, a damaged one, a first-bit offset, short outputs):
    # host records, the same in device memory, and -- for the whole ones -- chained to the encode plan
    # (every output a window of its own: a stream with carried bits in front, or padded with other bits than ones, may hold
    #  more symbols than went in -- back where the symbols came from, with room for five more, two items wrote the same
    #  byte, and which of them did last is nobody's to say: found by the round-5 soak, 3 runs in 270)
    ditems, dpos = [], 3
    for i, (it, r) in enumerate(zip(items, res_host)):
        produced = r[3]
        in_len = produced if i % 7 else produced // 2  # (every seventh cut in half)
        cap = [it["in_len"], it["in_len"] + 5, it["in_len"] // 2][i % 3] if it["in_len"] < 50_000 else it["in_len"]
        ditems.append(dict(in_offset=it["out_offset"], in_len=in_len, first_bit=0, out_offset=dpos, out_capacity=cap))
        dpos += cap + int(rng.integers(1, 9))
    damaged = int(np.argmax(lens))
    damage_at = items[damaged]["out_offset"] + min(1_000_000, res_host[damaged][3] // 2)  # (inside the longest item's output)
    bad = eng.download(d_a, 8, offset=damage_at)
    eng.upload(d_a, np.full(4, 0xFF, np.uint8), offset=damage_at)
    sym_total = max(host_in.size, dpos + 64)
    d_sa, d_sb = eng.alloc(sym_total), eng.alloc(sym_total)
    host_dplan = eng.decode_plan(ditems)
    dev_dplan, d_ditems = eng.decode_plan_from_device_items(ditems)
    assert eng.decode_stats(host_dplan) == eng.decode_stats(dev_dplan), (eng.decode_stats(host_dplan), eng.decode_stats(dev_dplan))
    st = eng.decode_stats(dev_dplan)
    assert st["by_pieces"] >= 4 and st["by_thread"] >= 3 and st["by_wave"] >= 1 and st["end_pieces_packed"] + st["end_pieces_single"] >= 4, st
    for d_out, plan in ((d_sa, host_dplan), (d_sb, dev_dplan)):
        eng.fill(d_out, SENTINEL, sym_total)
        eng.decode_launch(plan, d_a, d_out)
    dres_host, dres_dev = eng.decode_results(host_dplan, len(ditems)), eng.decode_results(dev_dplan, len(ditems))
    assert dres_host == dres_dev, [(i, a, b) for i, (a, b) in enumerate(zip(dres_host, dres_dev)) if a != b][:3]
    assert len({r[:2] for r in dres_host}) >= 3  # (whole, short of room, a symbol without a code)
    out_a, out_b = eng.download(d_sa, sym_total), eng.download(d_sb, sym_total)
    if not np.array_equal(out_a, out_b):  # (say which item, and which of the two plans the oracle agrees with)
        at = int(np.flatnonzero(out_a != out_b)[0])
        who = [i for i, d in enumerate(ditems) if d["out_offset"] <= at < d["out_offset"] + max(d["out_capacity"], 1)]
        detail = {"first_difference_at": at, "bytes_that_differ": int((out_a != out_b).sum()), "items": who}
        enc_now = eng.download(d_a, out_total)
        for i in who[:1]:
            d = ditems[i]
            dec = w.oracle.new_decoder(w.ocoder)
            dst = np.full(d["out_capacity"] + 1, SENTINEL, np.uint8)
            r = w.oracle.decode_call(dec, enc_now[d["in_offset"]:d["in_offset"] + d["in_len"]].copy(), 0, d["in_len"], dst, 0, d["out_capacity"])
            lo, hi = d["out_offset"], d["out_offset"] + d["out_capacity"]
            detail.update(item=d, oracle=(r.rc, r.err, r.produced), record=dres_host[i], offset_in_item=at - lo,
                          host_plan_is_the_oracles=bool(np.array_equal(out_a[lo:hi], dst[:d["out_capacity"]])),
                          device_plan_is_the_oracles=bool(np.array_equal(out_b[lo:hi], dst[:d["out_capacity"]])))
        raise AssertionError("decode: the device-made plan wrote other bytes: %r" % (detail,))
    eng.upload(d_a, bad[:4], offset=damage_at)
    # chained: what the encode launch of dev_plan left in d_b, decoded back to where it came from
    chained = eng.empty_decode_plan()
    eng.encode_launch(dev_plan, d_in, d_b)
    assert eng.decode_plan_from_encode(chained, dev_plan)
    st = eng.decode_stats(chained)
    assert st["items"] == len(items) and st["by_pieces"] >= 4, st
    eng.fill(d_sb, SENTINEL, sym_total)
    eng.decode_launch(chained, d_b, d_sb)
    cres = eng.decode_results(chained, len(items))
    back = eng.download(d_sb, sym_total)
    want = np.full(sym_total, SENTINEL, np.uint8)
    for i, (it, r) in enumerate(zip(items, res_host)):
        # (what was consumed comes back -- but for the symbol whose code was cut by a short output, and for carried bits,
        #  which are somebody else's symbols: only items encoded whole from nothing are compared byte by byte)
        # (nor items padded with other bits than ones: those may read as one more symbol, reference huffman.h "eos_padding")
        if r[0] == 0 and it["overflow_in"][1] == 0 and it["eos_padding"] == 0xFF:
            assert cres[i][:3] == (0, 0, it["in_len"]), (i, cres[i], it)
            want[it["in_offset"]:it["in_offset"] + it["in_len"]] = blobs[i]
        else:
            want[it["in_offset"]:it["in_offset"] + it["in_len"]] = back[it["in_offset"]:it["in_offset"] + it["in_len"]]
    assert np.array_equal(back, want), "chained decode: wrong symbols, or bytes outside an item"
    for plan in (host_plan, dev_plan):
        lib.aws_huffman_amd_encode_plan_destroy(plan)
    for plan in (host_dplan, dev_dplan, chained):
        lib.aws_huffman_amd_decode_plan_destroy(plan)
    for ptr in (d_in, d_a, d_b, d_sa, d_sb, d_items, d_ditems):
        eng.free(ptr)

    # ---- batches of equal buffers by stride: 16 KiB and 700 B, with room and with too little
    for count, size, cap in strided_batches:
        data = inputs(rng, count * size, "uniform")
        d_in, d_enc1, d_enc2, d_back = eng.alloc(data.size + 64), eng.alloc(count * cap + 64), eng.alloc(count * cap + 64), eng.alloc(data.size + 64)
        eng.upload(d_in, data)
        host_items = [dict(in_offset=k * size, in_len=size, out_offset=k * cap, out_capacity=cap, eos_padding=0xFF) for k in range(count)]
        host_plan = eng.encode_plan(host_items)
        strided = eng.plan_strided(True, count=count, in_offset=0, in_stride=size, in_len=size, out_offset=0, out_stride=cap,
                                   out_capacity=cap, eos_padding=0xFF)
        assert eng.encode_stats(host_plan) == eng.encode_stats(strided), (count, size, eng.encode_stats(host_plan), eng.encode_stats(strided))
        for d_out, plan in ((d_enc1, host_plan), (d_enc2, strided)):
            eng.fill(d_out, SENTINEL, count * cap + 64)
            eng.encode_launch(plan, d_in, d_out)
        r1, r2 = eng.encode_results(host_plan, count), eng.encode_results(strided, count)
        assert r1 == r2, (count, size, cap)
        assert np.array_equal(eng.download(d_enc1, count * cap + 64), eng.download(d_enc2, count * cap + 64)), (count, size, cap)
        # decode of the strided launch's output, chained (lengths on the device only) and by stride where every item is whole
        chained = eng.empty_decode_plan()
        assert eng.decode_plan_from_encode(chained, strided)
        eng.fill(d_back, SENTINEL, data.size + 64)
        eng.decode_launch(chained, d_enc2, d_back)
        cres = eng.decode_results(chained, count)
        back = eng.download(d_back, data.size + 64)
        for k in range(count):
            if r1[k][0] == 0:
                assert cres[k][:3] == (0, 0, size), (count, size, cap, k, cres[k])
                assert np.array_equal(back[k * size:(k + 1) * size], data[k * size:(k + 1) * size]), (count, size, cap, k)
            else:
                # the room ran out: the symbols whose codes went out whole come back (the cut one does not)
                assert r1[k][:2] == (-1, harness.AWS_ERROR_SHORT_BUFFER) and cres[k][2] in (r1[k][2] - 1, r1[k][2]), (k, r1[k], cres[k])
                assert np.array_equal(back[k * size:k * size + r1[k][2] - 1], data[k * size:k * size + r1[k][2] - 1])
        assert np.all(back[data.size:] == SENTINEL)
        if all(r[0] == 0 for r in r1) and len({r[3] for r in r1}) == 1:
            pass  # (equal encoded lengths happen for no random batch: the decode-by-stride case is made below)
        lib.aws_huffman_amd_encode_plan_destroy(host_plan)
        lib.aws_huffman_amd_encode_plan_destroy(strided)
        lib.aws_huffman_amd_decode_plan_destroy(chained)
        for ptr in (d_in, d_enc1, d_enc2, d_back):
            eng.free(ptr)
    # decode by stride: equal slots of encoded bytes, each holding a stream and then other bytes -- every item is "the slot",
    # and what lies behind a stream's end is decoded on (or stops it) exactly as for the oracle
    for count, slot, n_sym in ((40, 20000, 16384), (3000, 300, 200)):
        streams = [oracle_encode(w, inputs(rng, n_sym, "uniform"), eos=0x00) for _ in range(count)]
        assert max(e.size for e in streams) <= slot
        blob = rng.integers(0, 256, count * slot + 64, dtype=np.uint8)
        for k, e in enumerate(streams):
            blob[k * slot:k * slot + e.size] = e
        cap = n_sym + 40
        d_enc, d_s1, d_s2 = eng.alloc(blob.size), eng.alloc(count * cap + 64), eng.alloc(count * cap + 64)
        eng.upload(d_enc, blob)
        host_items = [dict(in_offset=k * slot, in_len=slot, out_offset=k * cap, out_capacity=cap) for k in range(count)]
        host_plan = eng.decode_plan(host_items)
        strided = eng.plan_strided(False, count=count, in_offset=0, in_stride=slot, in_len=slot, out_offset=0, out_stride=cap, out_capacity=cap)
        assert eng.decode_stats(host_plan) == eng.decode_stats(strided)
        for d_out, plan in ((d_s1, host_plan), (d_s2, strided)):
            eng.fill(d_out, SENTINEL, count * cap + 64)
            eng.decode_launch(plan, d_enc, d_out)
        assert eng.decode_results(host_plan, count) == eng.decode_results(strided, count)
        got = eng.download(d_s2, count * cap + 64)
        assert np.array_equal(eng.download(d_s1, count * cap + 64), got)
        for k in (0, 1, count - 1):
            d = w.oracle.new_decoder(w.ocoder)
            dst = np.full(cap + 1, SENTINEL, np.uint8)
            r = w.oracle.decode_call(d, blob[k * slot:(k + 1) * slot], 0, slot, dst, 0, cap)
            assert eng.decode_results(strided, count)[k][:3] == (r.rc, r.err, r.produced), (k, r)
            assert np.array_equal(got[k * cap:(k + 1) * cap], dst[:cap]), k
        lib.aws_huffman_amd_decode_plan_destroy(host_plan)
        lib.aws_huffman_amd_decode_plan_destroy(strided)
        for ptr in (d_enc, d_s1, d_s2):
            eng.free(ptr)
    # ---- what the host's loop refuses, the device's passes refuse: an item whose segments do not fit 32 bits (2^45 symbols
    # an item used to wrap to no pieces at all and the plan was accepted), a batch whose pieces together do not, and a decode
    # chained to a launch that only measured
    refused, refused_d = eng.empty_encode_plan(), eng.empty_decode_plan()
    for count, in_len in ((2, 1 << 45), (1, 1 << 46), (300, 1 << 46), (3, (0xFFFFFFFE - 1) * 16384 + 1), (70_000, 1 << 30), (1, (1 << 64) - 1)):
        desc = harness.StridedItems(count=count, in_offset=0, in_stride=0, in_len=in_len, out_offset=0, out_stride=0, out_capacity=16, eos_padding=0xFF)
        assert lib.aws_huffman_amd_encode_plan_reset_strided(refused, C.byref(desc), None) != 0, (count, in_len)
        assert lib.aws_last_error() == harness.AWS_ERROR_INVALID_ARGUMENT and eng.encode_stats(refused)["items"] == 0, (count, in_len)
    for count, in_len in ((2, 1 << 32), (70_000, (1 << 32) - 1)):
        desc = harness.StridedItems(count=count, in_offset=0, in_stride=0, in_len=in_len, out_offset=0, out_stride=0, out_capacity=16)
        assert lib.aws_huffman_amd_decode_plan_reset_strided(refused_d, C.byref(desc), None) != 0, (count, in_len)
        assert lib.aws_last_error() == harness.AWS_ERROR_INVALID_ARGUMENT and eng.decode_stats(refused_d)["items"] == 0, (count, in_len)
    data = inputs(rng, 40 * 3000, "uniform")
    d_in, d_enc = eng.alloc(data.size + 64), eng.alloc(40 * 6000 + 64)
    eng.upload(d_in, data)
    measured = eng.plan_strided(True, plan=refused, count=40, in_offset=0, in_stride=3000, in_len=3000, out_offset=0, out_stride=6000,
                                out_capacity=6000, eos_padding=0xFF)
    lib.aws_huffman_amd_decode_plan_from_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    assert lib.aws_huffman_amd_decode_plan_from_encode(refused_d, measured, None) != 0  # (never launched)
    assert lib.aws_last_error() == harness.AWS_ERROR_INVALID_ARGUMENT
    eng.encode_launch(measured, d_in, d_enc)
    eng.encode_launch(measured, d_in, d_enc, length_only=True)
    want_lens = [oracle_encode(w, data[k * 3000:(k + 1) * 3000]).size for k in range(40)]
    assert eng.encoded_lengths(measured, 40) == want_lens
    assert lib.aws_huffman_amd_decode_plan_from_encode(refused_d, measured, None) != 0  # (launched, but a length query last)
    assert lib.aws_last_error() == harness.AWS_ERROR_INVALID_ARGUMENT
    eng.encode_launch(measured, d_in, d_enc)
    assert eng.decode_plan_from_encode(refused_d, measured)
    # (a plan destroyed with its record-writing kernels still queued is the engine's spare: the next plan_new waits for them)
    lib.aws_huffman_amd_decode_plan_destroy(refused_d)
    again = eng.decode_plan([dict(in_offset=0, in_len=want_lens[0], out_offset=0, out_capacity=3000)])
    d_back = eng.alloc(3000 + 64)
    eng.decode_launch(again, d_enc, d_back)
    assert eng.decode_results(again, 1)[0][:3] == (0, 0, 3000)
    assert np.array_equal(eng.download(d_back, 3000), data[:3000])
    lib.aws_huffman_amd_decode_plan_destroy(again)
    lib.aws_huffman_amd_encode_plan_destroy(measured)
    for ptr in (d_in, d_enc, d_back):
        eng.free(ptr)
    if engine is None:
        eng.close()


def device_plans_of_other_coders(w, names=("hpack_lengths", "len8", "len2to30", "len1to16"), seed=149,
                                  batches=((60, 16384), (700, 600), (5000, 60))):
    """Plans from a stride and chained on the device, for the coders the chunk kernels do not take (codes of more than 12
    bits: items a thread / a wave / a workgroup each; one code length: no walk at all) and for encoders outside the one-pass
    kernel's codes: the same records and bytes as the plan made from host records, the oracle's bytes, and the symbols back."""
    rng = np.random.default_rng(seed)
    for name in names:
        oc, pcoder, lengths = profile_coders(w, name)
        eng = harness.Engine(w.product.lib, pcoder)
        lib = eng.lib
        longest = max(lengths)
        for count, size in batches:
            data = (32 + rng.integers(0, 95, count * size)).astype(np.uint8)
            cap = (longest * size + 7) // 8 + 8
            d_in, d_e1, d_e2, d_back = eng.alloc(data.size + 64), eng.alloc(count * cap + 64), eng.alloc(count * cap + 64), eng.alloc(data.size + 64)
            eng.upload(d_in, data)
            host_plan = eng.encode_plan([dict(in_offset=k * size, in_len=size, out_offset=k * cap, out_capacity=cap, eos_padding=0xFF) for k in range(count)])
            strided = eng.plan_strided(True, count=count, in_offset=0, in_stride=size, in_len=size, out_offset=0, out_stride=cap, out_capacity=cap, eos_padding=0xFF)
            assert eng.encode_stats(host_plan) == eng.encode_stats(strided), (name, count, size)
            for d_out, plan in ((d_e1, host_plan), (d_e2, strided)):
                eng.fill(d_out, SENTINEL, count * cap + 64)
                eng.encode_launch(plan, d_in, d_out)
            r1, r2 = eng.encode_results(host_plan, count), eng.encode_results(strided, count)
            assert r1 == r2 and all(r[0] == 0 and r[2] == size for r in r1), (name, count, size)
            enc = eng.download(d_e2, count * cap + 64)
            assert np.array_equal(eng.download(d_e1, count * cap + 64), enc), (name, count, size)
            for k in (0, count // 2, count - 1):
                want = w.oracle.encode_all(oc, data[k * size:(k + 1) * size], slack=64)
                assert r1[k][3] == want.size and np.array_equal(enc[k * cap:k * cap + want.size], want), (name, k)
            chained = eng.empty_decode_plan()
            assert eng.decode_plan_from_encode(chained, strided)
            eng.fill(d_back, SENTINEL, data.size + 64)
            eng.decode_launch(chained, d_e2, d_back)
            cres = eng.decode_results(chained, count)
            assert all(r[:3] == (0, 0, size) for r in cres), (name, count, size, [r for r in cres if r[:3] != (0, 0, size)][:2])
            back = eng.download(d_back, data.size + 64)
            assert np.array_equal(back[:data.size], data) and np.all(back[data.size:] == SENTINEL), (name, count, size)
            lib.aws_huffman_amd_encode_plan_destroy(host_plan)
            lib.aws_huffman_amd_encode_plan_destroy(strided)
            lib.aws_huffman_amd_decode_plan_destroy(chained)
            for ptr in (d_in, d_e1, d_e2, d_back):
                eng.free(ptr)
        eng.close()


def plans_one_after_another(w, seed=107):
    """An engine keeps the device arrays of ONE destroyed plan of each kind for the next aws_huffman_amd_*_plan_new: plans
    of different shapes made, launched and destroyed after one another on one engine (larger after smaller and the other
    way round, header-sized after chunked), and two plans alive at once of which only one can be an adopted one."""
    eng = harness.Engine(w.product.lib, w.pcoder)
    batched_device_api(w, n_items=5, item_len=40_000, engine=eng)
    tiny_encode_items(w, n_items=300, seed=seed, engine=eng)
    tiny_decode_items(w, n_items=300, seed=seed + 1, engine=eng)
    mid_sized_items(w, n_items=30, seed=seed + 2, engine=eng, modes=(None,))
    batched_device_api(w, n_items=12, item_len=3000, engine=eng)
    rng = np.random.default_rng(seed)
    data = inputs(rng, 50_000, "uniform")
    enc = oracle_encode(w, data)
    d_enc, d_a, d_b = eng.alloc(enc.size + 64), eng.alloc(data.size + 64), eng.alloc(data.size + 64)
    eng.upload(d_enc, enc)
    whole = [dict(in_offset=0, in_len=enc.size, out_offset=0, out_capacity=data.size)]
    cut = [dict(in_offset=0, in_len=enc.size // 2, out_offset=3, out_capacity=data.size)]
    a, b = eng.decode_plan(whole), eng.decode_plan(cut)
    eng.decode_launch(a, d_enc, d_a)
    eng.decode_launch(b, d_enc, d_b)
    ra, rb = eng.decode_results(a, 1)[0], eng.decode_results(b, 1)[0]
    assert ra[:3] == (0, 0, data.size) and np.array_equal(eng.download(d_a, data.size), data)
    assert rb[0] == 0 and 0 < rb[2] < data.size and np.array_equal(eng.download(d_b, rb[2] + 3)[3:], data[:rb[2]])
    eng.lib.aws_huffman_amd_decode_plan_destroy(a)
    eng.lib.aws_huffman_amd_decode_plan_destroy(b)
    a = eng.decode_plan(cut)  # (adopts what a was)
    eng.fill(d_b, SENTINEL, data.size + 64)
    eng.decode_launch(a, d_enc, d_b)
    assert eng.decode_results(a, 1)[0] == rb and np.array_equal(eng.download(d_b, rb[2] + 3)[3:], data[:rb[2]])
    eng.lib.aws_huffman_amd_decode_plan_destroy(a)
    for d in (d_enc, d_a, d_b):
        eng.free(d)
    eng.close()


def wide_long_code_items(w, n=150_000, seed=79, modes=(None, "wide-fails", "wide-fn-fails")):
    """Long items of coders with codes of more than 12 bits (decode through linked tables): a workgroup per 32 KiB
    block (dec_wide_*) from 64 KiB on here; with "wide-fails" those kernels give every item up as they do a stream whose
    walks never fall into step, and it goes by transfer functions (dec_wide_fn_*: every road a second time, that way);
    with "wide-fn-fails" those give it up as well and dec_deep, the last way back, takes it."""
    rng = np.random.default_rng(seed)
    w.product.lib.aws_huffman_amd_testing_set_wide_min_bytes(2 * 32768)
    try:
        for name in ("hpack_lengths", "len4to15"):
            ocoder, pcoder, lengths = profile_coders(w, name)
            prob = np.array([2.0 ** -min(l, 16) for l in lengths])
            prob /= prob.sum()
            eng = harness.Engine(w.product.lib, pcoder)
            data = rng.choice(256, size=n, p=prob).astype(np.uint8)
            good = w.oracle.encode_all(ocoder, data, slack=64 + 4 * n)
            assert good.size >= 3 * 32768, good.size
            damaged = good.copy()
            damaged[good.size // 3: good.size // 3 + 5] = 0xFF  # (hpack: 30 ones have no code)
            damaged2 = good.copy()
            damaged2[40000] ^= 0x10  # other symbols from here on, for a while: block 1's guess meets another walk
            streams = [
                (good, 0, n), (good, 0, n + 5), (good, 0, n // 2), (good, 0, 0), (damaged, 0, n), (damaged2, 0, 2 * n),
                (good[: good.size - 3], 0, n), (good[: 2 * 32768 + 1], 0, n), (good[: 2 * 32768], 0, n),
                (good[: 3 * 32768 - 1], 0, n), (good[:5000], 0, n), (good, 3, 2 * n), (good[100:], 5, 2 * n),
                (rng.integers(0, 256, 70000, dtype=np.uint8), 0, 70000),
            ]
            # symbols at random: mostly the long codes (len4to15: 15, 12 and 9 bits, walks fall into step only at one of the
            # few others: blocks that are left differently from their guess, a second and third fixing launch)
            flat = rng.integers(0, 256, 100_000).astype(np.uint8)
            streams.append((w.oracle.encode_all(ocoder, flat, slack=64 + 4 * flat.size), 0, flat.size))
            if name == "len4to15":
                # code lengths 9, 12 and 15 only: walks from different entries never fall into step, every block is left
                # differently from its guess and dec_wide_* give the item up by themselves
                apart = rng.integers(28, 256, 60000).astype(np.uint8)
                streams.append((w.oracle.encode_all(ocoder, apart, slack=64 + 4 * apart.size), 0, apart.size))
            decode_items_like_the_oracle(w, eng, ocoder, streams, rng, name, modes)
            # the reference's entry point on a host buffer of that length
            ddo, ddp = w.oracle.new_decoder(ocoder), w.product.new_decoder(pcoder)
            oo, op = np.full(n + 8, SENTINEL, np.uint8), np.full(n + 8, SENTINEL, np.uint8)
            paired_decode(w, ddo, ddp, good, 0, good.size, oo, op, 0, n)
            eng.close()
    finally:
        w.product.lib.aws_huffman_amd_testing_set_wide_min_bytes(0)


def never_in_step_stream(w, n=40_000_000, seed=113):
    """ONE long stream whose walks never fall into step (code lengths 9, 12 and 15 only: three phases, each valid for
    ever): dec_wide_settle gives it up and it goes by transfer functions (dec_wide_fn_*), every block a workgroup --
    whole, cut inside a block, damaged two thirds in, entered three bits into its first byte, short of room."""
    rng = np.random.default_rng(seed)
    ocoder, pcoder, _ = profile_coders(w, "len4to15")
    eng = harness.Engine(w.product.lib, pcoder)
    apart = rng.integers(28, 256, n).astype(np.uint8)
    good = w.oracle.encode_all(ocoder, apart, slack=64 + 2 * n)
    assert good.size > 300 * 32768 or n < 1_000_000  # (more blocks than dec_wide_fn_scan has threads: runs of several)
    damaged = good.copy()
    at = 2 * (good.size // 3)
    damaged[at:at + 4] = 0xFF  # (no code starts with 16 ones)
    streams = [(good, 0, n), (good[: good.size // 2 + 11], 0, n), (damaged, 0, n), (good[7:], 3, n), (good, 0, n // 3)]
    decode_items_like_the_oracle(w, eng, ocoder, streams, rng, "never in step", modes=(None,), kinds=2)
    eng.close()


def fixed_length_coders(w, n=70_000, seed=83):
    """Coders whose codes all have one length (8 bits: every window a code; 9 bits: half of them): dec_fixed_* for the
    items beyond a thread's work, whatever their number and size."""
    rng = np.random.default_rng(seed)
    for name in ("len8", "len9"):
        ocoder, pcoder, _ = profile_coders(w, name)
        eng = harness.Engine(w.product.lib, pcoder)
        data = rng.integers(0, 256, n).astype(np.uint8)
        good = w.oracle.encode_all(ocoder, data, slack=64 + 2 * n)
        damaged = good.copy()
        damaged[good.size // 2: good.size // 2 + 3] = 0xFF  # (len9: a window that starts with a one has no code)
        streams = [
            (good, 0, n), (good, 0, n + 5), (good, 0, n // 2), (good, 0, 0), (damaged, 0, n), (damaged, 0, 100),
            (good[: good.size - 1], 0, n), (good[: good.size - 3], 0, n), (good[:16384], 0, n), (good[:16385], 0, n),
            (good[:16383], 0, n), (good[:200], 0, 400), (good[:64], 0, 100), (good, 3, 2 * n), (good[77:], 5, 2 * n),
            (rng.integers(0, 256, 40000, dtype=np.uint8), 0, 40000), (np.full(20000, 0xFF, np.uint8), 1, 40000),
        ] + [(good[o: o + int(rng.integers(129, 3000))], int(rng.integers(0, 8)), 3000) for o in range(0, 40000, 2500)]
        decode_items_like_the_oracle(w, eng, ocoder, streams, rng, name, kinds=2 if name == "len8" else 3)
        # the reference's entry points on host buffers, in pieces too
        ddo, ddp = w.oracle.new_decoder(ocoder), w.product.new_decoder(pcoder)
        oo, op = np.full(n + 8, SENTINEL, np.uint8), np.full(n + 8, SENTINEL, np.uint8)
        r = paired_decode(w, ddo, ddp, good, 0, good.size // 3, oo, op, 0, n // 5)
        paired_decode(w, ddo, ddp, good, r.consumed, good.size, oo, op, r.produced, n)
        eng.close()


def first_bit_offsets(w, engine=None):
    """Decode items that start inside their first byte (what a carried decoder state turns into)."""
    rng = np.random.default_rng(19)
    eng = engine or harness.Engine(w.product.lib, w.pcoder)
    data = inputs(rng, 3000, "uniform")
    enc = oracle_encode(w, data)
    for skip_symbols in (1, 2, 3, 5, 8):
        # drop the first symbols on the oracle side to learn where symbol k starts
        d = w.oracle.new_decoder(w.ocoder)
        dst = np.zeros(skip_symbols, np.uint8)
        r = w.oracle.decode_call(d, enc, 0, enc.size, dst, 0, skip_symbols)
        start_bit = r.consumed * 8 - r.state[0]
        byte0, first_bit = start_bit // 8, start_bit % 8
        tail = enc[byte0:]
        d_in, d_out = eng.alloc(tail.size + 16), eng.alloc(data.size + 16)
        eng.upload(d_in, tail)
        plan = eng.decode_plan([dict(in_offset=0, in_len=tail.size, first_bit=first_bit, out_offset=0,
                                     out_capacity=data.size)])
        eng.decode_launch(plan, d_in, d_out)
        (rc, err, produced, bits), = eng.decode_results(plan, 1)
        assert (rc, err, produced) == (0, 0, data.size - skip_symbols)
        assert np.array_equal(eng.download(d_out, produced), data[skip_symbols:])
        eng.lib.aws_huffman_amd_decode_plan_destroy(plan)
        eng.free(d_in)
        eng.free(d_out)
    if engine is None:
        eng.close()


# ----------------------------------------------------------------------------- scenario: plans of several streams, damaged and short of room
ROAD_TWO_PASS, ROAD_ONE_PASS, ROAD_GAVE_UP = 0, 1, 2


def decode_roads(w, engine=None, sizes=(40_000, 300_000, 1_100_000), seed=53):
    """Plans of several streams (sizes in symbols) at odd offsets: as they come, with one stream damaged in its middle or
    near its front, with one output short by half or by one symbol.  Records, output bytes and guard bytes as the oracle
    has them; the road query answers TWO_PASS (sync + scan + emit: the decoder's one road since the one-pass decoder of
    rounds 3-4 was retired)."""
    own = engine is None
    eng = engine or harness.Engine(w.product.lib, w.pcoder)
    rng = np.random.default_rng(seed)
    plains = [inputs(rng, n, "uniform") for n in sizes]
    streams = [oracle_encode(w, p) for p in plains]

    def run(damage=None, short=None):
        encs = [s.copy() for s in streams]
        if damage is not None:
            k, at = damage
            encs[k][at:at + 4] = 0xFF  # the test coder has no code of ten one bits: an invalid window
        caps = [p.size for p in plains]
        if short is not None:
            k, cap = short
            caps[k] = cap
        in_offs = np.cumsum([0] + [e.size + 5 for e in encs])   # odd gaps: chunks at any alignment
        out_offs = np.cumsum([0] + [c + 16 for c in caps])
        blob = np.full(int(in_offs[-1]) + 64, 0xA5, np.uint8)
        for e, o in zip(encs, in_offs):
            blob[o:o + e.size] = e
        d_in, d_out = eng.alloc(blob.size), eng.alloc(int(out_offs[-1]) + 64)
        eng.upload(d_in, blob)
        eng.fill(d_out, SENTINEL, int(out_offs[-1]) + 64)
        items = [dict(in_offset=int(in_offs[i]), in_len=int(encs[i].size), out_offset=int(out_offs[i]),
                      out_capacity=int(caps[i])) for i in range(len(encs))]
        plan = eng.decode_plan(items)
        eng.decode_launch(plan, d_in, d_out)
        assert eng.decode_road(plan) == ROAD_TWO_PASS
        got = eng.decode_results(plan, len(items))
        back = eng.download(d_out, int(out_offs[-1]) + 64)
        for i, e in enumerate(encs):
            dec = w.oracle.new_decoder(w.ocoder)
            want_out = np.full(caps[i] + 16, SENTINEL, np.uint8)
            r = w.oracle.decode_call(dec, e, 0, e.size, want_out, 0, caps[i])
            assert got[i][0] == r.rc and got[i][1] == r.err and got[i][2] == r.produced, (damage, short, i, got[i], r)
            mine = back[out_offs[i]:out_offs[i] + caps[i] + 16]
            assert np.array_equal(mine, want_out), (damage, short, i, int(np.flatnonzero(mine != want_out)[0]))
        eng.lib.aws_huffman_amd_decode_plan_destroy(plan)
        eng.free(d_in)
        eng.free(d_out)

    run()
    big = int(np.argmax([s.size for s in streams]))
    run(damage=(big, streams[big].size // 2))
    run(damage=(big, 40))
    run(short=(big, plains[big].size // 2))
    run(short=(big, plains[big].size - 1))  # the edge lies in the stream's last chunk
    if own:
        eng.close()


# ----------------------------------------------------------------------------- scenario: which kernels encode
def encode_roads(w, sizes=(200_000, 16384, 40_000, 3_000_000), seed=57):
    """The one-pass encoder (enc_onepass) waits between workgroups, with bounds; when a wait runs out the three-kernel
    road, queued behind it on the same stream, does the launch over.  Every road must give the oracle's bytes and
    records: one plan of several streams (one with a short output: SHORT_BUFFER with the reference's consumed /
    overflow) through engines made
      as they come                                                  -> ONE_PASS
      with aws_huffman_amd_testing_set_encode_road(THREE_KERNEL)    -> TWO_PASS (count / scan / pack)
      with aws_huffman_amd_testing_set_encode_road(ONE_PASS_FAILS)  -> GAVE_UP  (a wave in the middle of the plan made to give up)
    and the output is read straight after the launch, before the records are fetched: it must be whole by then."""
    rng = np.random.default_rng(seed)
    plains = [inputs(rng, n, "uniform") for n in sizes]
    wants = [oracle_encode(w, p) for p in plains]
    caps = [e.size + 9 for e in wants]
    caps[1] = wants[1].size // 2  # a short output
    in_offs = np.cumsum([0] + [p.size + 3 for p in plains])
    out_offs = np.cumsum([0] + [c + 16 for c in caps])
    blob = np.concatenate([np.concatenate([p, np.zeros(3, np.uint8)]) for p in plains])
    for mode, want_road in ((None, ROAD_ONE_PASS), ("three-kernel", ROAD_TWO_PASS), ("one-pass-fails", ROAD_GAVE_UP)):
        with harness.encode_road(w.product.lib, mode):
            coder = w.product.lib.aws_huffman_amd_table_coder_new(*w.table)  # a fresh coder: a fresh engine that reads the switch
            eng = harness.Engine(w.product.lib, coder)
        d_in, d_out = eng.alloc(blob.size + 64), eng.alloc(int(out_offs[-1]) + 64)
        eng.upload(d_in, blob)
        eng.fill(d_out, SENTINEL, int(out_offs[-1]) + 64)
        items = [dict(in_offset=int(in_offs[i]), in_len=int(plains[i].size), out_offset=int(out_offs[i]),
                      out_capacity=int(caps[i])) for i in range(len(plains))]
        plan = eng.encode_plan(items)
        eng.encode_launch(plan, d_in, d_out)
        got = eng.download(d_out, int(out_offs[-1]) + 64)  # (behind the launch on the stream, before any record is read)
        res = eng.encode_results(plan, len(items))
        assert eng.encode_road(plan) == want_road, (mode, eng.encode_road(plan))
        for i, e in enumerate(wants):
            enc = w.oracle.new_encoder(w.ocoder)
            ref_out = np.full(caps[i] + 16, SENTINEL, np.uint8)
            r = w.oracle.encode_call(enc, plains[i], 0, ref_out, 0, caps[i])
            rc, err, consumed, produced, ob, op = res[i]
            assert (rc, err, consumed, produced) == (r.rc, r.err, r.consumed, r.produced), (mode, i, res[i], r)
            assert (ob, op if ob else 0) == r.state, (mode, i, res[i], r)
            mine = got[out_offs[i]:out_offs[i] + caps[i] + 16]
            assert np.array_equal(mine, ref_out), (mode, i, int(np.flatnonzero(mine != ref_out)[0]))
        eng.lib.aws_huffman_amd_encode_plan_destroy(plan)
        eng.free(d_in)
        eng.free(d_out)
        eng.close()
        w.product.lib.aws_huffman_amd_table_coder_destroy(coder)


# ----------------------------------------------------------------------------- scenario: a coder with one callback, or whose two disagree in range
def one_sided_coders(w, n=50_000, seed=63):
    """The reference's decoder only ever calls `decode` (source/huffman.c:235-238), its encoder only `encode` (:60):
    a coder object may lack the other callback, and the decoder may know more codes than the encoder uses.
      * decode-only: the test coder's decode callback alone (encode = NULL) decodes the oracle's streams, in one call
        and in pieces, and stops like the oracle on damaged ones;
      * encode-only: the encode callback alone encodes the oracle's stream;
      * a decoder that knows more: encode from the test coder WITH HOLES (symbols 7 and 200 have no code), decode
        from the full test coder -- streams that contain 7 and 200 still decode (the decode tables, the shortest code
        and the number of entry states come from the decode callback, not from the encode table)."""
    rng = np.random.default_rng(seed)
    data = inputs(rng, n, "uniform")
    want = oracle_encode(w, data)
    split = w.oracle.lib.oracle_split_coder_new
    split.restype, split.argtypes = C.POINTER(harness.SymbolCoder), [C.POINTER(harness.SymbolCoder)] * 2
    dec_only, enc_only = split(None, w.ocoder), split(w.ocoder, None)
    mixed = split(w.ocoder_holes, w.ocoder)
    assert dec_only and enc_only and mixed and not dec_only.contents.encode and not enc_only.contents.decode
    # decode-only
    r, back = w.product.decode_all(dec_only, want, n)
    assert r.rc == 0 and r.produced == n and np.array_equal(back, data)
    ddo, ddp = w.oracle.new_decoder(w.ocoder), w.product.new_decoder(dec_only)
    oo, op = np.full(n + 8, SENTINEL, np.uint8), np.full(n + 8, SENTINEL, np.uint8)
    pos_o = pos_p = 0
    for lo in range(0, want.size, 7001):  # in pieces: carried window bits
        hi = min(lo + 7001, want.size)
        ro = w.oracle.decode_call(ddo, want, lo, hi, oo, pos_o, n)
        rp = w.product.decode_call(ddp, want, lo, hi, op, pos_p, n)
        assert rp.key() == ro.key(), (lo, rp, ro)
        pos_o += ro.produced
        pos_p += rp.produced
    assert np.array_equal(oo, op)
    bad = want.copy()
    bad[bad.size // 3: bad.size // 3 + 5] = 0xFF
    ro, _ = w.oracle.decode_all(w.ocoder, bad, n)
    rp, _ = w.product.decode_all(dec_only, bad, n)
    assert rp.key()[:4] == ro.key()[:4], (rp, ro)
    enc = w.product.new_encoder(dec_only)
    dst = np.zeros(64, np.uint8)
    r = w.product.encode_call(enc, data[:10].copy(), 0, dst, 0, 64)
    assert r.rc != 0  # nothing to encode with
    # encode-only
    got = w.product.encode_all(enc_only, data)
    assert np.array_equal(got, want)
    # the decoder knows more than the encoder
    few = data.copy()
    few[(few == 7) | (few == 200)] = 9
    assert np.array_equal(w.product.encode_all(mixed, few), oracle_encode(w, few))
    r, back = w.product.decode_all(mixed, want, n)  # (a stream with 7s and 200s in it)
    assert r.rc == 0 and np.array_equal(back, data)
    # an encoder with codes of up to 30 bits beside a decoder whose table has 10: the decode kernels' entry states, walk
    # tables and certain steps follow the DECODE table (the chunk kernels hold 12 entry states, 16 on the long way; with
    # the encoder's 30 they would overrun), the packer and its images the encode table
    for name in ("len2to30", "hpack_lengths", "len1to16"):
        long_oc, _, lengths = profile_coders(w, name)
        odd = split(long_oc, w.ocoder)
        assert odd
        plain = rng.integers(0, 256, 20_000).astype(np.uint8)
        got = w.product.encode_all(odd, plain, slack=64 + 4 * plain.size)
        assert np.array_equal(got, w.oracle.encode_all(long_oc, plain, slack=64 + 4 * plain.size)), name
        for stream, size in ((want, n), (oracle_encode(w, inputs(rng, 70_000, "constant")), 70_000)):
            ro, back_o = w.oracle.decode_all(w.ocoder, stream, size)  # (regular chunks; a stream whose phases never meet: the long way)
            rp, back_p = w.product.decode_all(odd, stream, size)
            assert rp.key() == ro.key() and np.array_equal(back_p, back_o), (name, rp, ro)
        ro, _ = w.oracle.decode_all(w.ocoder, bad, n)
        rp, _ = w.product.decode_all(odd, bad, n)
        assert rp.key()[:4] == ro.key()[:4], (name, rp, ro)
        w.oracle.lib.oracle_split_coder_destroy(odd)
    for c in (dec_only, enc_only, mixed):
        w.oracle.lib.oracle_split_coder_destroy(c)


# ----------------------------------------------------------------------------- scenario: growth that fails
def failed_growth(w, seed=67):
    """allow_growth with a buffer that cannot grow (no allocator: aws_byte_buf_reserve_relative fails): the reference
    meets the failed reserve with the buffer full -- the symbols before it are stored, the decoder has moved past
    them, the call returns the reserve's error (source/huffman.c:257-264,275).  Same observable state, then a second
    call into a fresh buffer finishes the stream."""
    rng = np.random.default_rng(seed)
    for n, cap in ((100, 16), (5000, 64), (40000, 1000), (40000, 0)):
        data = inputs(rng, n, "uniform")
        enc = oracle_encode(w, data)
        outs = []
        for codec, coder in ((w.oracle, w.ocoder), (w.product, w.pcoder)):
            d = codec.new_decoder(coder)
            codec.decoder_allow_growth(d, True)
            dst = np.full(n + 8, SENTINEL, np.uint8)
            buf = harness.ByteBuf(0, dst.ctypes.data, cap, None)  # capacity `cap`, nothing to grow with
            cur = harness.ByteCursor(enc.size, enc.ctypes.data)
            codec.reset_error()
            rc = codec._decode(C.byref(d), C.byref(cur), C.byref(buf))
            err = codec.last_error() if rc else 0
            first = (rc, err, enc.size - cur.len, buf.len, buf.capacity, d.num_bits, d.working_bits, bytes(dst))
            # the rest of the stream into a buffer that is large enough
            rest = np.full(n + 8, SENTINEL, np.uint8)
            codec.decoder_allow_growth(d, False)
            r = codec.decode_call(d, enc, enc.size - cur.len, enc.size, rest, 0, n)
            outs.append((first, r.key(), bytes(rest)))
        assert outs[0] == outs[1], (n, cap, outs[0][0][:7], outs[1][0][:7], outs[0][1], outs[1][1])
        if cap:
            assert outs[0][0][0] != 0 and outs[0][0][3] == cap  # (the oracle itself: an error, the buffer full)
            assert outs[0][0][7][:cap] + outs[0][2][:n - cap] == bytes(data)


# ----------------------------------------------------------------------------- scenario: inputs longer than a device item
def long_inputs_in_pieces(w, n=60_000, seed=71):
    """aws_huffman_decode of an input longer than a device item takes (4 GiB) splits it inside the call (2 GiB
    pieces, the window carried over).  With the pieces made small the same call must still match the oracle's ONE
    call: whole streams, a damaged stream (the error and where it is raised), a short output, arbitrary bytes."""
    rng = np.random.default_rng(seed)
    data = inputs(rng, n, "uniform")
    good = oracle_encode(w, data)
    bad = good.copy()
    bad[good.size // 2: good.size // 2 + 6] = 0xFF
    noise = rng.integers(0, 256, 20_000).astype(np.uint8)
    try:
        for piece in (4099, 1000, 131, 9):
            w.product.lib.aws_huffman_amd_testing_set_decode_piece_bytes(piece)
            for stream, cap in ((good, n), (good, n // 2), (bad, n), (noise, n), (good[:good.size - 3], n)):
                oo, op = np.full(n + 8, SENTINEL, np.uint8), np.full(n + 8, SENTINEL, np.uint8)
                ddo, ddp = w.oracle.new_decoder(w.ocoder), w.product.new_decoder(w.pcoder)
                paired_decode(w, ddo, ddp, stream, 0, stream.size, oo, op, 0, cap)
    finally:
        w.product.lib.aws_huffman_amd_testing_set_decode_piece_bytes(0)
