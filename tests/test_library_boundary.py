"""CPU-only checks of the drop-in boundary: the shared library loads, exports every symbol
the headers under include/ declare, keeps the reference's struct layouts, and fails loudly
(no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re
import subprocess

import pytest

import harness


@pytest.fixture(scope="module")
def product_lib():
    if not os.path.exists(harness.PRODUCT_SO):
        import __graft_entry__

        __graft_entry__.build()
    return harness.load_product()


def declared_symbols():
    names = set()
    inc = os.path.join(harness.REPO, "include", "aws", "compression")
    for header in ("huffman.h", "huffman_amd.h", "compression.h"):
        text = open(os.path.join(inc, header)).read()
        for m in re.finditer(r"AWS_COMPRESSION_API\s+[^;]*?\b(aws_\w+)\s*\(", text, re.S):
            names.add(m.group(1))
    return names


def test_every_declared_symbol_is_exported(product_lib):
    declared = declared_symbols()
    assert len(declared) >= 35
    assert declared == set(harness.EXPORTED_SYMBOLS)
    out = subprocess.check_output(["nm", "-D", "--defined-only", harness.PRODUCT_SO], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    missing = declared - exported
    assert not missing, "declared in include/ but not exported: %s" % sorted(missing)
    # the reference's coder-testing helpers (include/aws/compression/private/huffman_testing.h) keep their names
    text = open(os.path.join(harness.REPO, "include", "aws", "compression", "private", "huffman_testing.h")).read()
    helpers = set(re.findall(r"AWS_COMPRESSION_API\s+int\s+(huffman_\w+)\s*\(", text))
    assert helpers == set(harness.TESTING_SYMBOLS) and helpers <= exported
    # nothing from the oracle or the emulator leaks into the product
    assert not any(s.startswith("oracle_") or "hip_emu" in s for s in exported)


def test_abi_layout():
    harness.check_abi_layout()


def test_init_and_reset_need_no_gpu(product_lib):
    codec = harness.Codec(product_lib, "aws_")
    patterns, lens = harness.load_table()
    coder = product_lib.aws_huffman_amd_table_coder_new(patterns, lens)
    assert coder
    e = codec.new_encoder(coder)
    assert e.eos_padding == 0xFF and e.overflow_bits.num_bits == 0
    e.overflow_bits.num_bits = 5
    codec.encoder_reset(e)
    assert e.overflow_bits.num_bits == 0
    d = codec.new_decoder(coder)
    assert not d.allow_growth and d.num_bits == 0 and d.working_bits == 0
    codec.decoder_allow_growth(d, True)
    d.num_bits, d.working_bits = 7, 1 << 63
    codec.decoder_reset(d)
    assert d.allow_growth and d.num_bits == 0 and d.working_bits == 0


def test_library_init_registers_the_error_name(product_lib):
    """reference tests/library_test.c:9-22: after aws_compression_library_init, aws_error_name of
    AWS_ERROR_COMPRESSION_UNKNOWN_SYMBOL is that very string -- stand-alone too, through the registry of
    csrc/host/common_compat.c; after clean-up the name is gone again, and both calls are idempotent."""
    lib = product_lib
    lib.aws_error_name.restype, lib.aws_error_name.argtypes = C.c_char_p, [C.c_int]
    lib.aws_error_str.restype, lib.aws_error_str.argtypes = C.c_char_p, [C.c_int]
    code = harness.AWS_ERROR_COMPRESSION_UNKNOWN_SYMBOL
    assert code == 0x0C00
    lib.aws_compression_library_clean_up()
    assert lib.aws_error_name(code) == b"Unknown Error Code"
    lib.aws_compression_library_init(lib.aws_default_allocator())
    lib.aws_compression_library_init(lib.aws_default_allocator())
    assert lib.aws_error_name(code) == b"AWS_ERROR_COMPRESSION_UNKNOWN_SYMBOL"
    assert lib.aws_error_str(code) == b"Compression encountered an unknown symbol."
    assert lib.aws_error_name(harness.AWS_ERROR_SHORT_BUFFER) == b"AWS_ERROR_SHORT_BUFFER"
    lib.aws_compression_library_clean_up()
    lib.aws_compression_library_clean_up()
    assert lib.aws_error_name(code) == b"Unknown Error Code"


def test_table_coder_matches_the_reference_table(product_lib):
    rows = harness.load_json("test_coder_table.json")["rows"]
    tree = harness.load_json("test_coder_decode_tree.json")
    patterns, lens = harness.load_table()
    coder = product_lib.aws_huffman_amd_table_coder_new(patterns, lens)
    enc = harness.ENCODE_FN(coder.contents.encode)
    dec = harness.DECODE_FN(coder.contents.decode)
    for r in rows:
        code = enc(r["symbol"], coder.contents.userdata)
        assert (code.pattern, code.num_bits) == (r["pattern"], r["num_bits"])
    sym = C.c_uint8()
    for leaf in tree["leaves"]:
        p = leaf["prefix"]
        for fill in (0, (1 << (32 - len(p))) - 1):
            assert dec((int(p, 2) << (32 - len(p))) | fill, C.byref(sym), coder.contents.userdata) == leaf["num_bits"]
            assert sym.value == leaf["symbol"]
    for p in tree["dead_ends"]:
        for fill in (0, (1 << (32 - len(p))) - 1):
            assert dec((int(p, 2) << (32 - len(p))) | fill, C.byref(sym), coder.contents.userdata) == 0
    # not a prefix code: rejected
    bad = (C.c_uint8 * 256)(*lens)
    bad_p = (C.c_uint32 * 256)(*patterns)
    bad_p[1], bad[1] = bad_p[0] >> 1, bad[0] - 1  # a prefix of symbol 0's code
    assert not product_lib.aws_huffman_amd_table_coder_new(bad_p, bad)


def def_text(rows, extra=""):
    """The text of a table .def file for these rows, laid out the way the reference's tables are."""
    lines = ["/**", " * a table", " */", "", "#ifndef HUFFMAN_CODE", '#error "Macro HUFFMAN_CODE must be defined before including this header file!"',
             "#endif", "", "/*           sym          bits   code len */"]
    for sym, pattern, n in rows:
        bits = format(pattern, "0%db" % n) if n else ""
        lines.append('HUFFMAN_CODE(%3d, %20s, 0x%x, %d)' % (sym, '"%s"' % bits, pattern, n))
    return ("\n".join(lines) + "\n" + extra).encode()


def test_table_coder_from_def_text(product_lib):
    """huffman_amd.h aws_huffman_amd_table_coder_from_def: the generator's input format (generator.c:42-104)."""
    rows = [(r["symbol"], r["pattern"], r["num_bits"]) for r in harness.load_json("test_coder_table.json")["rows"]]
    text = def_text(rows, extra="// HUFFMAN_CODE(1, \"0\", 0x0, 1) in a comment is not a row\nHUFFMAN_CODE(256, \"1\", 0x3fffffff, 30)\n")
    coder = product_lib.aws_huffman_amd_table_coder_from_def(text, len(text))
    assert coder
    enc = harness.ENCODE_FN(coder.contents.encode)
    dec = harness.DECODE_FN(coder.contents.decode)
    sym = C.c_uint8()
    for s_, pattern, n in rows:
        code = enc(s_, coder.contents.userdata)
        assert (code.pattern, code.num_bits) == (pattern, n)
        assert dec(pattern << (32 - n), C.byref(sym), coder.contents.userdata) == n and sym.value == s_
    product_lib.aws_huffman_amd_table_coder_destroy(coder)
    # a symbol listed twice, a malformed row, no rows at all, not a prefix code
    for bad in (def_text(rows + [rows[3]]), def_text(rows)[:-9], b"/* nothing */\n",
                def_text([(0, 0b0, 1), (1, 0b01, 2)])):
        product_lib.aws_reset_error()
        assert not product_lib.aws_huffman_amd_table_coder_from_def(bad, len(bad))
        assert product_lib.aws_last_error() == 34  # AWS_ERROR_INVALID_ARGUMENT
    # symbols missing from the file have no code
    some = def_text(rows[:100])
    coder = product_lib.aws_huffman_amd_table_coder_from_def(some, len(some))
    enc = harness.ENCODE_FN(coder.contents.encode)
    assert enc(99, coder.contents.userdata).num_bits == rows[99][2] and enc(100, coder.contents.userdata).num_bits == 0
    product_lib.aws_huffman_amd_table_coder_destroy(coder)


def test_fails_loudly_without_a_gpu(product_lib):
    if product_lib.aws_huffman_amd_device_count() > 0:
        pytest.skip("a GPU is present")
    patterns, lens = harness.load_table()
    coder = product_lib.aws_huffman_amd_table_coder_new(patterns, lens)
    h = C.c_void_p()
    assert product_lib.aws_huffman_amd_engine_new(C.byref(h), coder, -1) == -1
    assert product_lib.aws_last_error() == 6  # AWS_ERROR_UNSUPPORTED_OPERATION
    codec = harness.Codec(product_lib, "aws_")
    import numpy as np

    src = np.frombuffer(b"abc", dtype=np.uint8)
    dst = np.zeros(8, np.uint8)
    r = codec.encode_call(codec.new_encoder(coder), src, 0, dst, 0, 8)
    assert r.rc == -1 and r.err == 6 and r.consumed == 0 and r.produced == 0  # no silent CPU encode


@pytest.fixture(scope="module")
def kernel_listing(tmp_path_factory):
    """The kernels compiled to assembly for gfx950 (the code-object metadata is what the tests below read): every
    translation unit of csrc/hip, listings and sources concatenated."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    hip_dir = os.path.join(harness.REPO, "aws-c-compression_amd", "csrc", "hip")
    out_dir = tmp_path_factory.mktemp("asm")
    sources, jobs = [], []
    for name in sorted(os.listdir(hip_dir)):
        if name.endswith(".hip") and name != "hip_shim.hip":
            src = os.path.join(hip_dir, name)
            asm = out_dir / (name + ".s")
            jobs.append((asm, subprocess.Popen(
                [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S",
                 "-I" + os.path.join(harness.REPO, "include"), "-I" + os.path.join(harness.REPO, "include", "compat"),
                 src, "-o", str(asm)], stderr=subprocess.DEVNULL)))
            sources.append(open(src).read())
    for name in sorted(os.listdir(hip_dir)):
        if name.endswith(".hpp"):
            sources.append(open(os.path.join(hip_dir, name)).read())
    listing = ""
    for asm, job in jobs:
        assert job.wait() == 0, asm
        listing += open(asm).read()
    return "\n".join(sources), listing


def test_onepass_kernels_scalar_registers(kernel_listing):
    """The grid of enc_onepass is sized to be resident as a whole: by the occupancy query AND by the scalar-register rule
    the query does not know (persistent_grid, kOnepassSgprs in csrc/hip).  The constant must cover what the build really
    uses: `.sgpr_count` of every instantiation."""
    src, listing = kernel_listing
    declared = int(re.search(r"constexpr uint32_t kOnepassSgprs = (\d+);", src).group(1))
    counts = {}
    name = None
    for line in listing.splitlines():
        m = re.match(r"\s+\.name:\s+(\S+)", line)
        if m:
            name = m.group(1)
        m = re.match(r"\s+\.sgpr_count:\s+(\d+)", line)
        if m and name and "enc_onepass_kernel" in name:
            counts[name] = int(m.group(1))
    assert len(counts) >= 2, counts
    assert max(counts.values()) <= declared, (declared, counts)


def test_kernel_census(kernel_listing):
    """What a maintainer links: at most 90 kernels (round 4 shipped 130, 51 of them instantiations of experiments that had
    measured slower and of builds no coder selects)."""
    _, listing = kernel_listing
    names = set(re.findall(r"\.name:\s+(\S+)", listing))
    names = {n for n in names if not n.endswith(".kd")}
    assert 40 <= len(names) <= 90, len(names)
    assert not [n for n in names if "dec_onepass" in n or "dec_sync_bank" in n or "dec_sync_resident" in n or "dec_sync_fast" in n]


def test_kernels_spell_nothing_differently_for_the_cpu_build():
    """What the GPU build and the CPU test build (tests/emu) spell differently -- inline assembly, address-space casts,
    scalar broadcasts -- is in the two headers of primitives (kernels_common.hpp, decode_common.hpp), each with a plain C++
    twin beside it: no kernel source has an arm of its own for the emulator."""
    hip = os.path.join(harness.REPO, "aws-c-compression_amd", "csrc", "hip")
    for name in sorted(n for n in os.listdir(hip) if n.endswith((".hip", ".hpp", ".h", ".inc"))):
        text = open(os.path.join(hip, name)).read()
        arms = text.count("__HIP_DEVICE_COMPILE__")
        if name in ("kernels_common.hpp", "decode_common.hpp"):
            assert arms >= 1, name
        else:
            assert arms == 0, (name, arms)


def test_no_kernel_spills_vector_registers(kernel_listing):
    """No kernel of the library has scratch memory or a spilled vector register (profiles/tools/spill_census.py prints
    the same table): a value spilled inside these kernels' divergent walks once came back wrong, and a spill in a
    look-back kernel waits for every poll in flight.  Scalar registers parked in vector lanes are not memory."""
    _, listing = kernel_listing
    rows = re.findall(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", listing, re.S)
    assert len(rows) >= 60
    bad = [(n, int(scratch), int(spills)) for n, scratch, spills in rows if int(scratch) or int(spills)]
    assert not bad, bad
