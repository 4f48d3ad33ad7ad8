"""CPU-only logic tests of the HIP kernels through the fiber emulator (tests/emu).

The product sources are compiled unchanged against tests/emu/hip/hip_runtime.h with
UBSan and every parity scenario is run against the oracle.  This is the "sanitizers on
the CPU build" leg (GPU sanitizers are unavailable on the pool) and the way kernel logic
is debugged in a container without a GPU.  It is NOT a parity claim for the GPU build:
tests/test_gpu_parity.py makes that, on an MI355X.
"""
import os
import subprocess

import pytest

import harness
import parity_cases as pc

EMU_DIR = os.path.join(harness.REPO, "tests", "emu")
EMU_SO = os.path.join(EMU_DIR, "libaws-c-compression-emu.so")


@pytest.fixture(scope="module")
def world(oracle):
    subprocess.check_call(["make", "-s", "-C", EMU_DIR], stdout=subprocess.DEVNULL)
    product = harness.Codec(harness.load_product(EMU_SO), "aws_")
    return pc.World(oracle, product)


def test_reference_unit_tests(world):
    pc.reference_unit_tests(world.product, world.pcoder)


def test_one_shot_roundtrips(world):
    pc.one_shot_roundtrips(world, sizes=[1, 2, 15, 16, 17, 255, 4096, 16383, 16384, 16385, 40000])


def test_streaming_encode(world):
    pc.streaming_encode(world, sizes=[1, 40, 5000, 33000])


def test_streaming_decode(world):
    pc.streaming_decode(world, sizes=[1, 40, 5000, 60000])


def test_unknown_symbols(world):
    pc.unknown_symbols(world)


def test_garbage_decode(world):
    pc.garbage_decode(world)


def test_block_decode_calls(world):
    pc.block_decode_calls(world)


def test_wide_long_code_items(world):
    pc.wide_long_code_items(world)


def test_streams_out_of_step(world):
    pc.streams_out_of_step(world)


def test_never_in_step_stream(world):
    pc.never_in_step_stream(world, n=400_000)


def test_fixed_length_coders(world):
    pc.fixed_length_coders(world)


def test_eos_padding_values(world):
    pc.eos_padding_values(world)


def test_null_empty_cursors(world):
    pc.null_empty_cursors(world)


def test_many_header_sized_items(world):
    pc.many_header_sized_items(world)


def test_encode_then_decode_on_the_device(world):
    pc.encode_then_decode_on_the_device(world, batches=((40, 50), (7300, 90)))


def test_mid_sized_items(world):
    pc.mid_sized_items(world)


def test_plans_one_after_another(world):
    pc.plans_one_after_another(world)


def test_streams_with_two_last_chunks(world):
    pc.streams_with_two_last_chunks(world)


def test_few_ends_among_many_chunks(world):
    pc.few_ends_among_many_chunks(world)


def test_quiet_plans(world):
    pc.quiet_plans(world, n=150_000)


def test_plans_made_on_the_device(world):
    pc.plans_made_on_the_device(world, big=1_200_000, n_small=120)



def test_survey_records(world):
    pc.survey_records_on_product(world)


def test_transitive_helpers(world):
    pc.transitive_helpers(world)


def test_shared_coder_threads(world):
    pc.shared_coder_threads(world, calls=12)


def test_foreign_coder_callbacks(world):
    pc.foreign_coder_callbacks(world)


def test_recreated_coders(world):
    pc.recreated_coders(world)


def test_sharded_items(world):
    pc.sharded_items(world, devices=(0, 0, 0), n_items=14, item_len=4096)


def test_batched_device_api(world):
    pc.batched_device_api(world)


def test_tiny_encode_items(world):
    pc.tiny_encode_items(world, n_items=400)  # a handful: one thread only up to 128 symbols, segments above
    # 18 items per byte of the longest: one thread each (the GPU tests do this with items of up to 512 and 1200 symbols)
    pc.tiny_encode_items(world, n_items=3700, seed=40, max_len=200, edge_lens=False)
    pc.tiny_encode_items(world, n_items=400, seed=38, holes=True)
    # a thread per item up to 1024 symbols (the rule asks for 100 items per byte of the longest: set aside by the tests' switch)
    with harness.items_per_byte(world.product.lib, encode=1):
        pc.tiny_encode_items(world, n_items=1200, seed=46, max_len=1100, thread_limit=1024)
        pc.tiny_encode_items(world, n_items=2300, seed=47, holes=True, max_len=2100, thread_limit=2048)  # (a coder with holes keeps to count / scan / pack: its class ends at 2048)


def test_mid_sized_encode_items(world):
    """Items either side of one tile (HUFD_ENC_SOLO_BYTES = 4096 symbols): a wave each without segments below it where the
    coder encodes in one pass, segments above; every kind of stop of the short items' scenario, and the same items with
    the one-pass road's look-back given up and with a coder that takes count / scan / pack."""
    edges = (4095, 4096, 4097, 8191, 8192, 8193, 12288, 16383, 16384, 16385)
    # (256 items or more: up to a segment, four tiles one after the other by the same wave)
    pc.tiny_encode_items(world, n_items=300, seed=157, max_len=20000, edge_lens=False, wave_limit=16384, more_lens=edges)
    pc.tiny_encode_items(world, n_items=220, seed=151, max_len=9000, edge_lens=False, wave_limit=4096, more_lens=edges)
    # (codes of up to 15 bits: the packing kernel's other build)
    pc.tiny_encode_items(world, n_items=270, seed=171, max_len=20000, edge_lens=False, wave_limit=16384, more_lens=edges, profile="len4to15")
    pc.tiny_encode_items(world, n_items=90, seed=152, max_len=4000, edge_lens=False, wave_limit=4096)  # (no segments at all)
    pc.tiny_encode_items(world, n_items=120, seed=153, holes=True, max_len=9000, edge_lens=False, wave_limit=0, more_lens=edges)
    with harness.encode_road(world.product.lib, "one-pass-fails"):
        pc.tiny_encode_items(world, n_items=120, seed=154, max_len=9000, edge_lens=False, wave_limit=4096, more_lens=edges)
        pc.tiny_encode_items(world, n_items=260, seed=158, max_len=20000, edge_lens=False, wave_limit=16384, more_lens=edges)
    with harness.encode_road(world.product.lib, "three-kernel"):
        pc.tiny_encode_items(world, n_items=120, seed=155, max_len=9000, edge_lens=False, wave_limit=0, more_lens=edges)


def test_long_streams_of_other_coders(world):
    pc.long_streams_of_other_coders(world, names=("len4to12",), n=2_600_000)  # (runs of 32 chunks in this build: three of them)


def test_device_plans_of_other_coders(world):
    pc.device_plans_of_other_coders(world, names=("hpack_lengths", "len8", "len1to16"), batches=((9, 16384), (4200, 60)))


def test_walks_that_never_meet(world):
    pc.walks_that_never_meet(world, runs=(130, 420))


def test_tiny_decode_items(world):
    pc.tiny_decode_items(world, n_items=400)  # a handful: one thread up to 128 bytes, one wave up to 768, chunks above
    pc.tiny_decode_items(world, n_items=3300, seed=44)  # 6 items per byte of the longest: one thread up to 512
    pc.tiny_decode_items(world, n_items=400, seed=42, profile="hpack_lengths")
    with harness.items_per_byte(world.product.lib, decode=1):
        pc.tiny_decode_items(world, n_items=900, seed=48, max_len=900, thread_limit=768)


def test_first_bit_offsets(world):
    pc.first_bit_offsets(world)


def test_other_coders(world):
    pc.other_coders(world, n=60000)


def test_dense_symbols(world):
    pc.dense_symbols(world, n=200_000)


def test_cut_streams(world):
    pc.cut_streams(world, chunks=(1,), step=23, n=40_000)


def test_encode_roads(world):
    """enc_onepass, the three-kernel road when told so, and the three-kernel road behind a one-pass launch that gave up."""
    pc.encode_roads(world, sizes=(70_000, 16384, 40_000, 300_000))


def test_one_sided_coders(world):
    pc.one_sided_coders(world, n=50_000)


def test_failed_growth(world):
    pc.failed_growth(world)


def test_long_inputs_in_pieces(world):
    pc.long_inputs_in_pieces(world)


def test_decode_roads(world):
    """Plans of several streams, one damaged or short of room."""
    pc.decode_roads(world, sizes=(40_000, 90_000, 160_000))


def test_large_items_take_the_workgroup_scan(world):
    """More than HUFD_SCAN_SMALL_MAX (64) segments / chunks per item."""
    pc.one_shot_roundtrips(world, sizes=[16384 * 66 + 3, 32768 * 70], seed=21)


def test_three_kernel_encoder(oracle):
    """Coders the one-pass encoder takes (every symbol coded, codes of 4..15 bits -- the test coder) also have the
    count / scan / pack road (aws_huffman_amd_testing_set_encode_road): it is what the library falls back to when a
    look-back wait runs out, and what every other coder takes.  Same scenarios."""
    product = harness.Codec(harness.load_product(EMU_SO), "aws_")
    with harness.encode_road(product.lib, "three-kernel"):
        w = pc.World(oracle, product)  # fresh coder objects: fresh engines that read the switch
        pc.reference_unit_tests(w.product, w.pcoder)
        pc.one_shot_roundtrips(w, sizes=[1, 17, 4096, 16384, 16385, 40000])
        pc.streaming_encode(w, sizes=[40, 33000])
        pc.unknown_symbols(w)
        pc.batched_device_api(w)




def test_one_pass_encoder_across_rounds(world):
    """The emulator build keeps 4 tiles a look-back group and 4 groups a round (tests/emu/Makefile), so that a few
    hundred KiB cross many group and round boundaries of the one-pass encoder."""
    pc.one_shot_roundtrips(world, sizes=[16384 * 40, 16384 * 37 + 4097], seed=33)
