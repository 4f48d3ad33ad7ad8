"""Parity of the HIP build against the CPU oracle, on an MI355X (`pytest -m gpu`).

Everything goes through the C ABI of aws-c-compression_amd/libaws-c-compression-amd.so.
The scenarios are the ones of tests/parity_cases.py; the sizes here reach the
BASELINE.json configurations (1 GiB single stream, 65 536 x 16 KiB batch).
"""
import hashlib
import os

import numpy as np
import pytest

import harness
import parity_cases as pc

pytestmark = pytest.mark.gpu

PROBE = harness.load_json("survey_probe_records.json")


@pytest.fixture(scope="module")
def world(oracle):
    lib = harness.load_product()
    assert lib.aws_huffman_amd_device_count() >= 1, "no HIP device visible: the product has no CPU path"
    return pc.World(oracle, harness.Codec(lib, "aws_"))


@pytest.fixture(scope="module")
def engine(world):
    eng = harness.Engine(world.product.lib, world.pcoder)
    yield eng
    eng.close()


def test_reference_unit_tests(world):
    pc.reference_unit_tests(world.product, world.pcoder)


def test_one_shot_roundtrips(world):
    pc.one_shot_roundtrips(world, sizes=[1, 2, 15, 16, 17, 255, 4096, 16383, 16384, 16385, 40000, 200001])


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
    pc.never_in_step_stream(world)


def test_fixed_length_coders(world):
    pc.fixed_length_coders(world)


def test_damaged_long_streams(world):
    pc.damaged_long_streams(world)


def test_other_coders(world):
    pc.other_coders(world, n=1_500_000)


def test_cut_streams(world):
    pc.cut_streams(world)


def test_dense_symbols(world):
    pc.dense_symbols(world, n=3_000_000)


def test_eos_padding_values(world):
    pc.eos_padding_values(world)


def test_null_empty_cursors(world):
    pc.null_empty_cursors(world)


def test_many_header_sized_items(world):
    pc.many_header_sized_items(world)


def test_encode_then_decode_on_the_device(world):
    pc.encode_then_decode_on_the_device(world)


def test_mid_sized_items(world):
    pc.mid_sized_items(world)


def test_plans_one_after_another(world):
    pc.plans_one_after_another(world)


def test_streams_with_two_last_chunks(world):
    pc.streams_with_two_last_chunks(world)


def test_few_ends_among_many_chunks(world):
    pc.few_ends_among_many_chunks(world)


def test_quiet_plans(world):
    pc.quiet_plans(world)


def test_plans_made_on_the_device(world):
    pc.plans_made_on_the_device(world)



def test_survey_records(world):
    pc.survey_records_on_product(world, names=("G4K", "G16K", "G16KP", "G1M"))


def test_transitive_helpers(world):
    pc.transitive_helpers(world)


def test_shared_coder_threads(world):
    pc.shared_coder_threads(world)


def test_foreign_coder_callbacks(world):
    pc.foreign_coder_callbacks(world)


def test_one_sided_coders(world):
    pc.one_sided_coders(world, n=400_000)


def test_failed_growth(world):
    pc.failed_growth(world)


def test_long_inputs_in_pieces(world):
    pc.long_inputs_in_pieces(world)


def test_recreated_coders(world):
    pc.recreated_coders(world, rounds=8, n=40000)


def test_sharded_items(world):
    """The in-library multi-GPU driver, here with every shard on this box's one GPU: three engines, three host
    threads, the configs[3] split i mod G (on an 8-GPU node: devices=range(8))."""
    ndev = world.product.lib.aws_huffman_amd_device_count()
    pc.sharded_items(world, devices=tuple(g % ndev for g in range(3)), n_items=60)
    # every GPU of the node once (one on this box, eight on the node the scaling runs use): each engine on its own device,
    # the caller's thread left where it was
    pc.sharded_items(world, devices=tuple(range(min(8, ndev))), n_items=64, seed=72)


def test_batched_device_api(world, engine):
    pc.batched_device_api(world, n_items=40, engine=engine)


def test_tiny_encode_items(world, engine):
    pc.tiny_encode_items(world, engine=engine)  # a handful: one thread up to 128 symbols, segments above
    pc.tiny_encode_items(world, seed=38, holes=True)
    pc.tiny_encode_items(world, n_items=11000, seed=36, engine=engine)  # one thread up to 512
    pc.tiny_encode_items(world, n_items=11000, seed=35, holes=True)


def test_mid_sized_encode_items(world, engine):
    """Items either side of one tile and of one segment (4096 / 16384 symbols): a wave each without segments where the coder
    encodes in one pass -- up to a tile in a small plan (HUFD_ENC_SOLO_BYTES), up to a segment, four tiles one after the
    other, in a plan of 256 items or more (HUFD_ENC_SOLO_MANY_*) --, segments above (the encode twin of
    test_mid_sized_items); with every kind of stop of the short items' scenario, a plan of such items only, a coder that
    takes count / scan / pack, and the two test roads."""
    edges = (4095, 4096, 4097, 8191, 8192, 8193, 12288, 16383, 16384, 16385, 20000)
    pc.tiny_encode_items(world, n_items=2500, seed=151, engine=engine, max_len=20000, edge_lens=False, wave_limit=16384, more_lens=edges)
    pc.tiny_encode_items(world, n_items=5000, seed=152, engine=engine, max_len=4000, edge_lens=False, wave_limit=16384)
    pc.tiny_encode_items(world, n_items=200, seed=156, engine=engine, max_len=9000, edge_lens=False, wave_limit=4096, more_lens=edges)
    # (codes of up to 15 bits: the packing kernel's other build)
    pc.tiny_encode_items(world, n_items=1500, seed=171, max_len=20000, edge_lens=False, wave_limit=16384, more_lens=edges, profile="len4to15")
    pc.tiny_encode_items(world, n_items=150, seed=172, max_len=9000, edge_lens=False, wave_limit=4096, more_lens=edges, profile="len4to15")
    pc.tiny_encode_items(world, n_items=1200, seed=153, holes=True, max_len=9000, edge_lens=False, wave_limit=0, more_lens=edges)
    with harness.encode_road(world.product.lib, "one-pass-fails"):
        pc.tiny_encode_items(world, n_items=1200, seed=154, max_len=20000, edge_lens=False, wave_limit=16384, more_lens=edges)
    with harness.encode_road(world.product.lib, "three-kernel"):
        pc.tiny_encode_items(world, n_items=1200, seed=155, max_len=9000, edge_lens=False, wave_limit=0, more_lens=edges)


def test_long_streams_of_other_coders(world):
    pc.long_streams_of_other_coders(world)


def test_device_plans_of_other_coders(world):
    pc.device_plans_of_other_coders(world)


def test_walks_that_never_meet(world, engine):
    pc.walks_that_never_meet(world, engine=engine)
    pc.walks_that_never_meet(world, engine=engine, seed=138, runs=(200, 333, 1500), modes=(None, "long-way"))


def test_many_short_items_take_one_thread_each(world, engine):
    """A thread per item for a whole class of short items, by the library's own rule (HUFD_*_TINY_PER_BYTE items per byte of
    the class's longest item): the one-pass coder's enc_tiny on items of up to 1024 symbols, dec_tiny on items of up to 768
    encoded bytes -- every kind of stop of the scenarios among them, and the plans say that this is the road they took."""
    pc.tiny_encode_items(world, n_items=125000, seed=39, engine=engine, max_len=1200, thread_limit=1024)
    pc.tiny_decode_items(world, n_items=66000, seed=43, engine=engine, max_len=1000, thread_limit=768)
    # fewer items than the rule asks for, and the rule set aside by the tests' switch: the same road for them
    with harness.items_per_byte(world.product.lib, encode=1, decode=1):
        pc.tiny_encode_items(world, n_items=9000, seed=139, engine=engine, max_len=1100, thread_limit=1024)
        pc.tiny_encode_items(world, n_items=9000, seed=140, holes=True, max_len=2100, thread_limit=2048)  # (a coder with holes keeps to count / scan / pack: its class ends at 2048)
        pc.tiny_decode_items(world, n_items=9000, seed=143, engine=engine, max_len=900, thread_limit=768)
    # and by default such a handful goes by tiles / chunks above 128 symbols / bytes
    pc.tiny_encode_items(world, n_items=3000, seed=141, engine=engine, max_len=1100, thread_limit=128)
    pc.tiny_decode_items(world, n_items=300, seed=144, engine=engine, max_len=900, thread_limit=128)


def test_tiny_decode_items(world, engine):
    pc.tiny_decode_items(world, engine=engine)  # a handful: one thread up to 128 bytes, one wave up to 768
    pc.tiny_decode_items(world, n_items=4000, seed=45, engine=engine)  # one thread up to 512
    pc.tiny_decode_items(world, seed=42, profile="hpack_lengths")


def test_first_bit_offsets(world, engine):
    pc.first_bit_offsets(world, engine=engine)


def test_decode_roads(world, engine):
    """One stream's chunks by the default road and by the tests' roads (the long way for chunks whose walks never become
    one, a workgroup per end-of-stream chunk), whole, damaged and short of room: same records, same bytes, nothing written
    that should not be."""
    pc.decode_roads(world, engine=engine)
    pc.decode_roads(world, engine=engine, sizes=(5_000_000, 33_000, 12_000_000, 70_000), seed=54)


def test_large_items_take_the_workgroup_scan(world):
    pc.one_shot_roundtrips(world, sizes=[16384 * 66 + 3, 32768 * 70, 8 * 1024 * 1024 + 11], seed=21)


def test_config2_config3_one_gib_stream(world, engine):
    """BASELINE.json configs[1] and [2]: 1 GiB of splitmix64(seed 5) bytes, encode then decode.

    Pinned three ways: sha256 of the device-generated input and of the encoded stream against
    the digests SURVEY.md 8c records from the real reference, and the decoded stream against
    the input (encode -> decode round trip at full size).
    """
    rec = PROBE["streams"]["G1G"]
    n, e = rec["len"], rec["encoded_len"]
    d_in, d_enc, d_back = engine.alloc(n), engine.alloc(e + 64), engine.alloc(n + 64)
    engine.fill_splitmix64(d_in, n, rec["seed"])
    engine.fill(d_enc, 0x5A, e + 64)
    engine.fill(d_back, 0x5A, n + 64)
    plan = engine.encode_plan([dict(in_offset=0, in_len=n, out_offset=0, out_capacity=e + 64)])
    engine.encode_launch(plan, d_in, d_enc)
    (rc, err, consumed, produced, ob, op), = engine.encode_results(plan, 1)
    assert (rc, err, consumed, produced, ob) == (0, 0, n, e, 0)
    dplan = engine.decode_plan([dict(in_offset=0, in_len=e, out_offset=0, out_capacity=n)])
    engine.decode_launch(dplan, d_enc, d_back)
    (rc, err, symbols, bits), = engine.decode_results(dplan, 1)
    assert (rc, err, symbols) == (0, 0, n)
    assert e * 8 - bits == rec["decoder_tail_num_bits"]  # padding bits left over
    assert engine.decode_road(dplan) == pc.ROAD_TWO_PASS

    def digest(ptr, size):
        h = hashlib.sha256()
        step = 256 << 20
        for off in range(0, size, step):
            h.update(engine.download(ptr, min(step, size - off), offset=off).tobytes())
        return h.hexdigest()

    assert digest(d_in, n) == rec["sha256_input"]
    assert digest(d_enc, e) == rec["sha256_encoded"]
    assert digest(d_back, n) == rec["sha256_input"]
    assert np.all(engine.download(d_enc, 64, offset=e) == 0x5A)  # nothing past the stream
    assert np.all(engine.download(d_back, 64, offset=n) == 0x5A)
    engine.lib.aws_huffman_amd_encode_plan_destroy(plan)
    engine.lib.aws_huffman_amd_decode_plan_destroy(dplan)
    for p in (d_in, d_enc, d_back):
        engine.free(p)


def test_config4_batch_of_16k_buffers(world, engine):
    """BASELINE.json configs[3]: 65 536 buffers x 16 KiB, every fourth one capacity-limited.

    Buffer i is splitmix64(seed 2 + i); buffers with i % 4 == 0 get 16 384 bytes of room and
    must come back SHORT_BUFFER with the reference's (consumed, overflow) record -- buffer 0 is
    the one SURVEY.md 8c pins -- then finish in a second call; the rest encode in one call.
    A sample of buffers is compared byte for byte with the oracle, and the whole batch is
    decoded back and compared with the input.
    """
    count, size = 65536, 16384
    d_in = engine.alloc(count * size)
    host_in = np.empty((count, size), np.uint8)
    for i in range(count):
        host_in[i] = harness.splitmix64_bytes(2 + i, size)
    engine.upload(d_in, host_in.reshape(-1))
    stride = 2 * size
    d_out = engine.alloc(count * stride)
    engine.fill(d_out, 0x5A, count * stride)
    items = [dict(in_offset=i * size, in_len=size, out_offset=i * stride,
                  out_capacity=size if i % 4 == 0 else stride) for i in range(count)]
    plan = engine.encode_plan(items)
    engine.encode_launch(plan, d_in, d_out)
    res = engine.encode_results(plan, count)
    first = PROBE["G16K_partial_encode"][2]
    assert res[0] == (-1, harness.AWS_ERROR_SHORT_BUFFER, first["consumed"], first["out_len"],
                      first["overflow_num_bits"], first["overflow_pattern"])
    assert all(r[0] == 0 and r[2] == size for i, r in enumerate(res) if i % 4)
    assert all(r[0] == -1 and r[1] == harness.AWS_ERROR_SHORT_BUFFER and r[3] == size
               for i, r in enumerate(res) if i % 4 == 0)
    # second call for the capacity-limited ones: the rest of the input, carried overflow, fresh room
    resume = [dict(in_offset=i * size + res[i][2], in_len=size - res[i][2], out_offset=i * stride + size,
                   out_capacity=size, overflow_in=(res[i][5], res[i][4])) for i in range(0, count, 4)]
    plan2 = engine.encode_plan(resume)
    engine.encode_launch(plan2, d_in, d_out)
    res2 = engine.encode_results(plan2, len(resume))
    assert all(r[0] == 0 for r in res2)
    lengths = [res[i][3] + (res2[i // 4][3] if i % 4 == 0 else 0) for i in range(count)]
    # a sample against the oracle, byte for byte, plus the guard bytes behind each stream
    rng = np.random.default_rng(5)
    for i in [0, 1, 2, 3, 4, count - 1] + [int(x) for x in rng.integers(0, count, 40)]:
        want = world.oracle.encode_all(world.ocoder, host_in[i])
        got = engine.download(d_out, stride, offset=i * stride)
        assert lengths[i] == want.size and np.array_equal(got[: want.size], want), i
        assert np.all(got[want.size:] == 0x5A), i
    # decode everything back
    d_back = engine.alloc(count * size)
    ditems = [dict(in_offset=i * stride, in_len=lengths[i], out_offset=i * size, out_capacity=size)
              for i in range(count)]
    dplan = engine.decode_plan(ditems)
    engine.decode_launch(dplan, d_out, d_back)
    dres = engine.decode_results(dplan, count)
    assert all(r[0] == 0 and r[2] == size for r in dres)
    back = engine.download(d_back, count * size)
    assert np.array_equal(back, host_in.reshape(-1))
    for p in (plan, plan2):
        engine.lib.aws_huffman_amd_encode_plan_destroy(p)
    engine.lib.aws_huffman_amd_decode_plan_destroy(dplan)
    for p in (d_in, d_out, d_back):
        engine.free(p)


def test_config5_streams_of_other_ranks(engine):
    """BASELINE.json configs[4]: rank g of an N-GPU run encodes and decodes splitmix64(seed 5 + g).  The streams of the
    first and the last other rank (seeds 6 and 12), on this one GPU, against the pinned oracle's records
    (tests/golden/config5_stream_pins.json, made by make_config5_pins.py -- whose seed 5 reproduces the survey's record of
    the real reference): length and sha256 of the encoded stream, and the round trip."""
    pins = harness.load_json("config5_stream_pins.json")
    n = pins["len"]
    d_in, d_enc, d_back = engine.alloc(n), engine.alloc(n * 10 // 8 + 64), engine.alloc(n + 64)

    def digest(ptr, size):
        h = hashlib.sha256()
        for off in range(0, size, 256 << 20):
            h.update(engine.download(ptr, min(256 << 20, size - off), offset=off).tobytes())
        return h.hexdigest()

    for seed in (6, 12):
        rec = pins["streams"][str(seed)]
        e = rec["encoded_len"]
        engine.fill_splitmix64(d_in, n, seed)
        engine.fill(d_enc, 0x5A, e + 64)
        engine.fill(d_back, 0x5A, n + 64)
        plan = engine.encode_plan([dict(in_offset=0, in_len=n, out_offset=0, out_capacity=n * 10 // 8 + 64)])
        engine.encode_launch(plan, d_in, d_enc)
        (rc, err, consumed, produced, ob, op), = engine.encode_results(plan, 1)
        assert (rc, err, consumed, produced, ob) == (0, 0, n, e, 0), seed
        dplan = engine.decode_plan([dict(in_offset=0, in_len=e, out_offset=0, out_capacity=n)])
        engine.decode_launch(dplan, d_enc, d_back)
        (rc, err, symbols, bits), = engine.decode_results(dplan, 1)
        assert (rc, err, symbols) == (0, 0, n), seed
        assert digest(d_in, n) == rec["sha256_input"], seed
        assert digest(d_enc, e) == rec["sha256_encoded"], seed
        assert digest(d_back, n) == rec["sha256_input"], seed
        assert np.all(engine.download(d_enc, 64, offset=e) == 0x5A) and np.all(engine.download(d_back, 64, offset=n) == 0x5A)
        engine.lib.aws_huffman_amd_encode_plan_destroy(plan)
        engine.lib.aws_huffman_amd_decode_plan_destroy(dplan)
    for p in (d_in, d_enc, d_back):
        engine.free(p)


def test_three_kernel_encoder(oracle):
    """The count / scan / pack road (aws_huffman_amd_testing_set_encode_road) on the GPU: what every coder outside the
    one-pass kernel's range takes, and what is queued behind every one-pass launch in case a look-back wait runs out.
    Same scenarios as the default road, and the 1 GiB stream's digest."""
    lib = harness.load_product()
    with harness.encode_road(lib, "three-kernel"):
        w = pc.World(oracle, harness.Codec(lib, "aws_"))
        pc.one_shot_roundtrips(w, sizes=[1, 17, 16384, 16385, 40000, 3 * 1024 * 1024 + 5])
        pc.streaming_encode(w, sizes=[40, 33000])
        pc.unknown_symbols(w)
        eng = harness.Engine(w.product.lib, w.pcoder)
        assert not eng.lib.aws_huffman_amd_engine_encodes_in_one_pass(eng.h)
        pc.batched_device_api(w, n_items=40, engine=eng)
        rec = PROBE["streams"]["G1G"]
        n, e = rec["len"], rec["encoded_len"]
        d_in, d_enc = eng.alloc(n), eng.alloc(e + 64)
        eng.fill_splitmix64(d_in, n, rec["seed"])
        plan = eng.encode_plan([dict(in_offset=0, in_len=n, out_offset=0, out_capacity=e + 64)])
        eng.encode_launch(plan, d_in, d_enc)
        (rc, err, consumed, produced, ob, op), = eng.encode_results(plan, 1)
        assert (rc, consumed, produced) == (0, n, e)
        assert eng.encode_road(plan) == pc.ROAD_TWO_PASS
        h = hashlib.sha256()
        for off in range(0, e, 256 << 20):
            h.update(eng.download(d_enc, min(256 << 20, e - off), offset=off).tobytes())
        assert h.hexdigest() == rec["sha256_encoded"]
        eng.close()


def test_encode_roads(world):
    """enc_onepass; count / scan / pack when told so; and count / scan / pack queued behind an enc_onepass launch in
    which a wave gave up (a look-back wait that ran out is what it stands for): the output is read on the stream right
    behind the launch, before any record is fetched, and must be whole."""
    pc.encode_roads(world)
    pc.encode_roads(world, sizes=(33_000_000, 16384, 5_000_000, 70_000), seed=58)


def test_one_gib_stream_when_the_one_pass_encoder_gives_up(oracle):
    """The 1 GiB stream with a wave of enc_onepass made to give up half-way: the three-kernel road behind it on the
    stream must leave the pinned stream."""
    lib = harness.load_product()
    with harness.encode_road(lib, "one-pass-fails"):
        w = pc.World(oracle, harness.Codec(lib, "aws_"))
        eng = harness.Engine(w.product.lib, w.pcoder)
    rec = PROBE["streams"]["G1G"]
    n, e = rec["len"], rec["encoded_len"]
    d_in, d_enc = eng.alloc(n), eng.alloc(e + 64)
    eng.fill_splitmix64(d_in, n, rec["seed"])
    eng.fill(d_enc, 0x5A, e + 64)
    plan = eng.encode_plan([dict(in_offset=0, in_len=n, out_offset=0, out_capacity=e + 64)])
    eng.encode_launch(plan, d_in, d_enc)
    h = hashlib.sha256()  # (the stream, read behind the launch, before the records)
    for off in range(0, e, 256 << 20):
        h.update(eng.download(d_enc, min(256 << 20, e - off), offset=off).tobytes())
    (rc, err, consumed, produced, ob, op), = eng.encode_results(plan, 1)
    assert (rc, consumed, produced) == (0, n, e) and eng.encode_road(plan) == pc.ROAD_GAVE_UP
    assert h.hexdigest() == rec["sha256_encoded"]
    assert np.all(eng.download(d_enc, 64, offset=e) == 0x5A)
    eng.close()




def test_plan_launches_in_a_hip_graph():
    """an encode launch and the decode launch of its output, captured into one HIP graph on a stream of
    the caller's and replayed over scrambled outputs (profiles/tools/graph_capture.py: a batch of 16 KiB items and one
    64 MiB stream -- kernels and the second stream's fork and join are all nodes of the graph; and a stream whose replays
    take the ways back every time: what those leave in control words and counters must not outlive a replay)."""
    import subprocess
    import sys

    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "tools", "graph_capture.py")
    out = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("decoded back bit-exact") == 3 and out.stdout.count("(the ways back)") == 1, out.stdout


def test_two_one_pass_encoders_on_one_device():
    """The one-pass encoder's tile schedule wants the device to itself; two engines on one device launching at the same
    time from two threads take that away for real (profiles/tools/contention.py): whatever waits run out, every launch
    must leave the right bytes on its stream -- read back before the records are fetched."""
    import subprocess
    import sys

    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "tools", "contention.py")
    out = subprocess.run([sys.executable, tool, str(128 << 20), "8"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.count("wrong outputs [0, 0]") == 2, out.stdout + out.stderr
