import ctypes as C, os, sys
REPO = "/root/repo"
sys.path.insert(0, os.path.join(REPO, "tests"))
import harness, numpy as np
lib = harness.load_product(os.path.join(REPO, "aws-c-compression_amd", "libaws-c-compression-amd-stamps.so"))
lib.hufk_stamps_attach.argtypes = [C.c_void_p]
patterns, lens = harness.load_table()
eng = harness.Engine(lib, lib.aws_huffman_amd_table_coder_new(patterns, lens))
MAX_WG = 131072
d_rows = eng.alloc(3 * MAX_WG * 8 * 8)
assert lib.hufk_stamps_attach(d_rows) == 0
n = 1 << 30
d_in, d_enc = eng.alloc(n), eng.alloc(n * 10 // 8 + 64)
eng.fill_splitmix64(d_in, n, 5)
ep = eng.encode_plan([dict(in_offset=0, in_len=n, out_offset=0, out_capacity=n * 10 // 8 + 64)])
eng.encode_launch(ep, d_in, d_enc); eng.encode_results(ep, 1)
eng.upload(d_rows, np.zeros(3 * MAX_WG * 8, dtype=np.uint64).view(np.uint8))
eng.encode_launch(ep, d_in, d_enc); eng.encode_results(ep, 1)
raw = eng.download(d_rows, 4096 * 64, offset=2 * MAX_WG * 64).view(np.uint64).reshape(4096, 8).astype(np.float64)
r = raw[raw[:, 0] > 0]
tiles = (n + 4095) // 4096
per = tiles / 8.0 / len(r)
print("workgroups", len(r), "tiles per wave", per)
print("first look misses per tile: tile words %.3f, group words %.3f, round base %.3f; polls per tile %.2f" % (
    r[:, 3].mean() / per, r[:, 4].mean() / per, r[:, 5].mean() / per, r[:, 6].mean() / per))
idx = np.nonzero(raw[:, 0] > 0)[0]
polls = r[:, 2] - r[:, 7]
fresh = r[:, 1] - r[:, 0]
print("poll clocks per workgroup (wave 0): mean %.0f, percentiles 1/10/50/90/99: %s" % (polls.mean(), np.percentile(polls, [1, 10, 50, 90, 99]).round()))
print("fresh-phase clocks: mean %.0f, percentiles %s" % (fresh.mean(), np.percentile(fresh, [1, 10, 50, 90, 99]).round()))
for m in (8, 2):
    print("polls by blockIdx %% %d:" % m, [round(polls[idx % m == k].mean()) for k in range(m)])
    print("fresh by blockIdx %% %d:" % m, [round(fresh[idx % m == k].mean()) for k in range(m)])
lo = np.argsort(polls)[:16]
print("the 16 workgroups that poll least:", sorted(idx[lo].tolist()), "their polls", polls[lo].round().tolist())
