import sys, os, hashlib
sys.path.insert(0, "/root/repo/tests")
import harness
lib = harness.load_product(sys.argv[1])
patterns, lens = harness.load_table()
coder = lib.aws_huffman_amd_table_coder_new(patterns, lens)
eng = harness.Engine(lib, coder)
n = 1 << 30
cap = n * 10 // 8 + 64
d_in, d_out = eng.alloc(n), eng.alloc(cap)
eng.fill_splitmix64(d_in, n, 5)
plan = eng.encode_plan([dict(in_offset=0, in_len=n, out_offset=0, out_capacity=cap)])
ev = eng.new_events(2)
ts = []
for k in range(14):
    eng.record(ev[0]); eng.encode_launch(plan, d_in, d_out); eng.record(ev[1]); eng.sync()
    ts.append(eng.elapsed_ms(ev[0], ev[1]))
res = eng.encode_results(plan, 1)[0]
dg = hashlib.sha256(eng.download(d_out, res[3]).tobytes()).hexdigest()[:12]
print(os.path.basename(sys.argv[1]), "mode", os.environ.get("HUFD_DBG_MODE", "0"), "median %.3f" % sorted(ts[4:])[len(ts[4:])//2], "min %.3f" % min(ts[4:]), res[:4], dg, flush=True)
