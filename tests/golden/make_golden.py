#!/usr/bin/env python3
"""Extract golden DATA for the Huffman hot path from the reference checkout.

Runs only in the build container (needs /root/reference); the GPU box uses the
committed JSON.  Nothing here copies reference source text: the outputs are
input/output vectors and table rows (data), written to

  tests/golden/test_coder_table.json     256 (symbol, pattern, num_bits) rows of the test coder
                                         <- reference tests/test_huffman_static_table.def:11-266
  tests/golden/reference_vectors.json    known-answer vectors of the reference's unit tests
                                         <- reference tests/huffman_test.c:20-39,178-194,408
  tests/golden/test_coder_decode_tree.json  leaves and dead ends of the generated decoder's decision tree
                                         <- reference tests/test_huffman_static.c:276-2381, cross-checked
                                            against the output of the reference's generator tool
                                            (source/huffman_generator/generator.c) rebuilt by oracle/Makefile

Usage: python tests/golden/make_golden.py [--reference /root/reference]
"""
import argparse
import ast
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))


def parse_table(def_path):
    rows = []
    pat = re.compile(r'HUFFMAN_CODE\(\s*(\d+)\s*,\s*"([01]+)"\s*,\s*0x([0-9a-fA-F]+)\s*,\s*(\d+)\s*\)')
    with open(def_path) as f:
        for line in f:
            m = pat.search(line)
            if not m:
                continue
            sym, bits, code, n = int(m.group(1)), m.group(2), int(m.group(3), 16), int(m.group(4))
            assert len(bits) == n and int(bits, 2) == code, line
            rows.append({"symbol": sym, "pattern": code, "num_bits": n})
    assert [r["symbol"] for r in rows] == list(range(256))
    return rows


def c_string_literals(text):
    """Concatenate adjacent C string literals found in `text`."""
    out = b""
    for lit in re.findall(r'"(?:[^"\\]|\\.)*"', text):
        out += ast.literal_eval("b" + lit)
    return out


def parse_unit_test_vectors(test_c_path):
    src = open(test_c_path).read()

    def array_bytes(name):
        m = re.search(r"static uint8_t %s\[\] = \{(.*?)\};" % name, src, re.S)
        return bytes(int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{2})", m.group(1)))

    def string_bytes(name):
        m = re.search(r"static const char %s\[\] = (.*?);\nenum" % name, src, re.S)
        return c_string_literals(m.group(1))

    steps = re.search(r"s_step_sizes\[\] = \{(.*?)\};", src).group(1)
    vec = {
        "K1_url": {"plain": string_bytes("s_url_string").hex(), "encoded": array_bytes("s_encoded_url").hex()},
        "K2_all_codes": {"plain": string_bytes("s_all_codes").hex(), "encoded": array_bytes("s_encoded_codes").hex()},
        "step_sizes": [int(x) for x in steps.split(",")],
    }
    # exact-fit cases (tests/huffman_test.c:178-194) and the even-bytes case (:408)
    m1 = re.search(r'aws_byte_cursor_from_array\("(\?)", 1\);\s*uint8_t expected_1byte\[\] = \{(.*?)\};', src, re.S)
    m2 = re.search(r'aws_byte_cursor_from_array\("(yz)", 2\);\s*uint8_t expected_2byte\[\] = \{(.*?)\};', src, re.S)
    m4 = re.search(r'huffman_test_transitive\(test_get_coder\(\), "(\w+)", (\d+), (\d+),', src)
    vec["K3_exact_fit"] = [
        {"plain": m1.group(1).encode().hex(), "encoded": bytes(int(x, 16) for x in re.findall(r"0x([0-9a-f]{2})", m1.group(2))).hex()},
        {"plain": m2.group(1).encode().hex(), "encoded": bytes(int(x, 16) for x in re.findall(r"0x([0-9a-f]{2})", m2.group(2))).hex()},
    ]
    vec["K4_even_bytes"] = {"plain": m4.group(1).encode().hex(), "plain_len": int(m4.group(2)), "encoded_len": int(m4.group(3))}
    assert len(bytes.fromhex(vec["K1_url"]["plain"])) == 15 and len(bytes.fromhex(vec["K1_url"]["encoded"])) == 12
    assert len(bytes.fromhex(vec["K2_all_codes"]["plain"])) == 95 and len(bytes.fromhex(vec["K2_all_codes"]["encoded"])) == 103
    return vec


def parse_decision_tree(generated_c_path):
    """Walk the goto tree of decode_symbol: returns (leaves, dead_ends).

    leaves: {prefix_bits: (symbol, length)}; dead_ends: [prefix_bits] ("return 0; invalid node").
    """
    lines = open(generated_c_path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if "decode_symbol(uint32_t bits" in l)
    leaves, dead = {}, []
    prefix, branch, pending_symbol = "", None, None
    for l in lines[start:]:
        s = l.strip()
        m = re.match(r"node_([01]+):", s)
        if m:
            prefix = m.group(1)
            continue
        m = re.match(r"if \(bits & 0x([0-9a-f]+)\) \{", s)
        if m:
            assert int(m.group(1), 16) == 1 << (31 - len(prefix)), (prefix, s)
            branch = "1"
            continue
        if s == "} else {":
            branch = "0"
            continue
        if branch is None:
            continue
        m = re.match(r"\*symbol = (\d+);", s)
        if m:
            pending_symbol = int(m.group(1))
            continue
        m = re.match(r"return (\d+);", s)
        if m:
            n = int(m.group(1))
            path = prefix + branch
            if n == 0:
                dead.append(path)
            else:
                assert pending_symbol is not None and n == len(path), (path, n)
                leaves[path] = (pending_symbol, n)
                pending_symbol = None
            continue
        m = re.match(r"goto node_([01]+);", s)
        if m:
            assert m.group(1) == prefix + branch
            continue
        if s.startswith("struct aws_huffman_symbol_coder"):
            break
    return leaves, dead


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    ref = args.reference
    if not os.path.isdir(ref):
        sys.exit("reference checkout not found: %s" % ref)

    rows = parse_table(os.path.join(ref, "tests/test_huffman_static_table.def"))
    with open(os.path.join(HERE, "test_coder_table.json"), "w") as f:
        json.dump({"source": "reference tests/test_huffman_static_table.def:11-266", "rows": rows}, f, indent=0)

    vec = parse_unit_test_vectors(os.path.join(ref, "tests/huffman_test.c"))
    vec["source"] = "reference tests/huffman_test.c:20-39,178-194,408"
    with open(os.path.join(HERE, "reference_vectors.json"), "w") as f:
        json.dump(vec, f, indent=1)

    leaves, dead = parse_decision_tree(os.path.join(ref, "tests/test_huffman_static.c"))
    # every table row must be a leaf of the committed generated coder, and nothing else
    assert len(leaves) == 256
    for r in rows:
        path = format(r["pattern"], "0%db" % r["num_bits"])
        assert leaves[path] == (r["symbol"], r["num_bits"]), r
    regenerated = os.path.join(REPO, "oracle/_ref/generated_test_coder.c")
    rebuilt_matches = None
    if os.path.exists(regenerated):
        l2, d2 = parse_decision_tree(regenerated)
        rebuilt_matches = (l2 == leaves and sorted(d2) == sorted(dead))
        assert rebuilt_matches, "generator rebuilt from source disagrees with the committed generated coder"
    tree = {
        "source": "reference tests/test_huffman_static.c:276-2381 (decision tree of decode_symbol)",
        "generator_rebuild_matches": rebuilt_matches,
        "leaves": [{"prefix": p, "symbol": s, "num_bits": n} for p, (s, n) in sorted(leaves.items())],
        "dead_ends": sorted(dead, key=lambda p: (len(p), p)),
    }
    with open(os.path.join(HERE, "test_coder_decode_tree.json"), "w") as f:
        json.dump(tree, f, indent=0)
    print("table rows: %d  leaves: %d  dead ends: %s  generator rebuild matches: %s"
          % (len(rows), len(leaves), tree["dead_ends"], rebuilt_matches))


if __name__ == "__main__":
    main()
