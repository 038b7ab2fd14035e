#!/usr/bin/env python3
"""Where the depth of a traced inverse goes: every look-up is tagged with the host function that asked for it (a frame of
base_p_arrays / qfloat / qfloat_matrix_inversion, chosen by a priority list), then one critical path (deepest leaf, always the
deepest predecessor) is walked back and its levels are counted per tag.
usage: critical_path.py n len ints [division_bits]"""
import collections, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bounty-matrix-inversion_amd"))
from bmi_amd import circuit as cir, main

TAGS = ["_division_radix", "_division_bitserial", "carry_propagate_signed", "carry_propagate_nonneg", "is_greater_or_equal", "is_equal",
        "base_p_subtraction", "tidy", "base_tidy", "__imul__", "from_mul", "multi_from_mul", "__iadd__", "invert", "qfloat_argmax",
        "qfloat_pivot_matrix", "select", "abs", "__gt__"]
tag_of = {}


def wrap(name):
    orig = getattr(cir.Circuit, name)

    def f(self, *a, **k):
        before = len(self.leaf_level)
        r = orig(self, *a, **k)
        if len(self.leaf_level) > before:
            fr, names = sys._getframe(1), []
            while fr is not None and len(names) < 40:
                names.append(fr.f_code.co_name)
                fr = fr.f_back
            tag = next((t for t in TAGS if t in names), names[0])
            outer = next((t for t in ("invert", "__imul__", "from_mul", "multi_from_mul", "__iadd__", "tidy", "qfloat_argmax", "qfloat_pivot_matrix") if t in names), "")
            for leaf in range(before, len(self.leaf_level)):
                tag_of[leaf] = (tag, outer)
        return r
    setattr(cir.Circuit, name, f)


for nm in ("lut", "lut_odd", "lut_neg"):
    wrap(nm)
n, ln, ints = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
bits = int(sys.argv[4]) if len(sys.argv) > 4 else 3
c = main.trace_inverse(n, ln, ints, 2, False, False, bits)
producer = {leaf: i for i, (_, _, _, leaf) in enumerate(c.nodes)}
# outputs -> deepest leaf
best = max((t for terms, _ in c.outputs for t, _ in terms), key=lambda t: c.leaf_level[t])
path = []
leaf = best
while leaf in producer:
    path.append(leaf)
    terms = c.nodes[producer[leaf]][0]
    preds = [t for t, _ in terms if t in producer]
    if not preds:
        break
    leaf = max(preds, key=lambda t: c.leaf_level[t])
cnt, cnt2 = collections.Counter(), collections.Counter()
for leaf in path:
    cnt[tag_of.get(leaf, ("?", ""))[0]] += 1
    cnt2[tag_of.get(leaf, ("?", ""))] += 1
print("depth", max(c.leaf_level), "critical path length", len(path), "pbs", len(c.nodes))
for k, v in cnt.most_common():
    print(f"  {v:4d}  {k}")
print("by (function, enclosing operation):")
for k, v in cnt2.most_common(16):
    print(f"  {v:4d}  {k}")
allc = collections.Counter(tag_of.get(leaf, ("?", "")) for _, _, _, leaf in c.nodes)
print("all look-ups by (function, enclosing operation):")
for k, v in allc.most_common(14):
    print(f"  {v:6d}  {k}")
