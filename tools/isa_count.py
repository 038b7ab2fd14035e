#!/usr/bin/env python3
"""Static instruction counts of the main loop of a gfx950 kernel, from a `hipcc -S --cuda-device-only` listing.
The main loop = the backward branch that spans the most instructions (one CMUX step of a blind-rotation kernel).  Vector
instructions are split by issue cost on a SIMD-32 (MI355X_MICROARCH.md: a wave64 f32 instruction issues over 2 cycles; f64 and
64-bit integer instructions run at half that rate: 4 cycles):
    f64      v_*_f64 (incl. conversions to / from f64)
    int64    v_mad_u64_u32 / v_mad_i64_i32, v_lshl*_b64, v_lshr*_b64, v_ashr*_i64, v_mul_hi_*, v_mul_lo_u32 (quarter-rate multiplies count here)
    other    every other v_* instruction (32-bit ALU, moves, compares, cndmask ...)
usage: isa_count.py listing.s kernel_substring [kernel_substring ...]     -> one JSON object per kernel on stdout
       isa_count.py --build <out.json>      compiles the four blind-rotation TUs and writes the counts of the shipped kernels"""
import json, os, re, subprocess, sys, tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "bounty-matrix-inversion_amd", "csrc")


def kernels_in(listing):
    L = open(listing).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(L) if l.startswith("_Z") and ":" in l and "@" in l]
    out = {}
    for (i, name), nxt in zip(starts, starts[1:] + [(len(L), None)]):
        end = next((j for j in range(i, nxt[0]) if "s_endpgm" in L[j]), nxt[0])
        out[name] = L[i:end + 1]
    return out


def classify(op):
    if not op.startswith("v_"):
        if op.startswith("ds_"): return "lds"
        if op.startswith(("global_", "buffer_", "flat_")): return "vmem"
        if op.startswith("scratch_"): return "scratch"
        if op.startswith("s_barrier"): return "barrier"
        if op.startswith("s_waitcnt"): return "waitcnt"
        return "scalar"
    if "_f64" in op: return "f64"
    if op.startswith(("v_mad_u64", "v_mad_i64", "v_mul_hi", "v_mul_lo")) or re.search(r"_[bi]64", op): return "int64"
    return "other"


def main_loop(body):
    """(first, last) instruction index of the backward branch with the widest span"""
    ins, labels = [], {}
    for l in body:
        s = l.strip()
        if re.match(r"^\.LBB[0-9_]+:", s):
            labels[s.split(":")[0]] = len(ins)
        elif l.startswith("\t") and s and not s.startswith((".", ";")):
            ins.append(s)
    best = None
    for i, s in enumerate(ins):
        p = s.split()
        if p[0].startswith(("s_cbranch", "s_branch")) and len(p) > 1 and p[1] in labels and labels[p[1]] <= i:
            span = (labels[p[1]], i)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    return ins, best


def count(body):
    ins, span = main_loop(body)
    if span is None:
        return None
    c = {}
    for s in ins[span[0]:span[1] + 1]:
        k = classify(s.split()[0])
        c[k] = c.get(k, 0) + 1
    valu = c.get("f64", 0) + c.get("int64", 0) + c.get("other", 0)
    cyc = 4 * (c.get("f64", 0) + c.get("int64", 0)) + 2 * c.get("other", 0)
    return {"main_loop_instructions": span[1] - span[0] + 1, "valu": valu, "valu_f64": c.get("f64", 0), "valu_int64": c.get("int64", 0),
            "valu_other": c.get("other", 0), "lds": c.get("lds", 0), "vmem": c.get("vmem", 0), "scratch": c.get("scratch", 0),
            "barriers": c.get("barrier", 0), "valu_issue_cycles": cyc, "avg_issue_cycles_per_valu": round(cyc / max(valu, 1), 3)}


def demangle(names):
    try:
        out = subprocess.run(["c++filt"] + names, capture_output=True, text=True, check=True).stdout.split("\n")
        return [re.sub(r"^void \(anonymous namespace\)::", "", o).split("(")[0] for o in out[:len(names)]]
    except Exception:
        return names


def build(out_path):
    srcs = {"bmi_kernels_f64.hip": True, "bmi_kernels_t64.hip": True, "bmi_kernels_t64f.hip": True, "bmi_kernels_t64fu.hip": True,
            "bmi_kernels_t64w.hip": True, "bmi_kernels_t64w2.hip": True, "bmi_kernels_t64q.hip": True, "bmi_kernels_f64u.hip": False, "bmi_kernels_t64u.hip": False}   # True: max-ilp scheduler (csrc/Makefile)
    want = ("k_blind_rotate_tpx49", "k_blind_rotate_t64", "k_blind_rotate_lat2_49", "k_blind_rotate_lat2u_49", "k_blind_rotate_lat_t64",
            "k_blind_rotate_lat2u_t64", "k_blind_rotate_w_t64f", "k_blind_rotate_w2_t64f", "k_blind_rotate_q_t64f")   # (substring match: the t64f / t64fu kernels are picked up by the t64 entries)
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for src, ilp in srcs.items():
            lst = os.path.join(td, src + ".s")
            cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only", "-S", "-o", lst,
                   "-x", "hip", os.path.join(CSRC, src)] + (["-mllvm", "-amdgpu-sched-strategy=max-ilp"] if ilp else [])
            subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
            ks = kernels_in(lst)
            names = [n for n in ks if any(w in n for w in want)]
            for n, d in zip(names, demangle(names)):
                c = count(ks[n])
                if c:
                    res[d] = dict(c, source=src, mangled=n)
    json.dump({"what": "static counts of one iteration of the main (blind-rotation step) loop; every path of wave-conditional phases is counted once; "
                       "issue cycles: 4 per f64 / 64-bit-integer instruction, 2 per other vector instruction (SIMD-32, wave64)",
               "tool": "tools/isa_count.py --build", "kernels": res}, open(out_path, "w"), indent=1)
    for k, v in res.items():
        print(k, v["valu"], v["valu_f64"], v["valu_int64"], v["valu_other"], v["avg_issue_cycles_per_valu"])


if __name__ == "__main__":
    if sys.argv[1] == "--build":
        build(sys.argv[2])
    else:
        ks = kernels_in(sys.argv[1])
        for sub in sys.argv[2:]:
            for n in ks:
                if sub in n:
                    print(json.dumps({"kernel": demangle([n])[0], **(count(ks[n]) or {})}))
